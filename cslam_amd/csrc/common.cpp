#include "common.hpp"

#include <cstring>

namespace cslam {

std::string& last_error_slot() {
    static thread_local std::string slot;
    return slot;
}

int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    last_error_slot() = buf;
    return code;
}

int device_cu_count(int device_id, int* cus) {
    hipDeviceProp_t prop;
    CSLAM_HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    *cus = prop.multiProcessorCount;
    return CSLAM_OK;
}

}  // namespace cslam

extern "C" {

const char* cslam_last_error(void) { return cslam::last_error_slot().c_str(); }

int cslam_abi_version(void) { return 1; }

int cslam_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

}  // extern "C"
