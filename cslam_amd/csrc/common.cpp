#include "common.hpp"

#include <cstdlib>
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

namespace {
std::atomic<int> g_knob_generation{1};
}
int knob(Knob& k) {
    static const bool live = getenv("CSLAM_KNOBS_LIVE") != nullptr;
    const int g = g_knob_generation.load(std::memory_order_relaxed);
    if (live || k.generation.load(std::memory_order_acquire) != g) {
        const char* s = getenv(k.name);
        k.value.store(s ? atoi(s) : k.def, std::memory_order_relaxed);
        k.generation.store(g, std::memory_order_release);
    }
    return k.value.load(std::memory_order_relaxed);
}
void knobs_invalidate() { g_knob_generation.fetch_add(1, std::memory_order_acq_rel); }

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
