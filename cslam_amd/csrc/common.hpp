// Shared host-side helpers of libcslam_hip.so (error slot, HIP checks, device buffers).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>

#include "../../include/cslam_hip.h"

namespace cslam {

std::string& last_error_slot();
int fail(int code, const char* fmt, ...);

#define CSLAM_HIP_TRY(expr)                                                                  \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess)                                                                \
            return ::cslam::fail(CSLAM_E_HIP, "%s failed: %s (%s:%d)", #expr,                \
                                 hipGetErrorString(_e), __FILE__, __LINE__);                 \
    } while (0)

#define CSLAM_TRY(expr)                \
    do {                               \
        int _rc = (expr);              \
        if (_rc != CSLAM_OK) return _rc; \
    } while (0)

// Growable device allocation; contents are NOT preserved by reserve().
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int reserve(size_t want) {
        if (want <= bytes) return CSLAM_OK;
        if (p) {
            CSLAM_HIP_TRY(hipFree(p));
            p = nullptr;
            bytes = 0;
        }
        size_t grown = want + want / 4;
        CSLAM_HIP_TRY(hipMalloc(&p, grown));
        bytes = grown;
        return CSLAM_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <class T>
    T* as() const {
        return static_cast<T*>(p);
    }
};

int device_cu_count(int device_id, int* cus);

}  // namespace cslam
