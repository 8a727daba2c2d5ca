// Address audit of k_scan_mq / k_mq_combine (tools/build_mq_probes.sh, PROBES=AUDIT): every global address the
// kernels are about to touch is compared with what the launch arguments (MqArgs) say the operand holds -
//   0 pool rows      [X, X + (n - 1) xs + dp) floats      1 staged queries [Q, Q + 16 T dp) floats
//   2 query norms    [rnq, rnq + 16 T)                    3 row norms      [rnx, rnx + n)
//   4 chunk partials [part, part + strips chunks T 64) float4
//   5 composites     index < n_blocks p_max NW per query   6 dots           window row < n
// - a violation is counted, the first 16 are logged (kind, block, thread, address, operand range), and the access
// is redirected to the operand's first element (loads) or left out (stores), so that an addressing defect shows
// up as a log entry instead of a memory fault of the GPU.  Results of this build are the product's (the audit
// changes no address that is in range); it is slower.  The host side of the chain - MqArgs against the pool's
// allocations - is checked by knn.hip::mq_pass in every build.
// Read the log with cslam_probe_mq_audit_read (tools/mq_audit.py).  -DCSLAM_MQ_AUDIT_SELFTEST: negative control
// (the last row of the window counts as out of range: the log must fill, results must change).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

__device__ unsigned long long g_mq_audit_log[1 + 16 * 4];

__device__ inline bool mq_audit_note(int kind, const void* addr, const void* lo, const void* hi) {
    const unsigned long long n = atomicAdd(&g_mq_audit_log[0], 1ull);
    if (n < 16) {
        g_mq_audit_log[1 + 4 * n] = (unsigned long long)kind | ((unsigned long long)blockIdx.x << 8) |
                                    ((unsigned long long)threadIdx.x << 40);
        g_mq_audit_log[2 + 4 * n] = (unsigned long long)addr;
        g_mq_audit_log[3 + 4 * n] = (unsigned long long)lo;
        g_mq_audit_log[4 + 4 * n] = (unsigned long long)hi;
    }
    return false;
}
template <class Args>
__device__ inline void mq_audit_range(const Args& A, int kind, int tiles16, const char** lo, const char** hi) {
    switch (kind) {
        case 0:
            *lo = reinterpret_cast<const char*>(A.X);
            *hi = reinterpret_cast<const char*>(A.X + (A.n - 1) * A.xs + A.dp);
#if defined(CSLAM_MQ_AUDIT_SELFTEST)   // negative control: the window's last row is declared out of range
            *hi = reinterpret_cast<const char*>(A.X + (A.n - 2) * A.xs + A.dp);
#endif
            break;
        case 1:
            *lo = reinterpret_cast<const char*>(A.Q);
            *hi = reinterpret_cast<const char*>(A.Q + (int64_t)tiles16 * A.dp);
            break;
        case 2:
            *lo = reinterpret_cast<const char*>(A.rnq);
            *hi = reinterpret_cast<const char*>(A.rnq + tiles16);
            break;
        case 3:
            *lo = reinterpret_cast<const char*>(A.rnx);
            *hi = reinterpret_cast<const char*>(A.rnx + A.n);
            break;
        default: {
            const int64_t strips = (A.n + 15) >> 4, chunks = (A.dp + 511) / 512;
            *lo = reinterpret_cast<const char*>(A.part);
            *hi = reinterpret_cast<const char*>(A.part + strips * chunks * (tiles16 / 16) * 64);
        }
    }
}
template <class Args>
__device__ inline const char* mq_audit_src(const Args& A, int kind, int tiles16, const char* p, int bytes) {
    const char *lo, *hi;
    mq_audit_range(A, kind, tiles16, &lo, &hi);
    if (p >= lo && p + bytes <= hi) return p;
    mq_audit_note(kind, p, lo, hi);
    return lo;
}
template <class Args>
__device__ inline bool mq_audit_dst(const Args& A, int kind, int tiles16, const void* p, int bytes) {
    const char *lo, *hi;
    mq_audit_range(A, kind, tiles16, &lo, &hi);
    const char* c = static_cast<const char*>(p);
    if (c >= lo && c + bytes <= hi) return true;
    return mq_audit_note(kind, p, lo, hi);
}
__device__ inline bool mq_audit_idx(int kind, int64_t idx, int64_t bound) {
    if (idx >= 0 && idx < bound) return true;
    return mq_audit_note(kind, (const void*)idx, nullptr, (const void*)bound);
}

#define CSLAM_MQ_AUDIT_SRC(kind, ptr, bytes) ptr = mq_audit_src(A, kind, 16 * T, ptr, bytes);
#define CSLAM_MQ_AUDIT_DST(kind, ptr, bytes) if (mq_audit_dst(A, kind, 16 * T, ptr, bytes))
#define CSLAM_MQ_AUDIT_IDX(kind, idx, bound) if (mq_audit_idx(kind, (int64_t)(idx), (int64_t)(bound)))
// k_mq_combine: the chunk partials it reads, the norms, and the slots it writes
#define CSLAM_MQ_AUDIT_COMBINE                                                                                      \
    {                                                                                                               \
        bool ok = true;                                                                                             \
        for (int c = 0; c < nchunks; ++c)                                                                           \
            ok = ok && mq_audit_dst(A, 4, 16 * tiles, src + (int64_t)c * tiles * 64, 16);                           \
        ok = ok && mq_audit_idx(3, hi - 1, A.n) && (!BEST || mq_audit_idx(5, cand_off + strip, A.cand_stride));     \
        if (!ok) return;                                                                                            \
    }

// host side: copy the log out and clear it.  out[0] = violations since the last read, then 16 x 4 words.
extern "C" int cslam_probe_mq_audit_read(unsigned long long* out) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mq_audit_log), sizeof(unsigned long long) * (1 + 16 * 4)) != hipSuccess)
        return -1;
    static const unsigned long long zeros[1 + 16 * 4] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_mq_audit_log), zeros, sizeof(zeros)) != hipSuccess) return -1;
    return 0;
}
