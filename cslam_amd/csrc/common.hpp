// Shared host-side helpers of libcslam_hip.so (error slot, HIP checks, device buffers).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
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

// Growable pinned host allocation (device-visible under the same address: hipHostMalloc maps it);
// contents are NOT preserved by reserve().
struct PinBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int reserve(size_t want) {
        if (want <= bytes) return CSLAM_OK;
        if (p) {
            CSLAM_HIP_TRY(hipHostFree(p));
            p = nullptr;
            bytes = 0;
        }
        size_t grown = want + want / 4;
        CSLAM_HIP_TRY(hipHostMalloc(&p, grown, hipHostMallocDefault));
        bytes = grown;
        return CSLAM_OK;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <class T>
    T* as() const {
        return static_cast<T*>(p);
    }
};

int device_cu_count(int device_id, int* cus);

// Measurement / A-B knobs (environment variables, KNOBS.md): read ONCE per process, then cached.  Two ways to
// have them read again, both for tests and tools only: cslam_knobs_reload() (every knob re-reads its variable
// at its next use: a test changes one setting, reloads, and reloads again when it restores the environment)
// and CSLAM_KNOBS_LIVE (every use re-reads: tools that alternate settings launch by launch, inside one
// process - boxes and processes differ by more than the effects in question).  Fields are atomics: shards
// run as threads of one process (tests/test_shard_gpu.py).  Use: `static Knob k{"CSLAM_X", 1}; knob(k)`.
struct Knob {
    const char* name;
    int def;
    std::atomic<int> value{0};
    std::atomic<int> generation{0};
};
int knob(Knob& k);
void knobs_invalidate();   // what cslam_knobs_reload() does

// knn.hip → shard.hip (internal).  pool_append_on: cslam_pool_add_dev without the cross-stream
// ordering of the public entry points (the shard orders its streams with its own events);
// pool_order_on: make the pool's other entry points wait for what `st` has been given so far.
int pool_append_on(cslam_pool_t* p, const float* rows_dev, int64_t n_rows, int64_t row_stride,
                   hipStream_t st);
int pool_order_on(cslam_pool_t* p, hipStream_t st);
// While on, pool calls do not wait for an append queued on another stream (pool_append_on): the
// shard's own searches work on the explicit row windows of THEIR step and are ordered behind that
// step's publish event; the append they would wait for belongs to a later step of the pipeline.
void pool_internal_scope(cslam_pool_t* p, bool on);
struct PoolInternalScope {
    cslam_pool_t* p;
    explicit PoolInternalScope(cslam_pool_t* pool) : p(pool) { pool_internal_scope(p, true); }
    ~PoolInternalScope() { pool_internal_scope(p, false); }
};
// candidates of a split best-match scan, see knn.hip
struct BestCandidates {
    const uint64_t* a = nullptr;  // [m][stride] composites of rows < split, n valid per query
    const uint64_t* c = nullptr;  // ... of rows >= split
    int64_t stride = 0, n = 0;
};
// staged = the block pool_stage_queries_on() left for these m queries (17..80: the multi-tile scan then
// skips its own staging launch), or nullptr
int pool_best_split_candidates(cslam_pool_t* p, const float* queries_dev, int64_t query_stride, int m,
                               int64_t split_row, int64_t row_hi, hipStream_t st,
                               BestCandidates* out, bool* applied, const float* staged = nullptr,
                               const float* staged_rnq = nullptr);
// What a best-match scan with dots leaves for the rank select of topk_parts.hpp (search with 2 <= k <= 32): the
// best composite of every part of the window [row_lo, row_hi) - waves of k_scan_dma (kind 0) or panels / K-split
// strips of k_scan_mq (kind 1) - and the raw dots of every row.  The caller launches the select itself (the
// shard: its records kernel, in the same launch).  *applied = false (nothing launched) where that scan does
// not run for this shape: the caller then takes cslam_pool_search_range_dev.
struct TopkCandidates {
    const uint64_t* cand = nullptr;
    int64_t cand_stride = 0;
    int n_parts = 0, kind = 0;
    int64_t wq = 0, wrem = 0;                                   // kind 0: WaveParts
    int64_t n_main = 0;                                         // kind 1: PanelParts
    int n_blocks = 0, p_max = 0, nw = 0, ns = 0, n_main_parts = 0;
    const float* dots = nullptr;
    int64_t dots_stride = 0;
    const float* rnx = nullptr;   // 1 / ||x|| of the window's rows
    const float* rnq = nullptr;   // 1 / ||q|| of the m queries
    uint32_t row0 = 0;            // pool row of the window's first row
    int64_t n = 0;                // rows of the window
};
int pool_topk_candidates(cslam_pool_t* p, const float* queries_dev, int64_t query_stride, int m, int k,
                         int64_t row_lo, int64_t row_hi, hipStream_t st, TopkCandidates* out, bool* applied);
// The multi-tile scan reads its queries from a zero-padded block [16 ceil(m / 16)][padded dim] next to
// their inverse norms.  A caller that has the queries a step early (the shard: the exchange lands while
// the previous step is being scanned) stages them on ITS stream into buffers of its own.
int pool_mq_free_cus(const cslam_pool_t* p, int m, int64_t rows);   // CUs that scan leaves to other streams
int64_t pool_staged_floats(const cslam_pool_t* p, int m);   // floats of the staged block (norms: 16 ceil(m / 16))
// (group_rows > 0: row w sits at queries_dev + (w / group_rows) * group_stride + (w % group_rows) * query_stride)
int pool_stage_queries_on(cslam_pool_t* p, const float* queries_dev, int64_t query_stride, int m,
                          float* staged, float* staged_rnq, hipStream_t st, int group_rows = 0,
                          int64_t group_stride = 0);

}  // namespace cslam
