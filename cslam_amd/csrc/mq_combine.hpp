// The fold of the multi-tile scan's K-split pass (scan_mq.hip, MODE_PART), shared by
//   * k_mq_combine (scan_mq.hip): fold + epilogue, one composite per query and strip (or the raw dots), and
//   * k_mq_combine_final<Final> (this header; round 5): fold + epilogue + the FINAL level of a best-match search in
//     the same launch - what k_best_pair (knn.hip) / k_records_from_candidates (shard.hip) did in a launch of their
//     own behind the combine: ~4-6 us of kernel and ~6 us of kernel-to-kernel gap per step of BASELINE configs[3]'s
//     shape (8 x 10 gathered descriptors against a 25 000-row pool: cslam/loop_closure_detection_node.py:34,
//     cslam/loop_closure_sparse_matching.py:58-76).
// The K-split pass left, per strip and 512-float chunk, the chunk's partial products ([strip][chunk][tile][lane]
// float4 in the MFMA accumulator layout); tot = ((0 + c0) + c1) + ... in chunk order is exactly the fold of
// k_scan_mq's registers - same bits whoever computed the chunks.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "common.hpp"
#include "composite.hpp"
#include "scan_mq.hpp"

namespace cslam {

// tot of one (strip, tile) for this lane (4 rows x query lane & 15); src = part + ((strip nchunks) tiles + t) 64 + lane.
// The <= 16 chunk partials are requested together, folded in order.
__device__ inline f32x4 mq_fold_partials(const f32x4* __restrict__ src, int tiles, int nchunks) {
    f32x4 part[16];
#pragma unroll
    for (int c = 0; c < 16; ++c)
        part[c] = c < nchunks ? src[(int64_t)c * tiles * 64] : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 tot{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 16; ++c)
        if (c < nchunks) tot = tot + part[c];
    for (int c = 16; c < nchunks; ++c) tot = tot + src[(int64_t)c * tiles * 64];   // dp > 8192
    return tot;
}

// the lane's best composite(s) of its four rows of a strip: rows before / from the split row apart
struct MqBest2 {
    uint64_t a, c;
};
__device__ inline MqBest2 mq_strip_composites(const MqArgs& A, const f32x4& tot, int64_t lo, int64_t hi, int g, float rq) {
    const int rows = (int)(hi - lo);
    float rn[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int64_t row = lo + 4 * g + e;
        rn[e] = A.rnx[row < A.n ? row : A.n - 1];
    }
    const bool two = A.split < hi;
    const uint32_t rbase = A.row0 + (uint32_t)lo;
    const int srel = A.split > lo ? (int)((A.split < hi ? A.split : hi) - lo) : 0;
    uint64_t best = 0, best_c = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int rel = 4 * g + e;
        const float sim = (tot[e] * rn[e]) * rq;
        uint64_t c = make_comp(sim, rbase + (uint32_t)rel);
        c = rel < rows ? c : 0;
        if (two) {
            const uint64_t ca = rel < srel ? c : 0, cc = rel < srel ? 0 : c;
            best = ca > best ? ca : best;
            best_c = cc > best_c ? cc : best_c;
        } else {
            best = c > best ? c : best;
        }
    }
    return MqBest2{best, best_c};
}

// Scratch of the fused final level.  CSLAM_MQ_FINAL_VARIANT (compile time; A/B in profiles/r05_mq_final_ab.txt):
//   0  one agent-scope atomic maximum per query and block on ONE slot per query
//   1  ... spread over kMqFinalSlots slots per query (block's strip group modulo), the last block folds the slots
//   2  no atomics: every block leaves its maxima in an array [window][query][group]; the last block reduces it
#ifndef CSLAM_MQ_FINAL_VARIANT
#define CSLAM_MQ_FINAL_VARIANT 2
#endif
constexpr int kMqFinalSlots = CSLAM_MQ_FINAL_VARIANT == 1 ? 8 : 1;
constexpr int kMqFinalQ = 96;        // queries a pass takes at most, rounded up (16 x mq_max_tiles() = 80)
struct MqFinalScratch {
    uint64_t* best_a;    // variants 0 / 1: [slots][kMqFinalQ], zero between launches; variant 2: [kMqFinalQ][groups]
    uint64_t* best_c;
    uint32_t* ticket;    // zero between launches
    int64_t groups;      // strip groups of the launch (variant 2: row length of best_a / best_c)
};
constexpr int kMqFinalStrips = 4;   // strips a block folds for its tile: one per wave (all 7 800 waves of BASELINE
                                     // configs[3]'s shape resident at once: the fold is two dependent memory latencies)
inline size_t mq_final_scratch_bytes(int64_t strips) {
    const int64_t groups = (strips + kMqFinalStrips - 1) / kMqFinalStrips;
    const size_t arrays = CSLAM_MQ_FINAL_VARIANT == 2 ? (size_t)2 * kMqFinalQ * groups : (size_t)2 * kMqFinalSlots * kMqFinalQ;
    return arrays * sizeof(uint64_t) + 256;
}

// Fold + epilogue + final level.  Block = (group of kMqFinalStrips strips, query tile t); wave w folds strip w of the
// group for the 16 queries of the tile (as k_mq_combine does), the block's four waves meet in the LDS and leave ONE
// maximum per query and block (order-free: composites are totally ordered, ties -> lowest row), then a ticket: the
// last block of the launch reduces what the blocks left, hands every query's winners to `fin(q, best_a, best_c)`,
// cleans the scratch for the next launch and - host-pointer API - writes the completion word behind the results.
// Final = a POD functor with `__device__ void operator()(int, uint64_t, uint64_t) const`; the kernel is instantiated
// where the functor lives (knn.hip: rows + similarities; shard.hip: the inter-robot match records).
template <class Final>
__global__ __launch_bounds__(256) void k_mq_combine_final(const MqArgs A, int tiles, int nchunks, MqFinalScratch fs,
                                                          Final fin, uint32_t* done, uint32_t seq) {
    __shared__ uint64_t sh_a[4][16], sh_c[4][16];
    __shared__ int last_s;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 15, g = lane >> 4;
    const int t = (int)(blockIdx.x % (unsigned)tiles);
    const int64_t sg = blockIdx.x / (unsigned)tiles;
    const int64_t S = (A.n + 15) >> 4;
    const bool split = A.split != INT64_MAX;
    uint64_t best = 0, best_c = 0;
    const int64_t strip = sg * kMqFinalStrips + wave;
    if (strip < S) {
        const int64_t lo = strip << 4;
        const int64_t hi = lo + 16 < A.n ? lo + 16 : A.n;
        const f32x4 tot = mq_fold_partials(A.part + ((strip * nchunks) * tiles + t) * 64 + lane, tiles, nchunks);
        const MqBest2 b = mq_strip_composites(A, tot, lo, hi, g, A.rnq[16 * t + r]);
        best = b.a;
        best_c = b.c;
    }
    uint64_t o = __shfl_xor(best, 16, 64);
    best = o > best ? o : best;
    o = __shfl_xor(best, 32, 64);
    best = o > best ? o : best;
    if (split) {
        o = __shfl_xor(best_c, 16, 64);
        best_c = o > best_c ? o : best_c;
        o = __shfl_xor(best_c, 32, 64);
        best_c = o > best_c ? o : best_c;
    }
    if (g == 0) {
        sh_a[wave][r] = best;
        sh_c[wave][r] = best_c;
    }
    __syncthreads();
    if (tid < 16 && 16 * t + tid < A.m_valid) {
        uint64_t a = sh_a[0][tid], c = sh_c[0][tid];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            a = sh_a[w][tid] > a ? sh_a[w][tid] : a;
            c = sh_c[w][tid] > c ? sh_c[w][tid] : c;
        }
        const int q = 16 * t + tid;
#if CSLAM_MQ_FINAL_VARIANT == 2
        fs.best_a[(int64_t)q * fs.groups + sg] = a;
        if (split) fs.best_c[(int64_t)q * fs.groups + sg] = c;
#else
        const int slot = (int)(sg % kMqFinalSlots) * kMqFinalQ;
        if (a) __hip_atomic_fetch_max(&fs.best_a[slot + q], a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (split && c) __hip_atomic_fetch_max(&fs.best_c[slot + q], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    }
    // the last block to get here sees what every block left: every thread's stores / atomics are performed (fence)
    // before its block takes a ticket (the pattern of k_small_search, knn.hip)
    __threadfence();
    __syncthreads();
    if (tid == 0) {
        const uint32_t tk = __hip_atomic_fetch_add(fs.ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        last_s = tk == gridDim.x - 1 ? 1 : 0;
    }
    __syncthreads();
    if (!last_s) return;
    __threadfence();   // acquire: this block's caches may hold lines of an earlier launch
#if CSLAM_MQ_FINAL_VARIANT == 2
    // wave w reduces the queries w, w + 4, ...: 64 lanes stride the groups of a query's row, wave maximum
    for (int q = wave; q < A.m_valid; q += 4) {
        uint64_t a = 0, c = 0;
        for (int64_t i = lane; i < fs.groups; i += 64) {
            const uint64_t va = fs.best_a[(int64_t)q * fs.groups + i];
            a = va > a ? va : a;
            if (split) {
                const uint64_t vc = fs.best_c[(int64_t)q * fs.groups + i];
                c = vc > c ? vc : c;
            }
        }
        a = wave_max_u64(a);
        if (split) c = wave_max_u64(c);
        if (lane == 0) fin(q, a, c);
    }
#else
    for (int q = tid; q < A.m_valid; q += 256) {
        uint64_t a = 0, c = 0;
        for (int sl = 0; sl < kMqFinalSlots; ++sl) {
            const uint64_t va = __hip_atomic_load(&fs.best_a[sl * kMqFinalQ + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            a = va > a ? va : a;
            __hip_atomic_store(&fs.best_a[sl * kMqFinalQ + q], (uint64_t)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (split) {
                const uint64_t vc = __hip_atomic_load(&fs.best_c[sl * kMqFinalQ + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                c = vc > c ? vc : c;
                __hip_atomic_store(&fs.best_c[sl * kMqFinalQ + q], (uint64_t)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        fin(q, a, c);
    }
#endif
    if (tid == 0) __hip_atomic_store(fs.ticket, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    if (done != nullptr) {   // results first, system-wide, then the call's number (signal_done, topk_parts.hpp)
        __threadfence_system();
        __syncthreads();
        if (tid == 0) __hip_atomic_store(done, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

template <class Final>
int launch_mq_combine_final(const MqArgs& a, int tiles, int nchunks, const MqFinalScratch& fs, const Final& fin,
                            uint32_t* done, uint32_t seq, hipStream_t st) {
    const int64_t strips = (a.n + 15) >> 4;
    const int64_t groups = (strips + kMqFinalStrips - 1) / kMqFinalStrips;
    if (strips <= 0 || tiles < 1 || a.part == nullptr || fs.ticket == nullptr || fs.groups != groups ||
        a.m_valid > kMqFinalQ)
        return fail(CSLAM_E_INVALID, "mq_combine_final: bad launch");
    const unsigned grid = (unsigned)(groups * tiles);
    hipLaunchKernelGGL((k_mq_combine_final<Final>), dim3(grid), dim3(256), 0, st, a, tiles, nchunks, fs, fin, done, seq);
    CSLAM_HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}

// How a caller of the pass asks for the fused final level: `launch` is called INSTEAD of launch_mq_combine when the
// whole window went through the K-split launch in best-match mode without dots (the plan of BASELINE configs[3]'s
// shape); the pass reports whether it was.  A plain function pointer + context: the pass lives in knn.hip, the
// functor of the sharded step in shard.hip.
struct MqFinalHook {
    int (*launch)(const MqArgs& a, int tiles, int nchunks, const MqFinalScratch& fs, void* ctx, hipStream_t st);
    void* ctx;
};

}  // namespace cslam
