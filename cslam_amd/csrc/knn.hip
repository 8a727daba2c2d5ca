// knn.hip — descriptor pool + brute-force cosine kNN for gfx950 (MI355X).
//
// Replaces cslam/nns_matching.py:5-78 (NearestNeighborsMatching): `add_item` → cslam_pool_add*,
// `search`/`search_best` → cslam_pool_search*.  Numeric contract: DESIGN.md §3 (identical to
// oracle/knn_oracle.c): dot(x,q) is ONE fmaf chain in v_mfma_f32_16x16x4_f32 operand order,
// ||x||^2 is a wave64 butterfly sum cached at add time, sim = (dot * rn(x)) * rn(q).
//
// Kernels
//   k_append_rows   add_item: copy + zero-pad + cached 1/||row|| (wave64 order) in one launch
//   k_stage_queries pads the query block to [16*T][dp] in HBM (L2-resident) + 1/||q||
//   k_scan<T,U>     HBM-bound streaming scan: every wave owns a contiguous, byte-balanced row
//                   range, walks it in 16-row strips, A operand straight from HBM to VGPRs
//                   (U float4 deep, double-buffered), B operand (queries) from L2,
//                   fp32 MFMA 16x16x4, epilogue scales by the cached inverse norms
//   k_select<...>   exact top-k (k <= 128): per-part k rounds of block arg-max on
//                   (similarity desc, row asc) composites, wave-shuffle + LDS reduction
//   k_bitonic_step  any k: full bitonic sort of the composites in HBM (rare path)
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <vector>

#include "common.hpp"
#include "composite.hpp"
#include "dma_scan.hpp"
#include "scan_mq.hpp"
#include "mq_plan.hpp"
#include "topk_parts.hpp"
#include "probe_hooks.hpp"

using namespace cslam;

namespace {

constexpr int kWave = 64;
constexpr int kChunkKb = 32;       // fold the MFMA partial into the total every 32 k-blocks
constexpr int kSelectPart = 4096;  // elements per select block (256 threads x 16)
constexpr int kSelectMaxK = 128;   // above this the bitonic path is used
constexpr int kMaxQueriesPerPass = 64;
constexpr int64_t kQueryChunk = 1024;  // most queries per similarity-workspace chunk
// Queries per chunk for the current row window: as many as a 2-GiB dots workspace holds, between
// 256 and kQueryChunk, in whole 128-query GEMM tiles.  (Round 1 chunked at 256: a 100k-row GEMM
// launch then had 782 tiles for 512 tile slots — 3.05 waves of tiles, a quarter of the chip idle
// in the last one; at 1024 queries it is 12.2 waves.)
inline int64_t query_chunk(int64_t sims_stride) {
    int64_t c = ((int64_t)1 << 31) / (sims_stride * (int64_t)sizeof(float));
    c = c / 128 * 128;
    return c < 256 ? 256 : (c > kQueryChunk ? kQueryChunk : c);
}

__host__ __device__ inline int64_t pad16(int64_t d) { return (d + 15) / 16 * 16; }

// ---------------------------------------------------------------------------------------------
// 1 / max(||row||, 1e-8); the sum of squares arrives in wave64 order (oracle/knn_oracle.c:
// sumsq_wave_order): lane l takes floats 256t + 4l + c, xor butterfly 32 .. 1
// ---------------------------------------------------------------------------------------------
__device__ inline float inv_norm_from_sumsq(float ss) {
    // plain sqrtf and '/' are IEEE correctly rounded under hipcc's default
    // -fhip-fp32-correctly-rounded-divide-sqrt (the __fsqrt_rn/__fdiv_rn intrinsics are NOT).
    float nrm = __builtin_sqrtf(ss);
    if (!(nrm > 1e-8f)) nrm = 1e-8f;
    return 1.0f / nrm;
}

__device__ inline float ld_or_zero(const float* s, int k, int dim) { return k < dim ? s[k] : 0.0f; }

// append = copy + pad + cached inverse norm in ONE launch: one wave per row, lane l moves floats
// 256t + 4l .. +3 — exactly the lane assignment of the norm's wave64 order (wave_sumsq), so the
// sum of squares is accumulated from the registers that are being stored.  16-byte loads when
// the source allows it (vec_ok), 8 loads in flight per lane.
__global__ __launch_bounds__(256) void k_append_rows(const float* __restrict__ src,
                                                     int64_t src_stride, int dim,
                                                     float* __restrict__ dst, int64_t dst_stride,
                                                     int dp, int64_t first, int64_t count,
                                                     float* __restrict__ inv_norm, int vec_ok) {
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (w >= count) return;
    const float* s = src + w * src_stride;
    float* d = dst + (first + w) * dst_stride;
    float ss = 0.0f;
    for (int base0 = 4 * lane; base0 < dp; base0 += 256 * 8) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = base0 + 256 * u;
            if (k < dp) {
                if (vec_ok && k + 3 < dim)
                    v[u] = *reinterpret_cast<const f32x4*>(s + k);
                else
                    v[u] = f32x4{ld_or_zero(s, k, dim), ld_or_zero(s, k + 1, dim),
                                 ld_or_zero(s, k + 2, dim), ld_or_zero(s, k + 3, dim)};
            } else {
                v[u] = f32x4{0.f, 0.f, 0.f, 0.f};  // fmaf(0,0,ss) == ss: no-op past the row
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = base0 + 256 * u;
            if (k < dp) *reinterpret_cast<f32x4*>(d + k) = v[u];
            ss = __builtin_fmaf(v[u].x, v[u].x, ss);
            ss = __builtin_fmaf(v[u].y, v[u].y, ss);
            ss = __builtin_fmaf(v[u].z, v[u].z, ss);
            ss = __builtin_fmaf(v[u].w, v[u].w, ss);
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) ss = ss + __shfl_xor(ss, off, kWave);
    if (lane == 0) inv_norm[first + w] = inv_norm_from_sumsq(ss);
}

// 1 / max(||q||, 1e-8) of one query row (dim floats at s, taken as zero padded to dp) by ONE wave:
// ||q||^2 in wave64 order straight from the source values (loads batched 8 deep: the chain order is
// per lane, not per load).  Result in every lane.
__device__ inline float query_inv_norm_wave64(const float* __restrict__ s, int dim, int dp, int vec_ok, int lane) {
    float ss = 0.0f;
    for (int base0 = 4 * lane; base0 < dp; base0 += 256 * 8) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = base0 + 256 * u;
            if (k < dp) {
                if (vec_ok && k + 3 < dim)
                    v[u] = *reinterpret_cast<const f32x4*>(s + k);
                else
                    v[u] = f32x4{ld_or_zero(s, k, dim), ld_or_zero(s, k + 1, dim),
                                 ld_or_zero(s, k + 2, dim), ld_or_zero(s, k + 3, dim)};
            } else {
                v[u] = f32x4{0.f, 0.f, 0.f, 0.f};  // fmaf(0,0,ss) == ss: no-op past the row
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            ss = __builtin_fmaf(v[u].x, v[u].x, ss);
            ss = __builtin_fmaf(v[u].y, v[u].y, ss);
            ss = __builtin_fmaf(v[u].z, v[u].z, ss);
            ss = __builtin_fmaf(v[u].w, v[u].w, ss);
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) ss = ss + __shfl_xor(ss, off, kWave);
    return inv_norm_from_sumsq(ss);
}

// queries → staging [mq_pad][dp] (zero padded, rows >= m are zero) and rnq[mq_pad].
// One 256-thread block per query row; wave 0 also computes ||q||^2 in wave64 order straight
// from the source values (loads batched 8 deep: the chain order is per lane, not per load).

__global__ __launch_bounds__(256) void k_stage_queries(const float* __restrict__ q,
                                                       int64_t q_stride, int dim, int m,
                                                       int mq_pad, int dp,
                                                       float* __restrict__ stage,
                                                       float* __restrict__ rnq, int vec_ok,
                                                       int64_t copy_stride, int group_rows,
                                                       int64_t group_stride) {
    const int w = blockIdx.x;
    const int tid = threadIdx.x;
    if (w >= mq_pad) return;
    float* d = stage + (int64_t)blockIdx.y * copy_stride + (int64_t)w * dp;
    if (w >= m) {
        for (int k = tid * 4; k < dp; k += 1024)
            *reinterpret_cast<f32x4*>(d + k) = f32x4{0.f, 0.f, 0.f, 0.f};
        if (tid == 0 && blockIdx.y == 0) rnq[w] = 0.0f;
        return;
    }
    // (group_rows > 0: the rows come in groups of that many, group_stride floats apart - the sharded step's
    // exchange buffer, one part per rank: rows, then a header - and are staged straight from there)
    const float* s = group_rows > 0 ? q + (int64_t)(w / group_rows) * group_stride + (int64_t)(w % group_rows) * q_stride
                                    : q + (int64_t)w * q_stride;
    for (int k = tid * 4; k < dp; k += 1024) {
        f32x4 v;
        if (vec_ok && k + 3 < dim)
            v = *reinterpret_cast<const f32x4*>(s + k);
        else
            v = f32x4{ld_or_zero(s, k, dim), ld_or_zero(s, k + 1, dim), ld_or_zero(s, k + 2, dim),
                      ld_or_zero(s, k + 3, dim)};
        *reinterpret_cast<f32x4*>(d + k) = v;
    }
    if (tid < 64 && blockIdx.y == 0) {
        const float rn = query_inv_norm_wave64(s, dim, dp, vec_ok, tid);
        if (tid == 0) rnq[w] = rn;
    }
}

// ---------------------------------------------------------------------------------------------
// k_scan<T,S,U>: raw dot products dots[q][row] of 16*T staged queries against rows [0,n)
//   grid: W waves (4 per block); wave w owns the byte-balanced row range [w*n/W, (w+1)*n/W)
//   and walks it in groups of S 16-row strips (S*T independent MFMA accumulators).
//   lane l: r = l & 15 (row in strip / query column), g = l >> 4 (k slot)
//   A_s = X[row(s,r)][16kb + 4g .. +3], B_t = Q[col 16t + r][16kb + 4g .. +3]; MFMA j consumes
//   component j → chain order inside a 16-block: k = 16kb + 4g + j, j outer, g inner
//   (DESIGN.md §3); the chain restarts every 512 floats (32 k-blocks) and the chunk partials
//   are folded into a running total by plain fp32 adds — identical for every T/S/U.
//   Both operands run through a U-deep register ring that is refilled right after use and
//   keeps running ACROSS group boundaries, so a wave always has U*S KiB of X in flight.  The
//   loop body is branch-free apart from the group epilogue (stores only), which lets the
//   compiler emit counted s_waitcnt vmcnt(N) instead of draining the queue.
//   Why S strips share one B fragment: all waves of the chip walk the (L2-resident) query
//   block in near lockstep, which hot-spots one L2 channel per XCD; at S=1 the query stream
//   (1 byte per X byte) capped the scan at ~3.8 TB/s.  S=4 cuts it to 1/4.
//   The cosine scaling (x 1/||x|| x 1/||q||) is applied by the consumer (k_select/k_scale):
//   a load in the epilogue would sit behind the ring's HBM loads (loads return in order).
//   Requires nkb % U == 0 (the host picks U).
// ---------------------------------------------------------------------------------------------
template <int T, int S, int U, int G = 1>
__global__ __launch_bounds__(256) void k_scan(const float* __restrict__ X, int64_t xs, int64_t n,
                                              int dp, const float* __restrict__ Q, int m_valid,
                                              float* __restrict__ dots, int64_t dots_stride,
                                              int64_t n_waves, int q_copies,
                                              int64_t q_copy_stride) {
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= n_waves) return;
    // replicas of the staged query block at different addresses spread the lockstep query
    // reads of all waves over more L2 channels
    Q += (w % q_copies) * q_copy_stride;
    const int lane = threadIdx.x & 63;
    const int r = lane & 15;
    const int g = lane >> 4;
    const int64_t lo = (w * n) / n_waves;
    const int64_t hi = ((w + 1) * n) / n_waves;
    const int ipk = (dp >> 4) / U;  // ring turns per group
    const int64_t ngroups = (hi - lo + 16 * S - 1) / (16 * S);
    const int64_t total = ngroups * ipk;

    const f32x4* qp[T];
#pragma unroll
    for (int t = 0; t < T; ++t)
        qp[t] = reinterpret_cast<const f32x4*>(Q + (int64_t)(t * 16 + r) * dp + 4 * g);

    auto rowptr = [&](int64_t grp, int s) {
        int64_t row = lo + grp * (16 * S) + 16 * s + r;
        if (row >= hi) row = hi - 1;  // clamp: duplicates hit L1, their results are not stored
        return reinterpret_cast<const f32x4*>(X + row * xs + 4 * g);
    };

    f32x4 a[U][S];
    f32x4 b[U][T];
    f32x4 acc[S][T];  // partial of the current 512-float chunk (MFMA accumulator)
    f32x4 tot[S][T];  // running total of the folded chunks
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
        for (int t = 0; t < T; ++t) {
            acc[s][t] = f32x4{0.f, 0.f, 0.f, 0.f};
            tot[s][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    const int fold_turns = kChunkKb / U;  // U divides 32
    auto fold = [&]() {
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int t = 0; t < T; ++t) {
                tot[s][t] = tot[s][t] + acc[s][t];
                acc[s][t] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
    };

    auto store_group = [&](int64_t grp) {
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int64_t r0 = lo + grp * (16 * S) + 16 * s + 4 * g;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int q = t * 16 + r;
                if (q < m_valid) {
                    float* out = dots + (int64_t)q * dots_stride + r0;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (r0 + e < hi) out[e] = tot[s][t][e];
                }
                tot[s][t] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    auto mfma_slot = [&](int u) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int t = 0; t < T; ++t)
                    acc[s][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][s][j], b[u][t][j],
                                                                     acc[s][t], 0, 0, 0);
    };

    {
        const f32x4* x0[S];
#pragma unroll
        for (int s = 0; s < S; ++s) x0[s] = rowptr(0, s);
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int s = 0; s < S; ++s) a[u][s] = x0[s][u * 4];
#pragma unroll
            for (int t = 0; t < T; ++t) b[u][t] = qp[t][u * 4];
        }
    }
    int64_t grp = 0;  // group of the ring's current contents
    int turn = 0;     // ring turn inside the group
    for (int64_t it = 0; it + 1 < total; ++it) {
        int nturn = turn + 1;
        int64_t ngrp = grp;
        if (nturn == ipk) {
            nturn = 0;
            ngrp = grp + 1;
        }
        const f32x4* xn[S];
#pragma unroll
        for (int s = 0; s < S; ++s) xn[s] = rowptr(ngrp, s) + (int64_t)nturn * U * 4;
        const int qn = nturn * U * 4;
        // consume G slots, then refill those G slots (G = 2: the two 64-B halves of every
        // 128-B line are requested back to back)
#pragma unroll
        for (int u0 = 0; u0 < U; u0 += G) {
#pragma unroll
            for (int u = u0; u < u0 + G; ++u) mfma_slot(u);
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int u = u0; u < u0 + G; ++u) a[u][s] = xn[s][u * 4];
#pragma unroll
            for (int u = u0; u < u0 + G; ++u)
#pragma unroll
                for (int t = 0; t < T; ++t) b[u][t] = qp[t][qn + u * 4];
        }
        if (nturn == 0 || nturn % fold_turns == 0) fold();
        if (nturn == 0) store_group(grp);
        grp = ngrp;
        turn = nturn;
    }
    // last ring turn: nothing left to prefetch
#pragma unroll
    for (int u = 0; u < U; ++u) mfma_slot(u);
    fold();
    store_group(grp);
}

// ---------------------------------------------------------------------------------------------
// k_scan_dma<NS,NW,NST,AUX>: the HBM-bound scan (<= 16 queries) with X streamed through LDS-DMA.
//   Why: the MFMA A-operand shape (16 rows x 64 B per load instruction) touches 16 half lines
//   per instruction and tops out at 5.1-5.3 TB/s in k_scan; a linear 1-KiB-per-instruction
//   read streams at 6.0-6.3 TB/s on this part (profiles/r02_dma_scan_probe.txt).  Here every
//   global_load_lds_dwordx4 moves ONE row's 256-float slice (8 whole lines) straight into LDS,
//   and the wave reads the A fragments back with ds_read_b128 (row pitch 1024+32 B: the 16-lane
//   groups of ds_read_b128 hit 16 different 16-B slots, conflict free).
//   Structure: NW waves per block, one block per CU, NO barriers: every wave owns a private
//   ring of NST stages (stage = 16 rows x 256 floats) and only its own counted vmcnt orders its
//   ds_reads behind its DMAs.  A stage is read into 64 VGPRs at once and its slot re-issued
//   immediately, so a wave keeps NST-1..NST stages (16..32 KiB at NST=2) in flight.
//   Work order is k-slice-major inside a row panel (<= NS strips): the B fragments (queries) of
//   a 256-float slice sit in 64 VGPRs and serve all NS strips, so the L2 query stream is 1/NS
//   of the X stream; the next slice's fragments are prefetched by inline-asm global loads that
//   the compiler's waitcnt pass does not see (an ordinary load beside LDS-DMAs makes hipcc
//   drain vmcnt(0)), counted by hand together with the DMA pieces.
//   Numerics: per (row, query) the MFMA chain visits k in the canonical order, accumulators
//   are carried per strip across the two slices of a 512-float chunk and folded exactly like
//   k_scan -> identical bits (tests/test_knn_gpu.py).
//   Requires dp % 64 == 0 (slice width KW = 256 / 128 / 64, see DmaGeom below; the pool pads rows of
//   more than 16 floats to a multiple of 64, shorter ones take k_scan).
// ---------------------------------------------------------------------------------------------
constexpr int kDmaPieceB = 1024 + 32;          // bytes per row piece in LDS
constexpr int kDmaStageB = 16 * kDmaPieceB;    // 16 rows x 256 floats
// rows from which the DMA scan is the default.  It wins at every size (16-query search_best, us:
// 64 rows x 4096-d 47 vs 175 for the register ring, 32k rows 103 vs 281; 64 rows x 512-d 16 vs 33,
// tools/small_n_sweep.py) — the ring kernel stays for descriptors of <= 16 floats and for 17-64 queries.
constexpr int64_t kDmaMinRows = 1;

// NB B fragments (one KW-float slice of the lane's query column, NB = KW / 16) by inline-asm
// loads; the registers may only be read behind a counted wait followed by tie_b().  What the compiler
// does with them in between is checked on the emitted assembly at every build (Makefile: asm-check)
template <int NB>
__device__ inline void load_b(f32x4 (&b)[NB], const float* qrow) {
#define CSLAM_LB(I)                                                                     \
    if (I < NB)                                                                         \
        asm volatile("global_load_dwordx4 %0, %1, off offset:" #I "*64" : "=v"(b[I < NB ? I : 0]) : "v"(qrow) : "memory");
    CSLAM_LB(0) CSLAM_LB(1) CSLAM_LB(2) CSLAM_LB(3) CSLAM_LB(4) CSLAM_LB(5) CSLAM_LB(6) CSLAM_LB(7)
    CSLAM_LB(8) CSLAM_LB(9) CSLAM_LB(10) CSLAM_LB(11) CSLAM_LB(12) CSLAM_LB(13) CSLAM_LB(14) CSLAM_LB(15)
#undef CSLAM_LB
}
// behind the counted wait that covers the loads: tells the compiler the registers changed
template <int NB>
__device__ inline void tie_b(f32x4 (&b)[NB]) {
#pragma unroll
    for (int i = 0; i < NB; i += 4)
        asm volatile("" : "+v"(b[i]), "+v"(b[i + 1]), "+v"(b[i + 2]), "+v"(b[i + 3]));
}
// prologue: everything in flight has landed; wait and tie of the first fragments in ONE statement
template <int NB>
__device__ inline void wait0_tie_b(f32x4 (&b)[NB]) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) : : "memory");
#pragma unroll
    for (int i = 4; i < NB; i += 4)
        asm volatile("" : "+v"(b[i]), "+v"(b[i + 1]), "+v"(b[i + 2]), "+v"(b[i + 3]));
}
__device__ inline void load_b16(f32x4 (&b)[16], const float* qrow) { load_b<16>(b, qrow); }

struct DmaCursor {   // position of a stage in a wave's work list
    int64_t lo;      // first row of the panel
    int rows;        // rows of the panel (<= 16 * NS)
    int ns;          // stages per slice of the panel
    int p;           // panel index
    int kc;          // 256-float slice
    int s;           // strip inside the panel
};

//   BEST = true (search_best, k = 1): nothing but 16 x n_panels composites leaves the kernel.
//   1/||q|| is accumulated from the B fragments in the oracle's wave64 order (scan lane (r, g)
//   holds, for query r, exactly the floats of norm lanes 4i + g, i = 0..15: 16 fmaf chains per
//   lane, then the xor butterfly 32..4 inside the lane's registers and 2, 1 across g), the
//   epilogue applies (dot * rn(x)) * rn(q) and keeps the panel's best (similarity desc, row asc)
//   composite per query; the queries may be read in place (q_rows = m, any stride) — no
//   staging kernel, no dots round trip.  Every wave keeps a running best over its panels and
//   leaves ONE composite per query, a small select launch reduces the <= 1024 of them.
//   Split window (cand_c != nullptr): rows < split and rows >= split (window-relative) are kept
//   apart — two composites per query and wave: the sharded step asks for the best OLD row and
//   the best row of the block it has just appended in the same pass (csrc/shard.hip).
//   Measured and rejected (round 2, commit 775bacb holds the code): finishing inside the launch — waves
//   fold through LDS, block tickets, the last block's wave reduces and writes the results.  The
//   dependent chain at the tail (write-through stores complete → ticket atomic → invalidate →
//   loads, ~1.5 us of memory-side latency each) cost 7.4 us against 4.5 us for the select launch
//   behind the scan (265.9 vs 263.0 us per step); with a release fence per block (buffer_wbl2)
//   +8 us more.
//   Slice width KW (template): 256 floats = one row per DMA instruction, as described above.
//   KW = 128 / 64 (descriptor lengths that are a multiple of 128 / 64 but not of 256: the 128-d
//   descriptors of the reference's own configuration): one DMA instruction moves R = 256 / KW
//   ROWS x KW floats (lanes [k 64/R, (k+1) 64/R) fetch row k of the piece), a stage is 16 R rows
//   = R row groups of 16, a panel is still 128 rows = 8 row groups with one accumulator each.
//   The 16-B segments of row k of a piece are XOR-k swizzled on the SOURCE side (LDS position j
//   holds global segment j ^ k) and the piece pitch is 1024 + 32 B (R <= 2) or 1024 + 64 B
//   (R = 4), so that the 16 lanes of a ds_read_b128 pass — 16 rows at one k offset — hit 16
//   different 16-B slots.  Same chain order, same folds: same bits.
template <int NS, int NW, int NST, int AUX, bool BEST, int KW = 256, bool DOTS_TOO = false>
__global__ __launch_bounds__(NW * 64) void k_scan_dma(const float* __restrict__ X, int64_t xs,
                                                      int64_t n, int dp,
                                                      const float* __restrict__ Q, int64_t q_stride,
                                                      int q_rows, int m_valid,
                                                      float* __restrict__ dots,
                                                      int64_t dots_stride, int n_waves,
                                                      const float* __restrict__ rnx, uint32_t row0,
                                                      uint64_t* __restrict__ cand,
                                                      int64_t cand_stride,
                                                      float* __restrict__ rnq_out,
                                                      int64_t split, uint64_t* __restrict__ cand_c,
                                                      int even) {
    using GM = DmaGeom<KW>;
    constexpr int R = GM::R, NB = GM::NB, kPiece = GM::kPiece, kStage = GM::kStage;
    static_assert(NS == 8 && NS % R == 0, "a panel is 8 row groups");
    CSLAM_DMA_HOOK_KERNEL_BEGIN
    extern __shared__ __attribute__((aligned(16))) char dma_lds[];
    // the query norms are computed here when the epilogue needs them (BEST) or when the caller
    // skipped the staging kernel and wants them for the select (rnq_out)
    const bool norms = BEST || rnq_out != nullptr;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // scalar: uniform branches
    const int lane = threadIdx.x & 63;
    const int r = lane & 15, g = lane >> 4;
    // wave number -> rows: block-major, or (even & 2) dealt round-robin over the blocks so that a pool
    // of fewer strips than wave slots still puts waves on EVERY CU (a CU streams ~25-30 GB/s however
    // many of its waves ask: 625 strips as 157 full blocks ran at 3.5 TB/s)
    const int w = (even & 2) ? wave * (int)gridDim.x + (int)blockIdx.x : blockIdx.x * NW + wave;
    char* ring = dma_lds + wave * NST * kStage;
    // BEST: 1/||x|| of the wave's panel rows (<= 128 floats), fetched by LDS-DMA - a load without a
    // VGPR destination leaves nothing the compiler could copy or recycle while it is in flight
    char* rvbuf = dma_lds + NW * NST * kStage + wave * 512;
    const int nkc = dp / KW;
    if (w >= n_waves) return;
    // wave w owns the rows [wlo, whi) of an even split of the window over the waves and walks
    // them in p_end panels of <= 128 rows (8 strips)
    const int64_t wq = n / n_waves, wrem = n - wq * n_waves;
    const int64_t wlo = (int64_t)w * wq + (w < wrem ? w : wrem);
    const int64_t whi = wlo + wq + (w < wrem ? 1 : 0);
    const int p_begin = 0;
    const int p_end = (int)((whi - wlo + 16 * NS - 1) / (16 * NS));
    if (p_end == 0) return;
    // even = 1 (default): p_end equal panels; 0: panels of 128 rows, the remainder last
    const int pq = (int)((whi - wlo) / p_end), prem = (int)((whi - wlo) - (int64_t)pq * p_end);
    auto panel_lo = [&](int p) {
        const int64_t v = (even & 1) ? wlo + (int64_t)p * pq + (p < prem ? p : prem) : wlo + (int64_t)p * (16 * NS);
        return v < whi ? v : whi;
    };
    auto panel = [&](DmaCursor& c, int p) {
        c.p = p;
        c.lo = panel_lo(p);
        const int64_t hi = panel_lo(p + 1);
        c.rows = (int)(hi - c.lo);
        c.ns = (c.rows + 16 * R - 1) / (16 * R);   // stages per slice
        c.kc = 0;
        c.s = 0;
    };
    auto advance = [&](DmaCursor& c) {
        if (++c.s < c.ns) return;
        c.s = 0;
        if (++c.kc < nkc) return;
        panel(c, c.p + 1);  // may run to p_end: the caller checks before issuing
    };
    // one DMA instruction per piece (R rows) of the stage: the LDS image of a piece is linear
    // (base + 16 * lane).  R = 1: lane l moves bytes [16l, 16l+16) of the row's slice.  R > 1:
    // lane l moves segment (l mod 64/R) ^ k of row k = l / (64/R) of the piece; a row past the
    // end of the stage repeats the stage's last row (its results are never looked at).
    const int sub = R == 1 ? 0 : lane / (64 / R);
    const int seg = R == 1 ? lane : ((lane & (64 / R - 1)) ^ sub);
    const int64_t sub_off = (int64_t)sub * xs;
    auto issue = [&](const DmaCursor& c, char* slot) {
        int cnt = c.rows - 16 * R * c.s;   // valid rows of the stage (>= 1)
        cnt = cnt > 16 * R ? 16 * R : cnt;
        const int np = (cnt + R - 1) / R;  // pieces
        const float* src = X + (c.lo + 16 * R * c.s) * xs + c.kc * KW + seg * 4;
#pragma unroll
        for (int rr = 0; rr < 16; ++rr)
            if (rr < np) {
                const float* a = src + (int64_t)(R * rr) * xs;
                if (R > 1) {
                    if (R * rr + R <= cnt) {
                        a += sub_off;
                    } else {  // ragged last piece
                        const int k = R * rr + sub < cnt ? sub : cnt - 1 - R * rr;
                        a += (int64_t)k * xs;
                    }
                }
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)a,
                    (__attribute__((address_space(3))) void*)(slot + rr * kPiece), 16, 0, AUX);
            }
        return np;
    };
    // reader side: row r of row group h sits in piece (16 h + r) / R as row r % R
    const int rd_off = (r / R) * kPiece + (r % R) * (KW * 4) + ((g ^ (r % R)) * 16);

    // this lane's B fragment column (rows past q_rows repeat the last query: never stored)
    const float* qrow = Q + (int64_t)(r < q_rows ? r : q_rows - 1) * q_stride + 4 * g;
    f32x4 bc[NB], bn[NB];
    float ss[16];  // BEST: sum-of-squares chains of norm lanes 4i + g of query r
    float rnq = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) ss[i] = 0.f;
    DmaCursor ic;  // issue cursor: runs NST stages ahead of the consumer
    panel(ic, p_begin);
    int y[NST];  // vector-memory operations issued after the stage that sits in slot i
    int yq = 0;  // ... after the B loads in flight
#pragma unroll
    for (int i = 0; i < NST; ++i) y[i] = 0;
    auto bump = [&](int k) {
#pragma unroll
        for (int i = 0; i < NST; ++i) y[i] += k;
        yq += k;
    };
    load_b<NB>(bn, qrow);
    wait0_tie_b<NB>(bn);
    CSLAM_DMA_HOOK_QUERIES_LANDED
    // (Round 3 tried once more to request the first stages IN FRONT of this wait, for pools of a few
    // strips per wave - the time line of a 25 000 x 4096-d scan is 2.6-3.2 us until the queries have
    // landed, 2.5-3.5 more until the first stage has, then 66 us of streaming,
    // profiles/r03_dma_probe_clock.txt.  With the stage issue BETWEEN the asm loads and their wait it
    // bought 1 us of 78 (profiles/r03_size_sweep_early.txt) and faulted on its first small pool: the
    // window let the compiler recycle registers the loads still owed ("The inline-asm load hazard").
    // With the stages issued in front of the asm loads - no window - it is safe (121 tests) and buys
    // nothing (profiles/r03_size_sweep_early2.txt: 77.1-78.7 vs 75.8-79.2 us): not kept.)
#pragma unroll
    for (int i = 0; i < NST; ++i)
        if (ic.p < p_end) {
            const int c = issue(ic, ring + i * kStage);
            bump(c);
            y[i] = 0;
            advance(ic);
        }

    int slot = 0;  // ring slot of the stage the consumer is at
    uint64_t wbest_a = 0, wbest_c = 0;  // BEST: this wave's best composites (lanes g == 0)
    for (int p = p_begin; p < p_end; ++p) {
        const int64_t lo = panel_lo(p);
        const int64_t hi = panel_lo(p + 1);
        const int ns = (int)((hi - lo + 15) >> 4);   // row groups of the panel
        f32x4 acc[NS], tot[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            acc[s] = f32x4{0.f, 0.f, 0.f, 0.f};
            tot[s] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        for (int kc = 0; kc < nkc; ++kc) {
            // bn holds B(kc), landed and tied: hand it over and start B(kc+1) (slice 0 again for
            // the next panel); it is waited for at the end of this slice
#pragma unroll
            for (int i = 0; i < NB; ++i) bc[i] = bn[i];
            if (BEST) {
                // 1/||x|| of the panel's <= 128 rows for the epilogue: lane l moves the norms of rows
                // lo + l and lo + 64 + l (clamped into the window) into the wave's LDS area (two 256-B
                // LDS-DMA pieces).  Issued IN FRONT of the B loads, so the counted wait for those at
                // the end of the slice covers these two as well; the epilogue reads them back with
                // one ds_read_b128 per row group.
                // (Ordinary loads in the epilogue made hipcc wait vmcnt(0) 32 times per panel:
                // 32 dependent memory round trips with the ring drained.  Rounds 2-3 used inline-asm
                // loads into two VGPRs here: registers the compiler may not touch while the load is
                // in flight but does not know about - ADVICE r03.)
                int64_t i0 = lo + lane, i1 = lo + 64 + lane;
                i0 = i0 < n ? i0 : n - 1;
                i1 = i1 < n ? i1 : n - 1;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(rnx + i0),
                                                 (__attribute__((address_space(3))) void*)rvbuf, 4, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(rnx + i1),
                                                 (__attribute__((address_space(3))) void*)(rvbuf + 256), 4, 0, 0);
                bump(2);
            }
            load_b<NB>(bn, qrow + (kc + 1 < nkc ? kc + 1 : 0) * KW);
            bump(NB);
            yq = 0;
            if (norms && p == p_begin) {
                // fragment i of slice kc holds the floats of norm lane 4 (i + NB (kc mod R)) + g
                const int q4 = R == 1 ? 0 : (kc & (R - 1));
#pragma unroll
                for (int h = 0; h < R; ++h)
                    if (q4 == h) {
#pragma unroll
                        for (int i = 0; i < NB; ++i)
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                ss[h * NB + i] = __builtin_fmaf(bc[i][j], bc[i][j], ss[h * NB + i]);
                    }
            }
            const bool fold = ((kc + 1) & (GM::kFold - 1)) == 0 || kc + 1 == nkc;  // end of a 512-float chunk / of the row
#pragma unroll
            for (int s = 0; s < NS / R; ++s) {
                if (R * s < ns) {
                    char* sp = ring + slot * kStage;
                    int yc = y[0];
#pragma unroll
                    for (int i = 1; i < NST; ++i) yc = slot == i ? y[i] : yc;
                    wait_vm_dyn(yc);  // this stage's pieces have landed
                    CSLAM_DMA_HOOK_STAGE_LANDED
                    f32x4 a[16];
                    const char* ap = sp + rd_off;
#pragma unroll
                    for (int h = 0; h < R; ++h)
#pragma unroll
                        for (int i = 0; i < NB; ++i)
                            a[h * NB + i] = *reinterpret_cast<const f32x4*>(ap + h * (16 / R) * kPiece + i * 64);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    if (ic.p < p_end) {  // the slot is free again: refill it NST stages ahead
                        const int c = issue(ic, sp);
                        advance(ic);
                        bump(c);
#pragma unroll
                        for (int i = 0; i < NST; ++i) y[i] = slot == i ? 0 : y[i];
                    }
#pragma unroll
                    for (int h = 0; h < R; ++h) {
                        if (R == 1 || R * s + h < ns) {
#pragma unroll
                            for (int i = 0; i < NB; ++i)
#pragma unroll
                                for (int j = 0; j < 4; ++j)
                                    acc[R * s + h] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                        a[h * NB + i][j], bc[i][j], acc[R * s + h], 0, 0, 0);
                            if (fold) {
                                tot[R * s + h] = tot[R * s + h] + acc[R * s + h];
                                acc[R * s + h] = f32x4{0.f, 0.f, 0.f, 0.f};
                            }
                        }
                    }
                    slot = slot + 1 == NST ? 0 : slot + 1;
                }
            }
            wait_vm_dyn<true>(yq);   // the B loads (and the row norms in front of them) have landed
            tie_b<NB>(bn);
        }
        if (norms && p == p_begin) {
            // butterfly 32, 16, 8, 4 of the norm lanes = register index xor 8, 4, 2, 1;
            // then 2, 1 = g xor 2, 1 = lane xor 32, 16 (oracle/knn_oracle.c sumsq_wave_order)
#pragma unroll
            for (int off = 8; off >= 1; off >>= 1) {
                float t2[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) t2[i] = ss[i] + ss[i ^ off];
#pragma unroll
                for (int i = 0; i < 16; ++i) ss[i] = t2[i];
            }
            float v = ss[0];
            v = v + __shfl_xor(v, 32, kWave);
            v = v + __shfl_xor(v, 16, kWave);
            rnq = inv_norm_from_sumsq(__shfl(v, r, kWave));  // norm lane 0 = scan lane (r, 0)
            if (rnq_out != nullptr && w == 0 && g == 0 && r < m_valid) rnq_out[r] = rnq;
        }
        if (BEST) {
            // Branch-free per value (rows past the panel's end become the empty composite 0), row
            // numbers relative to the panel in 32 bits, the four norms of a row group fetched
            // before they are used, and the two-window bookkeeping only in panels the split
            // row falls into or lies in front of (wave-uniform): the epilogue is a tenth of a
            // panel's time at 64-d.
            uint64_t best = 0, best_c = 0;
            const bool two = split < hi;  // wave-uniform: rows >= split in this panel
            const int prow = (int)(hi - lo);
            const uint32_t rbase = row0 + (uint32_t)lo;
            const int srel = split > lo ? (int)((split < hi ? split : hi) - lo) : 0;  // first panel row of window c
            if (!two) {
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    if (s < ns) {
                        const f32x4 rn = *reinterpret_cast<const f32x4*>(rvbuf + (16 * s + 4 * g) * 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int rel = 16 * s + 4 * g + e;
                            const float sim = (tot[s][e] * rn[e]) * rnq;
                            uint64_t c = make_comp(sim, rbase + (uint32_t)rel);
                            c = rel < prow ? c : 0;
                            best = c > best ? c : best;
                        }
                    }
                }
            } else {
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    if (s < ns) {
                        const f32x4 rn = *reinterpret_cast<const f32x4*>(rvbuf + (16 * s + 4 * g) * 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int rel = 16 * s + 4 * g + e;
                            const float sim = (tot[s][e] * rn[e]) * rnq;
                            uint64_t c = make_comp(sim, rbase + (uint32_t)rel);
                            c = rel < prow ? c : 0;
                            const uint64_t ca = rel < srel ? c : 0, cc = rel < srel ? 0 : c;
                            best = ca > best ? ca : best;
                            best_c = cc > best_c ? cc : best_c;
                        }
                    }
                }
            }
            uint64_t o = __shfl_xor(best, 16, kWave);
            best = o > best ? o : best;
            o = __shfl_xor(best, 32, kWave);
            best = o > best ? o : best;
            wbest_a = best > wbest_a ? best : wbest_a;
            if (two) {
                o = __shfl_xor(best_c, 16, kWave);
                best_c = o > best_c ? o : best_c;
                o = __shfl_xor(best_c, 32, kWave);
                best_c = o > best_c ? o : best_c;
                wbest_c = best_c > wbest_c ? best_c : wbest_c;
            }
        }
        // raw dots out: the plain dots mode, or (DOTS_TOO, search with 2 <= k <= 16) next to the
        // per-wave best composites - the k best waves then bound where the k best rows can be and
        // the select reads the dots of those waves' rows only (k_topk_from_parts)
        if ((!BEST || DOTS_TOO) && r < m_valid) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                if (s < ns) {
                    const int64_t r0 = lo + 16 * s + 4 * g;
                    float* out = dots + (int64_t)r * dots_stride + r0;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (r0 + e < hi) out[e] = tot[s][e];
                }
            }
        }
    }
    if (BEST && g == 0 && r < m_valid) {
        cand[(int64_t)r * cand_stride + w] = wbest_a;
        if (cand_c != nullptr) cand_c[(int64_t)r * cand_stride + w] = wbest_c;
    }
    CSLAM_DMA_HOOK_KERNEL_END
}

// ---------------------------------------------------------------------------------------------
// k_scan_dma_ks: the scan for SMALL pools (fewer 16-row strips than CUs) and long descriptors.
//   k_scan_dma gives a strip to one wave, which then walks all dp/256 slices one after the other
//   with two stages in flight: at 4096-d that is 16 dependent HBM round trips for a kilobyte-sized
//   pool (47 us for 1000 rows) while most of the chip idles - and a CU streams ~30 GB/s only with
//   all four of its waves asking.  Here a BLOCK owns S = 1..3 strips (an even split of the strips
//   over min(strips, CUs) blocks) and its four waves split the S x nchunks 512-float chunks of those
//   strips evenly, in order: the chunks are independent fmaf chains, so a wave
//   multiplies its run of (strip, chunk) units - same LDS-DMA pieces, ring and MFMA order as
//   k_scan_dma, the ring running across strip boundaries - and leaves each chunk's partial in the
//   LDS; after ONE barrier wave s folds strip s in chunk order (tot = ((0 + c0) + c1) + ...),
//   exactly the sequential fold -> identical bits.
//   Queries are read from the staged block ([16][dp], zero padded rows); raw dots out.
//   S > 1 only under CSLAM_SCAN_KSPLIT_STRIPS (round 3: measured, not faster than k_scan_dma for
//   pools of 257..768 strips - ks_scan_applies; neither was a second ROUND of one-strip blocks,
//   profiles/r03_ks_sweep.txt).
// ---------------------------------------------------------------------------------------------
constexpr int kKsMaxChunks = 16;   // dp <= 8192
constexpr int kKsMaxParts = 28;    // S x nchunks partials of 1 KiB next to the 132-KiB ring
constexpr int kKsMaxStrips = 3;    // per block

template <int AUX>
__global__ __launch_bounds__(256) void k_scan_dma_ks(const float* __restrict__ X, int64_t xs, int64_t n,
                                                     int dp, const float* __restrict__ Qstage,
                                                     int m_valid, float* __restrict__ dots,
                                                     int64_t dots_stride, int n_strips) {
    extern __shared__ __attribute__((aligned(16))) char ks_lds[];
    constexpr int NW = 4, NST = 2;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int r = lane & 15, g = lane >> 4;
    char* ring = ks_lds + wave * NST * kDmaStageB;
    f32x4* part = reinterpret_cast<f32x4*>(ks_lds + NW * NST * kDmaStageB);   // [strip][chunk][lane]
    const int nkc = dp >> 8;
    const int nchunks = (nkc + 1) >> 1;
    // this block's strips: an even split of the strips over the blocks
    const int G = (int)gridDim.x, b = (int)blockIdx.x;
    const int sq = n_strips / G, srem = n_strips - sq * G;
    const int s_first = b * sq + (b < srem ? b : srem);
    const int S = sq + (b < srem ? 1 : 0);
    // ... and this wave's run of (strip, chunk) units, as a run of slices [v0, v1) of the block's
    // S x nkc slices (a chunk = two slices; the last chunk of an odd row length is one)
    const int U = S * nchunks;
    const int u0 = wave * U / NW, u1 = (wave + 1) * U / NW;
    auto slice_of_unit = [&](int u) {
        const int su = u / nchunks, cu = u - su * nchunks;
        return su * nkc + (2 * cu < nkc ? 2 * cu : nkc);
    };
    const int v0 = slice_of_unit(u0), v1 = slice_of_unit(u1);
    struct Cur { int s, kc; };
    auto advance = [&](Cur& c) {
        if (++c.kc == nkc) {
            c.kc = 0;
            ++c.s;
        }
    };
    auto rows_of = [&](int sb) {   // valid rows of the block's strip sb (>= 1)
        const int64_t left = n - (int64_t)(s_first + sb) * 16;
        return left > 16 ? 16 : (int)left;
    };

    if (v0 < v1) {
        auto issue = [&](const Cur& c, char* slot) {
            const int cnt = rows_of(c.s);
            const float* src = X + (int64_t)(s_first + c.s) * 16 * xs + c.kc * 256 + lane * 4;
#pragma unroll
            for (int rr = 0; rr < 16; ++rr)
                if (rr < cnt)
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void*)(src + rr * xs),
                        (__attribute__((address_space(3))) void*)(slot + rr * kDmaPieceB), 16, 0, AUX);
            return cnt;
        };
        const float* qrow = Qstage + (int64_t)r * dp + 4 * g;
        Cur cc{v0 / nkc, v0 % nkc};   // the consumer's stage
        Cur ci = cc;                  // the issue cursor, two stages ahead
        const int kc_first = cc.kc;
        f32x4 bc[16], bn[16];
        load_b16(bn, qrow + cc.kc * 256);
        wait0_tie_b<16>(bn);
        int y_next = 0;   // pieces of the stage behind the consumer's (0: none in flight)
        issue(ci, ring);
        advance(ci);
        if (v0 + 1 < v1) {
            y_next = issue(ci, ring + kDmaStageB);
            advance(ci);
        }
        f32x4 acc{0.f, 0.f, 0.f, 0.f};
        int slot = 0;
        for (int v = v0; v < v1; ++v) {
#pragma unroll
            for (int i = 0; i < 16; ++i) bc[i] = bn[i];
            // ALWAYS issued (the last one re-reads the wave's first slice and is never used): an
            // asm load inside a conditional leaves the compiler free to copy its not-yet-written
            // output registers at the join
            Cur cn = cc;
            advance(cn);
            load_b16(bn, qrow + (v + 1 < v1 ? cn.kc : kc_first) * 256);
            // younger than this stage: the next stage (if any) and the B loads just issued
            wait_vm_dyn(y_next + 16);
            char* sp = ring + slot * kDmaStageB;
            f32x4 a[16];
            const char* ap = sp + r * kDmaPieceB + g * 16;
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] = *reinterpret_cast<const f32x4*>(ap + i * 64);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            int y_refill = 0;
            if (v + 2 < v1) {
                y_refill = issue(ci, sp);
                advance(ci);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][j], bc[i][j], acc, 0, 0, 0);
            if ((cc.kc & 1) || cc.kc + 1 == nkc) {   // end of a 512-float chunk / of the row
                part[(cc.s * nchunks + (cc.kc >> 1)) * 64 + lane] = acc;
                acc = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            wait_vm_dyn<true>(y_refill);   // the B loads have landed (only the refill is younger)
            tie_b<16>(bn);
            slot ^= 1;
            y_next = y_refill;
            cc = cn;
        }
    }
    __syncthreads();
    if (wave < S) {   // wave s folds strip s
        f32x4 tot{0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < nchunks; ++c) tot = tot + part[(wave * nchunks + c) * 64 + lane];
        if (r < m_valid) {
            const int64_t r0 = (int64_t)(s_first + wave) * 16 + 4 * g;
            float* out = dots + (int64_t)r * dots_stride + r0;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (r0 + e < n) out[e] = tot[e];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_gemm32: raw dots for MANY queries (the MFMA-bound regime: from ~128 queries per call, see
// the cost model in compute_dots).
//   LDS-tiled fp32 GEMM on v_mfma_f32_32x32x2_f32: block = 4 waves (2 x 2), block tile
//   128 rows x 128 queries, K step = one 16-float block; every wave owns 64 x 64 outputs as
//   2 x 2 MFMA tiles.  The canonical chain order of DESIGN.md §3 maps onto the 2-slot MFMA as
//   (k, k+4) pairs: lane (row, h) holds v0 = X[row][16kb + 4h ..+3] and v1 = X[row][16kb + 8 +
//   4h ..+3]; the 8 MFMAs of a block consume v0[0], v1[0], v0[1], v1[1], ... — i.e. k = 0,4,8,12,
//   1,5,9,13, ... exactly the order of the 16x16x4 path, so similarities are bit-identical to the
//   streaming scan.  Chunk partials are folded every 32 k-blocks like everywhere else.
//   Operands go HBM/L2 → registers (prefetched one step ahead) → LDS (row pitch 20 floats:
//   conflict-free ds_read_b128) → MFMA.  Blocks are ordered so that the 128-row X tile is
//   consumed by ONE XCD for all its query tiles back to back (first read from HBM, the rest
//   from that XCD's L2).
// ---------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kGemmKb = 1;                         // k-blocks (of 16 floats) per LDS stage (2 measured 5 % slower)
constexpr int kGemmPitch = kGemmKb * 16 + 4;       // 36 floats: conflict-free ds_read_b128

__global__ __launch_bounds__(256, 2) void k_gemm32(const float* __restrict__ X, int64_t xs,
                                                   int64_t n, int dp,
                                                   const float* __restrict__ Q, int m_valid,
                                                   float* __restrict__ dots, int64_t dots_stride,
                                                   int n_rowblk, int n_colblk) {
    __shared__ __attribute__((aligned(16))) float As[2][128 * kGemmPitch];
    __shared__ __attribute__((aligned(16))) float Bs[2][128 * kGemmPitch];
    // XCD-aware tile order: block b runs on XCD b % 8 (dispatch is round-robin); give that XCD
    // row blocks xcd, xcd + 8, ... and walk the query tiles of a row block consecutively
    const int b = blockIdx.x;
    const int xcd = b & 7;
    const int t = b >> 3;
    const int colblk = t % n_colblk;
    const int rowblk = (t / n_colblk) * 8 + xcd;
    if (rowblk >= n_rowblk) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int r32 = lane & 31, h = lane >> 5;
    const int nkb = dp >> 4;
    const int nstep = (nkb + kGemmKb - 1) / kGemmKb;

    // staging assignment: thread → tile row tid/2, half of the stage row's kGemmKb*4 float4
    constexpr int kF4 = kGemmKb * 2;  // float4 per thread per operand per stage
    const int srow = tid >> 1;
    const int sc4 = (tid & 1) * kF4;
    int64_t xrow = (int64_t)rowblk * 128 + srow;
    if (xrow >= n) xrow = n - 1;  // clamp: results of clamped rows are not stored
    const f32x4* xg = reinterpret_cast<const f32x4*>(X + xrow * xs);
    const f32x4* qg = reinterpret_cast<const f32x4*>(Q + (int64_t)(colblk * 128 + srow) * dp);
    const int woff = srow * kGemmPitch + sc4 * 4;
    const int nf4 = dp >> 2;  // float4 per row

    f32x16 acc[2][2], tot[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                acc[i][j][e] = 0.f;
                tot[i][j][e] = 0.f;
            }

    // operands travel HBM/L2 → registers THREE stages ahead (one stage = 2048 MFMA cycles per
    // wave, less than a loaded HBM round trip) → LDS (double buffered) → MFMA
    f32x4 pa[3][kF4], pb[3][kF4];
    auto fetch = [&](f32x4 (&da)[kF4], f32x4 (&db)[kF4], int step) {
        if (step >= nstep) step = nstep - 1;  // past the end: harmless re-read, branch-free
#pragma unroll
        for (int c = 0; c < kF4; ++c) {
            int f = step * (kGemmKb * 4) + sc4 + c;  // float4 index in the row
            if (f >= nf4) f = nf4 - 1;               // ragged last stage: value unused (kb >= nkb)
            da[c] = xg[f];
            db[c] = qg[f];
        }
    };
    auto consume = [&](const f32x4 (&sa)[kF4], const f32x4 (&sb)[kF4], int step) {
        const int buf = step & 1;
#pragma unroll
        for (int c = 0; c < kF4; ++c) {
            *reinterpret_cast<f32x4*>(&As[buf][woff + 4 * c]) = sa[c];
            *reinterpret_cast<f32x4*>(&Bs[buf][woff + 4 * c]) = sb[c];
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < kGemmKb; ++kk) {
            const int kb = step * kGemmKb + kk;
            if (kb < nkb) {
                f32x4 a0[2], a1[2], b0[2], b1[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const float* ap =
                        &As[buf][(wr * 64 + i * 32 + r32) * kGemmPitch + kk * 16 + 4 * h];
                    a0[i] = *reinterpret_cast<const f32x4*>(ap);
                    a1[i] = *reinterpret_cast<const f32x4*>(ap + 8);
                    const float* bp =
                        &Bs[buf][(wc * 64 + i * 32 + r32) * kGemmPitch + kk * 16 + 4 * h];
                    b0[i] = *reinterpret_cast<const f32x4*>(bp);
                    b1[i] = *reinterpret_cast<const f32x4*>(bp + 8);
                }
#pragma unroll
                for (int j4 = 0; j4 < 4; ++j4) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                                a0[i][j4], b0[j][j4], acc[i][j], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                                a1[i][j4], b1[j][j4], acc[i][j], 0, 0, 0);
                }
                if ((kb + 1) % kChunkKb == 0 || kb + 1 == nkb) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            tot[i][j] = tot[i][j] + acc[i][j];
#pragma unroll
                            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
                        }
                }
            }
        }
        // the double buffer makes one barrier per stage enough: stage s+1 writes the other
        // buffer, and nobody reaches stage s+2's writes before passing stage s+1's barrier
    };
    fetch(pa[0], pb[0], 0);
    fetch(pa[1], pb[1], 1);
    fetch(pa[2], pb[2], 2);
    int step = 0;
    for (; step + 3 <= nstep; step += 3) {
        consume(pa[0], pb[0], step);
        fetch(pa[0], pb[0], step + 3);
        consume(pa[1], pb[1], step + 1);
        fetch(pa[1], pb[1], step + 4);
        consume(pa[2], pb[2], step + 2);
        fetch(pa[2], pb[2], step + 5);
    }
    if (step < nstep) consume(pa[0], pb[0], step);
    if (step + 1 < nstep) consume(pa[1], pb[1], step + 1);
    // epilogue: lane holds D[row = (e&3) + 8*(e>>2) + 4h][col = r32] of every 32x32 tile
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = colblk * 128 + wc * 64 + j * 32 + r32;
            if (q < m_valid) {
                float* out = dots + (int64_t)q * dots_stride;
                const int64_t rb = (int64_t)rowblk * 128 + wr * 64 + i * 32 + 4 * h;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int64_t row = rb + (e & 3) + 8 * (e >> 2);
                    if (row < n) out[row] = tot[i][j][e];
                }
            }
        }
}

// ---------------------------------------------------------------------------------------------
// k_gemm_dma: the MFMA-bound bulk path (>= 96 queries) with BOTH operands staged by LDS-DMA.
//   Block = 4 waves (one per SIMD, 2 x 2), block tile 256 rows x 128 queries, K stage = 32
//   floats: a stage tile is 384 "tile rows" (256 X rows then 128 query rows) x 128 B, i.e. whole
//   128-B lines, moved by 48 global_load_lds_dwordx4 of 8 rows each (12 per wave) into a 3-slot
//   LDS ring (3 x 48 KiB): two stages are in flight while the third is being multiplied, ONE
//   raw s_barrier per stage (8192 MFMA cycles per wave) behind a counted vmcnt — no register
//   staging, no ds_write, no vmcnt(0) in the loop.  The LDS image of a DMA instruction is linear,
//   so the 16-byte segments of every row are XOR-swizzled on the SOURCE side (segment ^ (row/2 & 7))
//   and un-swizzled by the ds_read_b128 address: the 16-lane groups of ds_read_b128 then touch 16
//   different 16-B slots (conflict free, checked by hand against the lane groups of the guide).
//   Every wave owns 128 x 64 outputs = 4 x 2 MFMA tiles of v_mfma_f32_32x32x2_f32; operand order
//   inside a 16-float block is the (k, k+4) pairing of k_gemm32, chunk folds every 32 k-blocks:
//   identical bits to every other dots kernel.  XCD-aware tile order as in k_gemm32.
// ---------------------------------------------------------------------------------------------
constexpr int kGdCols = 128, kGdK = 32;

// TM = 32-row MFMA tiles per wave along the rows (block rows = 64 * TM), SLOTS = LDS ring depth.
//   <4, 3>: 256 x 128 tile, 3 x 48 KiB, one block (4 waves, 1 per SIMD) per CU, two stages in flight
//   <2, 2>: 128 x 128 tile, 2 x 32 KiB, two blocks per CU (2 waves per SIMD cover each other's
//           barrier and DMA-issue stalls), one stage in flight — the default (122 vs 120 TF)
//   <1, 2>:  64 x 128 tile for the LAST, partial wave of tiles: the rows that do not fill a whole
//           number of tile waves are cut into half-height tiles so that they spread over the
//           chip (12.2 waves of tiles cost 12.5 instead of 13)
//   TJ = 32-column MFMA tiles per wave along the queries (block columns = 64 * TJ): <2, 2, 1> is the
//   128 x 64 tile for 33..64 queries (and a trailing 64-query block), 2 x 24 KiB per block.
template <int TM, int SLOTS, int TJ = 2>
__global__ __launch_bounds__(256, TM <= 2 ? 2 : 1) void k_gemm_dma(
    const float* __restrict__ X, int64_t xs, int64_t n, int dp, const float* __restrict__ Q,
    int m_valid, float* __restrict__ dots, int64_t dots_stride, int n_rowblk, int n_colblk) {
    constexpr int kRows = 64 * TM;                    // X rows per block
    constexpr int kCols = 64 * TJ;                    // queries per block
    constexpr int kTileRows = kRows + kCols;          // rows of 128 B per stage
    constexpr int kStageB = kTileRows * kGdK * 4;
    constexpr int kPW = kTileRows / 32;               // DMA instructions per wave and stage
    extern __shared__ __attribute__((aligned(16))) char gd_lds[];
    const int b = blockIdx.x;
    const int xcd = b & 7;
    const int t = b >> 3;
    const int colblk = t % n_colblk;
    const int rowblk = (t / n_colblk) * 8 + xcd;
    if (rowblk >= n_rowblk) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int r32 = lane & 31, h = lane >> 5;
    const int nkb = dp >> 4;
    const int nstage = (nkb + 1) >> 1;

    // ---- DMA side: this wave moves tile rows [8*kPW*wave, 8*kPW*(wave+1)) of every stage ------
    const int rr = lane >> 3, pos = lane & 7;
    const float* src[kPW];
    int seg_off[kPW];  // float offset of this lane's (swizzled) segment inside a stage
#pragma unroll
    for (int i = 0; i < kPW; ++i) {
        const int trow = (kPW * wave + i) * 8 + rr;
        const int seg = pos ^ ((trow >> 1) & 7);
        seg_off[i] = seg * 4;
        if (trow < kRows) {
            int64_t row = (int64_t)rowblk * kRows + trow;
            if (row >= n) row = n - 1;  // clamp: results of clamped rows are not stored
            src[i] = X + row * xs;
        } else {
            src[i] = Q + (int64_t)(colblk * kCols + trow - kRows) * dp;
        }
    }
    // quarter `part` of this wave's kPW pieces for `stage`
    auto issue_part = [&](int stage, int slot, int part) {
        char* dst = gd_lds + slot * kStageB + (kPW * wave) * 1024;
#pragma unroll
        for (int i = part * kPW / 4; i < (part + 1) * kPW / 4; ++i) {
            int k = stage * kGdK + seg_off[i];
            if (k >= dp) k -= 16;  // odd number of k-blocks: the missing half re-reads the first
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(src[i] + k),
                (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
        }
    };
    auto issue = [&](int stage, int slot) {
#pragma unroll
        for (int part = 0; part < 4; ++part) issue_part(stage, slot, part);
    };

    // ---- MFMA side ---------------------------------------------------------------------------
    const int swz = (r32 >> 1) & 7;
    int a_row[TM], b_row[TJ];
#pragma unroll
    for (int i = 0; i < TM; ++i) a_row[i] = (wr * 32 * TM + i * 32 + r32) * 128;
#pragma unroll
    for (int j = 0; j < TJ; ++j) b_row[j] = (kRows + wc * 32 * TJ + j * 32 + r32) * 128;
    int seg_rd[2][2];  // [k-block in stage][v0 / v1] byte offset of the un-swizzled segment
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int half = 0; half < 2; ++half) seg_rd[kb][half] = ((kb * 4 + h + 2 * half) ^ swz) * 16;

    f32x16 acc[TM][TJ], tot[TM][TJ];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                acc[i][j][e] = 0.f;
                tot[i][j][e] = 0.f;
            }

    // one k-block (16 floats) of the stage at `base`: 2*(TM+2) fragment reads, 16*TM MFMAs.  With
    // pre_stage >= 0 the wave's DMA pieces of that stage are issued a few at a time in the
    // shadow of the MFMAs (issued in one burst before them they kept the matrix pipe idle).
    auto kblock = [&](const char* base, int kb, int pre_stage, int pre_slot) {
        f32x4 a0[TM], a1[TM], b0[TJ], b1[TJ];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            a0[i] = *reinterpret_cast<const f32x4*>(base + a_row[i] + seg_rd[kb][0]);
            a1[i] = *reinterpret_cast<const f32x4*>(base + a_row[i] + seg_rd[kb][1]);
        }
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            b0[j] = *reinterpret_cast<const f32x4*>(base + b_row[j] + seg_rd[kb][0]);
            b1[j] = *reinterpret_cast<const f32x4*>(base + b_row[j] + seg_rd[kb][1]);
        }
#pragma unroll
        for (int j4 = 0; j4 < 4; ++j4) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i][j4], b0[j][j4], acc[i][j],
                                                                     0, 0, 0);
            if (pre_stage >= 0) {
                __builtin_amdgcn_sched_barrier(0);
                issue_part(pre_stage, pre_slot, j4);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i][j4], b1[j][j4], acc[i][j],
                                                                     0, 0, 0);
        }
    };
    auto fold = [&]() {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                tot[i][j] = tot[i][j] + acc[i][j];
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
            }
    };

    constexpr int kAhead = SLOTS - 1;  // stages issued ahead of the one being multiplied
#pragma unroll
    for (int i = 0; i < kAhead; ++i)
        if (i < nstage) issue(i, i);
    int slot = 0;
    const int nfull = nkb >> 1;  // stages with both k-blocks; an odd last k-block is peeled below
    // two loop levels (16 stages = 32 k-blocks = one 512-float chunk, then the fold) so that the
    // accumulators are carried by ONE plain inner loop and stay where the MFMA wants them
    for (int s0 = 0; s0 < nfull; s0 += 16) {
        const int s1 = s0 + 16 < nfull ? s0 + 16 : nfull;
        for (int s = s0; s < s1; ++s) {
            // this wave's pieces of stage s have landed (later stages' may still fly) ...
            if (SLOTS == 3 && s + 1 < nstage)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kPW) : "memory");
            else
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // ... and so have everybody's; everybody is also done reading stage s-1
            __builtin_amdgcn_s_barrier();
            const char* base = gd_lds + slot * kStageB;
            // stage s+kAhead goes into the slot of stage s-1 while k-block 0 is multiplied
            kblock(base, 0, s + kAhead < nstage ? s + kAhead : -1, slot == 0 ? SLOTS - 1 : slot - 1);
            kblock(base, 1, -1, 0);
            slot = slot == SLOTS - 1 ? 0 : slot + 1;
        }
        if (s1 == s0 + 16) fold();  // a whole chunk (kChunkKb k-blocks)
    }
    if (nkb & 1) {  // the row ends with half a stage
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        kblock(gd_lds + slot * kStageB, 0, -1, 0);
    }
    if ((nkb & 31) != 0) fold();  // the last, shorter chunk

    // epilogue: lane holds D[row = (e&3) + 8*(e>>2) + 4h][col = r32] of every 32x32 tile
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int q = colblk * kCols + wc * 32 * TJ + j * 32 + r32;
            if (q < m_valid) {
                float* out = dots + (int64_t)q * dots_stride;
                const int64_t rb = (int64_t)rowblk * kRows + wr * 32 * TM + i * 32 + 4 * h;
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) {
                    const int64_t row = rb + 8 * e4;
                    if (row + 3 < n) {
                        *reinterpret_cast<f32x4*>(out + row) =
                            f32x4{tot[i][j][4 * e4], tot[i][j][4 * e4 + 1], tot[i][j][4 * e4 + 2],
                                  tot[i][j][4 * e4 + 3]};
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (row + e < n) out[row + e] = tot[i][j][4 * e4 + e];
                    }
                }
            }
        }
}

// cosine scaling of raw dots in place: sims[q][i] = (dot * rnx[i]) * rnq[q]
__global__ __launch_bounds__(256) void k_scale(float* __restrict__ sims, int64_t stride, int64_t n,
                                               const float* __restrict__ rnx,
                                               const float* __restrict__ rnq) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t q = blockIdx.y;
    if (i >= n) return;
    float s = (sims[q * stride + i] * rnx[i]) * rnq[q];
    if (s == 0.0f) s = 0.0f;
    sims[q * stride + i] = s;
}

// ---------------------------------------------------------------------------------------------
// exact top-k by rounds of block arg-max over composites
// ---------------------------------------------------------------------------------------------
template <bool FROM_SIMS, bool FINAL>
__global__ __launch_bounds__(256) void k_select(const float* __restrict__ sims,
                                                int64_t sims_stride,
                                                const float* __restrict__ rnx,
                                                const float* __restrict__ rnq, uint32_t row0,
                                                const uint64_t* __restrict__ cin,
                                                int64_t cin_stride, int64_t n_in, int k,
                                                uint64_t* __restrict__ cout, int64_t cout_stride,
                                                int32_t* __restrict__ out_idx,
                                                float* __restrict__ out_sim) {
    __shared__ uint64_t wmax[2][4];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int64_t q = blockIdx.y;
    const int64_t base = (int64_t)blockIdx.x * kSelectPart;

    uint64_t v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        int64_t i = base + e * 256 + tid;
        uint64_t c = 0;
        if (i < n_in) {
            if (FROM_SIMS)
                c = make_comp((sims[q * sims_stride + i] * rnx[i]) * rnq[q], row0 + (uint32_t)i);
            else
                c = cin[q * cin_stride + i];
        }
        v[e] = c;
    }

    // every thread keeps the maximum of its 16 composites; a round only costs a block-wide
    // maximum of those, and the ONE thread that owned the winner (composites are unique: the row
    // is part of them) re-scans its 16 — the other three waves skip that altogether
    uint64_t mt = v[0];
#pragma unroll
    for (int e = 1; e < 16; ++e) mt = v[e] > mt ? v[e] : mt;
    for (int rd = 0; rd < k; ++rd) {
        const uint64_t m = wave_max_u64(mt);
        if (lane == 0) wmax[rd & 1][wave] = m;
        __syncthreads();
        uint64_t win = wmax[rd & 1][0];
#pragma unroll
        for (int x = 1; x < 4; ++x) {
            uint64_t o = wmax[rd & 1][x];
            win = o > win ? o : win;
        }
        if (win != 0 && mt == win) {
            mt = 0;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                if (v[e] == win) v[e] = 0;
                mt = v[e] > mt ? v[e] : mt;
            }
        }
        if (tid == 0) {
            if (FINAL) {
                if (win != 0) {
                    out_idx[q * k + rd] = (int32_t)(0xFFFFFFFFu - (uint32_t)(win & 0xFFFFFFFFu));
                    out_sim[q * k + rd] = ord2f((uint32_t)(win >> 32));
                } else {
                    out_idx[q * k + rd] = -1;
                    out_sim[q * k + rd] = -INFINITY;
                }
            } else {
                cout[q * cout_stride + (int64_t)blockIdx.x * k + rd] = win;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// any-k path: bitonic sort of composites (descending) in HBM
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_make_comp(const float* __restrict__ sims,
                                                   int64_t sims_stride,
                                                   const float* __restrict__ rnx,
                                                   const float* __restrict__ rnq, uint32_t row0,
                                                   int64_t n, int64_t npow2,
                                                   uint64_t* __restrict__ comp) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t q = blockIdx.y;
    if (i >= npow2) return;
    comp[q * npow2 + i] =
        (i < n) ? make_comp((sims[q * sims_stride + i] * rnx[i]) * rnq[q], row0 + (uint32_t)i)
                : 0ull;
}

__global__ __launch_bounds__(256) void k_bitonic_step(uint64_t* __restrict__ comp, int64_t npow2,
                                                      int64_t size, int64_t stride) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t q = blockIdx.y;
    if (i >= npow2 / 2) return;
    int64_t pos = 2 * i - (i & (stride - 1));
    uint64_t* p = comp + q * npow2;
    uint64_t a = p[pos], b = p[pos + stride];
    bool desc = (pos & size) == 0;
    if (desc ? (a < b) : (a > b)) {
        p[pos] = b;
        p[pos + stride] = a;
    }
}

__global__ __launch_bounds__(256) void k_emit_sorted(const uint64_t* __restrict__ comp,
                                                     int64_t npow2, int k,
                                                     int32_t* __restrict__ out_idx,
                                                     float* __restrict__ out_sim) {
    int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t q = blockIdx.y;
    if (j >= k) return;
    uint64_t c = (j < npow2) ? comp[q * npow2 + j] : 0ull;
    if (c != 0) {
        out_idx[q * k + j] = (int32_t)(0xFFFFFFFFu - (uint32_t)(c & 0xFFFFFFFFu));
        out_sim[q * k + j] = ord2f((uint32_t)(c >> 32));
    } else {
        out_idx[q * k + j] = -1;
        out_sim[q * k + j] = -INFINITY;
    }
}

// queries of a small host-pointer call: pinned bounce buffer (device-visible host memory) -> HBM by a kernel - a
// copy OPERATION costs ~8 us of fixed latency in the stream, this launch ~3 (CSLAM_HOST_Q_COPY=1: hipMemcpyAsync)
__global__ __launch_bounds__(256) void k_copy_f4(const f32x4* __restrict__ src, f32x4* __restrict__ dst, int64_t n4,
                                                 const float* __restrict__ src_tail, float* __restrict__ dst_tail,
                                                 int n_tail) {
    const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
    for (int64_t i = i0; i < n4; i += stride) dst[i] = src[i];
    if (i0 < n_tail) dst_tail[i0] = src_tail[i0];
}

__global__ __launch_bounds__(256) void k_fill_empty(int64_t total, int32_t* __restrict__ out_idx,
                                                    float* __restrict__ out_sim) {
    int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= total) return;
    out_idx[j] = -1;
    out_sim[j] = -INFINITY;
}

template <class Parts>
__global__ __launch_bounds__(256) void k_topk_from_parts(const uint64_t* __restrict__ cand, int64_t cand_stride,
                                                         int n_parts, const Parts parts,
                                                         const float* __restrict__ dots, int64_t dots_stride,
                                                         const float* __restrict__ rnx,
                                                         const float* __restrict__ rnq, uint32_t row0, int64_t n,
                                                         int k, int32_t* __restrict__ out_idx,
                                                         float* __restrict__ out_sim, uint32_t* done, uint32_t seq,
                                                         uint32_t* ticket) {
    const int64_t q = blockIdx.x;
    topk_from_parts_body(cand + q * cand_stride, n_parts, parts, dots + q * dots_stride, rnx, rnq[q], row0, n, k,
                         out_idx + q * k, out_sim + q * k);
    signal_done(done, seq, ticket);
}

// ---------------------------------------------------------------------------------------------
// k_small_search<TOPK> (reported as k_small_best / k_small_topk): search_best - or search with 2 <= k <= 32,
// the intra-robot match of lcsm.py:78-97 - of <= 16 queries over a SMALL window (<= 4096 rows) in ONE launch - the
// per-keyframe call of the reference at its own scale (nns_matching.py:66-78 on a pool of a few
// hundred 128-d descriptors: BASELINE configs[0]).  The general path is a copy of the queries into
// HBM + k_stage_queries + a scan + k_best_pair: four dependent operations of 4-8 us each around
// ~2 us of work.  Here the queries are read where the caller left them (device memory, or the pinned
// bounce buffer of the host-pointer API: no copy), every block restates the stage kernel's 1/||q||,
// and the final level is an atomic maximum of the composites (order-free) plus a ticket: the last
// block to finish converts the winners and leaves the scratch zeroed for the next call.
//   A block owns `spb` strips of 16 rows; its (strip, 512-float chunk) units are dealt over the four
//   waves; a unit is the MFMA chain of DESIGN.md section 3 from 0 (operands straight from global memory, two
//   batches of eight k-blocks in flight); chunk partials meet in the LDS and wave s folds strip s's in
//   chunk order - the arithmetic of every other scan, same bits.
// ---------------------------------------------------------------------------------------------
constexpr int kSmallBestMaxRows = 4096, kSmallBestMaxChunks = 16, kSmallBestMaxQFloats = 8192;
constexpr int64_t kSmallBestMaxFloats = (int64_t)6 << 20;
template <bool TOPK>
__global__ __launch_bounds__(256) void k_small_search(const float* __restrict__ X, int64_t xs, int64_t n, int dp,
                                                      const float* __restrict__ Q, int64_t q_stride, int dim, int m,
                                                      int vec_ok, const float* __restrict__ rnx, uint32_t row0,
                                                      int spb, int nchunks, uint64_t* __restrict__ best,
                                                      uint32_t* __restrict__ ticket, int32_t* __restrict__ out_idx,
                                                      float* __restrict__ out_sim, uint32_t* __restrict__ done,
                                                      uint32_t seq, int k, uint64_t* cand, int64_t cand_stride,
                                                      float* dots, int64_t dots_stride) {
    extern __shared__ f32x4 sb_part[];   // [spb * nchunks][64]
    __shared__ float rq_s[16];
    __shared__ int last_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int64_t S = (n + 15) >> 4;
    const int64_t s_first = (int64_t)blockIdx.x * spb;
    const int nkb_all = dp >> 4;
    const float* qrow = Q + (int64_t)r * q_stride;
    auto b_at = [&](int kb) -> f32x4 {   // B fragment: Q[r][16 kb + 4 g .. + 3], zero past m / dim
        const int k = 16 * kb + 4 * g;
        if (r >= m) return f32x4{0.f, 0.f, 0.f, 0.f};
        if (vec_ok && k + 3 < dim) return *reinterpret_cast<const f32x4*>(qrow + k);
        return f32x4{ld_or_zero(qrow, k, dim), ld_or_zero(qrow, k + 1, dim), ld_or_zero(qrow, k + 2, dim),
                     ld_or_zero(qrow, k + 3, dim)};
    };
    for (int u = wave; u < spb * nchunks; u += 4) {
        const int s = u / nchunks, c = u - s * nchunks;
        const int64_t strip = s_first + s;
        if (strip >= S) break;   // (units are strip-major: nothing behind it either)
        int64_t row = (strip << 4) + r;
        row = row < n ? row : n - 1;   // rows past the window repeat its last row; masked below
        const f32x4* ap = reinterpret_cast<const f32x4*>(X + row * xs + 4 * g);
        const int kb0 = 32 * c, kb1 = kb0 + 32 < nkb_all ? kb0 + 32 : nkb_all;
        f32x4 acc{0.f, 0.f, 0.f, 0.f};
        f32x4 a[2][8], b[2][8];
        auto load = [&](int set, int kb) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const bool in = kb + i < kb1;
                a[set][i] = in ? ap[(int64_t)(kb + i) * 4] : f32x4{0.f, 0.f, 0.f, 0.f};   // fmaf(0, 0, acc) = acc
                b[set][i] = in ? b_at(kb + i) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        };
        auto mul = [&](int set) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[set][i][j], b[set][i][j], acc, 0, 0, 0);
        };
        load(0, kb0);
#pragma unroll 1
        for (int kb = kb0; kb < kb1; kb += 16) {
            load(1, kb + 8);
            mul(0);
            if (kb + 8 < kb1) {   // wave-uniform (a 128-float row is one batch)
                load(0, kb + 16);
                mul(1);
            }
        }
        sb_part[u * 64 + lane] = acc;
    }
    // 1 / ||q||: query j by wave (j + 1) % 4 (one query: beside wave 0's unit)
    for (int j = 0; j < m; ++j)
        if (((j + 1) & 3) == wave) {
            const float rn = query_inv_norm_wave64(Q + (int64_t)j * q_stride, dim, dp, vec_ok, lane);
            if (lane == 0) rq_s[j] = rn;
        }
    __syncthreads();
    for (int s = wave; s < spb; s += 4) {
        const int64_t strip = s_first + s;
        if (strip >= S) break;
        f32x4 tot{0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < nchunks; ++c) tot = tot + sb_part[(s * nchunks + c) * 64 + lane];
        const int64_t lo = strip << 4;
        const int rows = (int)(n - lo < 16 ? n - lo : 16);
        const float rq = r < m ? rq_s[r] : 0.0f;
        uint64_t bst = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int rel = 4 * g + e;
            const int64_t row = lo + rel < n ? lo + rel : n - 1;
            const float sim = (tot[e] * rnx[row]) * rq;
            const uint64_t cmp = make_comp(sim, row0 + (uint32_t)(lo + rel));
            bst = (rel < rows && cmp > bst) ? cmp : bst;
        }
        uint64_t o = __shfl_xor(bst, 16, 64);
        bst = o > bst ? o : bst;
        o = __shfl_xor(bst, 32, 64);
        bst = o > bst ? o : bst;
        if (TOPK) {
            // search with 2 <= k <= 32: the strip's composite and the raw dots of its rows stay in memory for
            // the last block's rank select (topk_from_parts_body over the strips)
            if (r < m) {
                if (g == 0) cand[(int64_t)r * cand_stride + strip] = bst;
                float* out = dots + (int64_t)r * dots_stride + lo + 4 * g;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (4 * g + e < rows) out[e] = tot[e];
            }
        } else if (g == 0 && r < m) {
            __hip_atomic_fetch_max(&best[r], bst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // the last block to get here sees every block's maxima: every thread's atomics are performed
    // (fence) before its block takes a ticket
    __threadfence();
    __syncthreads();
    if (tid == 0) {
        const uint32_t t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        last_s = t == gridDim.x - 1 ? 1 : 0;
    }
    __syncthreads();
    if (!last_s) return;
    if (TOPK) {
        // every block's composites and dots were released before its ticket; acquire them (this block's
        // caches may hold lines of an earlier call), then one rank select per query
        __threadfence();
        const StripParts parts{n};
        for (int j = 0; j < m; ++j) {
            topk_from_parts_body(cand + (int64_t)j * cand_stride, (int)S, parts, dots + (int64_t)j * dots_stride, rnx,
                                 rq_s[j], row0, n, k, out_idx + (int64_t)j * k, out_sim + (int64_t)j * k);
            __syncthreads();
        }
    } else if (tid < m) {
        const uint64_t w = __hip_atomic_load(&best[tid], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        out_idx[tid] = comp_row(w);
        out_sim[tid] = comp_sim(w);
        __hip_atomic_store(&best[tid], (uint64_t)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    if (done != nullptr) {
        // host-pointer API: the results sit in pinned host memory and the host polls `done` for this call's
        // number instead of paying a stream synchronisation (results first, system-wide, then the number)
        __threadfence_system();
        __syncthreads();
        if (tid == 0) __hip_atomic_store(done, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// final level of search_best: one block per (query, window) reduces the waves' composites.
// window 1 (blockIdx.y == 1) exists for the split search only.
__global__ __launch_bounds__(256) void k_best_pair(const uint64_t* __restrict__ cand_a,
                                                   const uint64_t* __restrict__ cand_c,
                                                   int64_t stride, int64_t n_cand,
                                                   int32_t* __restrict__ idx_a,
                                                   float* __restrict__ sim_a,
                                                   int32_t* __restrict__ idx_c,
                                                   float* __restrict__ sim_c, uint32_t* done, uint32_t seq,
                                                   uint32_t* ticket) {
    __shared__ uint64_t scratch[4];
    const int q = blockIdx.x;
    const bool second = blockIdx.y == 1;
    const uint64_t best = block_max_u64((second ? cand_c : cand_a) + (int64_t)q * stride, n_cand, scratch);
    if (threadIdx.x == 0) {
        (second ? idx_c : idx_a)[q] = comp_row(best);
        (second ? sim_c : sim_a)[q] = comp_sim(best);
    }
    signal_done(done, seq, ticket);
}

}  // namespace

// =============================================================================================
// host side
// =============================================================================================
struct cslam_pool {
    int dim = 0, dp = 0, device = 0, cus = 256;
    int64_t n = 0, cap = 0;
    int64_t xs = 0;             // row stride in floats (dp + anti-camping pad)
    int64_t v0 = 0, vn = 0;     // row window [v0, v0+vn) of the search in flight
    int q_copies = 1;           // replicas of the staged query block (T = 1 path)
    int64_t q_copy_stride = 0;  // floats between replicas
    float* rows = nullptr;      // [cap][dp]
    float* inv_norm = nullptr;  // [cap]
    hipStream_t stream = nullptr;
    DevBuf qstage, rnq, sims, cand_a, cand_b, cand_c, comp, part;
    DevBuf small_scratch;              // k_small_best: 16 composites + the ticket, zero between calls
    uint32_t small_seq = 0;            // call number the host polls for (host-pointer API)
    uint32_t* done_word = nullptr;     // armed by cslam_pool_search's small path: the call's final launch signals it
    bool done_taken = false;
    DevBuf h_q, h_idx, h_sim, h_rows;  // device staging of the host-pointer API
    PinBuf pin_q, pin_out;             // pinned bounce buffers of cslam_pool_search
    PinBuf pin_rows;                   // pinned ring of cslam_pool_add's small appends
    size_t pin_rows_used = 0;
    // profiling of the scan kernel
    std::vector<hipEvent_t> ev;
    int prof_max = 0, prof_n = 0;
    int prof_every = 1, prof_tick = 0;  // bracket every prof_every-th dots launch
    bool prof_on = false;
    const char* last_scan_kernel = "";  // variant of the last dots kernel launched (bench.py)
    // ordering between the device-pointer API (caller's stream) and the host-pointer API (the
    // pool's own non-blocking stream): every *_dev call records dev_ev on its stream, the next
    // call on a different stream waits for it first
    // (round 4: the event is recorded LAZILY - by the first call that needs it, on dev_stream, i.e. behind
    // everything enqueued there since - so that a loop of *_dev calls on one stream carries no event packet
    // between its kernels)
    hipEvent_t dev_ev = nullptr;
    hipStream_t dev_stream = nullptr;
    bool dev_pending = false;
    bool dev_deferred = false;
    // ... and the sharded step's append, which runs on the exchange stream while scans of earlier
    // steps are still marked on the compute stream: an event of its own, so that neither mark
    // hides the other
    hipEvent_t app_ev = nullptr;
    hipStream_t app_stream = nullptr;
    bool app_pending = false;
};

namespace {

// The pool whose searches the CALLING THREAD is running as the shard (PoolInternalScope, shard.hip):
// they work on the explicit row windows of their own step, behind that step's publish event, and must
// not wait for the append of a LATER step that the pipelined exchange has already queued.  Per thread
// and per pool (round 3 kept a flag on the shared pool object: another thread's host-API search on the
// same pool skipped its ordering while a shard was matching - ADVICE r03).
thread_local const cslam_pool* tls_shard_scope = nullptr;
inline bool shard_scope(const cslam_pool* p) { return tls_shard_scope == p; }

// the event behind the last device-API work, recorded now if that was put off (mark_dev)
int dev_event_now(cslam_pool* p) {
    if (p->dev_pending && p->dev_deferred) {
        p->dev_deferred = false;
        if (hipEventRecord(p->dev_ev, p->dev_stream) != hipSuccess) {
            // the caller's stream is gone (destroyed since): whatever ran on it is ordered by the device
            (void)hipGetLastError();
            CSLAM_HIP_TRY(hipDeviceSynchronize());
            p->dev_pending = false;
        }
    }
    return CSLAM_OK;
}
// make `st` wait for the last device-API work enqueued on another stream
int order_after_dev(cslam_pool* p, hipStream_t st) {
    if (p->dev_pending && p->dev_stream != st) {
        CSLAM_TRY(dev_event_now(p));
        CSLAM_HIP_TRY(hipStreamWaitEvent(st, p->dev_ev, 0));
    }
    if (p->app_pending && p->app_stream != st && !shard_scope(p)) {
        // the append was issued a whole step ahead of the scan that needs it: normally it is
        // complete, and then the stream is spared the cross-stream barrier packet (~10 us in
        // front of a scan, profiles/r02_shard_step_timeline.txt)
        if (hipEventQuery(p->app_ev) == hipSuccess)
            p->app_pending = false;
        else
            CSLAM_HIP_TRY(hipStreamWaitEvent(st, p->app_ev, 0));
    }
    return CSLAM_OK;
}
// remember that device-API work is in flight on `st`
// (the event itself is recorded by whoever needs it first - dev_event_now - unless the work moves to another
// stream: an event packet behind every search cost the loop of one-stream callers ~1-2 us per call, measured
// in profiles/r04_dev_event_ab.txt; CSLAM_DEV_EVENT=1 records at once, A/B)
bool dev_event_eager();   // (the knob, defined with the other knobs below)
int mark_dev(cslam_pool* p, hipStream_t st) {
    if (p->dev_pending && p->dev_deferred && p->dev_stream != st)
        CSLAM_TRY(dev_event_now(p));   // earlier work on another stream: its event before the mark moves on
    if (dev_event_eager()) {
        CSLAM_HIP_TRY(hipEventRecord(p->dev_ev, st));
        p->dev_deferred = false;
    } else {
        p->dev_deferred = true;
    }
    p->dev_stream = st;
    p->dev_pending = true;
    return CSLAM_OK;
}

int pool_grow(cslam_pool* p, int64_t want_rows, hipStream_t st) {
    if (want_rows <= p->cap) return CSLAM_OK;
    int64_t ncap = p->cap > 0 ? p->cap : 1000;
    while (ncap < want_rows) ncap *= 2;  // nns_matching.py:38 doubles
    float* nrows = nullptr;
    float* nnorm = nullptr;
    CSLAM_HIP_TRY(hipMalloc(&nrows, (size_t)ncap * p->xs * sizeof(float)));
    CSLAM_HIP_TRY(hipMalloc(&nnorm, (size_t)ncap * sizeof(float)));
    // A growth waits for the whole device, BEFORE the old rows are copied: appends may still be
    // pending on other streams (cslam_pool_add's pinned-ring kernel on the handle's stream, the
    // sharded step's append on its exchange stream) and `st` is not ordered behind them; scans
    // on other streams may still read the old buffers (rare: capacities double)
    if (p->rows) CSLAM_HIP_TRY(hipDeviceSynchronize());
    if (p->n > 0) {
        CSLAM_HIP_TRY(hipMemcpyAsync(nrows, p->rows, (size_t)p->n * p->xs * sizeof(float),
                                     hipMemcpyDeviceToDevice, st));
        CSLAM_HIP_TRY(hipMemcpyAsync(nnorm, p->inv_norm, (size_t)p->n * sizeof(float),
                                     hipMemcpyDeviceToDevice, st));
    }
    CSLAM_HIP_TRY(hipMemsetAsync(nrows + p->n * p->xs, 0,
                                 (size_t)(ncap - p->n) * p->xs * sizeof(float), st));
    CSLAM_HIP_TRY(hipMemsetAsync(nnorm + p->n, 0, (size_t)(ncap - p->n) * sizeof(float), st));
    CSLAM_HIP_TRY(hipStreamSynchronize(st));   // the copies are done before the old buffers go
    if (p->rows) CSLAM_HIP_TRY(hipFree(p->rows));
    if (p->inv_norm) CSLAM_HIP_TRY(hipFree(p->inv_norm));
    p->rows = nrows;
    p->inv_norm = nnorm;
    p->cap = ncap;
    return CSLAM_OK;
}

template <int T, int S, int U>
void launch_scan_t(cslam_pool* p, const float* Q, int m_valid, float* dots, int64_t dots_stride,
                   int64_t n_waves, hipStream_t st) {
    unsigned blocks = (unsigned)((n_waves + 3) / 4);
    // refill granularity of the register ring: G = U (consume the whole ring, then refill it in
    // one burst) measured best for the HBM-bound T = 1 scan (profiles/r01_scan_sweep*.txt);
    // the MFMA-bound T > 1 variants keep the rolling refill (G = 1).
    constexpr int G = (T == 1) ? U : 1;
    hipLaunchKernelGGL((k_scan<T, S, U, G>), dim3(blocks), dim3(256), 0, st,
                       p->rows + p->v0 * p->xs, p->xs, p->vn, p->dp, Q, m_valid, dots,
                       dots_stride, n_waves, p->q_copies, p->q_copy_stride);
}

template <int T, int S, int UMAX>
void launch_scan_u(cslam_pool* p, const float* Q, int m_valid, float* dots, int64_t dots_stride,
                   int64_t n_waves, hipStream_t st) {
    const int nkb = p->dp / 16;
    if (UMAX >= 8 && nkb % 8 == 0)
        launch_scan_t<T, S, (UMAX >= 8 ? 8 : UMAX)>(p, Q, m_valid, dots, dots_stride, n_waves, st);
    else if (UMAX >= 4 && nkb % 4 == 0)
        launch_scan_t<T, S, (UMAX >= 4 ? 4 : UMAX)>(p, Q, m_valid, dots, dots_stride, n_waves, st);
    else if (UMAX >= 2 && nkb % 2 == 0)
        launch_scan_t<T, S, (UMAX >= 2 ? 2 : UMAX)>(p, Q, m_valid, dots, dots_stride, n_waves, st);
    else
        launch_scan_t<T, S, 1>(p, Q, m_valid, dots, dots_stride, n_waves, st);
}

// bracket this dots launch with HIP events?  (every prof_every-th one: an event pair costs ~8 us
// of step time — tools/event_cost.py — so bench.py samples instead of bracketing every launch)
bool prof_take(cslam_pool* p) {
    if (!p->prof_on || p->prof_n >= p->prof_max) return false;
    return (p->prof_tick++ % p->prof_every) == 0;
}

// does the DMA scan apply to the current row window?  (env: CSLAM_SCAN_KERNEL=ring keeps the
// round-1 register-ring kernel for A/B runs)
bool dma_scan_applies(const cslam_pool* p) {
    static const char* kern_env = getenv("CSLAM_SCAN_KERNEL");
    if (p->dp % 64 != 0 || p->vn <= 0) return false;
    if (kern_env && strcmp(kern_env, "dma") == 0) return true;
    if (kern_env && strcmp(kern_env, "ring") == 0) return false;
    return p->vn >= kDmaMinRows;
}

// measurement knobs: read once — or at every launch when CSLAM_KNOBS_LIVE is set, so that a tool can
// alternate settings inside one process (tools/reserve_ab.py; boxes and processes differ by
// more than the effects in question)
// (measurement / A-B knobs: `Knob` + `knob()` of common.hpp - read once per process, cslam_knobs_reload() and
// CSLAM_KNOBS_LIVE have them read again)
bool dev_event_eager() {   // CSLAM_DEV_EVENT=1: record the device-API event behind every *_dev call (mark_dev)
    static Knob k_eager{"CSLAM_DEV_EVENT", 0};
    return knob(k_eager) != 0;
}

// small pool (fewer strips than CUs) with long descriptors: k_scan_dma_ks (K split over the waves
// of a block) instead of one wave walking every slice of its strip.  CSLAM_SCAN_KSPLIT=0 keeps
// k_scan_dma.  CSLAM_SCAN_KSPLIT_STRIPS=2|3 lets a block take that many strips (pools of up to
// 3 x CUs strips): measured SLOWER than k_scan_dma with its fused arg-max - 5000..8192 rows x
// 4096-d 48.5 vs 45.2 us, 10 000..12 288 rows 62 vs 50 us end to end: 14 us per extra strip of a
// block (profiles/r03_ks_strips_per_block.txt) - so the default stays one strip per block.
int ks_strips_per_block(const cslam_pool* p) {
    const int64_t strips = (p->vn + 15) / 16;
    const int64_t blocks = strips < p->cus ? strips : p->cus;
    return (int)((strips + blocks - 1) / blocks);
}
bool ks_scan_applies(const cslam_pool* p, int m) {
    static const bool ks_env = !(getenv("CSLAM_SCAN_KSPLIT") && atoi(getenv("CSLAM_SCAN_KSPLIT")) == 0);
    if (!ks_env || m > 16 || !dma_scan_applies(p) || p->dp % 256 != 0) return false;
    const int nchunks = (p->dp / 256 + 1) / 2;
    static Knob k_max{"CSLAM_SCAN_KSPLIT_STRIPS", 1};
    const int S = ks_strips_per_block(p);
    const int cap = knob(k_max) < kKsMaxStrips ? knob(k_max) : kKsMaxStrips;
    return S <= cap && S * nchunks <= kKsMaxParts && nchunks >= 4 && nchunks <= kKsMaxChunks;
}

// k_scan_dma over the row window [v0, v0+vn): either raw dots (best = false) or one (similarity,
// row) composite per query and panel (best = true, *n_panels_out candidates per query)
int launch_scan_dma(cslam_pool* p, const float* Q, int64_t q_stride, int q_rows, int m, float* dots,
                    int64_t dots_stride, bool best, uint64_t* cand, int64_t cand_stride,
                    int64_t* n_panels_out, hipStream_t st, float* rnq_out = nullptr,
                    int64_t split = -1, uint64_t* cand_c = nullptr, bool dots_too = false) {
    constexpr int NS = 8, NW = 4, NST = 2;
    if (dots_too && (!best || dots == nullptr || rnq_out == nullptr || cand_c != nullptr))
        return fail(CSLAM_E_INVALID, "scan_dma: dots_too needs best, dots, rnq_out and one window");
    // CSLAM_SCAN_RESERVE_CUS=n leaves n CUs to kernels of other streams (the scan's blocks fill a
    // CU each — LDS and registers — so a collective's workgroup that sits on one when the scan
    // starts pushes a scan block into a second round: tools/shard_step_time.py with
    // CSLAM_LOOPBACK_HOG)
    static Knob k_reserve{"CSLAM_SCAN_RESERVE_CUS", 0}, k_even{"CSLAM_SCAN_EVEN_PANELS", 1};
    const int reserve_env = knob(k_reserve);
    const int cus_eff = reserve_env > 0 && reserve_env < p->cus ? p->cus - reserve_env : p->cus;
    const int64_t nw = (int64_t)cus_eff * NW;
    const int64_t strips = (p->vn + 15) / 16;
    int64_t n_waves = nw < strips ? nw : strips;   // small pools: one full strip per wave
    // One panel per wave (<= 128 rows each): every wave walks S = ceil(strips / waves) stage rounds
    // per slice whether its last strip holds 16 rows or 2, so use just enough waves to fill S
    // strips each (100k rows: 893 waves x 112 rows on 224 CUs instead of 1024 x 97.7 on 256 —
    // 0.2553 -> 0.2534 ms per step in alternating runs, tools/reserve_ab.py; the scan is
    // HBM-bound and the idle CUs are free for a collective's kernels).  Only when that idles less
    // than 15 % of the waves: 8192-d x 50k rows would drop to 782 waves (4 full strips each) and
    // runs 7 % SLOWER that way (0.2964 vs 0.2766 ms).  Splitting in whole strips over ALL waves
    // (6 or 7 strips each) is slower too: 0.2700 ms (profiles/r02_scan_partition_ab.txt).
    if (n_waves == nw && strips <= nw * NS && reserve_env == 0) {
        const int64_t s_per_wave = (strips + nw - 1) / nw;
        const int64_t waves = (strips + s_per_wave - 1) / s_per_wave;
        if (waves * 100 >= nw * 85) n_waves = waves;
    }
    if (best && cand_stride < n_waves)   // one composite per query and wave
        return fail(CSLAM_E_INVALID, "scan_dma: candidate stride");
    unsigned blocks = (unsigned)((n_waves + NW - 1) / NW);
    // one strip per wave and fewer strips than wave slots (< 16k rows): deal the waves round-robin
    // over min(CUs, waves) blocks instead of filling block after block - 4200..10 000 rows x 4096-d:
    // 45-48 -> 42-45 us, profiles/r03_size_sweep_spread.txt (CSLAM_SCAN_SPREAD=0: block-major as
    // before).  NOT for the 893-wave partition of ~100k rows below: 250.5 -> 260.4 us there.
    static Knob k_spread{"CSLAM_SCAN_SPREAD", 1};
    int spread = 0;
    if (knob(k_spread) != 0 && n_waves < nw && n_waves == strips && reserve_env == 0) {
        blocks = (unsigned)(n_waves < cus_eff ? n_waves : cus_eff);
        spread = 2;
    }
    // a wave's rows in equal panels (default) or in panels of 128 rows and a remainder
    // (CSLAM_SCAN_EVEN_PANELS=0): fewer stage rounds (390 rows: 8 + 8
    // + 8 + 1 instead of 4 x 7) but a remainder panel pays the whole per-panel cost — equal panels
    // win or tie from 256-d to 2048-d (1024-d 0.2597 vs 0.2633 ms), lose 2 % at 128-d
    const int even = (knob(k_even) == 0 ? 0 : 1) | spread;
    // slice width: the widest of 256 / 128 / 64 floats that divides the padded row length
    const int kw = p->dp % 256 == 0 ? 256 : (p->dp % 128 == 0 ? 128 : 64);

    if (cand_c == nullptr || split < 0) {
        split = INT64_MAX;  // one window
        cand_c = nullptr;
    }
    // pool rows are read once per search: non-temporal unless the pool could stay resident
    // in the 256-MiB Infinity Cache between searches
    const bool nt = (size_t)p->vn * p->xs * sizeof(float) > ((size_t)192 << 20);
#define CSLAM_DMA_LAUNCH(AUX, BEST, KW, DT)                                                         \
    do {                                                                                          \
        const size_t lds = (size_t)NW * NST * DmaGeom<KW>::kStage + (size_t)NW * 512; /* + row norms */ \
        static bool attr_set[64] = {};  /* per device: the attribute is per device code object */ \
        if (!attr_set[p->device & 63]) {                                                          \
            CSLAM_HIP_TRY(hipFuncSetAttribute((const void*)k_scan_dma<NS, NW, NST, AUX, BEST, KW, DT>, \
                                              hipFuncAttributeMaxDynamicSharedMemorySize,         \
                                              (int)lds));                                         \
            attr_set[p->device & 63] = true;                                                      \
        }                                                                                         \
        hipLaunchKernelGGL((k_scan_dma<NS, NW, NST, AUX, BEST, KW, DT>), dim3(blocks), dim3(NW * 64), \
                           lds, st, p->rows + p->v0 * p->xs, p->xs, p->vn, p->dp, Q, q_stride,    \
                           q_rows, m, dots, dots_stride, (int)n_waves,                            \
                           p->inv_norm + p->v0, (uint32_t)p->v0, cand, cand_stride, rnq_out,      \
                           split, cand_c, even);                                                  \
    } while (0)
#define CSLAM_DMA_LAUNCH_KW(AUX, BEST, DT)               \
    do {                                                 \
        if (kw == 256) CSLAM_DMA_LAUNCH(AUX, BEST, 256, DT); \
        else if (kw == 128) CSLAM_DMA_LAUNCH(AUX, BEST, 128, DT); \
        else CSLAM_DMA_LAUNCH(AUX, BEST, 64, DT);           \
    } while (0)
    static const char* const names[3][4] = {
        {"k_scan_dma<NS=8,NW=4,NST=2,nt,best>", "k_scan_dma<NS=8,NW=4,NST=2,best>",
         "k_scan_dma<NS=8,NW=4,NST=2,nt>", "k_scan_dma<NS=8,NW=4,NST=2>"},
        {"k_scan_dma<NS=8,NW=4,NST=2,nt,best,KW=128>", "k_scan_dma<NS=8,NW=4,NST=2,best,KW=128>",
         "k_scan_dma<NS=8,NW=4,NST=2,nt,KW=128>", "k_scan_dma<NS=8,NW=4,NST=2,KW=128>"},
        {"k_scan_dma<NS=8,NW=4,NST=2,nt,best,KW=64>", "k_scan_dma<NS=8,NW=4,NST=2,best,KW=64>",
         "k_scan_dma<NS=8,NW=4,NST=2,nt,KW=64>", "k_scan_dma<NS=8,NW=4,NST=2,KW=64>"}};
    const int ki = kw == 256 ? 0 : (kw == 128 ? 1 : 2);
    if (best && dots_too) {   // search with 2 <= k <= 16: composites per wave AND raw dots
        static const char* const names_dt[3][2] = {
            {"k_scan_dma<NS=8,NW=4,NST=2,nt,best+dots>", "k_scan_dma<NS=8,NW=4,NST=2,best+dots>"},
            {"k_scan_dma<NS=8,NW=4,NST=2,nt,best+dots,KW=128>", "k_scan_dma<NS=8,NW=4,NST=2,best+dots,KW=128>"},
            {"k_scan_dma<NS=8,NW=4,NST=2,nt,best+dots,KW=64>", "k_scan_dma<NS=8,NW=4,NST=2,best+dots,KW=64>"}};
        if (nt) CSLAM_DMA_LAUNCH_KW(2, true, true);
        else CSLAM_DMA_LAUNCH_KW(0, true, true);
        p->last_scan_kernel = names_dt[ki][nt ? 0 : 1];
    } else if (best) {
        if (nt) CSLAM_DMA_LAUNCH_KW(2, true, false);
        else CSLAM_DMA_LAUNCH_KW(0, true, false);
        p->last_scan_kernel = names[ki][nt ? 0 : 1];
    } else {
        if (nt) CSLAM_DMA_LAUNCH_KW(2, false, false);
        else CSLAM_DMA_LAUNCH_KW(0, false, false);
        p->last_scan_kernel = names[ki][nt ? 2 : 3];
    }
#undef CSLAM_DMA_LAUNCH_KW
#undef CSLAM_DMA_LAUNCH
    CSLAM_HIP_TRY(hipGetLastError());
    // BEST: candidates per query = waves that hold panels
    *n_panels_out = n_waves;
    return CSLAM_OK;
}

// raw dots of `m` staged queries (m <= 64) → dots[q*stride + row]
int launch_scan(cslam_pool* p, const float* Qstage, int m, float* dots, int64_t dots_stride,
                hipStream_t st, int64_t q_stride = 0, float* rnq_out = nullptr) {
    static const int wpc_env = getenv("CSLAM_SCAN_WPC") ? atoi(getenv("CSLAM_SCAN_WPC")) : 0;
    static const int s_env = getenv("CSLAM_SCAN_S") ? atoi(getenv("CSLAM_SCAN_S")) : 0;
    // waves per CU and strips per wave of the variant that will run (profiles/r01_scan_sweep*.txt,
    // r01_scan_m_sweep.txt): 17..32 queries <T=2,S=8,U=4> at 4 waves/CU, 33..64 <T=4,S=4,U=4> at 8
    const int wpc = wpc_env > 0 ? wpc_env : (m > 32 ? 8 : 4);
    const int S = (s_env == 1 || s_env == 2 || s_env == 4 || s_env == 16) ? s_env : 8;  // strips per query fragment (T = 1)
    const int s_eff = (m <= 16) ? S : (m <= 32 ? 8 : 4);
    int64_t groups = (p->vn + 16 * s_eff - 1) / (16 * s_eff);
    int64_t resident = (int64_t)p->cus * wpc;
    int64_t n_waves = groups < resident ? groups : resident;
    if (n_waves < 1) n_waves = 1;
    bool prof = prof_take(p);
    if (prof) CSLAM_HIP_TRY(hipEventRecord(p->ev[2 * p->prof_n], st));
    if (m <= 16) {
        // X through LDS-DMA (k_scan_dma, 6.0-6.3 TB/s) for pools that are worth it
        const bool use_dma = dma_scan_applies(p);
        // small pool, long descriptors, staged queries: the block-per-strip K-split kernel
        const int64_t strips = (p->vn + 15) / 16;
        const int nchunks = (p->dp / 256 + 1) / 2;
        if (rnq_out == nullptr && ks_scan_applies(p, m)) {
            const int S = ks_strips_per_block(p);
            const size_t lds = (size_t)4 * 2 * kDmaStageB + (size_t)S * nchunks * 64 * sizeof(f32x4);
            static bool attr_set[64] = {};
            if (!attr_set[p->device & 63]) {
                CSLAM_HIP_TRY(hipFuncSetAttribute((const void*)k_scan_dma_ks<0>,
                                                  hipFuncAttributeMaxDynamicSharedMemorySize,
                                                  (int)((size_t)4 * 2 * kDmaStageB +
                                                        (size_t)kKsMaxParts * 64 * sizeof(f32x4))));
                attr_set[p->device & 63] = true;
            }
            // (default cache policy: a pool this small lives in the Infinity Cache between searches)
            const unsigned blocks = (unsigned)(strips < p->cus ? strips : p->cus);
            hipLaunchKernelGGL((k_scan_dma_ks<0>), dim3(blocks), dim3(256), lds, st,
                               p->rows + p->v0 * p->xs, p->xs, p->vn, p->dp, Qstage, m, dots,
                               dots_stride, (int)strips);
            p->last_scan_kernel = "k_scan_dma_ks (K split over the block's waves)";
        } else if (use_dma) {
            int64_t n_panels = 0;
            // rnq_out set: Qstage are the caller's queries read in place (no staging ran)
            CSLAM_TRY(launch_scan_dma(p, Qstage, rnq_out ? q_stride : p->dp, rnq_out ? m : 16, m, dots,
                                      dots_stride, false, nullptr, 0, &n_panels, st, rnq_out));
        } else {
            static const int u_env = getenv("CSLAM_SCAN_U") ? atoi(getenv("CSLAM_SCAN_U")) : 0;
            const int U = u_env == 8 ? 8 : 4;
            p->last_scan_kernel = "k_scan<T=1> (register ring)";
            if (S == 1 && U == 8)
                launch_scan_u<1, 1, 8>(p, Qstage, m, dots, dots_stride, n_waves, st);
            else if (S == 2 && U == 8)
                launch_scan_u<1, 2, 8>(p, Qstage, m, dots, dots_stride, n_waves, st);
            else if (S == 2)
                launch_scan_u<1, 2, 4>(p, Qstage, m, dots, dots_stride, n_waves, st);
            else if (S == 4 && U == 8)
                launch_scan_u<1, 4, 8>(p, Qstage, m, dots, dots_stride, n_waves, st);
            else if (S == 4)
                launch_scan_u<1, 4, 4>(p, Qstage, m, dots, dots_stride, n_waves, st);
            else if (S == 16)
                launch_scan_u<1, 16, 2>(p, Qstage, m, dots, dots_stride, n_waves, st);
            else
                launch_scan_u<1, 8, 4>(p, Qstage, m, dots, dots_stride, n_waves, st);
        }
    } else if (m <= 32) {
        p->last_scan_kernel = "k_scan<T=2,S=8,U=4>";
        launch_scan_u<2, 8, 4>(p, Qstage, m, dots, dots_stride, n_waves, st);
    } else {
        p->last_scan_kernel = "k_scan<T=4,S=4,U=4>";
        launch_scan_u<4, 4, 4>(p, Qstage, m, dots, dots_stride, n_waves, st);
    }
    CSLAM_HIP_TRY(hipGetLastError());
    if (prof) {
        CSLAM_HIP_TRY(hipEventRecord(p->ev[2 * p->prof_n + 1], st));
        p->prof_n++;
    }
    return CSLAM_OK;
}

// ---- the multi-tile LDS-DMA scan (scan_mq.hip): 17..80 queries per pass, the pool read once ----
// CSLAM_SCAN_MQ=0 keeps round 2's register-ring passes (A/B runs).
bool mq_scan_applies(const cslam_pool* p, int m) {
    static const bool mq_env = !(getenv("CSLAM_SCAN_MQ") && atoi(getenv("CSLAM_SCAN_MQ")) == 0);
    return mq_env && m > 16 && p->dp % 64 == 0 && p->vn > 0;
}
// queries of the next pass when `left` are still to do: as many tiles as a pass takes, the passes
// of one call balanced (96 queries = 48 + 48, not 80 + 16: the pass is MFMA-bound from 3 tiles on)
int mq_pass_queries(int left) {
    const int cap = 16 * mq_max_tiles();
    const int passes = (left + cap - 1) / cap;
    const int per = (left + passes - 1) / passes;
    const int mb = (per + 15) / 16 * 16;
    return mb < left ? mb : left;
}
// How a pass deals its strips over the waves - one launch, main + K-split tail, or the whole window through the
// K-split launch: mq_plan.hpp (plain host arithmetic, checked without a GPU by tests/test_mq_plan_cpu.py).  Here:
// its inputs from the pool and from the A/B knobs.
// (rows >= 0: plan for a window of that many rows instead of the current one)
MqPlan mq_plan(const cslam_pool* p, int m, int64_t rows = -1, int shard_caller = -1) {
    static Knob k_tail{"CSLAM_MQ_TAIL", 1}, k_kw{"CSLAM_MQ_KW", 0}, k_waves{"CSLAM_MQ_WAVES", 4};
    static Knob k_rounds{"CSLAM_MQ_TAIL_ROUNDS", 6}, k_shape{"CSLAM_MQ_TAIL_SHAPE", 0}, k_cost{"CSLAM_MQ_TAIL_MAXCOST", 80};
    static Knob k_allpart{"CSLAM_MQ_ALLPART", 1}, k_allpart_g{"CSLAM_MQ_ALLPART_GROUPS", 0}, k_allpart_k{"CSLAM_MQ_ALLPART_KS", 0};
    static Knob k_part_waves{"CSLAM_MQ_PART_WAVES", 8}, k_pack{"CSLAM_MQ_PACK_BLOCKS", 1}, k_blocks2{"CSLAM_MQ_PART_BLOCKS2", 1};
    MqPlanKnobs kn;
    kn.tail = knob(k_tail);
    kn.kw = knob(k_kw);
    kn.waves = knob(k_waves);
    kn.tail_rounds = knob(k_rounds);
    kn.tail_shape = knob(k_shape);
    kn.tail_maxcost = knob(k_cost);
    kn.allpart = knob(k_allpart);
    kn.allpart_groups = knob(k_allpart_g);
    kn.allpart_ks = knob(k_allpart_k);
    kn.part_waves = knob(k_part_waves);
    kn.part_blocks2 = knob(k_blocks2);
    kn.pack_blocks = knob(k_pack);
    MqPlanIn in;
    in.vn = rows >= 0 ? rows : p->vn;
    in.dp = p->dp;
    in.cus = p->cus;
    in.xs = p->xs;
    in.m = m;
    in.shard = shard_caller >= 0 ? shard_caller != 0 : shard_scope(p);
    in.waves8_available = mq_waves_available(8) != 0;
    return mq_plan_compute(in, kn);
}
const char* mq_name(const MqPlan& g, bool best) {
    static const char* const names[3][2][4] = {
        {{"k_scan_mq<T=2,KW=128,NST=3,NW=4>", "k_scan_mq<T=3,KW=128,NST=3,NW=4>", "k_scan_mq<T=4,KW=128,NST=2,NW=4>",
          "k_scan_mq<T=5,KW=128,NST=2,NW=4>"},
         {"k_scan_mq<T=2,KW=128,NST=3,NW=4,best>", "k_scan_mq<T=3,KW=128,NST=3,NW=4,best>",
          "k_scan_mq<T=4,KW=128,NST=2,NW=4,best>", "k_scan_mq<T=5,KW=128,NST=2,NW=4,best>"}},
        {{"k_scan_mq<T=2,KW=64,NST=6,NW=4>", "k_scan_mq<T=3,KW=64,NST=6,NW=4>", "k_scan_mq<T=4,KW=64,NST=6,NW=4>",
          "k_scan_mq<T=5,KW=64,NST=6,NW=4>"},
         {"k_scan_mq<T=2,KW=64,NST=6,NW=4,best>", "k_scan_mq<T=3,KW=64,NST=6,NW=4,best>",
          "k_scan_mq<T=4,KW=64,NST=6,NW=4,best>", "k_scan_mq<T=5,KW=64,NST=6,NW=4,best>"}},
        {{"k_scan_mq<T=2,KW=64,NST=4,NW=8>", "k_scan_mq<T=3,KW=64,NST=3,NW=8>", "k_scan_mq<T=4,KW=64,NST=3,NW=8>",
          "k_scan_mq<T=5,KW=64,NST=3,NW=8>"},
         {"k_scan_mq<T=2,KW=64,NST=4,NW=8,best>", "k_scan_mq<T=3,KW=64,NST=3,NW=8,best>",
          "k_scan_mq<T=4,KW=64,NST=3,NW=8,best>", "k_scan_mq<T=5,KW=64,NST=3,NW=8,best>"}}};
    if (g.n_main == 0 && g.tail_strips > 0) {   // the whole window through the K-split launch + k_mq_combine
        static const char* const part[2][2][4] = {
            {{"k_scan_mq<T=2,KW=64,NST=3,NW=8,part> + k_mq_combine", "k_scan_mq<T=3,KW=64,NST=3,NW=8,part> + k_mq_combine",
              "k_scan_mq<T=4,KW=64,NST=3,NW=8,part> + k_mq_combine", "k_scan_mq<T=5,KW=64,NST=3,NW=8,part> + k_mq_combine"},
             {"k_scan_mq<T=2,KW=64,NST=3,NW=8,part> + k_mq_combine<best>", "k_scan_mq<T=3,KW=64,NST=3,NW=8,part> + k_mq_combine<best>",
              "k_scan_mq<T=4,KW=64,NST=3,NW=8,part> + k_mq_combine<best>", "k_scan_mq<T=5,KW=64,NST=3,NW=8,part> + k_mq_combine<best>"}},
            {{"k_scan_mq<T=2,NW=4,part> + k_mq_combine", "k_scan_mq<T=3,NW=4,part> + k_mq_combine",
              "k_scan_mq<T=4,NW=4,part> + k_mq_combine", "k_scan_mq<T=5,NW=4,part> + k_mq_combine"},
             {"k_scan_mq<T=2,NW=4,part> + k_mq_combine<best>", "k_scan_mq<T=3,NW=4,part> + k_mq_combine<best>",
              "k_scan_mq<T=4,NW=4,part> + k_mq_combine<best>", "k_scan_mq<T=5,NW=4,part> + k_mq_combine<best>"}}};
        if (g.part_per_cu == 2) {
            static const char* const two[2][4] = {
                {"k_scan_mq<T=2,KW=64,NST=2,NW=4,2 per CU,part> + k_mq_combine", "k_scan_mq<T=3,KW=64,NST=2,NW=4,2 per CU,part> + k_mq_combine",
                 "k_scan_mq<T=4,KW=64,NST=2,NW=4,2 per CU,part> + k_mq_combine", "k_scan_mq<T=5,KW=64,NST=2,NW=4,2 per CU,part> + k_mq_combine"},
                {"k_scan_mq<T=2,KW=64,NST=2,NW=4,2 per CU,part> + k_mq_combine<best>", "k_scan_mq<T=3,KW=64,NST=2,NW=4,2 per CU,part> + k_mq_combine<best>",
                 "k_scan_mq<T=4,KW=64,NST=2,NW=4,2 per CU,part> + k_mq_combine<best>", "k_scan_mq<T=5,KW=64,NST=2,NW=4,2 per CU,part> + k_mq_combine<best>"}};
            return two[best ? 1 : 0][g.tiles - 2];
        }
        return part[g.part_waves == 8 ? 0 : 1][best ? 1 : 0][g.tiles - 2];
    }
    return names[g.waves == 8 ? 2 : (g.kw == 128 ? 0 : 1)][best ? 1 : 0][g.tiles - 2];
}
// stage `m` queries (17..80) and scan the window [v0, v0 + vn): raw dots, or (best) one composite
// per query and panel in cand (and cand_c behind `split`, window-relative, when given)
// (staged != nullptr: the caller staged the queries itself - pool_stage_queries_on - with their norms in rnq)
int mq_pass(cslam_pool* p, const float* q_dev, int64_t q_stride, int m, float* rnq, float* dots,
            int64_t dots_stride, bool best, uint64_t* cand, int64_t cand_stride, int64_t split,
            uint64_t* cand_c, const MqPlan& g, hipStream_t st, const float* staged = nullptr) {
    const int mq_pad = 16 * g.tiles;
    if (staged == nullptr) {
        const int vec_ok = ((reinterpret_cast<uintptr_t>(q_dev) & 15) == 0 && (q_stride & 3) == 0) ? 1 : 0;
        hipLaunchKernelGGL(k_stage_queries, dim3(mq_pad, 1), dim3(256), 0, st, q_dev, q_stride, p->dim, m,
                           mq_pad, p->dp, p->qstage.as<float>(), rnq, vec_ok, (int64_t)0, 0, (int64_t)0);
        CSLAM_HIP_TRY(hipGetLastError());
    }
    MqArgs a{};
    a.X = p->rows + p->v0 * p->xs;
    a.xs = p->xs;
    a.n = g.n_main;
    a.dp = p->dp;
    a.Q = staged != nullptr ? staged : p->qstage.as<float>();
    a.rnq = rnq;
    a.m_valid = m;
    a.dots = dots;
    a.dots_stride = dots_stride;
    a.n_blocks = g.n_blocks;
    a.p_max = g.p_max;
    a.device = p->device;
    a.rnx = p->inv_norm + p->v0;
    a.row0 = (uint32_t)p->v0;
    a.cand = cand;
    a.cand_stride = cand_stride;
    a.split = (cand_c != nullptr && split >= 0) ? split : INT64_MAX;
    a.cand_c = (cand_c != nullptr && split >= 0) ? cand_c : nullptr;
    if (g.tail_strips > 0)
        CSLAM_TRY(p->part.reserve((size_t)g.tail_strips * g.nchunks * g.tiles * 64 * sizeof(f32x4)));
    // The launch geometry against what the buffers hold, on the host, before anything is launched (the kernels
    // clamp their addresses into the window MqArgs describes - tools/probes/mq_probe_audit.hpp checks that
    // half on the device; a window or a plan that does not fit the allocations must never reach them).
    {
        const int64_t strips = (p->vn + 15) / 16, strips_main = (g.n_main + 15) / 16;
        const bool window_ok = p->v0 >= 0 && p->vn > 0 && p->v0 + p->vn <= p->n && p->n <= p->cap;
        const bool plan_ok = g.tiles >= 2 && g.tiles <= mq_max_tiles() && m <= mq_pad && g.n_main >= 0 &&
                             g.n_main <= p->vn && (g.n_main == p->vn || g.n_main % 16 == 0) &&
                             (g.n_main == p->vn ? g.tail_strips == 0 : g.tail_strips == strips - g.n_main / 16) &&
                             (g.n_main == 0 ||
                              (g.n_blocks >= 1 && (int64_t)g.p_max * g.waves * mq_panel_strips(g.tiles, g.waves) >=
                                                      (strips_main + g.n_blocks - 1) / g.n_blocks)) &&
                             (g.tail_strips == 0 || (g.tail_groups >= 1 && g.tail_ks >= 1 && g.tail_ks <= g.nchunks &&
                                                     (int64_t)g.tail_groups * g.tail_ks <= 65535)) &&
                             g.nchunks == (p->dp + 511) / 512 && p->dp % g.kw == 0 && p->dp % g.part_kw == 0;
        const bool stage_ok = staged != nullptr || p->qstage.bytes >= (size_t)mq_pad * p->dp * sizeof(float);
        const bool cand_ok = !best || (cand != nullptr && cand_stride >= g.n_cand &&
                                       g.n_cand == (int64_t)g.n_blocks * g.p_max * g.waves + g.tail_strips);
        const bool dots_ok = dots == nullptr || dots_stride >= p->vn;
        if (!(window_ok && plan_ok && stage_ok && cand_ok && dots_ok && (best || dots != nullptr)))
            return fail(CSLAM_E_INVALID,
                        "mq_pass: launch does not fit its buffers (window %d plan %d stage %d cand %d dots %d; rows %lld+%lld of "
                        "%lld/%lld, main %lld, tail %lld as %d x %d, blocks %d x %d rounds)",
                        (int)window_ok, (int)plan_ok, (int)stage_ok, (int)cand_ok, (int)dots_ok, (long long)p->v0,
                        (long long)p->vn, (long long)p->n, (long long)p->cap, (long long)g.n_main,
                        (long long)g.tail_strips, g.tail_groups, g.tail_ks, g.n_blocks, g.p_max);
    }
    bool prof = prof_take(p);
    if (prof) CSLAM_HIP_TRY(hipEventRecord(p->ev[2 * p->prof_n], st));
    if (g.n_main > 0)
        CSLAM_TRY(launch_scan_mq(a, g.tiles, g.kw, g.waves, best ? kMqModeBest : kMqModeDots, g.nt, st));
    if (g.tail_strips > 0) {
        // the tail window [n_main, vn): K-split launch, then the fold + epilogue of its strips
        MqArgs b = a;
        b.X = a.X + g.n_main * p->xs;
        b.n = p->vn - g.n_main;
        b.rnx = a.rnx + g.n_main;
        b.row0 = a.row0 + (uint32_t)g.n_main;
        b.dots = dots ? dots + g.n_main : nullptr;
        if (a.split != INT64_MAX) b.split = a.split - g.n_main;   // <= 0: every tail row is behind it
        b.n_blocks = g.tail_groups;
        b.p_max = 1;
        b.part = p->part.as<f32x4>();
        b.part_ks = g.tail_ks;
        b.part_per_cu = g.part_per_cu;
        static Knob k_part_nt{"CSLAM_MQ_PART_NT", 1};   // A/B: non-temporal pool reads in the all-K-split launch
        CSLAM_TRY(launch_scan_mq(b, g.tiles, g.part_kw, g.part_waves, kMqModePart,
                                 g.n_main == 0 && g.nt && knob(k_part_nt) != 0, st));
        CSLAM_TRY(launch_mq_combine(b, g.tiles, g.nchunks, best, (int64_t)g.n_blocks * g.p_max * g.waves, st));
    }
    if (prof) {
        CSLAM_HIP_TRY(hipEventRecord(p->ev[2 * p->prof_n + 1], st));
        p->prof_n++;
    }
    p->last_scan_kernel = mq_name(g, best);
    return CSLAM_OK;
}

// raw dots for queries [0,m) (device, arbitrary stride), m <= kQueryChunk; also fills
// p->rnq[0..m) with 1/max(||q||,1e-8)
int compute_dots(cslam_pool* p, const float* q_dev, int64_t q_stride, int m, float* dots,
                 int64_t dots_stride, hipStream_t st) {
    static const int qrep_env = getenv("CSLAM_SCAN_QREP") ? atoi(getenv("CSLAM_SCAN_QREP")) : 0;
    const int qrep_max = qrep_env > 0 ? qrep_env : 1;
    // replica stride: the block plus an odd number of 256-B lines so replicas start on
    // different channels
    const int64_t copy_stride = (int64_t)kMaxQueriesPerPass * p->dp + 64 * 7;
    CSLAM_TRY(p->qstage.reserve(std::max((size_t)qrep_max * copy_stride,
                                         (size_t)(kQueryChunk + 128) * p->dp) * sizeof(float)));
    CSLAM_TRY(p->rnq.reserve((size_t)(kQueryChunk + kMaxQueriesPerPass) * sizeof(float)));
    static const int gemm_env = getenv("CSLAM_GEMM") ? atoi(getenv("CSLAM_GEMM")) : 1;
    // Scan passes or the 128-query GEMM tiles?  Relative costs measured at 100k x 4096
    // (profiles/r01_scan_m_sweep.txt; one 16-query pass = 1): a 32-query pass 1.12, a 64-query
    // pass 1.85, a GEMM tile 3.2.  The scan wins up to ~200 queries except where the tiles fit
    // exactly (128), the GEMM beyond.
    const bool mq = mq_scan_applies(p, m);
    auto scan_cost = [mq](int mm) {
        if (mq) {   // balanced passes of <= 80 queries on k_scan_mq (profiles/r03_mq_time.txt)
            static const double tile_cost[6] = {0.0, 1.0, 1.17, 1.44, 1.72, 2.07};
            double c = 0.0;
            for (int left = mm; left > 0;) {
                const int mb = mq_pass_queries(left);
                c += tile_cost[(mb + 15) / 16];
                left -= mb;
            }
            return c;
        }
        const int rem = mm % 64;
        return 1.85 * (mm / 64) + (rem == 0 ? 0.0 : rem <= 16 ? 1.0 : rem <= 32 ? 1.12 : 1.85);
    };
    // 64-column tiles (k_gemm_dma<2,2,1>) where padding to 128 would waste half a tile.  Measured at
    // 100k x 4096 (tools/m_sweep2.py): a 128 x 64 tile pass runs at 91 TFLOP/s (half the reuse of the
    // query tile per barrier) = 1.61 units per 64 queries — no better than the 64-query scan pass
    // (1.85) for 33..64 queries, 13 % better for 129..192 (1.51 vs 1.74 ms at 192)
    static const int gemm64_env = getenv("CSLAM_GEMM64") ? atoi(getenv("CSLAM_GEMM64")) : 1;
    const bool narrow = gemm64_env && gemm_env != 32 && gemm_env != 4 && m > 128 && m <= 192;
    // a 128-query GEMM tile pass costs 3.05 units where the tiles fill whole waves of the chip (1024
    // queries: 124 TFLOP/s), 3.7 for one or two tiles (128 queries x 100k rows: 782 tiles on 512 slots)
    const double gemm_cost = narrow ? 1.8 * ((m + 63) / 64) : (m >= 384 ? 3.05 : 3.7) * ((m + 127) / 128);
    const bool use_gemm = gemm_env && m >= 96 && p->vn > 0 && gemm_cost < scan_cost(m);
    if (use_gemm) {
        // MFMA-bound regime: stage the whole chunk once (padded to whole query tiles), one GEMM
        const int gd_cols = narrow ? 64 : kGdCols;
        const int mq_pad = (m + gd_cols - 1) / gd_cols * gd_cols;
        CSLAM_TRY(p->qstage.reserve((size_t)(kQueryChunk + 128) * p->dp * sizeof(float)));
        const int vec_ok = ((reinterpret_cast<uintptr_t>(q_dev) & 15) == 0 && (q_stride & 3) == 0) ? 1 : 0;
        hipLaunchKernelGGL(k_stage_queries, dim3(mq_pad, 1), dim3(256), 0, st, q_dev, q_stride, p->dim,
                           m, mq_pad, p->dp, p->qstage.as<float>(), p->rnq.as<float>(), vec_ok,
                           (int64_t)0, 0, (int64_t)0);
        CSLAM_HIP_TRY(hipGetLastError());
        bool prof = prof_take(p);
        if (prof) CSLAM_HIP_TRY(hipEventRecord(p->ev[2 * p->prof_n], st));
        // CSLAM_GEMM=32 keeps the register-staged 128 x 128 kernel (A/B runs); the LDS-DMA kernel
        // needs dots rows it can store 16 bytes at a time (the sims workspace is 64-float aligned)
        const bool use_dma_gemm = gemm_env != 32 && (dots_stride & 3) == 0 &&
                                  (reinterpret_cast<uintptr_t>(dots) & 15) == 0;
        if (use_dma_gemm) {
            // default: 128 x 128 tiles, two blocks per CU; CSLAM_GEMM=4: 256 x 128, one per CU
            const int tm = gemm_env == 4 ? 4 : 2;
            const int n_colblk = mq_pad / gd_cols;
            auto launch = [&](int tmv, int64_t row0, int64_t nrows) -> int {
                const int rows = 64 * tmv;
                const int n_rowblk = (int)((nrows + rows - 1) / rows);
                const int64_t nblocks = (int64_t)((n_rowblk + 7) / 8) * 8 * n_colblk;
                const size_t lds = (size_t)(tmv == 4 ? 3 : 2) * (rows + gd_cols) * kGdK * 4;
                const float* xr = p->rows + (p->v0 + row0) * p->xs;
                float* dr = dots + row0;
#define CSLAM_GD_LAUNCH(TMV, SL, TJV)                                                              \
    do {                                                                                           \
        static bool attr_set[64] = {};                                                             \
        if (!attr_set[p->device & 63]) {                                                           \
            CSLAM_HIP_TRY(hipFuncSetAttribute((const void*)k_gemm_dma<TMV, SL, TJV>,               \
                                              hipFuncAttributeMaxDynamicSharedMemorySize,          \
                                              (int)lds));                                          \
            attr_set[p->device & 63] = true;                                                       \
        }                                                                                          \
        hipLaunchKernelGGL((k_gemm_dma<TMV, SL, TJV>), dim3((unsigned)nblocks), dim3(256), lds, st, \
                           xr, p->xs, nrows, p->dp, p->qstage.as<float>(), m, dr, dots_stride,     \
                           n_rowblk, n_colblk);                                                    \
    } while (0)
                if (tmv == 4) CSLAM_GD_LAUNCH(4, 3, 2);
                else if (tmv == 2 && narrow) CSLAM_GD_LAUNCH(2, 2, 1);
                else if (tmv == 2) CSLAM_GD_LAUNCH(2, 2, 2);
                else if (narrow) CSLAM_GD_LAUNCH(1, 2, 1);
                else CSLAM_GD_LAUNCH(1, 2, 2);
#undef CSLAM_GD_LAUNCH
                return CSLAM_OK;
            };
            // whole waves of tiles with the main shape, the remainder with half-height tiles
            const int rows = 64 * tm;
            const int64_t slots = (int64_t)p->cus * (tm == 4 ? 1 : 2);
            const int64_t n_rowblk = (p->vn + rows - 1) / rows;
            int64_t rb_main = n_rowblk;
            if (tm == 2 && slots % n_colblk == 0) {
                const int64_t full_waves = n_rowblk * n_colblk / slots;
                const int64_t rb = full_waves * slots / n_colblk;
                if (full_waves >= 1 && rb < n_rowblk) rb_main = rb;
            }
            const int64_t rows_main = std::min<int64_t>(p->vn, rb_main * rows);
            p->last_scan_kernel = tm == 4 ? "k_gemm_dma<TM=4,SLOTS=3>"
                                          : (narrow ? "k_gemm_dma<TM=2,SLOTS=2,TJ=1>" : "k_gemm_dma<TM=2,SLOTS=2>");
            CSLAM_TRY(launch(tm, 0, rows_main));
            if (rows_main < p->vn) CSLAM_TRY(launch(1, rows_main, p->vn - rows_main));
        } else {
            const int n_rowblk = (int)((p->vn + 127) / 128);
            const int n_colblk = mq_pad / 128;
            const int64_t nblocks = (int64_t)((n_rowblk + 7) / 8) * 8 * n_colblk;
            p->last_scan_kernel = "k_gemm32";
            hipLaunchKernelGGL(k_gemm32, dim3((unsigned)nblocks), dim3(256), 0, st,
                               p->rows + p->v0 * p->xs, p->xs, p->vn, p->dp, p->qstage.as<float>(), m,
                               dots, dots_stride, n_rowblk, n_colblk);
        }
        CSLAM_HIP_TRY(hipGetLastError());
        if (prof) {
            CSLAM_HIP_TRY(hipEventRecord(p->ev[2 * p->prof_n + 1], st));
            p->prof_n++;
        }
        return CSLAM_OK;
    }
    for (int q0 = 0, mb = 0; q0 < m; q0 += mb) {
        if (mq && m - q0 > 16) {
            // 17..80 queries per pass on the multi-tile LDS-DMA scan: the pool is read once
            mb = mq_pass_queries(m - q0);
            CSLAM_TRY(mq_pass(p, q_dev + (int64_t)q0 * q_stride, q_stride, mb, p->rnq.as<float>() + q0,
                              dots + (int64_t)q0 * dots_stride, dots_stride, false, nullptr, 0, -1,
                              nullptr, mq_plan(p, mb), st));
            continue;
        }
        mb = m - q0 < kMaxQueriesPerPass ? m - q0 : kMaxQueriesPerPass;
        int mq_pad = mb <= 16 ? 16 : (mb <= 32 ? 32 : 64);
        // the staging area is reused by every pass: stream order keeps pass i+1's staging
        // behind pass i's scan
        const float* qsrc = q_dev + (int64_t)q0 * q_stride;
        const int vec_ok = ((reinterpret_cast<uintptr_t>(qsrc) & 15) == 0 && (q_stride & 3) == 0) ? 1 : 0;
        p->q_copies = mb <= 16 ? qrep_max : 1;
        p->q_copy_stride = copy_stride;
        if (mb <= 16 && vec_ok && p->dim == p->dp && dma_scan_applies(p) && !ks_scan_applies(p, mb)) {
            // the LDS-DMA scan reads the queries where they are and leaves 1/||q|| for the select
            CSLAM_TRY(launch_scan(p, qsrc, mb, dots + (int64_t)q0 * dots_stride, dots_stride, st,
                                  q_stride, p->rnq.as<float>() + q0));
            continue;
        }
        hipLaunchKernelGGL(k_stage_queries, dim3(mq_pad, p->q_copies), dim3(256), 0, st, qsrc,
                           q_stride, p->dim, mb, mq_pad, p->dp, p->qstage.as<float>(),
                           p->rnq.as<float>() + q0, vec_ok, copy_stride, 0, (int64_t)0);
        CSLAM_HIP_TRY(hipGetLastError());
        CSLAM_TRY(launch_scan(p, p->qstage.as<float>(), mb, dots + (int64_t)q0 * dots_stride,
                              dots_stride, st));
    }
    return CSLAM_OK;
}

int select_from_candidates(cslam_pool* p, DevBuf* src, DevBuf* dst, int64_t n_in, int64_t in_stride,
                           int m, int k, int32_t* out_idx, float* out_sim, hipStream_t st);

int select_small_k(cslam_pool* p, const float* sims, int64_t sims_stride, int m, int k,
                   int32_t* out_idx, float* out_sim, hipStream_t st) {
    const float* rnx = p->inv_norm + p->v0;
    const float* rnq = p->rnq.as<float>();
    int64_t n_in = p->vn;
    int64_t parts = (n_in + kSelectPart - 1) / kSelectPart;
    if (parts <= 1) {
        hipLaunchKernelGGL((k_select<true, true>), dim3(1, m), dim3(256), 0, st, sims,
                           sims_stride, rnx, rnq, (uint32_t)p->v0, nullptr, (int64_t)0, n_in, k,
                           nullptr, (int64_t)0, out_idx, out_sim);
        CSLAM_HIP_TRY(hipGetLastError());
        return CSLAM_OK;
    }
    // level 1: sims → candidates
    int64_t cstride = parts * k;
    CSLAM_TRY(p->cand_a.reserve((size_t)m * cstride * sizeof(uint64_t)));
    hipLaunchKernelGGL((k_select<true, false>), dim3((unsigned)parts, m), dim3(256), 0, st, sims,
                       sims_stride, rnx, rnq, (uint32_t)p->v0, nullptr, (int64_t)0, n_in, k,
                       p->cand_a.as<uint64_t>(), cstride, nullptr, nullptr);
    CSLAM_HIP_TRY(hipGetLastError());
    return select_from_candidates(p, &p->cand_a, &p->cand_b, cstride, cstride, m, k, out_idx,
                                  out_sim, st);
}

// merge levels: candidates [m][in_stride] (n_in valid per query) → top-k
int select_from_candidates(cslam_pool* p, DevBuf* src, DevBuf* dst, int64_t n_in, int64_t in_stride,
                           int m, int k, int32_t* out_idx, float* out_sim, hipStream_t st) {
    int64_t parts;
    while (true) {
        parts = (n_in + kSelectPart - 1) / kSelectPart;
        if (parts <= 1) {
            hipLaunchKernelGGL((k_select<false, true>), dim3(1, m), dim3(256), 0, st, nullptr,
                               (int64_t)0, nullptr, nullptr, 0u, src->as<uint64_t>(), in_stride,
                               n_in, k, nullptr, (int64_t)0, out_idx, out_sim);
            CSLAM_HIP_TRY(hipGetLastError());
            return CSLAM_OK;
        }
        int64_t ostride = parts * k;
        CSLAM_TRY(dst->reserve((size_t)m * ostride * sizeof(uint64_t)));
        hipLaunchKernelGGL((k_select<false, false>), dim3((unsigned)parts, m), dim3(256), 0, st,
                           nullptr, (int64_t)0, nullptr, nullptr, 0u, src->as<uint64_t>(),
                           in_stride, n_in, k, dst->as<uint64_t>(), ostride, nullptr, nullptr);
        CSLAM_HIP_TRY(hipGetLastError());
        DevBuf* t = src;
        src = dst;
        dst = t;
        n_in = ostride;
        in_stride = ostride;
    }
}

int select_any_k(cslam_pool* p, const float* sims, int64_t sims_stride, int m, int k,
                 int32_t* out_idx, float* out_sim, hipStream_t st) {
    const float* rnx = p->inv_norm + p->v0;
    const float* rnq = p->rnq.as<float>();
    int64_t npow2 = 1;
    while (npow2 < p->vn) npow2 <<= 1;
    if (npow2 < 2) npow2 = 2;
    CSLAM_TRY(p->comp.reserve((size_t)m * npow2 * sizeof(uint64_t)));
    uint64_t* comp = p->comp.as<uint64_t>();
    hipLaunchKernelGGL(k_make_comp, dim3((unsigned)((npow2 + 255) / 256), m), dim3(256), 0, st,
                       sims, sims_stride, rnx, rnq, (uint32_t)p->v0, p->vn, npow2, comp);
    for (int64_t size = 2; size <= npow2; size <<= 1)
        for (int64_t stride = size >> 1; stride > 0; stride >>= 1)
            hipLaunchKernelGGL(k_bitonic_step, dim3((unsigned)((npow2 / 2 + 255) / 256), m),
                               dim3(256), 0, st, comp, npow2, size, stride);
    hipLaunchKernelGGL(k_emit_sorted, dim3((unsigned)((k + 255) / 256), m), dim3(256), 0, st, comp,
                       npow2, k, out_idx, out_sim);
    CSLAM_HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}

int append_rows_dev(cslam_pool* p, const float* rows_dev, int64_t n_rows, int64_t row_stride,
                    hipStream_t st) {
    CSLAM_TRY(pool_grow(p, p->n + n_rows, st));
    const int vec_ok = ((reinterpret_cast<uintptr_t>(rows_dev) & 15) == 0 && (row_stride & 3) == 0) ? 1 : 0;
    hipLaunchKernelGGL(k_append_rows, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, st, rows_dev,
                       row_stride, p->dim, p->rows, p->xs, p->dp, p->n, n_rows, p->inv_norm, vec_ok);
    CSLAM_HIP_TRY(hipGetLastError());
    p->n += n_rows;
    return CSLAM_OK;
}

// The completion word of a small host-pointer call (cslam_pool_search arms it): taken by the launch site of
// the call's FINAL kernel - one-shot, so that no other launch of the call signals early.
struct DoneArgs {
    uint32_t* word = nullptr;
    uint32_t seq = 0;
    uint32_t* ticket = nullptr;
};
DoneArgs take_done(cslam_pool* p, bool final_launch) {
    DoneArgs d;
    if (!final_launch || p->done_word == nullptr || p->done_taken || p->small_scratch.p == nullptr) return d;
    d.word = p->done_word;
    d.seq = p->small_seq;
    d.ticket = reinterpret_cast<uint32_t*>(p->small_scratch.as<uint64_t>() + 17);
    p->done_taken = true;
    return d;
}

// search_best (lcsm.py:49,69) of <= 16 queries over the row window [v0, v0+vn): the LDS-DMA scan
// with the fused arg-max epilogue; leaves one composite per query and wave in cand_a (and, with
// split >= 0 (window-relative row), the rows at or behind `split` apart in cand_c).  Queries are
// read in place when they need no padding.
int best_scan_candidates(cslam_pool* p, const float* q_dev, int64_t q_stride, int m, int64_t split,
                         int64_t* stride_out, int64_t* n_cand_out, hipStream_t st,
                         float* dots = nullptr, int64_t dots_stride = 0) {
    const bool direct = p->dim == p->dp && (reinterpret_cast<uintptr_t>(q_dev) & 15) == 0 &&
                        (q_stride & 3) == 0;
    const float* qsrc = q_dev;
    int64_t qs = q_stride;
    int q_rows = m;
    if (!direct) {
        CSLAM_TRY(p->qstage.reserve((size_t)(kQueryChunk + 128) * p->dp * sizeof(float)));
        CSLAM_TRY(p->rnq.reserve((size_t)(kQueryChunk + kMaxQueriesPerPass) * sizeof(float)));
        const int vec_ok = ((reinterpret_cast<uintptr_t>(q_dev) & 15) == 0 && (q_stride & 3) == 0) ? 1 : 0;
        hipLaunchKernelGGL(k_stage_queries, dim3(16, 1), dim3(256), 0, st, q_dev, q_stride,
                           p->dim, m, 16, p->dp, p->qstage.as<float>(), p->rnq.as<float>(),
                           vec_ok, (int64_t)0, 0, (int64_t)0);
        CSLAM_HIP_TRY(hipGetLastError());
        qsrc = p->qstage.as<float>();
        qs = p->dp;
        q_rows = 16;
    }
    const int64_t cstride = ((int64_t)p->cus * 4 * ((p->vn + (int64_t)p->cus * 4 * 128 - 1) /
                                                    ((int64_t)p->cus * 4 * 128)) + 63) / 64 * 64;
    CSLAM_TRY(p->cand_a.reserve((size_t)16 * cstride * sizeof(uint64_t)));
    if (split >= 0) CSLAM_TRY(p->cand_c.reserve((size_t)16 * cstride * sizeof(uint64_t)));
    bool prof = prof_take(p);
    if (prof) CSLAM_HIP_TRY(hipEventRecord(p->ev[2 * p->prof_n], st));
    // dots != nullptr (search with 2 <= k <= 16): the raw dots next to the composites, and the
    // query norms the kernel used in p->rnq for the select
    if (dots != nullptr) CSLAM_TRY(p->rnq.reserve((size_t)(kQueryChunk + kMaxQueriesPerPass) * sizeof(float)));
    CSLAM_TRY(launch_scan_dma(p, qsrc, qs, q_rows, m, dots, dots_stride, true, p->cand_a.as<uint64_t>(),
                              cstride, n_cand_out, st, dots != nullptr ? p->rnq.as<float>() : nullptr, split,
                              split >= 0 ? p->cand_c.as<uint64_t>() : nullptr, dots != nullptr));
    if (prof) {
        CSLAM_HIP_TRY(hipEventRecord(p->ev[2 * p->prof_n + 1], st));
        p->prof_n++;
    }
    *stride_out = cstride;
    return CSLAM_OK;
}

// ... and the final level: (row, similarity) per query, for both windows of a split search
int best_scan(cslam_pool* p, const float* q_dev, int64_t q_stride, int m, int64_t split,
              int32_t* idx_a, float* sim_a, int32_t* idx_c, float* sim_c, hipStream_t st,
              bool final_launch = false) {
    int64_t cstride = 0, n_cand = 0;
    CSLAM_TRY(best_scan_candidates(p, q_dev, q_stride, m, split, &cstride, &n_cand, st));
    const DoneArgs dn = take_done(p, final_launch);
    hipLaunchKernelGGL(k_best_pair, dim3(m, split >= 0 ? 2 : 1), dim3(256), 0, st,
                       p->cand_a.as<uint64_t>(), p->cand_c.as<uint64_t>(), cstride, n_cand, idx_a,
                       sim_a, idx_c, sim_c, dn.word, dn.seq, dn.ticket);
    CSLAM_HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}

// search_best of 17..~100 queries: k_scan_mq passes with the fused arg-max epilogue (one composite
// per query and panel), then k_best_pair per pass - no dots round trip.  Both windows of a split
// search (rows before / from `split`, window-relative) in the same pass.
bool mq_best_applies(const cslam_pool* p, int m) {
    // up to four passes: beyond that the 128-query GEMM tiles win even with their dots round trip and
    // select launches (cost model in compute_dots)
    return mq_scan_applies(p, m) && m <= 4 * 16 * mq_max_tiles();
}
int best_scan_mq(cslam_pool* p, const float* q_dev, int64_t q_stride, int m, int64_t split,
                 int32_t* idx_a, float* sim_a, int32_t* idx_c, float* sim_c, hipStream_t st,
                 bool final_call = false) {
    CSLAM_TRY(p->qstage.reserve((size_t)(kQueryChunk + 128) * p->dp * sizeof(float)));
    CSLAM_TRY(p->rnq.reserve((size_t)(kQueryChunk + kMaxQueriesPerPass) * sizeof(float)));
    for (int q0 = 0, mb = 0; q0 < m; q0 += mb) {
        mb = mq_pass_queries(m - q0);
        if (mb <= 16) {   // a last pass of <= 16 queries: the single-tile kernel
            CSLAM_TRY(best_scan(p, q_dev + (int64_t)q0 * q_stride, q_stride, mb, split, idx_a + q0, sim_a + q0,
                                idx_c ? idx_c + q0 : nullptr, sim_c ? sim_c + q0 : nullptr, st,
                                final_call && q0 + mb >= m));
            continue;
        }
        const MqPlan g = mq_plan(p, mb);
        const int64_t cstride = (g.n_cand + 63) / 64 * 64;
        CSLAM_TRY(p->cand_a.reserve((size_t)16 * g.tiles * cstride * sizeof(uint64_t)));
        if (split >= 0) CSLAM_TRY(p->cand_c.reserve((size_t)16 * g.tiles * cstride * sizeof(uint64_t)));
        CSLAM_TRY(mq_pass(p, q_dev + (int64_t)q0 * q_stride, q_stride, mb, p->rnq.as<float>() + q0, nullptr, 0,
                          true, p->cand_a.as<uint64_t>(), cstride, split,
                          split >= 0 ? p->cand_c.as<uint64_t>() : nullptr, g, st));
        const DoneArgs dn = take_done(p, final_call && q0 + mb >= m);
        hipLaunchKernelGGL(k_best_pair, dim3(mb, split >= 0 ? 2 : 1), dim3(256), 0, st,
                           p->cand_a.as<uint64_t>(), p->cand_c.as<uint64_t>(), cstride, g.n_cand,
                           idx_a + q0, sim_a + q0, idx_c ? idx_c + q0 : nullptr,
                           sim_c ? sim_c + q0 : nullptr, dn.word, dn.seq, dn.ticket);
        CSLAM_HIP_TRY(hipGetLastError());
    }
    return CSLAM_OK;
}

// search_best (k = 1) or search with 2 <= k <= 32 of a few queries over a small window in ONE launch
// (k_small_search).  CSLAM_SMALL_BEST=0: the general path (stage + scan + final level / rank select) - A/B
// runs and the tests that compare the two; CSLAM_SMALL_BEST=1: k = 1 only.
bool small_search_applies(const cslam_pool* p, int m, int k) {
    static Knob k_small{"CSLAM_SMALL_BEST", 2};
    const int nchunks = (p->dp + 511) / 512;
    // (24 MB of rows at most: 4000 x 4096-d is 1.7 us faster on the general path's LDS-DMA K-split scan,
    // 1000 x 4096-d 5 us slower, profiles/r04_small_best_ab.txt)
    return knob(k_small) != 0 && (k == 1 || (k >= 2 && k <= kTopkWavesMaxK && knob(k_small) >= 2)) && m >= 1 &&
           m <= 16 && p->vn >= 1 && p->vn <= kSmallBestMaxRows && nchunks <= kSmallBestMaxChunks &&
           (int64_t)m * p->dp <= kSmallBestMaxQFloats && p->vn * p->dp <= kSmallBestMaxFloats;
}
// (q may be any device-visible memory: HBM, or pinned host memory read over the host link)
int small_search(cslam_pool* p, const float* q, int64_t q_stride, int m, int k, int32_t* out_idx, float* out_sim,
                 hipStream_t st, uint32_t* done = nullptr, uint32_t seq = 0) {
    if (p->small_scratch.p == nullptr) {
        CSLAM_TRY(p->small_scratch.reserve(256));
        CSLAM_HIP_TRY(hipMemsetAsync(p->small_scratch.p, 0, 256, st));
    }
    const int nchunks = (p->dp + 511) / 512;
    const int spb = nchunks >= 4 ? 1 : 4 / nchunks;   // four (strip, chunk) units per block at least
    const int64_t strips = (p->vn + 15) / 16;
    const unsigned grid = (unsigned)((strips + spb - 1) / spb);
    const int vec_ok = ((reinterpret_cast<uintptr_t>(q) & 15) == 0 && (q_stride & 3) == 0) ? 1 : 0;
    const int64_t cstride = (strips + 63) / 64 * 64, dstride = (p->vn + 63) / 64 * 64;
    if (k > 1) {   // the strips' composites and the raw dots, for the last block's rank select
        CSLAM_TRY(p->cand_a.reserve((size_t)16 * cstride * sizeof(uint64_t)));
        CSLAM_TRY(p->sims.reserve((size_t)16 * dstride * sizeof(float)));
    }
    const bool prof = prof_take(p);
    if (prof) CSLAM_HIP_TRY(hipEventRecord(p->ev[2 * p->prof_n], st));
    const size_t lds = (size_t)spb * nchunks * 64 * sizeof(f32x4);
    uint64_t* best = p->small_scratch.as<uint64_t>();
    uint32_t* ticket = reinterpret_cast<uint32_t*>(best + 16);
    if (k == 1)
        hipLaunchKernelGGL((k_small_search<false>), dim3(grid), dim3(256), lds, st, p->rows + p->v0 * p->xs, p->xs,
                           p->vn, p->dp, q, q_stride, p->dim, m, vec_ok, p->inv_norm + p->v0, (uint32_t)p->v0, spb,
                           nchunks, best, ticket, out_idx, out_sim, done, seq, 1, (uint64_t*)nullptr, (int64_t)0,
                           (float*)nullptr, (int64_t)0);
    else
        hipLaunchKernelGGL((k_small_search<true>), dim3(grid), dim3(256), lds, st, p->rows + p->v0 * p->xs, p->xs,
                           p->vn, p->dp, q, q_stride, p->dim, m, vec_ok, p->inv_norm + p->v0, (uint32_t)p->v0, spb,
                           nchunks, best, ticket, out_idx, out_sim, done, seq, k, p->cand_a.as<uint64_t>(), cstride,
                           p->sims.as<float>(), dstride);
    CSLAM_HIP_TRY(hipGetLastError());
    if (prof) {
        CSLAM_HIP_TRY(hipEventRecord(p->ev[2 * p->prof_n + 1], st));
        p->prof_n++;
    }
    p->last_scan_kernel = k == 1 ? "k_small_best" : "k_small_topk";
    return CSLAM_OK;
}

int search_dev_impl(cslam_pool* p, const float* q_dev, int64_t q_stride, int m, int k, int32_t* out_idx,
                    float* out_sim, hipStream_t st, int64_t row_lo = 0, int64_t row_hi = -1);

// search with 2 <= k <= 32 of 17..~100 queries over the window [v0, v0 + vn): per pass one k_scan_mq
// launch (or its main + K-split forms) that leaves the best composite of every panel / strip AND the
// raw dots, then k_topk_from_parts over the rows of the k best parts - instead of the dots pass and
// two select levels over every row.
int topk_scan_mq(cslam_pool* p, const float* q_dev, int64_t q_stride, int m, int k, int32_t* out_idx,
                 float* out_sim, hipStream_t st) {
    const int64_t v0 = p->v0, vn = p->vn;
    const int64_t sims_stride = (vn + 63) / 64 * 64;
    CSLAM_TRY(p->qstage.reserve((size_t)(kQueryChunk + 128) * p->dp * sizeof(float)));
    CSLAM_TRY(p->rnq.reserve((size_t)(kQueryChunk + kMaxQueriesPerPass) * sizeof(float)));
    CSLAM_TRY(p->sims.reserve((size_t)16 * mq_max_tiles() * sims_stride * sizeof(float)));
    for (int q0 = 0, mb = 0; q0 < m; q0 += mb) {
        mb = mq_pass_queries(m - q0);
        if (mb <= 16) {   // a last pass of <= 16 queries: the single-tile path of search
            CSLAM_TRY(search_dev_impl(p, q_dev + (int64_t)q0 * q_stride, q_stride, mb, k, out_idx + (int64_t)q0 * k,
                                      out_sim + (int64_t)q0 * k, st, v0, v0 + vn));
            continue;
        }
        const MqPlan g = mq_plan(p, mb);
        const int64_t cstride = (g.n_cand + 63) / 64 * 64;
        if (g.n_cand > kTopkPartsMax) return fail(CSLAM_E_INVALID, "topk_scan_mq: %lld parts", (long long)g.n_cand);
        CSLAM_TRY(p->cand_a.reserve((size_t)16 * g.tiles * cstride * sizeof(uint64_t)));
        CSLAM_TRY(mq_pass(p, q_dev + (int64_t)q0 * q_stride, q_stride, mb, p->rnq.as<float>() + q0,
                          p->sims.as<float>(), sims_stride, true, p->cand_a.as<uint64_t>(), cstride, -1, nullptr,
                          g, st));
        const PanelParts parts{g.n_main,  vn,      g.n_blocks, g.p_max, g.waves, mq_panel_strips(g.tiles, g.waves),
                               g.n_blocks * g.p_max * g.waves};
        const DoneArgs dn = take_done(p, q0 + mb >= m);
        hipLaunchKernelGGL((k_topk_from_parts<PanelParts>), dim3(mb), dim3(256), 0, st, p->cand_a.as<uint64_t>(),
                           cstride, (int)g.n_cand, parts, p->sims.as<float>(), sims_stride, p->inv_norm + v0,
                           p->rnq.as<float>() + q0, (uint32_t)v0, vn, k, out_idx + (int64_t)q0 * k,
                           out_sim + (int64_t)q0 * k, dn.word, dn.seq, dn.ticket);
        CSLAM_HIP_TRY(hipGetLastError());
    }
    return CSLAM_OK;
}

int search_dev_impl(cslam_pool* p, const float* q_dev, int64_t q_stride, int m, int k, int32_t* out_idx,
                    float* out_sim, hipStream_t st, int64_t row_lo, int64_t row_hi) {
    if (m <= 0 || k <= 0) return CSLAM_OK;
    if (row_hi < 0 || row_hi > p->n) row_hi = p->n;
    if (row_lo < 0) row_lo = 0;
    p->v0 = row_lo;
    p->vn = row_hi > row_lo ? row_hi - row_lo : 0;
    if (p->vn == 0) {
        int64_t total = (int64_t)m * k;
        hipLaunchKernelGGL(k_fill_empty, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                           total, out_idx, out_sim);
        CSLAM_HIP_TRY(hipGetLastError());
        return CSLAM_OK;
    }
    if (small_search_applies(p, m, k)) return small_search(p, q_dev, q_stride, m, k, out_idx, out_sim, st);
    if (k == 1 && m <= 16 && dma_scan_applies(p) && !ks_scan_applies(p, m))
        return best_scan(p, q_dev, q_stride, m, -1, out_idx, out_sim, nullptr, nullptr, st, true);
    if (k == 1 && mq_best_applies(p, m))
        return best_scan_mq(p, q_dev, q_stride, m, -1, out_idx, out_sim, nullptr, nullptr, st, true);
    int64_t sims_stride = (p->vn + 63) / 64 * 64;
    const int64_t qc = query_chunk(sims_stride);
    int64_t chunk = m < qc ? m : qc;
    CSLAM_TRY(p->sims.reserve((size_t)chunk * sims_stride * sizeof(float)));
    float* sims = p->sims.as<float>();
    // 2 <= k <= 32 (lcsm.py:69's k = 10 at a handful of queries; the sharded step's fused intra + inter
    // search, K = block + nb_best_matches): the best-match scan that also leaves the raw dots, then ONE
    // select launch over the rows of the k best parts of the window instead of two select levels over
    // every row (k_topk_from_parts).  CSLAM_TOPK_WAVES=0 keeps the two-level select.
    static Knob k_tw{"CSLAM_TOPK_WAVES", 1};
    if (k >= 2 && k <= kTopkWavesMaxK && m <= 16 && dma_scan_applies(p) && !ks_scan_applies(p, m) &&
        (int64_t)p->cus * 4 <= kTopkWavesMax && knob(k_tw) != 0) {
        int64_t cstride = 0, n_cand = 0;
        CSLAM_TRY(best_scan_candidates(p, q_dev, q_stride, m, -1, &cstride, &n_cand, st, sims, sims_stride));
        const WaveParts parts{p->vn / n_cand, p->vn % n_cand};
        const DoneArgs dn = take_done(p, true);
        hipLaunchKernelGGL((k_topk_from_parts<WaveParts>), dim3(m), dim3(256), 0, st, p->cand_a.as<uint64_t>(),
                           cstride, (int)n_cand, parts, sims, sims_stride, p->inv_norm + p->v0,
                           p->rnq.as<float>(), (uint32_t)p->v0, p->vn, k, out_idx, out_sim, dn.word, dn.seq,
                           dn.ticket);
        CSLAM_HIP_TRY(hipGetLastError());
        return CSLAM_OK;
    }
    // ... and for 17+ queries: k_scan_mq passes that leave one composite per query and panel (or strip of
    // the K-split launch) beside the dots
    if (k >= 2 && k <= kTopkWavesMaxK && m > 16 && mq_best_applies(p, m) && knob(k_tw) != 0)
        return topk_scan_mq(p, q_dev, q_stride, m, k, out_idx, out_sim, st);
    for (int q0 = 0; q0 < m; q0 += (int)qc) {
        int mb = m - q0 < (int)qc ? m - q0 : (int)qc;
        CSLAM_TRY(compute_dots(p, q_dev + (int64_t)q0 * q_stride, q_stride, mb, sims, sims_stride,
                               st));
        if (k <= kSelectMaxK)
            CSLAM_TRY(select_small_k(p, sims, sims_stride, mb, k, out_idx + (int64_t)q0 * k,
                                     out_sim + (int64_t)q0 * k, st));
        else
            CSLAM_TRY(select_any_k(p, sims, sims_stride, mb, k, out_idx + (int64_t)q0 * k,
                                   out_sim + (int64_t)q0 * k, st));
    }
    return CSLAM_OK;
}

}  // namespace

extern "C" {

int cslam_pool_create(int dim, int64_t capacity_hint, int device_id, cslam_pool_t** out) {
    if (!out || dim <= 0 || capacity_hint < 0) return fail(CSLAM_E_INVALID, "pool_create: bad args");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(CSLAM_E_NODEVICE, "pool_create: no HIP device visible");
    if (device_id < 0 || device_id >= ndev)
        return fail(CSLAM_E_INVALID, "pool_create: device %d out of range (%d)", device_id, ndev);
    CSLAM_HIP_TRY(hipSetDevice(device_id));
    cslam_pool* p = new cslam_pool();
    p->dim = dim;
    {
        // padded row length: a multiple of 64 floats, the LDS-DMA scan's narrowest slice (16 — the
        // MFMA k-group — for descriptors of <= 16 floats, where 64 would quadruple the pool).
        // Zero padding never changes a bit of the result: fmaf(0, 0, acc) = acc, and a 512-float
        // fold boundary is never crossed by the extra zeros.  Payload TB/s of a 16-query
        // best-match step, 16- vs 64-float alignment: 100-d 2.69 -> 4.44, 300-d 3.59 -> 5.68,
        // 1000-d 3.79 -> 5.92, 33-d 1.33 -> 2.34 (profiles/r02_dim_align.txt).
        // CSLAM_DIM_ALIGN=16 keeps the tight padding (register-ring scan for such lengths).
        const char* e = getenv("CSLAM_DIM_ALIGN");
        const int al = (e && atoi(e) == 16) || dim <= 16 ? 16 : 64;
        p->dp = (int)((dim + al - 1) / al * al);
    }
    {
        // Row stride.  A stride that is a multiple of 4 KiB (1024-, 2048-, 4096-, 8192-d rows) lines
        // the 16 rows of a scan stage up on the same few HBM channels: 512 B of padding per row
        // (never read) spreads them — same process, alternating pools, 16 best-match queries over
        // 1.6 GB: 4096-d 0.2615 -> 0.2555 ms, 1024-d 0.289 -> 0.2575, 2048-d 0.2735 -> 0.2558,
        // 8192-d 0.2825 -> 0.2629 (tools/pad_ab.py, profiles/r02_pad_ab.txt).  CSLAM_ROW_PAD
        // (floats) overrides, 0 = dense rows.
        const char* e = getenv("CSLAM_ROW_PAD");
        int64_t pad = e ? atoll(e) : (((int64_t)p->dp * 4) % 4096 == 0 ? 128 : 0);
        p->xs = p->dp + (pad > 0 ? pad16(pad) : 0);
    }
    p->device = device_id;
    int rc = device_cu_count(device_id, &p->cus);
    if (rc != CSLAM_OK) {
        delete p;
        return rc;
    }
    hipError_t e = hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete p;
        return fail(CSLAM_E_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    e = hipEventCreateWithFlags(&p->dev_ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&p->app_ev, hipEventDisableTiming);
    if (e != hipSuccess) {
        cslam_pool_destroy(p);
        return fail(CSLAM_E_HIP, "hipEventCreate: %s", hipGetErrorString(e));
    }
    rc = pool_grow(p, capacity_hint > 0 ? capacity_hint : 1000, p->stream);
    if (rc != CSLAM_OK) {
        cslam_pool_destroy(p);
        return rc;
    }
    *out = p;
    return CSLAM_OK;
}

int cslam_pool_destroy(cslam_pool_t* p) {
    if (!p) return CSLAM_OK;
    (void)hipSetDevice(p->device);
    (void)dev_event_now(p);
    if (p->dev_pending) (void)hipEventSynchronize(p->dev_ev);
    if (p->stream) (void)hipStreamSynchronize(p->stream);
    if (p->app_pending) (void)hipEventSynchronize(p->app_ev);
    if (p->dev_ev) (void)hipEventDestroy(p->dev_ev);
    if (p->app_ev) (void)hipEventDestroy(p->app_ev);
    for (hipEvent_t e : p->ev) (void)hipEventDestroy(e);
    if (p->rows) (void)hipFree(p->rows);
    if (p->inv_norm) (void)hipFree(p->inv_norm);
    p->qstage.release();
    p->rnq.release();
    p->sims.release();
    p->cand_a.release();
    p->cand_b.release();
    p->cand_c.release();
    p->comp.release();
    p->part.release();
    p->small_scratch.release();
    p->h_q.release();
    p->h_idx.release();
    p->h_sim.release();
    p->pin_q.release();
    p->pin_out.release();
    p->pin_rows.release();
    p->h_rows.release();
    if (p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
    return CSLAM_OK;
}

int cslam_pool_add(cslam_pool_t* p, const float* rows, int64_t n_rows) {
    if (!p || (!rows && n_rows > 0) || n_rows < 0) return fail(CSLAM_E_INVALID, "pool_add: bad args");
    if (n_rows == 0) return CSLAM_OK;
    CSLAM_HIP_TRY(hipSetDevice(p->device));
    CSLAM_TRY(order_after_dev(p, p->stream));
    CSLAM_TRY(pool_grow(p, p->n + n_rows, p->stream));
    // Small appends (add_item: one row per keyframe): the rows go into a pinned ring that the
    // append kernel reads in place (device-visible host memory) — no copy call, no
    // synchronisation; the ring is drained when it wraps.  Later calls on this handle's stream are
    // ordered behind the kernel, calls on other streams through the handle's event.
    const size_t small_bytes = (size_t)n_rows * p->dim * sizeof(float);
    constexpr size_t kPinRing = (size_t)1 << 20;
    if (small_bytes <= kPinRing / 8) {
        CSLAM_TRY(p->pin_rows.reserve(kPinRing));
        const size_t need = (small_bytes + 255) / 256 * 256;
        if (p->pin_rows_used + need > kPinRing) {
            CSLAM_HIP_TRY(hipStreamSynchronize(p->stream));   // every earlier append has read its slot
            p->pin_rows_used = 0;
        }
        float* slot = reinterpret_cast<float*>(static_cast<char*>(p->pin_rows.p) + p->pin_rows_used);
        memcpy(slot, rows, small_bytes);
        p->pin_rows_used += need;
        hipLaunchKernelGGL(k_append_rows, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, p->stream, slot,
                           (int64_t)p->dim, p->dim, p->rows, p->xs, p->dp, p->n, n_rows, p->inv_norm,
                           (p->dim & 3) == 0 ? 1 : 0);
        CSLAM_HIP_TRY(hipGetLastError());
        p->n += n_rows;
        return mark_dev(p, p->stream);
    }
    // dense host rows → device staging with plain 1-D copies (a strided 2-D copy out of pageable
    // memory cost ~200 us for ONE 16 KB row), then the device-side pad-and-append of add_dev;
    // staged in chunks of at most 64 MB
    const int64_t chunk_rows = std::max<int64_t>(1, (int64_t)(64u << 20) / ((int64_t)p->dim * 4));
    for (int64_t done = 0; done < n_rows; done += chunk_rows) {
        const int64_t c = std::min(chunk_rows, n_rows - done);
        const size_t bytes = (size_t)c * p->dim * sizeof(float);
        CSLAM_TRY(p->h_rows.reserve(bytes));
        CSLAM_HIP_TRY(hipMemcpyAsync(p->h_rows.p, rows + done * p->dim, bytes, hipMemcpyHostToDevice,
                                     p->stream));
        hipLaunchKernelGGL(k_append_rows, dim3((unsigned)((c + 3) / 4)), dim3(256), 0, p->stream,
                           p->h_rows.as<float>(), (int64_t)p->dim, p->dim, p->rows, p->xs, p->dp,
                           p->n + done, c, p->inv_norm, (p->dim & 3) == 0 ? 1 : 0);
        CSLAM_HIP_TRY(hipGetLastError());
        // the staging buffer is reused by the next chunk
        if (done + c < n_rows) CSLAM_HIP_TRY(hipStreamSynchronize(p->stream));
    }
    CSLAM_HIP_TRY(hipStreamSynchronize(p->stream));
    p->dev_pending = false;  // the stream waited for it and has drained
    p->n += n_rows;
    return CSLAM_OK;
}

int cslam_pool_add_dev(cslam_pool_t* p, const float* rows_dev, int64_t n_rows, int64_t row_stride,
                       void* stream) {
    if (!p || (!rows_dev && n_rows > 0) || n_rows < 0 || row_stride < (p ? p->dim : 0))
        return fail(CSLAM_E_INVALID, "pool_add_dev: bad args");
    if (n_rows == 0) return CSLAM_OK;
    CSLAM_HIP_TRY(hipSetDevice(p->device));
    hipStream_t st = (hipStream_t)stream;  // NULL = the default (null) stream
    CSLAM_TRY(order_after_dev(p, st));
    CSLAM_TRY(append_rows_dev(p, rows_dev, n_rows, row_stride, st));
    return mark_dev(p, st);
}

int64_t cslam_pool_size(const cslam_pool_t* p) { return p ? p->n : 0; }
int64_t cslam_pool_capacity(const cslam_pool_t* p) { return p ? p->cap : 0; }
int cslam_pool_dim(const cslam_pool_t* p) { return p ? p->dim : 0; }
int cslam_pool_device(const cslam_pool_t* p) { return p ? p->device : 0; }

int cslam_pool_read_rows(const cslam_pool_t* p, int64_t first, int64_t n, float* out) {
    if (!p || first < 0 || n < 0 || first + n > p->cap || (!out && n > 0))
        return fail(CSLAM_E_INVALID, "pool_read_rows: range [%lld,+%lld) outside capacity %lld",
                    (long long)first, (long long)n, (long long)(p ? p->cap : 0));
    if (n == 0) return CSLAM_OK;
    CSLAM_HIP_TRY(hipSetDevice(p->device));
    CSLAM_TRY(dev_event_now(const_cast<cslam_pool_t*>(p)));
    if (p->dev_pending) CSLAM_HIP_TRY(hipEventSynchronize(p->dev_ev));
    if (p->app_pending) CSLAM_HIP_TRY(hipEventSynchronize(p->app_ev));
    CSLAM_HIP_TRY(hipStreamSynchronize(p->stream));
    CSLAM_HIP_TRY(hipMemcpy2D(out, (size_t)p->dim * sizeof(float), p->rows + first * p->xs,
                              (size_t)p->xs * sizeof(float), (size_t)p->dim * sizeof(float),
                              (size_t)n, hipMemcpyDeviceToHost));
    return CSLAM_OK;
}

int cslam_pool_read_inv_norms(const cslam_pool_t* p, int64_t first, int64_t n, float* out) {
    if (!p || first < 0 || n < 0 || first + n > p->cap || (!out && n > 0))
        return fail(CSLAM_E_INVALID, "pool_read_inv_norms: bad range");
    if (n == 0) return CSLAM_OK;
    CSLAM_HIP_TRY(hipSetDevice(p->device));
    CSLAM_TRY(dev_event_now(const_cast<cslam_pool_t*>(p)));
    if (p->dev_pending) CSLAM_HIP_TRY(hipEventSynchronize(p->dev_ev));
    if (p->app_pending) CSLAM_HIP_TRY(hipEventSynchronize(p->app_ev));
    CSLAM_HIP_TRY(hipStreamSynchronize(p->stream));
    CSLAM_HIP_TRY(hipMemcpy(out, p->inv_norm + first, (size_t)n * sizeof(float),
                            hipMemcpyDeviceToHost));
    return CSLAM_OK;
}

int cslam_pool_device_view(const cslam_pool_t* p, const float** rows_dev, int64_t* row_stride) {
    if (!p || !rows_dev || !row_stride) return fail(CSLAM_E_INVALID, "pool_device_view: bad args");
    *rows_dev = p->rows;
    *row_stride = p->xs;
    return CSLAM_OK;
}

int cslam_pool_search_dev(cslam_pool_t* p, const float* queries_dev, int64_t query_stride,
                          int n_queries, int k, int32_t* out_idx_dev, float* out_sim_dev,
                          void* stream) {
    if (!p || n_queries < 0 || k < 0 || (n_queries > 0 && (!queries_dev || query_stride < p->dim)) ||
        ((int64_t)n_queries * k > 0 && (!out_idx_dev || !out_sim_dev)))
        return fail(CSLAM_E_INVALID, "pool_search_dev: bad args");
    CSLAM_HIP_TRY(hipSetDevice(p->device));
    hipStream_t st = (hipStream_t)stream;  // NULL = the default (null) stream
    CSLAM_TRY(order_after_dev(p, st));
    CSLAM_TRY(search_dev_impl(p, queries_dev, query_stride, n_queries, k, out_idx_dev, out_sim_dev,
                              st));
    return mark_dev(p, st);
}

int cslam_pool_search_range_dev(cslam_pool_t* p, const float* queries_dev, int64_t query_stride,
                                int n_queries, int k, int64_t row_lo, int64_t row_hi,
                                int32_t* out_idx_dev, float* out_sim_dev, void* stream) {
    if (!p || n_queries < 0 || k < 0 || row_lo < 0 || row_hi < row_lo || row_hi > p->n ||
        (n_queries > 0 && (!queries_dev || query_stride < p->dim)) ||
        ((int64_t)n_queries * k > 0 && (!out_idx_dev || !out_sim_dev)))
        return fail(CSLAM_E_INVALID, "pool_search_range_dev: bad args");
    CSLAM_HIP_TRY(hipSetDevice(p->device));
    hipStream_t st = (hipStream_t)stream;  // NULL = the default (null) stream
    CSLAM_TRY(order_after_dev(p, st));
    CSLAM_TRY(search_dev_impl(p, queries_dev, query_stride, n_queries, k, out_idx_dev, out_sim_dev,
                              st, row_lo, row_hi));
    return mark_dev(p, st);
}

int cslam_pool_search_best_split_dev(cslam_pool_t* p, const float* queries_dev, int64_t query_stride,
                                     int n_queries, int64_t split_row, int64_t row_hi,
                                     int32_t* idx_a_dev, float* sim_a_dev, int32_t* idx_c_dev,
                                     float* sim_c_dev, void* stream) {
    if (p && row_hi < 0) row_hi = p->n;
    if (!p || n_queries < 0 || split_row < 0 || split_row > row_hi || row_hi > p->n ||
        (n_queries > 0 && (!queries_dev || query_stride < p->dim || !idx_a_dev || !sim_a_dev ||
                           !idx_c_dev || !sim_c_dev)))
        return fail(CSLAM_E_INVALID, "pool_search_best_split_dev: bad args");
    if (n_queries == 0) return CSLAM_OK;
    CSLAM_HIP_TRY(hipSetDevice(p->device));
    hipStream_t st = (hipStream_t)stream;  // NULL = the default (null) stream
    CSLAM_TRY(order_after_dev(p, st));
    p->v0 = 0;
    p->vn = row_hi;
    if (split_row > 0 && split_row < row_hi && n_queries > 16 && mq_best_applies(p, n_queries)) {
        // 17..80 queries per pass, both sides in the same pass
        CSLAM_TRY(best_scan_mq(p, queries_dev, query_stride, n_queries, split_row, idx_a_dev, sim_a_dev,
                               idx_c_dev, sim_c_dev, st));
    } else if (split_row > 0 && split_row < row_hi && dma_scan_applies(p)) {
        // one scan launch per 16 queries answers both sides
        for (int q0 = 0; q0 < n_queries; q0 += 16) {
            const int mb = n_queries - q0 < 16 ? n_queries - q0 : 16;
            p->v0 = 0;
            p->vn = row_hi;
            CSLAM_TRY(best_scan(p, queries_dev + (int64_t)q0 * query_stride, query_stride, mb,
                                split_row, idx_a_dev + q0, sim_a_dev + q0, idx_c_dev + q0,
                                sim_c_dev + q0, st));
        }
    } else {
        CSLAM_TRY(search_dev_impl(p, queries_dev, query_stride, n_queries, 1, idx_a_dev, sim_a_dev, st,
                                  0, split_row));
        CSLAM_TRY(search_dev_impl(p, queries_dev, query_stride, n_queries, 1, idx_c_dev, sim_c_dev, st,
                                  split_row, row_hi));
    }
    return mark_dev(p, st);
}

int cslam_pool_similarities_dev(cslam_pool_t* p, const float* queries_dev, int64_t query_stride,
                                int n_queries, float* sims_dev, int64_t sims_stride,
                                void* stream) {
    if (!p || n_queries < 0 || (n_queries > 0 && (!queries_dev || query_stride < p->dim || !sims_dev)) ||
        sims_stride < p->n)
        return fail(CSLAM_E_INVALID, "pool_similarities_dev: bad args");
    if (n_queries == 0 || p->n == 0) return CSLAM_OK;
    CSLAM_HIP_TRY(hipSetDevice(p->device));
    hipStream_t st = (hipStream_t)stream;  // NULL = the default (null) stream
    CSLAM_TRY(order_after_dev(p, st));
    p->v0 = 0;
    p->vn = p->n;
    for (int q0 = 0; q0 < n_queries; q0 += (int)kQueryChunk) {
        int mb = n_queries - q0 < (int)kQueryChunk ? n_queries - q0 : (int)kQueryChunk;
        float* out = sims_dev + (int64_t)q0 * sims_stride;
        CSLAM_TRY(compute_dots(p, queries_dev + (int64_t)q0 * query_stride, query_stride, mb, out,
                               sims_stride, st));
        hipLaunchKernelGGL(k_scale, dim3((unsigned)((p->n + 255) / 256), mb), dim3(256), 0, st, out,
                           sims_stride, p->n, p->inv_norm, p->rnq.as<float>());
        CSLAM_HIP_TRY(hipGetLastError());
    }
    return mark_dev(p, st);
}

namespace {
// (both buffers come from hipHostMalloc / hipMalloc: 16-byte aligned)
int pinned_queries_to_device(cslam_pool* p, size_t qbytes) {
    static Knob k_copy{"CSLAM_HOST_Q_COPY", 0};
    if (knob(k_copy) != 0) {
        CSLAM_HIP_TRY(hipMemcpyAsync(p->h_q.p, p->pin_q.p, qbytes, hipMemcpyHostToDevice, p->stream));
        return CSLAM_OK;
    }
    const int64_t n4 = (int64_t)(qbytes / 16);
    const int n_tail = (int)((qbytes % 16) / 4);
    int64_t blocks = (n4 + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 256 ? 256 : blocks);
    hipLaunchKernelGGL(k_copy_f4, dim3((unsigned)blocks), dim3(256), 0, p->stream, p->pin_q.as<f32x4>(),
                       p->h_q.as<f32x4>(), n4, p->pin_q.as<float>() + n4 * 4, p->h_q.as<float>() + n4 * 4, n_tail);
    CSLAM_HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}
}  // namespace

int cslam_pool_search(cslam_pool_t* p, const float* queries, int n_queries, int k,
                      int32_t* out_idx, float* out_sim, int32_t* out_count) {
    if (!p || n_queries < 0 || k < 0 || (n_queries > 0 && !queries) ||
        ((int64_t)n_queries * k > 0 && (!out_idx || !out_sim)))
        return fail(CSLAM_E_INVALID, "pool_search: bad args");
    if (out_count)
        for (int q = 0; q < n_queries; ++q) out_count[q] = (int32_t)(k < p->n ? k : p->n);
    if (n_queries == 0 || k == 0) return CSLAM_OK;
    CSLAM_HIP_TRY(hipSetDevice(p->device));
    size_t qbytes = (size_t)n_queries * p->dim * sizeof(float);
    size_t obytes = (size_t)n_queries * k;
    CSLAM_TRY(p->h_q.reserve(qbytes));
    CSLAM_TRY(order_after_dev(p, p->stream));
    // Small calls (the per-keyframe pattern of the reference): queries through a pinned bounce
    // buffer (a pageable hipMemcpyAsync stages inside the runtime, ~30 us for 256 KB), results
    // written by the select kernels STRAIGHT into pinned host memory (device-visible under the
    // same address) — no device-to-host copies at all, one stream synchronisation.
    const bool small = qbytes <= ((size_t)8 << 20) && obytes <= ((size_t)1 << 16);
    if (small) {
        const size_t sim_off = (obytes * sizeof(int32_t) + 63) / 64 * 64;
        const size_t done_off = (sim_off + obytes * sizeof(float) + 63) / 64 * 64;
        CSLAM_TRY(p->pin_q.reserve(qbytes));
        CSLAM_TRY(p->pin_out.reserve(done_off + 64));
        memcpy(p->pin_q.p, queries, qbytes);
        int32_t* idx_pin = p->pin_out.as<int32_t>();
        float* sim_pin = reinterpret_cast<float*>(static_cast<char*>(p->pin_out.p) + sim_off);
        p->v0 = 0;
        p->vn = p->n;
        // Completion: the call's final kernel writes this call's number into a pinned word behind its results
        // (signal_done / k_small_search) and the host polls it - a stream synchronisation costs ~4 us more.
        // Launch sites that do not signal (two-level selects, GEMM tiles, ...) leave done_taken false and the
        // call synchronises the stream as before.  CSLAM_SMALL_BEST_POLL=0: always synchronise (A/B).
        // Polled up to 512 KiB of queries (=2: always, =0: never).  With the queries moved by hipMemcpyAsync
        // (CSLAM_HOST_Q_COPY=1, rounds 1-3) polling paid only up to 32 KiB: the next call's copy operation, queued
        // on a stream the runtime has not seen idle, made 16 x 4096-d calls 6 us and 80 x 4096-d calls 12 us
        // SLOWER; with the copy kernel (pinned_queries_to_device) the polled call wins or ties at every size
        // measured (16 x 4096-d: 276.5 -> 268.9 us; 80 x 4096-d: +0.5 %, hence the bound) - profiles/r04_host_call_ab_3.txt
        static Knob k_poll{"CSLAM_SMALL_BEST_POLL", 1};
        static Knob k_qcopy{"CSLAM_HOST_Q_COPY", 0};
        const size_t poll_max = knob(k_qcopy) != 0 ? ((size_t)32 << 10) : ((size_t)512 << 10);
        const bool poll = knob(k_poll) == 2 || (knob(k_poll) != 0 && qbytes <= poll_max);
        if (p->small_scratch.p == nullptr) {
            CSLAM_TRY(p->small_scratch.reserve(256));
            CSLAM_HIP_TRY(hipMemsetAsync(p->small_scratch.p, 0, 256, p->stream));
        }
        uint32_t* done = reinterpret_cast<uint32_t*>(static_cast<char*>(p->pin_out.p) + done_off);
        const uint32_t seq = ++p->small_seq ? p->small_seq : ++p->small_seq;   // never 0
        __atomic_store_n(done, 0u, __ATOMIC_RELAXED);
        bool signalled = false;
        if (small_search_applies(p, n_queries, k)) {
            // A small pool - the reference's per-keyframe calls at its own scale - is ONE launch
            // (k_small_search).  Queries of up to 1 KiB are read by the kernel from the pinned buffer itself
            // (no copy operation in front of the launch; larger ones would cross the host link once per
            // block).  CSLAM_SMALL_BEST_Q=0/1: A/B.
            static Knob k_q{"CSLAM_SMALL_BEST_Q", -1};
            const bool from_pin = knob(k_q) < 0 ? qbytes <= 1024 : knob(k_q) != 0;
            if (!from_pin)
                CSLAM_TRY(pinned_queries_to_device(p, qbytes));
            CSLAM_TRY(small_search(p, from_pin ? p->pin_q.as<float>() : p->h_q.as<float>(), p->dim, n_queries, k,
                                   idx_pin, sim_pin, p->stream, poll ? done : nullptr, seq));
            signalled = poll;
        } else {
            CSLAM_TRY(pinned_queries_to_device(p, qbytes));
            p->done_word = poll ? done : nullptr;
            p->done_taken = false;
            const int rc = search_dev_impl(p, p->h_q.as<float>(), p->dim, n_queries, k, idx_pin, sim_pin, p->stream);
            p->done_word = nullptr;
            signalled = p->done_taken;
            if (rc != CSLAM_OK) return rc;
        }
        bool seen = false;
        if (signalled) {
            // (a kernel that never writes the word - a fault - ends in the synchronisation below, which reports it;
            // long scans: poll for as long as the scan may take)
            const auto t0 = std::chrono::steady_clock::now();
            for (int spins = 1; !seen; ++spins) {
                seen = __atomic_load_n(done, __ATOMIC_ACQUIRE) == seq;
                if (!seen && (spins & 255) == 0 &&
                    std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20))
                    break;
            }
        }
        // (every 4096th call synchronises whatever it saw: the runtime gets to retire the stream's commands for good)
        if (!seen || (seq & 4095u) == 0) CSLAM_HIP_TRY(hipStreamSynchronize(p->stream));
        memcpy(out_idx, idx_pin, obytes * sizeof(int32_t));
        memcpy(out_sim, sim_pin, obytes * sizeof(float));
        p->dev_pending = false;
        return CSLAM_OK;
    }
    CSLAM_TRY(p->h_idx.reserve(obytes * sizeof(int32_t)));
    CSLAM_TRY(p->h_sim.reserve(obytes * sizeof(float)));
    CSLAM_HIP_TRY(hipMemcpyAsync(p->h_q.p, queries, qbytes, hipMemcpyHostToDevice, p->stream));
    CSLAM_TRY(search_dev_impl(p, p->h_q.as<float>(), p->dim, n_queries, k, p->h_idx.as<int32_t>(),
                              p->h_sim.as<float>(), p->stream));
    CSLAM_HIP_TRY(hipMemcpyAsync(out_idx, p->h_idx.p, obytes * sizeof(int32_t),
                                 hipMemcpyDeviceToHost, p->stream));
    CSLAM_HIP_TRY(hipMemcpyAsync(out_sim, p->h_sim.p, obytes * sizeof(float), hipMemcpyDeviceToHost,
                                 p->stream));
    CSLAM_HIP_TRY(hipStreamSynchronize(p->stream));
    p->dev_pending = false;
    return CSLAM_OK;
}

int cslam_pool_profile_begin(cslam_pool_t* p, int max_launches) {
    if (!p || max_launches <= 0) return fail(CSLAM_E_INVALID, "profile_begin: bad args");
    CSLAM_HIP_TRY(hipSetDevice(p->device));
    while ((int)p->ev.size() < 2 * max_launches) {
        hipEvent_t e;
        CSLAM_HIP_TRY(hipEventCreate(&e));
        p->ev.push_back(e);
    }
    p->prof_max = max_launches;
    p->prof_n = 0;
    p->prof_every = 1;
    p->prof_tick = 0;
    p->prof_on = true;
    return CSLAM_OK;
}

int cslam_pool_profile_begin_sampled(cslam_pool_t* p, int max_samples, int every) {
    if (every < 1) return fail(CSLAM_E_INVALID, "profile_begin_sampled: every < 1");
    CSLAM_TRY(cslam_pool_profile_begin(p, max_samples));
    p->prof_every = every;
    return CSLAM_OK;
}


int cslam_knobs_reload(void) {
    knobs_invalidate();
    return CSLAM_OK;
}

// The ordering event of the device-pointer API is recorded lazily (mark_dev): by the first later call that
// needs it, on the stream the last *_dev call went to - i.e. behind whatever the caller has queued there
// since.  A caller that is about to queue unrelated work on that stream, or to destroy it, records it here.
int cslam_pool_flush_dev(cslam_pool_t* p) {
    if (!p) return fail(CSLAM_E_INVALID, "pool_flush_dev: bad args");
    CSLAM_HIP_TRY(hipSetDevice(p->device));
    return dev_event_now(p);
}

const char* cslam_pool_scan_kernel(const cslam_pool_t* p) { return p ? p->last_scan_kernel : ""; }

int cslam_pool_profile_end(cslam_pool_t* p, double* avg_scan_ms, int* n_launches) {
    if (!p || !avg_scan_ms || !n_launches) return fail(CSLAM_E_INVALID, "profile_end: bad args");
    CSLAM_HIP_TRY(hipSetDevice(p->device));
    p->prof_on = false;
    double total = 0.0;
    for (int i = 0; i < p->prof_n; ++i) {
        CSLAM_HIP_TRY(hipEventSynchronize(p->ev[2 * i + 1]));
        float ms = 0.f;
        CSLAM_HIP_TRY(hipEventElapsedTime(&ms, p->ev[2 * i], p->ev[2 * i + 1]));
        total += ms;
    }
    *n_launches = p->prof_n;
    *avg_scan_ms = p->prof_n > 0 ? total / p->prof_n : 0.0;
    return CSLAM_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// internal (not ABI, csrc/shard.hip): the scan half of the split best-match search of <= 16
// queries over rows [0, row_hi) — stops at one composite per query, window and wave, so that
// the caller's own kernel can reduce them and go on (match records) in the same launch.
// *applied = false (nothing launched) where the LDS-DMA scan does not run.
// ---------------------------------------------------------------------------------------------
namespace cslam {
int pool_append_on(cslam_pool_t* p, const float* rows_dev, int64_t n_rows, int64_t row_stride,
                   hipStream_t st) {
    if (!p || (!rows_dev && n_rows > 0) || n_rows < 0 || row_stride < (p ? p->dim : 0))
        return fail(CSLAM_E_INVALID, "pool_append_on: bad args");
    if (n_rows == 0) return CSLAM_OK;
    CSLAM_HIP_TRY(hipSetDevice(p->device));
    // Not ordered behind earlier scans (they read rows below their own window only, and a growth
    // synchronises the device before it copies); every LATER pool call on another stream is
    // ordered behind this append through app_ev (order_after_dev).
    CSLAM_TRY(append_rows_dev(p, rows_dev, n_rows, row_stride, st));
    CSLAM_HIP_TRY(hipEventRecord(p->app_ev, st));
    p->app_stream = st;
    p->app_pending = true;
    return CSLAM_OK;
}
void pool_internal_scope(cslam_pool_t* p, bool on) {
    tls_shard_scope = on ? p : nullptr;
}
int pool_order_on(cslam_pool_t* p, hipStream_t st) {
    if (!p) return fail(CSLAM_E_INVALID, "pool_order_on: null");
    CSLAM_HIP_TRY(hipSetDevice(p->device));
    return mark_dev(p, st);
}
// CUs a multi-tile scan of `m` queries over `rows` rows leaves to other streams (0: none, or another kernel)
int pool_mq_free_cus(const cslam_pool_t* p, int m, int64_t rows) {
    if (!p || rows <= 0 || m <= 16 || m > 16 * mq_max_tiles() || p->dp % 64 != 0) return 0;
    const MqPlan g = mq_plan(p, m, rows, 1);   // as the shard's match() will plan it
    // The all-K-split pass puts a block on every CU, but its waves take well under half of a SIMD's
    // registers (188-224 of 512): the exchange stream's small kernels run beside it (measured: pack,
    // append, unpack of the next step inside the 151-us scan, profiles/r04_bench_cfg4_timeline.txt) -
    // reported as an eighth of the chip, enough for the caller to stage at publish time.
    if (g.n_main == 0 && g.tail_strips > 0) return p->cus / 8;
    return g.tail_strips == 0 && g.n_main > 0 ? p->cus - g.n_blocks : 0;
}
int64_t pool_staged_floats(const cslam_pool_t* p, int m) {
    return p && m > 0 ? (int64_t)((m + 15) / 16 * 16) * p->dp : 0;
}
int pool_stage_queries_on(cslam_pool_t* p, const float* queries_dev, int64_t query_stride, int m,
                          float* staged, float* staged_rnq, hipStream_t st, int group_rows, int64_t group_stride) {
    if (!p || !queries_dev || !staged || !staged_rnq || m < 1 || query_stride < p->dim || group_rows < 0 ||
        (group_rows > 0 && group_stride < (int64_t)group_rows * query_stride))
        return fail(CSLAM_E_INVALID, "pool_stage_queries_on: bad args");
    CSLAM_HIP_TRY(hipSetDevice(p->device));
    const int mq_pad = (m + 15) / 16 * 16;
    const int vec_ok = ((reinterpret_cast<uintptr_t>(queries_dev) & 15) == 0 && (query_stride & 3) == 0 &&
                        (group_stride & 3) == 0) ? 1 : 0;
    hipLaunchKernelGGL(k_stage_queries, dim3(mq_pad, 1), dim3(256), 0, st, queries_dev, query_stride, p->dim, m,
                       mq_pad, p->dp, staged, staged_rnq, vec_ok, (int64_t)0, group_rows, group_stride);
    CSLAM_HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}
int pool_best_split_candidates(cslam_pool_t* p, const float* queries_dev, int64_t query_stride, int m,
                               int64_t split_row, int64_t row_hi, hipStream_t st,
                               BestCandidates* out, bool* applied, const float* staged,
                               const float* staged_rnq) {
    *applied = false;
    if (!p || m < 1 || m > 16 * mq_max_tiles() || split_row <= 0 || split_row >= row_hi || row_hi > p->n ||
        !queries_dev || query_stride < p->dim)
        return CSLAM_OK;
    CSLAM_HIP_TRY(hipSetDevice(p->device));
    p->v0 = 0;
    p->vn = row_hi;
    if (m > 16) {   // one k_scan_mq pass: 17..80 gathered descriptors (8 robots x 10 per message)
        if (!mq_scan_applies(p, m)) return CSLAM_OK;
        CSLAM_TRY(order_after_dev(p, st));
        CSLAM_TRY(p->qstage.reserve((size_t)(kQueryChunk + 128) * p->dp * sizeof(float)));
        CSLAM_TRY(p->rnq.reserve((size_t)(kQueryChunk + kMaxQueriesPerPass) * sizeof(float)));
        const MqPlan g = mq_plan(p, m);
        const int64_t cstride = (g.n_cand + 63) / 64 * 64;
        CSLAM_TRY(p->cand_a.reserve((size_t)16 * g.tiles * cstride * sizeof(uint64_t)));
        CSLAM_TRY(p->cand_c.reserve((size_t)16 * g.tiles * cstride * sizeof(uint64_t)));
        const bool pre = staged != nullptr && staged_rnq != nullptr;
        CSLAM_TRY(mq_pass(p, queries_dev, query_stride, m,
                          pre ? const_cast<float*>(staged_rnq) : p->rnq.as<float>(), nullptr, 0, true,
                          p->cand_a.as<uint64_t>(), cstride, split_row, p->cand_c.as<uint64_t>(), g, st,
                          pre ? staged : nullptr));
        out->a = p->cand_a.as<uint64_t>();
        out->c = p->cand_c.as<uint64_t>();
        out->stride = cstride;
        out->n = g.n_cand;
        *applied = true;
        return mark_dev(p, st);
    }
    if (!dma_scan_applies(p)) return CSLAM_OK;
    CSLAM_TRY(order_after_dev(p, st));
    CSLAM_TRY(best_scan_candidates(p, queries_dev, query_stride, m, split_row, &out->stride, &out->n, st));
    out->a = p->cand_a.as<uint64_t>();
    out->c = p->cand_c.as<uint64_t>();
    *applied = true;
    return mark_dev(p, st);
}

// the scan half of search with 2 <= k <= 32 (search_dev_impl's two fused top-k branches, stopped in front of
// k_topk_from_parts): see TopkCandidates in common.hpp
int pool_topk_candidates(cslam_pool_t* p, const float* queries_dev, int64_t query_stride, int m, int k,
                         int64_t row_lo, int64_t row_hi, hipStream_t st, TopkCandidates* out, bool* applied) {
    *applied = false;
    if (!p || !out || m < 1 || k < 2 || k > kTopkWavesMaxK || !queries_dev || query_stride < p->dim ||
        row_lo < 0 || row_hi > p->n || row_hi <= row_lo)
        return CSLAM_OK;
    static Knob k_tw{"CSLAM_TOPK_WAVES", 1};
    if (knob(k_tw) == 0) return CSLAM_OK;
    CSLAM_HIP_TRY(hipSetDevice(p->device));
    p->v0 = row_lo;
    p->vn = row_hi - row_lo;
    const int64_t sims_stride = (p->vn + 63) / 64 * 64;
    out->rnx = p->inv_norm + p->v0;
    out->row0 = (uint32_t)p->v0;
    out->n = p->vn;
    out->dots_stride = sims_stride;
    if (m <= 16) {
        if (!dma_scan_applies(p) || ks_scan_applies(p, m) || (int64_t)p->cus * 4 > kTopkWavesMax) return CSLAM_OK;
        CSLAM_TRY(order_after_dev(p, st));
        CSLAM_TRY(p->sims.reserve((size_t)16 * sims_stride * sizeof(float)));
        int64_t cstride = 0, n_cand = 0;
        CSLAM_TRY(best_scan_candidates(p, queries_dev, query_stride, m, -1, &cstride, &n_cand, st,
                                       p->sims.as<float>(), sims_stride));
        out->kind = 0;
        out->cand = p->cand_a.as<uint64_t>();
        out->cand_stride = cstride;
        out->n_parts = (int)n_cand;
        out->wq = p->vn / n_cand;
        out->wrem = p->vn % n_cand;
        out->dots = p->sims.as<float>();
        out->rnq = p->rnq.as<float>();
        *applied = true;
        return mark_dev(p, st);
    }
    if (!mq_scan_applies(p, m) || mq_pass_queries(m) != m) return CSLAM_OK;   // one k_scan_mq pass: 17..80 queries
    const MqPlan g = mq_plan(p, m);
    if (g.n_cand > kTopkPartsMax) return CSLAM_OK;
    CSLAM_TRY(order_after_dev(p, st));
    CSLAM_TRY(p->qstage.reserve((size_t)(kQueryChunk + 128) * p->dp * sizeof(float)));
    CSLAM_TRY(p->rnq.reserve((size_t)(kQueryChunk + kMaxQueriesPerPass) * sizeof(float)));
    CSLAM_TRY(p->sims.reserve((size_t)16 * mq_max_tiles() * sims_stride * sizeof(float)));
    const int64_t cstride = (g.n_cand + 63) / 64 * 64;
    CSLAM_TRY(p->cand_a.reserve((size_t)16 * g.tiles * cstride * sizeof(uint64_t)));
    CSLAM_TRY(mq_pass(p, queries_dev, query_stride, m, p->rnq.as<float>(), p->sims.as<float>(), sims_stride, true,
                      p->cand_a.as<uint64_t>(), cstride, -1, nullptr, g, st));
    out->kind = 1;
    out->cand = p->cand_a.as<uint64_t>();
    out->cand_stride = cstride;
    out->n_parts = (int)g.n_cand;
    out->n_main = g.n_main;
    out->n_blocks = g.n_blocks;
    out->p_max = g.p_max;
    out->nw = g.waves;
    out->ns = mq_panel_strips(g.tiles, g.waves);
    out->n_main_parts = g.n_blocks * g.p_max * g.waves;
    out->dots = p->sims.as<float>();
    out->rnq = p->rnq.as<float>();
    *applied = true;
    return mark_dev(p, st);
}
}  // namespace cslam
