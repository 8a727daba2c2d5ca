// Exact top-k of one query from what a fused best-match scan left - the best composite of every PART of the
// row window and the raw dots of every row - as a device function (one 256-thread block per query), the
// partitions it is used with, and the completion word of the host-pointer API.  Shared by knn.hip (search with
// 2 <= k <= 32) and shard.hip (the sharded step's fused intra + inter search, which goes on to the match records
// in the same launch).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "composite.hpp"

namespace cslam {

// ---------------------------------------------------------------------------------------------
// k_topk_from_parts: exact top-k (2 <= k <= 32) of ONE query per block from what a fused scan left:
// the best (similarity, row) composite of every PART of the window (cand) and the raw dots of every
// row.  A part is what one composite covers: a wave's share of the window (k_scan_dma: WaveParts),
// a panel of the multi-tile scan's main launch or a strip of its K-split launch (k_scan_mq +
// k_mq_combine: PanelParts).
//   The k-th best row is at least as good as the k-th best PART maximum (k distinct parts hold k
//   distinct rows that good), and a part whose maximum is below it holds no row of the top k: the
//   candidates are the rows of the k best parts - ~k x 112 rows of the 100 000.  Round 2 ran k
//   block-wide arg-max rounds over every 4096 rows of the pool plus a merge level (14 + 9 us and a
//   6.4-MB dots read); this is one launch over ~1100 rows, and no serial rounds: selections are
//   by RANK (composites are unique: #{others greater} is the output position).
//   A1  every thread takes the best of its part maxima (parts tid, tid + 256, ...); a hardware wave
//       ranks its 64 thread maxima by their similarity word (64 readlanes) -> the k-th largest is a
//       lower bound of the k-th best part maximum; T1 = the best of the four bounds.
//   A2  part maxima >= T1 (a few times k of the ~1000) go to an LDS list with their part number.
//   A3  rank inside the list: ranks < k are the k best parts, rank k-1 is the bound Tk.
//   B1  composites of those parts' rows (dot x 1/||x|| x 1/||q||, as every other path; the row
//       ranges follow from the scan's partition of the window, Parts::range); those >= Tk (k of
//       them at least, rarely many more) go to the list.
//   B2  rank inside the list = output position.  A list overflow (> 1024 rows at or above the
//       bound: an all-zero query over parts of > 64 rows - or > 1024 parts at or above T1: the same
//       query over a window of many strips) takes k rounds of block arg-max instead.
// ---------------------------------------------------------------------------------------------
constexpr int kTopkWavesMaxK = 32;   // (the shard's fused intra + inter search asks for block + nb_best_matches <= 26)
constexpr int kTopkWavesMax = 1024;  // list entries (and the parts of WaveParts: 4 x CUs at most)
constexpr int kTopkPartsMax = 65535; // part numbers are kept in 16 bits

// k_scan_dma: the window's n rows split evenly over n_waves (launch_scan_dma)
struct WaveParts {
    int64_t wq, wrem;
    __host__ __device__ void range(int c, int64_t& lo, int& cnt) const {
        lo = c * wq + (c < wrem ? c : wrem);
        cnt = (int)(wq + (c < wrem ? 1 : 0));
    }
};
// k_scan_mq: parts [0, n_main_parts) are the panels of the main launch - number
// (block * p_max + round) * nw + wave, the strips of a block split evenly over its nw x rounds
// panels (scan_mq.hip, "Rows are dealt out in whole 16-row strips") - and part n_main_parts + s is
// strip s of the K-split launch's window [n_main, n) (k_mq_combine)
struct PanelParts {
    int64_t n_main, n;
    int n_blocks, p_max, nw, ns, n_main_parts;
    __host__ __device__ void range(int c, int64_t& lo, int& cnt) const {
        if (c >= n_main_parts) {
            lo = n_main + ((int64_t)(c - n_main_parts) << 4);
            const int64_t left = n - lo;
            cnt = left >= 16 ? 16 : (left > 0 ? (int)left : 0);
            return;
        }
        const int wave = c % nw, br = c / nw, rd = br % p_max, bidx = br / p_max;
        const int64_t S = (n_main + 15) >> 4;
        const int64_t bq = S / n_blocks, brem = S - bq * n_blocks;
        const int64_t bs_lo = (int64_t)bidx * bq + (bidx < brem ? bidx : brem);
        const int bs = (int)(bq + (bidx < brem ? 1 : 0));
        const int P = (bs + nw * ns - 1) / (nw * ns);
        lo = 0;
        cnt = 0;
        if (bs == 0 || rd >= P) return;
        const int npan = nw * P, pq = bs / npan, prem = bs - pq * npan;
        const int q0 = nw * rd + wave, q1 = q0 + 1;
        lo = (bs_lo + (int64_t)q0 * pq + (q0 < prem ? q0 : prem)) << 4;
        int64_t hi = (bs_lo + (int64_t)q1 * pq + (q1 < prem ? q1 : prem)) << 4;
        hi = hi < n_main ? hi : n_main;
        cnt = hi > lo ? (int)(hi - lo) : 0;
    }
};

// k_small_search<true>: part s is strip s of the window (16 rows)
struct StripParts {
    int64_t n;
    __host__ __device__ void range(int c, int64_t& lo, int& cnt) const {
        lo = (int64_t)c << 4;
        const int64_t left = n - lo;
        cnt = left >= 16 ? 16 : (left > 0 ? (int)left : 0);
    }
};

// (the body: ONE query - its part maxima cq[0..n_parts), its dots dq[0..n), 1/||q|| = rq - by one 256-thread
// block; results to oi/os[0..k).  Block-uniform control flow; callers that run it for several queries in a row
// put a barrier between two calls.)
template <class Parts>
__device__ inline void topk_from_parts_body(const uint64_t* cq, int n_parts, const Parts& parts,
                                            const float* dq, const float* __restrict__ rnx,
                                            const float rq, uint32_t row0, int64_t n, int k,
                                            int32_t* __restrict__ oi, float* __restrict__ os) {
    __shared__ uint64_t lst[kTopkWavesMax];
    __shared__ uint16_t lst_w[kTopkWavesMax];
    __shared__ uint64_t wmax[2][4];
    __shared__ uint64_t tk_s;
    __shared__ uint32_t tw[4];
    __shared__ int64_t w_lo[kTopkWavesMaxK];
    __shared__ int w_cnt[kTopkWavesMaxK], w_pre[kTopkWavesMaxK + 1];
    __shared__ int cnt1, cnt2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) {
        cnt1 = 0;
        cnt2 = 0;
        tk_s = 0;
    }
    // ---- A1: a lower bound of the k-th best part maximum --------------------------------------
    uint64_t gm = 0;
    for (int c = tid; c < n_parts; c += 256) {
        const uint64_t v = cq[c];
        gm = v > gm ? v : gm;
    }
    const uint32_t ghi = (uint32_t)(gm >> 32);   // an empty thread has 0, below every real similarity word
    int above = 0;
#pragma unroll
    for (int j = 0; j < 64; ++j) above += (uint32_t)__builtin_amdgcn_readlane((int)ghi, j) > ghi ? 1 : 0;
    // the lanes with fewer than k similarity words above theirs hold the k largest: the least of those
    uint32_t kth = above < k ? ghi : 0xFFFFFFFFu;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t v = (uint32_t)__shfl_xor((int)kth, o, 64);
        kth = v < kth ? v : kth;
    }
    if (lane == 0) tw[wave] = kth;
    __syncthreads();
    uint32_t t1 = tw[0];
#pragma unroll
    for (int x = 1; x < 4; ++x) t1 = tw[x] > t1 ? tw[x] : t1;
    // ---- A2: the part maxima at or above the bound ---------------------------------------------
    for (int c = tid; c < n_parts; c += 256) {
        const uint64_t v = cq[c];
        if (v != 0 && (uint32_t)(v >> 32) >= t1) {
            const int pos = atomicAdd(&cnt1, 1);
            if (pos < kTopkWavesMax) {
                lst[pos] = v;
                lst_w[pos] = (uint16_t)c;
            }
        }
    }
    __syncthreads();
    const int s1 = cnt1;
    const bool every_row = s1 > kTopkWavesMax;   // too many parts that good: every row is a candidate
    const int nwin = every_row ? 1 : (s1 < k ? s1 : k);
    // ---- A3: rank -> the k best parts and the bound Tk -----------------------------------------
    if (every_row) {
        if (tid == 0) {
            w_lo[0] = 0;
            w_cnt[0] = (int)n;
        }
    } else {
        for (int i = tid; i < s1; i += 256) {
            const uint64_t c = lst[i];
            int rank = 0;
            for (int j = 0; j < s1; ++j) rank += lst[j] > c ? 1 : 0;
            if (rank < k) {
                int64_t lo;
                int cnt;
                parts.range((int)lst_w[i], lo, cnt);
                w_lo[rank] = lo;
                w_cnt[rank] = cnt;
                if (rank == k - 1) tk_s = c;
            }
        }
    }
    __syncthreads();
    if (tid <= nwin) {
        int pre = 0;
        for (int j = 0; j < tid; ++j) pre += w_cnt[j];
        w_pre[tid] = pre;
    }
    __syncthreads();
    const int total = w_pre[nwin];
    const uint64_t tk = tk_s;
    // window-relative row of candidate number idx: the last winner whose first candidate is <= idx
    // (nwin <= 32: five steps)
    auto row_at = [&](int idx) -> int64_t {
        int j = 0;
#pragma unroll
        for (int step = 16; step > 0; step >>= 1)
            if (j + step < nwin && idx >= w_pre[j + step]) j += step;
        return w_lo[j] + (idx - w_pre[j]);
    };
    auto comp_at = [&](int idx) -> uint64_t {     // composite of candidate number idx
        const int64_t row = row_at(idx);
        return make_comp((dq[row] * rnx[row]) * rq, row0 + (uint32_t)row);
    };
    int s2 = kTopkWavesMax + 1;
    if (!every_row) {
        // ---- B1: the rows of those parts at or above Tk ----------------------------------------
        // (up to eight candidates per thread with all their loads in flight together: one memory round
        // trip instead of one per candidate)
        constexpr int kBatch = 8;
        for (int base = 0; base < total; base += kBatch * 256) {
            int64_t row[kBatch];
            float dv[kBatch], rx[kBatch];
#pragma unroll
            for (int e = 0; e < kBatch; ++e) {
                const int idx = base + e * 256 + tid;
                row[e] = idx < total ? row_at(idx) : -1;
            }
#pragma unroll
            for (int e = 0; e < kBatch; ++e) {
                const int64_t r_ = row[e] >= 0 ? row[e] : 0;
                dv[e] = dq[r_];
                rx[e] = rnx[r_];
            }
#pragma unroll
            for (int e = 0; e < kBatch; ++e) {
                const uint64_t c = row[e] >= 0 ? make_comp((dv[e] * rx[e]) * rq, row0 + (uint32_t)row[e]) : 0;
                if (row[e] >= 0 && c >= tk) {
                    const int pos = atomicAdd(&cnt2, 1);
                    if (pos < kTopkWavesMax) lst[pos] = c;
                }
            }
        }
        __syncthreads();
        s2 = cnt2;
    }
    if (s2 <= kTopkWavesMax) {
        // ---- B2: rank = output position ---------------------------------------------------------
        for (int i = tid; i < s2; i += 256) {
            const uint64_t c = lst[i];
            int rank = 0;
            for (int j = 0; j < s2; ++j) rank += lst[j] > c ? 1 : 0;
            if (rank < k) {
                oi[rank] = (int32_t)(0xFFFFFFFFu - (uint32_t)(c & 0xFFFFFFFFu));
                os[rank] = ord2f((uint32_t)(c >> 32));
            }
        }
        for (int j = s2 + tid; j < k; j += 256) {   // fewer rows than k
            oi[j] = -1;
            os[j] = -INFINITY;
        }
        return;
    }
    // ---- overflow: k rounds of block arg-max over the candidates, each below the previous winner
    uint64_t last = ~0ull;
    for (int rd = 0; rd < k; ++rd) {
        uint64_t mt = 0;
        for (int idx = tid; idx < total; idx += 256) {
            const uint64_t c = comp_at(idx);
            mt = (c < last && c > mt) ? c : mt;
        }
        const uint64_t m = wave_max_u64(mt);
        if (lane == 0) wmax[rd & 1][wave] = m;
        __syncthreads();
        uint64_t win = wmax[rd & 1][0];
#pragma unroll
        for (int x = 1; x < 4; ++x) win = wmax[rd & 1][x] > win ? wmax[rd & 1][x] : win;
        if (tid == 0) {
            oi[rd] = win ? (int32_t)(0xFFFFFFFFu - (uint32_t)(win & 0xFFFFFFFFu)) : -1;
            os[rd] = win ? ord2f((uint32_t)(win >> 32)) : -INFINITY;
        }
        last = win;   // 0 when nothing is left: the remaining rounds write the empty result
    }
}

// Completion word of the host-pointer API's small calls (cslam_pool_search): the results sit in pinned host
// memory and the host polls `done` for the call's number instead of paying a stream synchronisation.  Every
// block calls this behind its last result store; the LAST block of the launch to get here writes the word.
__device__ inline void signal_done(uint32_t* done, uint32_t seq, uint32_t* ticket) {
    if (done == nullptr) return;   // grid-uniform
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t total = gridDim.x * gridDim.y;
        bool last = true;
        if (total > 1) {
            last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == total - 1;
            if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (last) __hip_atomic_store(done, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

}  // namespace cslam
