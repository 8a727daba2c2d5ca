// The fold of the multi-tile scan's K-split pass (scan_mq.hip, MODE_PART) and the epilogue of a folded strip, as
// device functions of their own (k_mq_combine, scan_mq.hip).
// The K-split pass left, per strip and 512-float chunk, the chunk's partial products ([strip][chunk][tile][lane]
// float4 in the MFMA accumulator layout); tot = ((0 + c0) + c1) + ... in chunk order is exactly the fold of
// k_scan_mq's registers - same bits whoever computed the chunks.
// (Round 5 tried to END the pass here - fold, epilogue and the caller's final level (k_best_pair / the sharded step's
// records) in one launch, tickets telling the last block - and did not keep it: an agent-scope release fence per
// block is a write-back of the XCD's L2 (250 us for 2 000 blocks against 12 us of k_mq_combine), 2 000 device-scope
// atomics on ONE ticket word are performed one after the other at the memory side (72 us), and with write-through
// stores + two-level tickets the last block's reduction over the blocks' maxima - device-scope loads, a few in flight
// per lane - still made the launch 55 us: slower than the two launches and their 6-us gap together.
// profiles/r05_mq_final_ab_fences.txt, r05_mq_final_ab_3.txt, r05_mq_final_ab_4.txt.)
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

}  // namespace cslam
