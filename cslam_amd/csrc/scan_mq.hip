// scan_mq.hip — k_scan_mq: the pool scan for 17..80 queries per pass (T = 2..5 tiles of 16), X
// streamed ONCE through LDS-DMA whatever the number of query tiles.
//
// Replaces, for mid-size batches, what cslam/nns_matching.py:57-63 does per query (one full read of
// the pool each): the sharded step of SURVEY.md §8e matches all R x detection_publication_max_elems
// gathered descriptors (8 x 10 = 80, cslam/loop_closure_detection_node.py:34) against the pool in one
// pass.  Round 2 ran 17..64 queries on the register-ring kernel (16 rows x 64 B per load
// instruction, pool read twice for 80 queries: 0.83 ms = 0.50 of the fp32-MFMA peak).
//
// Structure (one 256-thread block per CU, four waves = one per SIMD):
//   * every wave owns a row panel of <= NS strips (16 rows each) per round and a PRIVATE ring of
//     NST stages (16 rows x KW floats, LDS-DMA pieces of 1 KiB, DmaGeom<KW>), ordered only by its own
//     counted vmcnt - exactly the X side of k_scan_dma;
//   * the query block is SHARED: the KW-float slice of all 16 T staged queries sits in LDS (double
//     buffered, moved by LDS-DMA too, every wave a quarter of the pieces), one raw s_barrier per
//     slice step.  B fragments go LDS -> VGPR per visit (one tile x HK <= 4 k-blocks, prefetched a
//     visit ahead), so they cost 2 x 4 HK registers instead of the 128 T of k_scan_dma's
//     register-resident slice - that is what makes T > 1 fit;
//   * work order is slice-major inside the panel (the step = one slice for all strips of all four
//     waves), accumulators acc/tot[NS][T] stay in registers (AccVGPRs) for the whole panel;
//   * the stage body is tile-major: visits (k-block group, tile), HK MFMAs in a row on one
//     accumulator.  The A fragments of a stage (NB float4 per lane) are refreshed IN PLACE in the
//     stage's last visits: once k-block i of stage j is multiplied for the last tile, a[i] is
//     re-read from stage j+1's slot - no second register set, no exposed LDS latency between
//     stages; stage j+NST is issued into stage j's slot in pieces between the MFMAs of the first
//     visits (a burst of DMA issues keeps the matrix pipe idle).
// Numerics: per (row, query) the MFMA chain visits k in the canonical order of DESIGN.md §3, T
// tiles interleave their independent chains, folds every 512 floats -> bit-identical to every
// other scan (tests/test_knn_gpu.py).
//
// MODE_DOTS raw dots out; MODE_BEST: one (similarity, row) composite per query and panel
// (+ the split window of csrc/shard.hip; + the raw dots when A.dots is given: search with
// 2 <= k <= 32, whose select then reads the rows of the k best panels only); MODE_PART: the block multiplies only the 512-float
// chunks [chunk_lo, chunk_hi) of its rows and leaves every chunk's partial in a scratch array -
// the K-split tail / small-pool pass, folded in chunk order by k_mq_combine (same bits).
#include <type_traits>
#include <utility>

#include "common.hpp"
#include "composite.hpp"
#include "dma_scan.hpp"
#include "scan_mq.hpp"
#include "mq_combine.hpp"
#include "mq_plan.hpp"
#include "probe_hooks.hpp"

using namespace cslam;

namespace {

constexpr int MODE_DOTS = 0, MODE_BEST = 1, MODE_PART = 2;

// f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>): a loop whose index is a constant
// expression in the body (the scheduling builtins take nothing else)
template <int... I, class F>
__device__ inline void static_for(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}

// A chunk partial leaves for the scratch array k_mq_combine reads right behind this launch.
// CSLAM_MQ_PART_STORE (compile time, A/B): 0 plain store, 1 non-temporal, 2 write-through (sc0 sc1).
#ifndef CSLAM_MQ_PART_STORE
#define CSLAM_MQ_PART_STORE 0
#endif
__device__ inline void mq_store_partial(f32x4* dst, const f32x4& v) {
#if CSLAM_MQ_PART_STORE == 1
    __builtin_nontemporal_store(v, dst);
#elif CSLAM_MQ_PART_STORE == 2
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(v) : "memory");
#else
    *dst = v;
#endif
}

// (two waves per SIMD - at most 256 registers per wave - for the form that runs as two four-wave blocks per CU)
template <int KW, int NST, int MODE, int NW>
constexpr int mq_min_waves_per_simd() {
    return MODE == MODE_PART && NW == 4 && NST == 2 && KW == 64 ? 2 : 1;
}
template <int T, int KW, int NST, int NS, int AUX, int MODE, int NW>
__global__ __launch_bounds__(NW * 64, (mq_min_waves_per_simd<KW, NST, MODE, NW>())) void k_scan_mq(const MqArgs A) {
    using GM = DmaGeom<KW>;
    constexpr int R = GM::R, NB = GM::NB, kPiece = GM::kPiece;
    constexpr int NP = 16 / R;            // DMA pieces per 16-row stage (== NB)
    constexpr int kStage = NP * kPiece;   // 16 rows x KW floats
    constexpr int kFold = 512 / KW;       // slices per 512-float chunk
    constexpr int NBP = T * NP / NW;      // B pieces per wave and slice (at least: wave w takes pieces w, w + NW, ...)
    constexpr int NBPX = (T * NP + NW - 1) / NW;
    constexpr int RV = MODE == MODE_BEST ? 2 : 0;   // row-norm loads per step
    // A stage's T x NB k-blocks are walked tile-major in NG groups of HK k-blocks (visit = group x tile:
    // 4 HK MFMAs in a row on one accumulator); position pos = visit HK + k-block counts them 0 .. T NB - 1.
    constexpr int HK = NB < 4 ? NB : 4;
    constexpr int NG = NB / HK;
    // the refill of a slot (stage j + NST into stage j's slot) is issued one piece behind each of the first
    // NP positions, the next slice's B pieces (first stage of a step) behind the following NBPX; in front of
    // position kPosW - the first k-block of the last visit - the NEXT stage must have landed: the A fragments
    // of the groups that every tile is through with are refreshed from it there, the others behind their
    // last k-block
    constexpr int kPosW = (NG * T - 1) * HK;
    constexpr bool kWrap = (NG * T) % 2 == 0;   // the last visit of a stage prefetches the first of the next
    // B pieces every wave has certainly issued in front of kPosW (pieces x < NBP at positions NP + x)
    constexpr int kBBeforeW = kPosW - NP < 0 ? 0 : (kPosW - NP < NBP ? kPosW - NP : NBP);
    static_assert(NP == NB && NB >= 4 && (NW == 4 || NW == 8), "geometry");
    static_assert(NP + NBPX <= T * NB, "every B piece needs a position behind the X pieces of the first stage");
    static_assert(NP <= kPosW, "the refill must be out before the wait for the next stage");
    extern __shared__ __attribute__((aligned(16))) char mq_lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int r = lane & 15, g = lane >> 4;
    char* ring = mq_lds + wave * NST * kStage;
    char* bbuf = mq_lds + NW * NST * kStage;   // [2][T][kStage]
    // MODE_BEST: 1/||x|| of the wave's panel rows (<= 128 floats), fetched by LDS-DMA - no VGPR
    // destination an in-flight load could owe
    char* rvbuf = bbuf + 2 * T * kStage + wave * 1024;   // + 1 / ||q|| of the 16 T queries behind them
    const int nkc_all = A.dp / KW;
    // MODE_PART (K-split launch): grid = n_blocks strip groups x part_ks K parts; this block multiplies
    // the whole 512-float chunks [chunk_lo, chunk_hi) of its rows only
    const int bidx = MODE == MODE_PART ? (int)(blockIdx.x % (unsigned)A.n_blocks) : (int)blockIdx.x;
    const int kpart = MODE == MODE_PART ? (int)(blockIdx.x / (unsigned)A.n_blocks) : 0;
    const int nchunks_all = (nkc_all + kFold - 1) / kFold;
    const int chunk_lo = MODE == MODE_PART ? kpart * nchunks_all / A.part_ks : 0;
    const int chunk_hi = MODE == MODE_PART ? (kpart + 1) * nchunks_all / A.part_ks : nchunks_all;
    const int kc_begin = chunk_lo * kFold;
    const int kc_end = chunk_hi * kFold < nkc_all ? chunk_hi * kFold : nkc_all;
    const int nkc = kc_end - kc_begin;

    // Rows are dealt out in whole 16-row strips (a partial strip costs a whole one on the matrix
    // pipe): the block's strips are an even split of the window's, walked in P rounds of four
    // panels (one per wave) of <= NS strips.  Only the window's last strip can be partial.
    const int64_t S = (A.n + 15) >> 4;
    const int64_t bq = S / A.n_blocks, brem = S - bq * A.n_blocks;
    const int64_t bs_lo = (int64_t)bidx * bq + (bidx < brem ? bidx : brem);
    const int bs = (int)(bq + (bidx < brem ? 1 : 0));   // strips of this block
    if (bs == 0 || nkc <= 0) return;   // block-uniform
    const int P = (bs + NW * NS - 1) / (NW * NS);
    const int npan = NW * P;
    const int pq = bs / npan, prem = bs - pq * npan;
    auto panel_s0 = [&](int q) { return bs_lo + (int64_t)q * pq + (q < prem ? q : prem); };   // first strip
    auto panel_lo = [&](int q) { return panel_s0(q) << 4; };
    auto panel_rows = [&](int rd) {
        const int q = NW * rd + wave;
        const int64_t lo = panel_lo(q);
        int64_t hi = panel_lo(q + 1);
        hi = hi < A.n ? hi : A.n;
        return (int)(hi > lo ? hi - lo : 0);
    };

    // stages of this wave: sum over rounds of strips x slices
    int J = 0;
    for (int rd = 0; rd < P; ++rd) J += ((panel_rows(rd) + 15) >> 4) * nkc;

    // DMA lane roles: lane l moves segment seg of row `sub` of a piece (R rows x KW floats)
    const int sub = R == 1 ? 0 : lane / (64 / R);
    const int seg = R == 1 ? lane : ((lane & (64 / R - 1)) ^ sub);
    // reader: row r of a stage sits in piece r / R as row r % R
    const int rd_off = (r / R) * kPiece + (r % R) * (KW * 4) + ((g ^ (r % R)) * 16);
    const uint32_t xs_b = (uint32_t)(A.xs * 4);   // row stride in bytes (< 2^20)

    // ---- issue side ---------------------------------------------------------------------------
    // Every wave issues the SAME operations per step and per stage whatever its rows (the refill
    // of a slot nobody will read again repeats the wave's last stage), so the waits of the stage
    // loop are compile-time counts; only the wait for the B pieces (the number of stages a step
    // holds varies) and the one for the row norms are counted at run time.
    int issued = 0;       // vector-memory operations issued by this wave so far
    int mark_b = 0, mark_rv = 0;
    // issue cursor: (round, slice, strip) of the next stage to request; stays on the last one.
    // i_base (uniform) = first row of that stage at its slice; i_last = rows of the panel behind the
    // stage's first row, minus one (the clamp of a ragged last strip)
    int i_rd = 0, i_kc = 0, i_s = 0, i_left = J;
    int i_ns = 1, i_last = 0;
    const char* i_base = reinterpret_cast<const char*>(A.X + (int64_t)kc_begin * KW);   // J == 0: row 0, dummy
    const int64_t strip_b = 16 * (int64_t)xs_b;
    auto i_panel = [&]() {   // first stage of the next round in which this wave has rows
        while (i_rd < P) {
            const int rows = panel_rows(i_rd);
            if (rows > 0) {
                i_ns = (rows + 15) >> 4;
                i_last = rows - 1;
                i_base = reinterpret_cast<const char*>(A.X + panel_lo(NW * i_rd + wave) * A.xs + (int64_t)kc_begin * KW);
                break;
            }
            ++i_rd;
        }
    };
    if (J > 0) i_panel();
    // piece pp of the stage under the issue cursor -> slot `sl`; rows past the panel's end repeat
    // its last row (clamped: never an address outside the window, results never looked at).
    // The byte offsets of a FULL strip's pieces are loop invariants, kept in NP registers (made opaque so that
    // they are not re-derived at every issue); the clamp is ONE v_min against the offset of the panel's last row
    // (i_climit, refreshed when the cursor moves: min(rel, last) xs + seg = min(rel xs + seg, last xs + seg)).
    // A piece then costs a v_min, an M0 write and the DMA instruction - round 3 paid a v_min and a 64-bit v_mad
    // per piece, ~20 clocks that no MFMA shadow covers (an LDS-DMA issue already costs ~35, tools/mfma_mix_probe.hip).
    uint32_t full_off[NP];
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) {
        full_off[pp] = (uint32_t)(R * pp + sub) * xs_b + (uint32_t)seg * 16u;
        asm volatile("" : "+v"(full_off[pp]));
    }
    uint32_t i_climit = (uint32_t)i_last * xs_b + (uint32_t)seg * 16u;
    auto issue_piece = [&](int pp, int sl) {
        const uint32_t off = full_off[pp] < i_climit ? full_off[pp] : i_climit;
        const char* src = i_base + off;
        CSLAM_MQ_AUDIT_SRC(0, src, 16)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(ring + sl * kStage + pp * kPiece),
                                         16, 0, AUX);
    };
    auto i_advance_cursor = [&]() {
        // (the likely path first: every taken branch between two stages costs the wave an instruction
        // buffer refill with nothing on the matrix pipe)
        if (__builtin_expect(i_left <= 1, 0)) {   // nothing behind the last stage: the cursor stays (dummy refills)
            i_left = 0;
            return;
        }
        --i_left;
        if (__builtin_expect(++i_s < i_ns, 1)) {            // next strip of the slice
            i_base += strip_b;
            i_last -= 16;
            return;
        }
        i_base -= (int64_t)(i_s - 1) * strip_b;   // back to the panel's first strip
        i_last += 16 * (i_s - 1);
        i_s = 0;
        if (__builtin_expect(++i_kc < nkc, 1)) {            // next slice
            i_base += KW * 4;
            return;
        }
        i_kc = 0;
        ++i_rd;
        i_panel();
    };
    auto i_advance = [&]() {
        i_advance_cursor();
        i_climit = (uint32_t)i_last * xs_b + (uint32_t)seg * 16u;   // offset of the last row the next stage may touch
    };
    // this wave's share of the B pieces of slice kc -> buffer bb: pieces wave, wave + NW, ... of the
    // slice's T NP (piece pi = tile pi / NP, piece pi % NP of the tile)
    // B pieces: piece pi = wave + NW x of a slice holds the query rows R pi .. R pi + R - 1, so its source is
    //   [Q + R wave dp + slice]  (uniform, a scalar pair re-made per step and opaque to the optimiser)
    //   + x (R NW dp floats)     (uniform, 32 bits)        + this lane's row / segment (one VGPR).
    // Written any simpler, the loop optimiser turns the T NP / NW pieces into as many loop-invariant
    // 64-bit per-lane addresses: 20 registers alive through the whole kernel at T = 5, one of them
    // spilled - and reloaded, with a vmcnt(0), in the middle of every step.
    const uint32_t b_lane_off = (uint32_t)(sub * A.dp + seg * 4) * 4u;
    const uint32_t b_stride = (uint32_t)(R * NW * A.dp) * 4u;
    const char* const b_wave = reinterpret_cast<const char*>(A.Q + (int64_t)(R * wave) * A.dp + (int64_t)kc_begin * KW);
    const char* b_cur = b_wave;   // scalar pair: source of this wave's piece 0 of the slice being requested
    auto b_slice = [&](int kc) {
        b_cur = b_wave + (int64_t)kc * (KW * 4);
        asm volatile("" : "+s"(b_cur));   // opaque: no per-piece address is derived from the loop counter
    };
    auto issue_b_piece = [&](int x, int kc, int bb) {
        const int pi = wave + NW * x;
        // (compile-time true when the T NP pieces of a slice divide evenly over the waves: no branch then)
        if ((T * NP) % NW == 0 || pi < T * NP) {   // wave-uniform
            // the lane offset enters LAST and opaque: scalar base + 32-bit VGPR offset is an addressing mode of
            // the DMA instruction; a 64-bit per-lane address costs a v_lshl_add_u64 in front of every piece
            uint32_t lo = b_lane_off;
            asm volatile("" : "+v"(lo));
            const char* ub = b_cur + (uint32_t)x * b_stride;
            const char* src = ub + lo;
            CSLAM_MQ_AUDIT_SRC(1, src, 16)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)src,
                (__attribute__((address_space(3))) void*)(bbuf + (bb * T * NP + pi) * kPiece), 16, 0, 0);
            ++issued;
        }
    };
    auto issue_b = [&](int kc, int bb) {
        b_slice(kc);
#pragma unroll
        for (int x = 0; x < NBPX; ++x) issue_b_piece(x, kc, bb);
        mark_b = issued;
    };

    // ---- prologue -----------------------------------------------------------------------------
    // MODE_BEST: 1 / ||q|| of the 16 T staged queries goes to the LDS once (every wave its own copy behind
    // its row norms: no barrier), by LDS-DMA; the epilogue reads the lane's T values back.  Held in
    // registers they were T more live through the stage loop of a kernel that sits at the register
    // ceiling; loaded in the epilogue they would drain the ring (hipcc waits vmcnt(0) for an ordinary load).
    if (MODE == MODE_BEST) {
        const int qi = lane < 16 * T ? lane : 16 * T - 1;
        const char* src = reinterpret_cast<const char*>(A.rnq + qi);
        CSLAM_MQ_AUDIT_SRC(2, src, 4)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(rvbuf + 512), 4, 0, 0);
        if (16 * T > 64) {
            const int qj = 64 + lane < 16 * T ? 64 + lane : 16 * T - 1;
            const char* src2 = reinterpret_cast<const char*>(A.rnq + qj);
            CSLAM_MQ_AUDIT_SRC(2, src2, 4)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src2,
                                             (__attribute__((address_space(3))) void*)(rvbuf + 512 + 256), 4, 0, 0);
        }
        issued += 16 * T > 64 ? 2 : 1;   // older than everything the waits of the stage loop count
    }
    issue_b(0, 0);
#pragma unroll
    for (int i = 0; i < NST; ++i) {
#pragma unroll
        for (int pp = 0; pp < NP; ++pp) issue_piece(pp, i);
        issued += NP;
        i_advance();
    }
    f32x4 a[NB];
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 1) * NP) : "memory");   // stage 0 has landed
#pragma unroll
    for (int i = 0; i < NB; ++i) a[i] = *reinterpret_cast<const f32x4*>(ring + rd_off + i * 64);

    int slot = 0;      // ring slot of the stage the consumer is at
    int gstep = 0;
    CSLAM_MQ_HOOK_KERNEL_BEGIN
    // both accumulator sets live in AccVGPRs across a panel (left to itself hipcc parks them in architectural
    // VGPRs between a strip's stages - 8 NS T of them - and spills)
    f32x4 acc[NS][T], tot[NS][T];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int t = 0; t < T; ++t) {
            acc[s][t] = f32x4{0.f, 0.f, 0.f, 0.f};
            tot[s][t] = f32x4{0.f, 0.f, 0.f, 0.f};
            asm volatile("" : "+a"(acc[s][t]));
        }
    for (int rd = 0; rd < P; ++rd) {
        const int rows = panel_rows(rd);
        const int64_t lo = panel_lo(NW * rd + wave);
        const int ns = (rows + 15) >> 4;
        // the running totals restart with every panel (K-split mode has none: its chunk partials leave at
        // every fold); the chain accumulators were zeroed once in front of the round loop and every round
        // ends on a fold, which leaves them zero again - 8 NS T v_accvgpr_write per round less
        if (MODE != MODE_PART) {
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    tot[s][t] = f32x4{0.f, 0.f, 0.f, 0.f};
                    asm volatile("" : "+a"(tot[s][t]));
                }
        }
        for (int kc = 0; kc < nkc; ++kc, ++gstep) {
            const int bb = gstep & 1;
            // my B pieces of this slice have landed ... and so have everybody's; everybody is also
            // done with the other buffer (slice kc - 1)
            CSLAM_MQ_HOOK_STEP_BEGIN
            wait_vm_dyn(issued - mark_b);
            CSLAM_MQ_HOOK_B_LANDED
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            CSLAM_MQ_HOOK_BARRIER_PASSED
            if (MODE == MODE_BEST) {
                // 1/||x|| of the panel's <= 128 rows for the epilogue: lane l moves the norms of rows
                // lo + l and lo + 64 + l (clamped into the window) into the wave's LDS area, two 256-B
                // LDS-DMA pieces.  Issued at EVERY step (the same values again: the waits of the stage
                // loop stay compile-time counts); waited for in front of the epilogue, which reads
                // them back with one ds_read_b128 per strip.
                // (uniform base + a 32-bit lane offset: two registers live through the step loop, not two
                // 64-bit addresses)
                // (a wave without rows in this round has lo at or behind the window's end: clamped too)
                const int64_t lo_c = lo < A.n ? lo : A.n - 1;
                const int64_t last = A.n - 1 - lo_c;   // >= 0
                const uint32_t o0 = (uint32_t)(lane < last ? lane : last) * 4u;
                const uint32_t o1 = (uint32_t)(64 + lane < last ? 64 + lane : last) * 4u;
                const char* rb = reinterpret_cast<const char*>(A.rnx + lo_c);
                const char* src0 = rb + o0;
                const char* src1 = rb + o1;
                CSLAM_MQ_AUDIT_SRC(3, src0, 4)
                CSLAM_MQ_AUDIT_SRC(3, src1, 4)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src0,
                                                 (__attribute__((address_space(3))) void*)rvbuf, 4, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src1,
                                                 (__attribute__((address_space(3))) void*)(rvbuf + 256), 4, 0, 0);
                issued += 2;
                mark_rv = issued;
            }
            // slice kc + 1 (slice 0 again behind the last one: the next round's first, or nobody's) is
            // requested piece by piece between the chains of this step's FIRST stage (a burst of
            // DMA issues at the top of the step kept the matrix pipe idle ~600 clocks per step)
            const int kc_nx = kc + 1 == nkc ? 0 : kc + 1;
            if (ns == 0) issue_b(kc_nx, bb ^ 1);
            else b_slice(kc_nx);
            const char* bs = bbuf + bb * T * kStage + rd_off;
            // end of a 512-float chunk / of the row: fold (or, MODE_PART, hand the partial over)
            const bool fold = (((kc_begin + kc + 1) & (kFold - 1)) == 0) || kc_begin + kc + 1 == nkc_all;
            f32x4 bq[2][HK];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                if (__builtin_expect(s < ns, 1)) {
                    // `a` holds this stage (read while the previous one was multiplied).  Its slot is
                    // refilled piece by piece BEHIND the k-blocks: the first DMA follows MFMAs that
                    // waited for B fragments read after the last of those `a` reads - LDS returns in
                    // order, so every read of the slot has returned before anything overwrites it.
                    const int nslot = slot + 1 == NST ? 0 : slot + 1;
                    const char* nx = ring + nslot * kStage + rd_off;
                    CSLAM_MQ_HOOK_STAGE_BEGIN
                    // ---- the stage's T x NB k-blocks, TILE-MAJOR in groups of HK k-blocks (round 4) ----
                    // Visit v = (group gi, tile t): HK k-blocks = 4 HK MFMAs in a row on ONE accumulator.
                    // The matrix pipe forwards the accumulator between back-to-back MFMAs of a chain (32
                    // clocks each) and pays ~8 clocks at every change of accumulator: round-robin over the
                    // tiles ran at 38-40 clocks per MFMA, chains of four (round 3: one k-block per tile) at
                    // 34 isolated / 35.9 in this loop, chains of sixteen leave 0.5 clocks of it
                    // (tools/mfma_rate2.hip; per-stage stamps in profiles/r04_mq_stage_probe_*.txt).
                    // Same chains, same order inside each: same bits.  The B fragments of a visit (HK float4)
                    // sit in bq[v & 1]; those of the next visit - of visit 0 again behind the last: the next
                    // strip of this step starts with them - are requested behind this visit's first k-block.
                    // (an odd number of visits per stage - 64-float slices with three or five tiles - would
                    // hand the next stage its first fragments in the wrong set: every stage then loads them
                    // itself; those variants run two waves per SIMD or lean on the HBM)
                    if (s == 0 || !kWrap) {
#pragma unroll
                        for (int ii = 0; ii < HK; ++ii) bq[0][ii] = *reinterpret_cast<const f32x4*>(bs + ii * 64);
                    }
                    static_for(std::make_integer_sequence<int, NG * T>{}, [&](auto vc) {
                        constexpr int v = decltype(vc)::value;
                        constexpr int gi = v / T, t = v % T;
                        constexpr int vn = v + 1 == NG * T ? 0 : v + 1;      // the visit behind this one
                        constexpr int gn = vn / T, tn = vn % T;
                        constexpr int cur = v & 1;
                        static_for(std::make_integer_sequence<int, HK>{}, [&](auto ic) {
                            constexpr int ii = decltype(ic)::value;
                            constexpr int i = gi * HK + ii;                      // k-block of the slice
                            constexpr int pos = v * HK + ii;                     // position in the stage
                            f32x4(&bc)[HK] = bq[cur];
                            f32x4(&bn)[HK] = bq[cur ^ 1];
                            if (pos == kPosW) {
                                // The next stage has landed.  Younger than its last piece: the stages behind it
                                // in the ring, ALL of this stage's refill (issued at positions 0 .. NP - 1), and
                                // what a step's top and first stage add when they lie in between (row norms, B
                                // pieces).  Every term is what EVERY wave has certainly issued.
                                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * NP + NP +
                                                                        (s == 0 ? RV + kBBeforeW
                                                                                : (s <= NST - 2 ? RV + NBP : (s == NST - 1 ? NBP : 0))))
                                             : "memory");
                                // A fragments of the groups every tile is through with, from the next stage
#pragma unroll
                                for (int i2 = 0; i2 < (NG - 1) * HK; ++i2)
                                    a[i2] = *reinterpret_cast<const f32x4*>(nx + i2 * 64);
                            }
                            // The four MFMAs of the k-block depend on each other through the accumulator: the
                            // wave cannot issue the next one for ~32 clocks - and in those clocks it can issue
                            // anything ELSE.  Everything a k-block has to do besides (next visit's B fragments, a
                            // DMA piece with its address arithmetic, the issue cursor, the A refresh) is therefore
                            // placed BETWEEN its MFMAs, one item per gap; behind the fourth MFMA the same work
                            // (round 3, and the compiler left to itself) holds up the first MFMA of the next chain.
                            __builtin_amdgcn_sched_barrier(0);
                            acc[s][t] = CSLAM_MQ_MFMA(a[i][0], bc[ii][0], acc[s][t]);
                            __builtin_amdgcn_sched_barrier(0);
                            if (ii == 0 && (kWrap || v + 1 < NG * T)) {
                                // next visit's B fragments (hipcc waits lgkmcnt(0) in front of a chain whenever an
                                // LDS-DMA was issued since the last wait: requested here they have the rest of
                                // the visit to land)
#pragma unroll
                                for (int i2 = 0; i2 < HK; ++i2)
                                    bn[i2] = *reinterpret_cast<const f32x4*>(bs + tn * kStage + (gn * HK + i2) * 64);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                            acc[s][t] = CSLAM_MQ_MFMA(a[i][1], bc[ii][1], acc[s][t]);
                            __builtin_amdgcn_sched_barrier(0);
                            // the refill of this stage's slot: one piece in each of the first NP k-blocks
                            if (CSLAM_MQ_REFILL && pos < NP) {
                                issue_piece(pos, slot);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                            // the last piece is out: the issue cursor moves on
                            if (pos == NP) {
                                i_advance();
                                __builtin_amdgcn_sched_barrier(0);
                            }
                            // in the step's first stage, the next slice's B pieces behind the X pieces
                            if (s == 0 && pos >= NP && pos - NP < NBPX) {
                                issue_b_piece(pos - NP, kc_nx, bb ^ 1);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                            acc[s][t] = CSLAM_MQ_MFMA(a[i][2], bc[ii][2], acc[s][t]);
                            __builtin_amdgcn_sched_barrier(0);
                            acc[s][t] = CSLAM_MQ_MFMA(a[i][3], bc[ii][3], acc[s][t]);
                            __builtin_amdgcn_sched_barrier(0);
                            // the last visit of a k-block: its A fragment comes from the next stage (behind the
                            // last MFMA that reads it; the next chain's first MFMA covers the read's issue)
                            if (v == NG * T - 1) {
                                a[i] = *reinterpret_cast<const f32x4*>(nx + i * 64);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        });
                    });
                    CSLAM_MQ_HOOK_STAGE_END
                    issued += NP;
                    if (s == 0) mark_b = issued;   // an upper bound of the last B piece's position: a longer wait at worst
#pragma unroll
                    for (int t = 0; t < T; ++t) asm volatile("" : "+a"(acc[s][t]));
                    if (__builtin_expect(fold, 0)) {
                        // the zero the accumulators restart from, made HERE: as a loop invariant the
                        // compiler kept one zero float4 alive through the whole kernel, spilled it, and
                        // reloaded it in every fold - and a scratch reload is followed by vmcnt(0): the
                        // wave's whole DMA ring drained once per strip and chunk
                        float z;
                        asm volatile("v_mov_b32 %0, 0" : "=v"(z));
                        const f32x4 zero4{z, z, z, z};
                        if (MODE == MODE_PART) {
                            // the partial of chunk c of this strip, in the layout k_mq_combine folds
                            const int64_t strip = (lo >> 4) + s;   // panels are strip-aligned
                            const int c = (kc_begin + kc) / kFold;
                            f32x4* out = A.part + ((strip * nchunks_all + c) * T) * 64 + lane;
#pragma unroll
                            for (int t = 0; t < T; ++t) {
                                CSLAM_MQ_AUDIT_DST(4, out + t * 64, 16) { mq_store_partial(out + t * 64, acc[s][t]); }
                                acc[s][t] = zero4;
                                asm volatile("" : "+a"(acc[s][t]));
                            }
                        } else {
#pragma unroll
                            for (int t = 0; t < T; ++t) {
                                tot[s][t] = tot[s][t] + acc[s][t];
                                acc[s][t] = zero4;
                                asm volatile("" : "+a"(acc[s][t]), "+a"(tot[s][t]));
                            }
                        }
                    }
                    slot = nslot;
                }
            }
        }
        // ---- panel epilogue ---------------------------------------------------------------
        if (MODE == MODE_BEST) {
            wait_vm_dyn(issued - mark_rv);   // the row norms of this panel sit in rvbuf
            // This lane's position, opaque from here on: everything the epilogue derives from (r, g) alone -
            // the 4 NS row offsets 16 s + 4 g + e, the T candidate addresses (16 t + r) cand_stride, the
            // LDS offsets of the norms - is loop-invariant, and hoisted out of the round loop it stays
            // live through the stage loop: ~60 registers, i.e. spills, which the allocator put INTO the
            // stage loop (scratch reload = vmcnt(0) = the wave's DMA ring drained at every fold).
            int er = r, eg = g;
            asm volatile("" : "+v"(er), "+v"(eg));
            // One query tile after the other (scheduling barriers in between): two composites, four row
            // norms and the values of one strip are all that is live - with every tile's running best
            // held at once the epilogue was the register peak of the kernel.
            const int64_t hi = lo + rows;
            const bool two = A.split < hi;   // wave-uniform: rows >= split in this panel
            const uint32_t rbase = A.row0 + (uint32_t)lo;
            const int srel = A.split > lo ? (int)((A.split < hi ? A.split : hi) - lo) : 0;
            const int64_t pidx = ((int64_t)bidx * A.p_max + rd) * NW + wave;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                __builtin_amdgcn_sched_barrier(0);
                uint64_t best = 0, best_c = 0;
                const float rq = *reinterpret_cast<const float*>(rvbuf + 512 + (16 * t + er) * 4);
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    if (s < ns) {
                        const f32x4 rn = *reinterpret_cast<const f32x4*>(rvbuf + (16 * s + 4 * eg) * 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int rel = 16 * s + 4 * eg + e;
                            const float sim = (tot[s][t][e] * rn[e]) * rq;
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
                    }
                }
                uint64_t o = __shfl_xor(best, 16, 64);
                best = o > best ? o : best;
                o = __shfl_xor(best, 32, 64);
                best = o > best ? o : best;
                if (two) {
                    o = __shfl_xor(best_c, 16, 64);
                    best_c = o > best_c ? o : best_c;
                    o = __shfl_xor(best_c, 32, 64);
                    best_c = o > best_c ? o : best_c;
                }
                // one composite per query and panel; a panel without rows leaves the empty composite
                if (eg == 0 && 16 * t + er < A.m_valid) {
                    CSLAM_MQ_AUDIT_IDX(5, pidx, (int64_t)A.n_blocks * A.p_max * NW) {
                        A.cand[(int64_t)(16 * t + er) * A.cand_stride + pidx] = best;
                        if (A.cand_c != nullptr) A.cand_c[(int64_t)(16 * t + er) * A.cand_stride + pidx] = best_c;
                    }
                }
                // search with k > 1 (k_topk_from_parts): the raw dots beside the composites (wave-uniform)
                if (A.dots != nullptr && 16 * t + er < A.m_valid) {
                    float* out = A.dots + (int64_t)(16 * t + er) * A.dots_stride + lo + 4 * eg;
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        if (s < ns) {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (16 * s + 4 * eg + e < rows) {
                                    CSLAM_MQ_AUDIT_IDX(6, lo + 16 * s + 4 * eg + e, A.n) { out[16 * s + e] = tot[s][t][e]; }
                                }
                        }
                    }
                }
            }
        } else if (MODE == MODE_DOTS) {
            const int64_t hi = lo + rows;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                if (s < ns) {
                    const int64_t r0 = lo + 16 * s + 4 * g;
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        if (16 * t + r < A.m_valid) {
                            float* out = A.dots + (int64_t)(16 * t + r) * A.dots_stride + r0;
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (r0 + e < hi) {
                                    CSLAM_MQ_AUDIT_IDX(6, r0 + e, A.n) { out[e] = tot[s][t][e]; }
                                }
                        }
                    }
                }
            }
        }
    }
    CSLAM_MQ_HOOK_KERNEL_END
    if (MODE == MODE_BEST && g == 0) {   // rounds this block does not have (blocks differ by one row)
        for (int rd = P; rd < A.p_max; ++rd) {
            const int64_t pidx = ((int64_t)bidx * A.p_max + rd) * NW + wave;
#pragma unroll
            for (int t = 0; t < T; ++t)
                if (16 * t + r < A.m_valid) {
                    CSLAM_MQ_AUDIT_IDX(5, pidx, (int64_t)A.n_blocks * A.p_max * NW) {
                        A.cand[(int64_t)(16 * t + r) * A.cand_stride + pidx] = 0;
                        if (A.cand_c != nullptr) A.cand_c[(int64_t)(16 * t + r) * A.cand_stride + pidx] = 0;
                    }
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_mq_combine: the K-split pass left, per strip and 512-float chunk, the chunk's partial products
// ([strip][chunk][tile][lane] float4 in the MFMA accumulator layout).  One wave per strip folds them
// in chunk order - tot = ((0 + c0) + c1) + ... exactly the fold of k_scan_mq's registers, same bits -
// and runs the same epilogue: raw dots, or the strip's (similarity, row) composite per query.
// ---------------------------------------------------------------------------------------------
template <bool BEST>
__global__ __launch_bounds__(256) void k_mq_combine(const MqArgs A, int tiles, int nchunks, int64_t cand_off) {
    // one wave per (strip, tile): its <= 16 chunk partials are requested together, folded in order
    const int64_t wv = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t strip = wv / tiles;
    const int t = (int)(wv % tiles);
    const int lane = threadIdx.x & 63;
    const int r = lane & 15, g = lane >> 4;
    const int64_t lo = strip << 4;
    if (lo >= A.n) return;
    const int64_t hi = lo + 16 < A.n ? lo + 16 : A.n;
    const f32x4* src = A.part + ((strip * nchunks) * tiles + t) * 64 + lane;
    CSLAM_MQ_AUDIT_COMBINE
    const f32x4 tot = mq_fold_partials(src, tiles, nchunks);   // (mq_combine.hpp: shared with the fused final level)
    if (BEST) {
        MqBest2 b = mq_strip_composites(A, tot, lo, hi, g, A.rnq[16 * t + r]);
        uint64_t o = __shfl_xor(b.a, 16, 64);
        b.a = o > b.a ? o : b.a;
        o = __shfl_xor(b.a, 32, 64);
        b.a = o > b.a ? o : b.a;
        o = __shfl_xor(b.c, 16, 64);
        b.c = o > b.c ? o : b.c;
        o = __shfl_xor(b.c, 32, 64);
        b.c = o > b.c ? o : b.c;
        if (g == 0 && 16 * t + r < A.m_valid) {
            A.cand[(int64_t)(16 * t + r) * A.cand_stride + cand_off + strip] = b.a;
            if (A.cand_c != nullptr) A.cand_c[(int64_t)(16 * t + r) * A.cand_stride + cand_off + strip] = b.c;
        }
    }
    // raw dots: dots mode, and beside the composites for search with k > 1 (k_topk_from_parts)
    if ((!BEST || A.dots != nullptr) && 16 * t + r < A.m_valid) {
        const int64_t r0 = lo + 4 * g;
        float* out = A.dots + (int64_t)(16 * t + r) * A.dots_stride + r0;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (r0 + e < hi) out[e] = tot[e];
    }
}

// LDS bytes of a variant (K-split mode has no epilogue: no area for the row / query norms)
template <int T, int KW, int NST, int NW, int MODE>
constexpr size_t mq_lds_bytes() {
    return (size_t)(NW * NST + 2 * T) * (16 / DmaGeom<KW>::R) * DmaGeom<KW>::kPiece + (MODE == MODE_PART ? 0 : (size_t)NW * 1024);
}

template <int T, int KW, int NST, int NS, int AUX, int MODE, int NW>
int launch_one(const MqArgs& a, hipStream_t st) {
    constexpr size_t lds = mq_lds_bytes<T, KW, NST, NW, MODE>();
    static_assert(lds <= 160 * 1024, "LDS budget");
    static bool attr_set[64] = {};
    if (!attr_set[a.device & 63]) {
        CSLAM_HIP_TRY(hipFuncSetAttribute((const void*)k_scan_mq<T, KW, NST, NS, AUX, MODE, NW>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[a.device & 63] = true;
    }
    const unsigned grid = (unsigned)a.n_blocks * (MODE == MODE_PART ? (unsigned)a.part_ks : 1u);
    hipLaunchKernelGGL((k_scan_mq<T, KW, NST, NS, AUX, MODE, NW>), dim3(grid), dim3(NW * 64), lds, st, a);
    CSLAM_HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}

template <int T, int KW, int NST, int NS, int NW>
int launch_mode(const MqArgs& a, int mode, bool nt, hipStream_t st) {
    if (mode == MODE_PART) return launch_one<T, KW, NST, NS, 0, MODE_PART, NW>(a, st);
    if (mode == MODE_BEST)
        return nt ? launch_one<T, KW, NST, NS, 2, MODE_BEST, NW>(a, st) : launch_one<T, KW, NST, NS, 0, MODE_BEST, NW>(a, st);
    return nt ? launch_one<T, KW, NST, NS, 2, MODE_DOTS, NW>(a, st) : launch_one<T, KW, NST, NS, 0, MODE_DOTS, NW>(a, st);
}

}  // namespace

namespace cslam {

// Variants.  Four waves per block (one per SIMD), 128-float slices where the row length allows, is
// what ships.  The eight-wave form (two per SIMD, eight private rings of 4-KiB stages: 64-float
// slices, which stream as fast as 128-float ones - profiles/r03_mq_probes.txt) is the same template
// with NW = 8; measured equal or slightly slower at every tile count once the MFMAs run in chains
// (0.567 vs 0.560 ms for 80 queries x 100k x 4096-d, 0.322 vs 0.310 for 32), so it is only
// compiled with -DCSLAM_MQ_WAVES8 (CSLAM_MQ_WAVES=8 then selects it, A/B runs).
int mq_max_tiles() { return 5; }
int mq_waves_available(int waves) {   // for the main launch; the K-split launch always has its eight-wave form
#if defined(CSLAM_MQ_WAVES8)
    return waves == 8 || waves == 4;
#else
    return waves == 4;
#endif
}
int mq_panel_strips(int tiles, int waves) { return mq_panel_strips_of(tiles, waves); }   // (mq_plan.hpp)

int launch_scan_mq(const MqArgs& a, int tiles, int kw, int waves, int mode, bool nt, hipStream_t st) {
    if (mode == kMqModePart && a.part_ks < 1) return fail(CSLAM_E_INVALID, "scan_mq: K parts");
    // The K-split launch with two waves per SIMD (eight private rings of 4-KiB stages, 64-float slices):
    // no running totals, so NS strips x T tiles of accumulators fit 256 registers per wave
    if (mode == kMqModePart && waves == 8 && kw == 64 && tiles >= 2 && tiles <= 5 && a.dp % 64 == 0 && a.n > 0 &&
        a.n_blocks >= 1) {
        static Knob k_ns{"CSLAM_MQ_PART_NS", 0};   // A/B: strips per panel
        const int ns_env = knob(k_ns);
        if (tiles == 5 && ns_env == 3) return launch_one<5, 64, 3, 3, 0, MODE_PART, 8>(a, st);
        if (tiles == 5 && ns_env == 2) return launch_one<5, 64, 3, 2, 0, MODE_PART, 8>(a, st);
        if (tiles == 5 && ns_env == 4) return launch_one<5, 64, 3, 4, 0, MODE_PART, 8>(a, st);
        // (nt: the pool rows are read once and must not push the chunk partials, which k_mq_combine reads
        // right behind this launch, out of the Infinity Cache)
        if (nt) {
            switch (tiles) {
                case 2: return launch_one<2, 64, 3, 8, 2, MODE_PART, 8>(a, st);
                case 3: return launch_one<3, 64, 3, 8, 2, MODE_PART, 8>(a, st);
                case 4: return launch_one<4, 64, 3, 6, 2, MODE_PART, 8>(a, st);
                default: return launch_one<5, 64, 3, 6, 2, MODE_PART, 8>(a, st);
            }
        }
        switch (tiles) {
            case 2: return launch_one<2, 64, 3, 8, 0, MODE_PART, 8>(a, st);
            case 3: return launch_one<3, 64, 3, 8, 0, MODE_PART, 8>(a, st);
            case 4: return launch_one<4, 64, 3, 6, 0, MODE_PART, 8>(a, st);
            default: return launch_one<5, 64, 3, 6, 0, MODE_PART, 8>(a, st);
        }
    }
    // ... and as TWO four-wave blocks per CU (A/B, CSLAM_MQ_PART_BLOCKS2): rings of two 4-KiB stages, 78 KiB of LDS
    // per block at five tiles, <= 256 registers per wave - a block's barrier is covered by the other block's MFMAs
    if (mode == kMqModePart && a.part_per_cu == 2 && waves == 4 && kw == 64 && tiles >= 2 && tiles <= 5 && a.dp % 64 == 0 &&
        a.n > 0 && a.n_blocks >= 1) {
        static_assert(2 * mq_lds_bytes<5, 64, 2, 4, MODE_PART>() <= 160 * 1024, "two blocks per CU");
        if (nt) {
            switch (tiles) {
                case 2: return launch_one<2, 64, 2, 8, 2, MODE_PART, 4>(a, st);
                case 3: return launch_one<3, 64, 2, 8, 2, MODE_PART, 4>(a, st);
                case 4: return launch_one<4, 64, 2, 6, 2, MODE_PART, 4>(a, st);
                default: return launch_one<5, 64, 2, 6, 2, MODE_PART, 4>(a, st);
            }
        }
        switch (tiles) {
            case 2: return launch_one<2, 64, 2, 8, 0, MODE_PART, 4>(a, st);
            case 3: return launch_one<3, 64, 2, 8, 0, MODE_PART, 4>(a, st);
            case 4: return launch_one<4, 64, 2, 6, 0, MODE_PART, 4>(a, st);
            default: return launch_one<5, 64, 2, 6, 0, MODE_PART, 4>(a, st);
        }
    }
    if (tiles < 2 || tiles > 5 || (kw != 128 && kw != 64) || a.dp % kw != 0 || a.n <= 0 || a.n_blocks < 1 ||
        !mq_waves_available(waves) || (waves == 8 && kw != 64))
        return fail(CSLAM_E_INVALID, "scan_mq: bad launch (tiles %d, slice %d, waves %d, rows %lld)", tiles, kw,
                    waves, (long long)a.n);
#if defined(CSLAM_MQ_DEV_T5)   // development builds: one variant only (seconds to compile; tools/mq_dev_build.sh)
    return launch_mode<5, 128, 2, 6, 4>(a, mode, nt, st);
#else
#if defined(CSLAM_MQ_WAVES8)
    if (waves == 8) {
        switch (tiles) {
            case 2: return launch_mode<2, 64, 3, 8, 8>(a, mode, nt, st);
            case 3: return launch_mode<3, 64, 3, 5, 8>(a, mode, nt, st);
            case 4: return launch_mode<4, 64, 3, 4, 8>(a, mode, nt, st);
            default: return launch_mode<5, 64, 3, 3, 8>(a, mode, nt, st);
        }
    }
#endif
    if (kw == 128) {
        switch (tiles) {
            case 2: return launch_mode<2, 128, 3, 8, 4>(a, mode, nt, st);
            case 3: return launch_mode<3, 128, 3, 8, 4>(a, mode, nt, st);
            case 4: return launch_mode<4, 128, 2, 6, 4>(a, mode, nt, st);
            default: return launch_mode<5, 128, 2, 6, 4>(a, mode, nt, st);
        }
    }
    switch (tiles) {
        case 2: return launch_mode<2, 64, 6, 8, 4>(a, mode, nt, st);
        case 3: return launch_mode<3, 64, 6, 8, 4>(a, mode, nt, st);
        case 4: return launch_mode<4, 64, 6, 6, 4>(a, mode, nt, st);
        default: return launch_mode<5, 64, 6, 6, 4>(a, mode, nt, st);
    }
#endif
}

int launch_mq_combine(const MqArgs& a, int tiles, int nchunks, bool best, int64_t cand_off, hipStream_t st) {
    const int64_t strips = (a.n + 15) >> 4;
    if (strips <= 0) return CSLAM_OK;
    const unsigned grid = (unsigned)((strips * tiles + 3) / 4);   // one wave per (strip, tile)
    if (best)
        hipLaunchKernelGGL((k_mq_combine<true>), dim3(grid), dim3(256), 0, st, a, tiles, nchunks, cand_off);
    else
        hipLaunchKernelGGL((k_mq_combine<false>), dim3(grid), dim3(256), 0, st, a, tiles, nchunks, cand_off);
    CSLAM_HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}

}  // namespace cslam
