// How the multi-tile scan (scan_mq.hip) deals a window's 16-row strips over the waves of the chip: plain host
// arithmetic, kept apart from the pool so that it can be compiled and checked without a GPU
// (tests/native/mq_plan_check.cpp); knn.hip::mq_plan fills the inputs from the pool and the environment knobs.
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>

namespace cslam {

// strips (of 16 rows) per wave and round of k_scan_mq: acc + tot = 8 NS T AccVGPRs of 256
inline int mq_panel_strips_of(int tiles, int waves) {
    if (waves == 8) return tiles >= 5 ? 3 : (tiles == 4 ? 4 : (tiles == 3 ? 5 : 8));
    return tiles >= 4 ? 6 : 8;
}

struct MqPlanIn {
    int64_t vn;             // rows of the window
    int dp;                 // padded row length (floats)
    int cus;                // compute units
    int64_t xs;             // row stride (floats)
    int m;                  // queries of the pass (17..80)
    bool shard;             // called for the sharded step (packs two-tile scans on fewer CUs too)
    bool waves8_available;  // the main launch's eight-wave build (-DCSLAM_MQ_WAVES8) is compiled in
};
// the A/B knobs of KNOBS.md, defaults = what ships
struct MqPlanKnobs {
    int tail = 1, kw = 0, waves = 4, tail_rounds = 6, tail_shape = 0, tail_maxcost = 80;
    int allpart = 1, allpart_groups = 0, allpart_ks = 0, part_waves = 8, pack_blocks = 1;
    int part_blocks2 = 1;   // the all-K-split launch as TWO four-wave blocks per CU (round 5; CSLAM_MQ_PART_BLOCKS2=0: one eight-wave block)
};

// A pass is either ONE launch (every wave whole strips of its block's even share) or, when the
// strips do not divide evenly - 100k rows are 6250 strips for 1024 waves: 6.1 each, and the wave
// with 7 sets the pace - a MAIN launch in which every wave takes exactly q strips, a K-SPLIT launch
// for the remaining tail strips (blocks of 4 strips x 1/ks of the 512-float chunks, chunk partials
// to a scratch array) and a combine launch that folds those in chunk order (same bits).
struct MqPlan {
    int tiles, kw, waves, n_blocks, p_max;
    int part_kw, part_waves;   // slice width and waves per block of the K-split launch
    int part_per_cu;           // blocks of the K-split launch a CU holds at once (1; 2 = the two-block form)
    int64_t n_main;      // rows of the main launch (all of them when there is no tail)
    int tail_groups, tail_ks, nchunks;   // K-split tail: strip groups x K parts
    int64_t tail_strips;
    int64_t n_cand;      // composites per query (best-match mode)
    bool nt;
};
inline MqPlan mq_plan_compute(const MqPlanIn& in, const MqPlanKnobs& kn) {
    const int64_t vn = in.vn;
    const int m = in.m;
    MqPlan g{};
    g.tiles = (m + 15) / 16;
    // A/B: force 64-float slices where 128 would do  (CSLAM_MQ_KW)
    g.kw = in.dp % 128 == 0 && kn.kw != 64 ? 128 : 64;
    // A/B: 8 = two waves per SIMD (builds with -DCSLAM_MQ_WAVES8 only)  (CSLAM_MQ_WAVES)
    g.waves = kn.waves == 8 && in.waves8_available ? 8 : 4;
    if (g.waves == 8) g.kw = 64;                // eight private rings fit the LDS with 4-KiB stages only
    g.part_kw = g.kw;
    g.part_waves = g.waves;
    g.part_per_cu = 1;
    g.nchunks = (in.dp + 511) / 512;
    const int ns = mq_panel_strips_of(g.tiles, g.waves);
    const int64_t strips = (vn + 15) / 16;
    const int64_t want = (strips + g.waves - 1) / g.waves;   // a strip per wave at least
    g.n_blocks = (int)(want < in.cus ? (want < 1 ? 1 : want) : in.cus);
    g.n_main = vn;
    const int64_t waves = (int64_t)g.waves * g.n_blocks;
    const int64_t q = strips / waves, tail = strips - q * waves;
    // q == 0: fewer strips than waves - one strip per wave leaves CUs idle and every wave walks
    // the whole K range alone (5000 rows x 80 queries: 117 us, as long as 16 000 rows); the whole
    // pool then goes the tail's way when the K split pays
    if (kn.tail && tail > 0) {
        // cost in strip times (one strip through all slices): q + 1 as one launch; with the tail
        // K-split: q + (strip, chunk) units of the busiest tail wave / chunks + the extra launches.
        // Two shapes of tail launch:
        //  (a) ONE WAVE OF BLOCKS (round 4): groups x ks <= CUs blocks, each a group of spb strips
        //      (ceil(spb / waves) per wave, walked in rounds inside the block: the DMA ring and the
        //      query slices keep streaming across them) x the chunks of its K part.  25 000 rows:
        //      539 tail strips = 32 groups x 8 parts, 5 units per wave = 0.625 strip times behind
        //      the main launch's one strip - against a second whole strip for every wave;
        //  (b) groups of one strip per wave, several rounds of small blocks handed out by the
        //      dispatcher (round 3): each block a chunk or two of work behind ~4 us of start-up.
        // (b): most rounds of tail blocks  (CSLAM_MQ_TAIL_ROUNDS)
        // A/B: 1 = (a) only, 2 = (b) only  (CSLAM_MQ_TAIL_SHAPE)
        int64_t best_groups = 0, best_ks = 1;
        double part_cost = 1e9;
        const double startup = 0.4 / g.nchunks, launches = 0.2 / g.nchunks;
        if (kn.tail_shape != 2) {
            for (int64_t k = 1; k <= g.nchunks; ++k) {
                int64_t groups = in.cus / k;
                const int64_t most = (tail + g.waves - 1) / g.waves;   // a strip per wave at least
                if (groups > most) groups = most;
                if (groups < 1) continue;
                const int64_t spb = (tail + groups - 1) / groups;          // strips of the largest group
                const int64_t spw = (spb + g.waves - 1) / g.waves;         // ... of its busiest wave
                const int64_t cpb = (g.nchunks + k - 1) / k;               // chunks of the largest K part
                const double c = (double)(spw * cpb) / g.nchunks + startup + launches;
                if (c < part_cost - 1e-9) {
                    part_cost = c;
                    best_groups = groups;
                    best_ks = k;
                }
            }
        }
        if (kn.tail_shape != 1) {
            const int64_t groups = (tail + g.waves - 1) / g.waves;
            for (int64_t k = 1; k <= g.nchunks; ++k) {   // K parts per strip group: the cheapest split
                const int64_t r = (groups * k + in.cus - 1) / in.cus;
                const double c = r * ((double)((g.nchunks + k - 1) / k) / g.nchunks + startup) + launches;
                if (r <= kn.tail_rounds && c < part_cost - 1e-9) {
                    part_cost = c;
                    best_groups = groups;
                    best_ks = k;
                }
            }
        }
        // percent of a strip time the split may cost (the model is optimistic about multi-round tails of
        // shape (b): at 90 / 100 the 25 000-row pool's 80-query pass went 207 -> 216 us, 60 000 rows
        // 368 -> 381, profiles/r03_mq_tail_cost_ab.txt)
        if (best_groups > 0 && part_cost < (q >= 1 ? kn.tail_maxcost * 0.01 : 0.85)) {
            g.n_main = 16 * q * waves;
            g.tail_strips = tail;
            g.tail_groups = (int)best_groups;
            g.tail_ks = (int)best_ks;
        }
    }
    // THE WHOLE WINDOW THROUGH THE K-SPLIT LAUNCH (round 4): no main launch; (strip, 512-float chunk)
    // units dealt over all waves of ONE wave of blocks - groups x ks <= CUs - and folded by
    // k_mq_combine.  A pool of few strips per wave loses whole strip times to quantisation (25 000
    // rows: 1563 strips = two per wave on 784 waves where 1.53 would do; as 32 groups x 8 K parts
    // the busiest wave has 13 units of 12.2 = 1.625 strip times) and the K-split kernel carries
    // no running totals (half the accumulators, no fold arithmetic).  Against that: every chunk
    // partial travels through memory - T / 32 of the pool's bytes written and read again (+31 % HBM
    // traffic at five tiles, which a matrix-pipe-bound pass has room for) - and one more launch.
    // Measured, 80 queries x 4096-d, kernel time of the pass (profiles/r04_mq_plan_ab_*.txt):
    //   12 000 rows 122 -> 81 us, 25 000: 200 -> 164, 40 000: 266 -> 237, 60 000: 364 -> 359-364,
    //   100 000: 553 -> 590: taken up to 3.3 strips per wave, from three query tiles on (two tiles
    //   lean on the HBM, where the extra traffic costs what the quantisation saves).
    // CSLAM_MQ_ALLPART=0: never; =2: whenever a split exists (A/B).  CSLAM_MQ_ALLPART_GROUPS / _KS force one.
    const int64_t most_groups = (strips + g.waves - 1) / g.waves;   // a strip per wave at least
    if (kn.allpart_groups > 0 && kn.allpart_ks > 0 && kn.allpart_ks <= g.nchunks) {
        g.n_main = 0;
        g.tail_strips = strips;
        g.tail_groups = (int)std::min<int64_t>(kn.allpart_groups, most_groups);
        g.tail_ks = kn.allpart_ks;
    } else if (kn.allpart != 0 && kn.tail && strips > 0 &&
               ((g.tiles >= 3 && 10 * strips <= 33 * (int64_t)g.waves * in.cus) || kn.allpart == 2)) {
        // TWO WAVES PER SIMD for this launch (CSLAM_MQ_PART_WAVES=4: one): the K-split kernel carries no
        // running totals, so eight waves of <= 256 registers fit a CU (64-float slices: eight private
        // rings of 4-KiB stages), and the partner wave's MFMAs fill what one wave per SIMD leaves idle at
        // every stage boundary, LDS wait and barrier (in-kernel clock probe: 40 clocks per MFMA with one
        // wave per SIMD, 36 inside the k-blocks, against 32 of the pipe; profiles/r04_mq_clock_probe_1.txt).
        // Round 5: the same two waves per SIMD as TWO independent four-wave blocks per CU - 64-float slices, rings
        // of two stages, 78 KiB of LDS each - so that one block's barrier at a slice step is covered by the other
        // block's MFMAs (the eight waves of one block all stand at it together); the B slices are fetched twice
        // per CU.  Pass time, 80 queries x 4096-d, alternating in one process (profiles/r05_blocks2_ab.txt):
        // 8 000 rows 60.0 -> 57.4 us, 12 000: 80.0 -> 77.9, 25 000: 155.4 -> 151.7, 40 000: 226.9 -> 223.3
        // (48 / 64 queries: -2 ... -4 % on 8 000 - 25 000 rows, even on 40 000).  CSLAM_MQ_PART_BLOCKS2=0: one
        // eight-wave block per CU (round 4).
        const bool two = kn.part_blocks2 != 0 && kn.part_waves == 8 && in.dp % 64 == 0 && g.tiles >= 2;
        const int pw = two ? 4 : (kn.part_waves == 8 && in.dp % 64 == 0 ? 8 : g.waves);
        const int per_cu = two ? 2 : 1, simd_waves = two ? 2 : pw / 4;
        const int64_t most = (strips + pw - 1) / pw;
        // the split with the fewest units on the busiest wave; among equals the fewest K parts
        // (longer rounds, fewer partial stores per step: 40 000 rows 237 us as 64 x 4 or 128 x 2, 245 as 32 x 8)
        int64_t best_u = INT64_MAX, bg = 0, bk = 0;
        for (int64_t k = 1; k <= g.nchunks; ++k) {
            int64_t groups = std::min<int64_t>((int64_t)in.cus * per_cu / k, most);
            if (groups < 1) continue;
            const int64_t spb = (strips + groups - 1) / groups;
            const int64_t spw = (spb + pw - 1) / pw;
            const int64_t cpb = (g.nchunks + k - 1) / k;
            // units per SIMD: the waves of a SIMD take spw strips at most each
            if (spw * cpb * simd_waves < best_u) {
                best_u = spw * cpb * simd_waves;
                bg = groups;
                bk = k;
            }
        }
        // units of the plan made above: whole strips in the main launch + the tail's
        const int64_t q_units = (g.tail_strips > 0 ? q : q + (tail > 0 ? 1 : 0)) * g.nchunks;
        int64_t tail_units = 0;
        if (g.tail_strips > 0) {
            const int64_t spb = (g.tail_strips + g.tail_groups - 1) / g.tail_groups;
            tail_units = ((spb + g.waves - 1) / g.waves) * ((g.nchunks + g.tail_ks - 1) / g.tail_ks);
        }
        if (bg > 0 && (best_u <= q_units + tail_units + (simd_waves == 2 ? 1 : 0) || kn.allpart == 2)) {
            g.n_main = 0;
            g.tail_strips = strips;
            g.tail_groups = (int)bg;
            g.tail_ks = (int)bk;
            g.part_waves = pw;
            g.part_per_cu = per_cu;
            if (pw == 8 || two) g.part_kw = 64;
        }
    }
    const int64_t strips_main = (g.n_main + 15) / 16;
    // A wave's time is its number of whole strips: 1563 strips need two per wave on 256 CUs (6 or 7 per
    // block) and STILL two per wave on 196 CUs (8 per block).  Just enough blocks for that many strips per
    // wave: the makespan is the same and the other CUs stay free for what the caller runs beside the scan -
    // the exchange stream's pack / append / unpack kernels and a collective's own kernel, which cannot
    // share a CU with a scan block (512 registers per lane, 144 KiB of LDS) and otherwise pile up in
    // front of the next scan.  CSLAM_MQ_PACK_BLOCKS=0: one block per CU whenever there are strips for it.
    // Two tiles lean on the HBM and lose 4-6 % of kernel time on fewer CUs: packed only for the shard's
    // own searches, where the exchange stream's kernels beside the scan are worth more (30 gathered
    // descriptors x 25 000 rows: 0.131-0.133 -> 0.126-0.127 ms per step), or with CSLAM_MQ_PACK_BLOCKS=2.
    const bool shard = in.shard;
    if (kn.pack_blocks && strips_main > 0 && (g.tiles >= 3 || shard || kn.pack_blocks == 2)) {
        const int64_t per_wave = (strips_main + (int64_t)g.waves * g.n_blocks - 1) / ((int64_t)g.waves * g.n_blocks);
        const int64_t blocks = (strips_main + g.waves * per_wave - 1) / (g.waves * per_wave);
        if (blocks >= 1 && blocks < g.n_blocks) g.n_blocks = (int)blocks;
    }
    const int64_t strips_max = (strips_main + g.n_blocks - 1) / g.n_blocks;   // blocks take whole strips
    g.p_max = (int)((strips_max + g.waves * ns - 1) / (g.waves * ns));   // 0: everything in the K-split launch
    g.n_cand = (int64_t)g.n_blocks * g.p_max * g.waves + g.tail_strips;
    g.nt = (size_t)vn * in.xs * sizeof(float) > ((size_t)192 << 20);
    return g;
}

}  // namespace cslam
