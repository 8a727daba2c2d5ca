// Host-side check of the multi-tile scan's planning (csrc/mq_plan.hpp; tests/test_mq_plan_cpu.py compiles and runs
// this with hipcc, no GPU needed): invariants over a grid of pool sizes / descriptor lengths / query counts / knob
// settings, and the plans DESIGN.md quotes for the benchmark shapes on a 256-CU part.
#include <cstdio>
#include <initializer_list>

#include "mq_plan.hpp"

using namespace cslam;

static int fails = 0;
#define EXPECT(c, ...) do { if (!(c)) { if (fails++ < 25) { printf("FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } } } while (0)

static void invariants(const MqPlanIn& in, const MqPlanKnobs& kn, const MqPlan& g) {
    const int64_t strips = (in.vn + 15) / 16;
    const int64_t strips_main = (g.n_main + 15) / 16;
    const char* tag = "vn %lld dp %d m %d";
#define T_ tag, (long long)in.vn, in.dp, in.m
    EXPECT(g.tiles == (in.m + 15) / 16 && g.tiles >= 2 && g.tiles <= 5, T_);
    EXPECT((g.kw == 128 || g.kw == 64) && in.dp % g.kw == 0 && g.nchunks == (in.dp + 511) / 512, T_);
    EXPECT(g.n_main >= 0 && g.n_main <= in.vn && g.tail_strips >= 0, T_);
    EXPECT(strips_main + g.tail_strips == strips, "strips %lld = %lld main + %lld tail; " "vn %lld dp %d m %d", (long long)strips,
           (long long)strips_main, (long long)g.tail_strips, (long long)in.vn, in.dp, in.m);
    if (g.tail_strips > 0) {
        EXPECT(g.n_main % 16 == 0, T_);                                            // the tail starts on a strip boundary
        EXPECT(g.tail_groups >= 1 && g.tail_ks >= 1 && g.tail_ks <= g.nchunks, T_);
        EXPECT((int64_t)g.tail_groups * g.tail_ks <= (int64_t)in.cus * g.part_per_cu || kn.tail_shape == 2 || g.n_main > 0, T_);   // one wave of blocks
        EXPECT(g.part_per_cu == 1 || (g.part_per_cu == 2 && g.part_waves == 4 && g.part_kw == 64 && g.n_main == 0 && in.dp % 64 == 0), T_);
        EXPECT(g.tail_groups <= g.tail_strips, T_);
        EXPECT((g.part_waves == 4 || g.part_waves == 8) && (g.part_kw == 64 || g.part_kw == g.kw), T_);
        if (g.part_waves == 8) EXPECT(g.part_kw == 64 && in.dp % 64 == 0, T_);
    } else {
        EXPECT(g.n_main == in.vn, T_);
    }
    if (g.n_main == 0) EXPECT(g.p_max == 0, T_);
    else {
        const int ns = mq_panel_strips_of(g.tiles, g.waves);
        EXPECT(g.n_blocks >= 1 && g.n_blocks <= in.cus, T_);
        EXPECT(g.p_max >= 1 && (int64_t)g.p_max * g.waves * ns * g.n_blocks >= strips_main, T_);   // every strip has a panel
        EXPECT(strips_main >= g.n_blocks, T_);                                                       // no block without strips
    }
    EXPECT(g.n_cand == (int64_t)g.n_blocks * g.p_max * g.waves + g.tail_strips, T_);
    EXPECT(g.nt == ((size_t)in.vn * in.xs * 4 > ((size_t)192 << 20)), T_);
#undef T_
}

int main() {
    long cases = 0;
    const MqPlanKnobs def;
    for (int dp : {64, 128, 192, 256, 512, 1024, 2048, 4096, 8192})
        for (int m : {17, 32, 33, 48, 64, 80})
            for (int64_t vn : {1LL, 15LL, 16LL, 17LL, 640LL, 4100LL, 5000LL, 9000LL, 12000LL, 16383LL, 20011LL, 25000LL, 40000LL,
                               60000LL, 70000LL, 100000LL, 100001LL, 300000LL, 1000000LL})
                for (int shard : {0, 1})
                    for (int cus : {256, 304, 64}) {
                        MqPlanIn in{vn, dp, cus, dp + (dp * 4 % 4096 == 0 ? 128 : 0), m, shard != 0, false};
                        invariants(in, def, mq_plan_compute(in, def));
                        ++cases;
                        for (int variant = 0; variant < 7; ++variant) {
                            MqPlanKnobs kn;
                            if (variant == 0) kn.tail = 0;
                            if (variant == 1) kn.allpart = 0;
                            if (variant == 2) kn.allpart = 2;
                            if (variant == 3) { kn.allpart = 0; kn.tail_shape = 2; }
                            if (variant == 4) kn.part_waves = 4;
                            if (variant == 5) { kn.kw = 64; kn.pack_blocks = 0; }
                            if (variant == 6) kn.part_blocks2 = 0;
                            const MqPlan g = mq_plan_compute(in, kn);
                            invariants(in, kn, g);
                            if (variant == 0) EXPECT(g.tail_strips == 0 && g.n_main == vn, "CSLAM_MQ_TAIL=0 keeps one launch");
                            // (a window of fewer strips than waves goes the tail's way whole when the K split pays - the tail rule's q == 0 case)
                            if (variant == 1 && (vn + 15) / 16 >= (int64_t)4 * g.n_blocks && (vn + 15) / 16 >= (int64_t)4 * cus)
                                EXPECT(g.n_main > 0, "CSLAM_MQ_ALLPART=0 keeps a main launch: vn %lld dp %d m %d cus %d", (long long)vn, dp, m, cus);
                            ++cases;
                        }
                    }
    // the benchmark shapes on the MI355X (256 CUs), as DESIGN.md section 5 quotes them
    auto plan = [&](int64_t vn, int dp, int m, bool shard = false) {
        MqPlanIn in{vn, dp, 256, dp + (dp * 4 % 4096 == 0 ? 128 : 0), m, shard, false};
        return mq_plan_compute(in, def);
    };
    {   // cfg4: 25 000 rows x 80 gathered descriptors: the whole window through the K-split launch, 64 groups x 8 K parts
        // of four-wave blocks, two per CU (round 4: 32 x 8 eight-wave blocks)
        const MqPlan g = plan(25000, 4096, 80, true);
        EXPECT(g.n_main == 0 && g.tail_strips == 1563 && g.tail_groups == 64 && g.tail_ks == 8 && g.part_waves == 4 && g.part_kw == 64 &&
                   g.part_per_cu == 2,
               "cfg4 plan: n_main %lld, %d x %d, %d waves, %d per CU", (long long)g.n_main, g.tail_groups, g.tail_ks, g.part_waves, g.part_per_cu);
        MqPlanKnobs r4;
        r4.part_blocks2 = 0;
        MqPlanIn in{25000, 4096, 256, 4096 + 128, 80, true, false};
        const MqPlan g4 = mq_plan_compute(in, r4);
        EXPECT(g4.n_main == 0 && g4.tail_groups == 32 && g4.tail_ks == 8 && g4.part_waves == 8 && g4.part_kw == 64 && g4.part_per_cu == 1,
               "cfg4 plan, one eight-wave block per CU: %d x %d, %d waves", g4.tail_groups, g4.tail_ks, g4.part_waves);
    }
    {   // 40 000 rows (2500 strips): 128 groups x 4 K parts - 20 units on the busiest SIMD (64 x 8 has 20 too: the fewer K parts win)
        const MqPlan g = plan(40000, 4096, 80);
        EXPECT(g.n_main == 0 && g.tail_groups == 128 && g.tail_ks == 4 && g.part_per_cu == 2, "40 000-row plan: n_main %lld, %d x %d", (long long)g.n_main,
               g.tail_groups, g.tail_ks);
    }
    {   // 100 000 rows: every wave exactly six strips in the main launch, 106 strips in the K-split tail
        const MqPlan g = plan(100000, 4096, 80);
        EXPECT(g.n_main == 16 * 6 * 1024 && g.tail_strips == 106 && g.n_blocks == 256 && g.p_max == 1 && g.waves == 4 && g.kw == 128,
               "100 000-row plan: n_main %lld, tail %lld, blocks %d, p_max %d", (long long)g.n_main, (long long)g.tail_strips, g.n_blocks,
               g.p_max);
        EXPECT((int64_t)g.tail_groups * g.tail_ks <= 256, "100 000-row tail: %d x %d", g.tail_groups, g.tail_ks);
    }
    {   // two query tiles lean on the HBM: never all K-split
        const MqPlan g = plan(25000, 1024, 32);
        EXPECT(g.n_main > 0, "two tiles took the all-K-split form");
    }
    {   // three tiles, fewer strips than waves: all K-split
        const MqPlan g = plan(12000, 1024, 48);
        EXPECT(g.n_main == 0 && g.tail_strips == 750, "12 000 x 48: n_main %lld", (long long)g.n_main);
    }
    {   // 4.3 strips per wave: beyond the all-K-split bound
        const MqPlan g = plan(70000, 1024, 80);
        EXPECT(g.n_main > 0, "70 000 rows took the all-K-split form");
    }
    printf("%s %ld plans, %d failures\n", fails ? "FAILED" : "ok", cases, fails);
    return fails ? 1 : 0;
}
