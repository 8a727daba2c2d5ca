// Host-side check of the partitions of topk_parts.hpp (tests/test_topk_parts_cpu.py compiles and runs this with
// hipcc; no GPU needed): for many window sizes and launch geometries the parts must tile the window exactly - in
// part order for the waves and strips, as a disjoint cover for the panels (blocks with fewer rounds than p_max
// own empty parts) - and a panel's rows must be the ones the scan's own arithmetic gives that panel
// (scan_mq.hip: the block's strips split evenly over nw x P panels, panel q = nw * round + wave).
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "topk_parts.hpp"

using namespace cslam;

static int fails = 0;
#define EXPECT(c, ...) do { if (!(c)) { if (fails++ < 20) { printf("FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } } } while (0)

static uint64_t rng_state = 88172645463325252ull;
static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

int main() {
    long cases = 0;
    // WaveParts: n rows over n_waves, in order, sizes differ by at most one
    for (int it = 0; it < 4000; ++it) {
        const int64_t n = 1 + (int64_t)(rnd() % 300000);
        const int nw = 1 + (int)(rnd() % 1024);
        const WaveParts p{n / nw, n % nw};
        int64_t next = 0;
        for (int c = 0; c < nw; ++c) {
            int64_t lo; int cnt;
            p.range(c, lo, cnt);
            EXPECT(lo == next && cnt >= 0 && (cnt == n / nw || cnt == n / nw + 1), "wave parts n=%lld nw=%d c=%d", (long long)n, nw, c);
            next = lo + cnt;
        }
        EXPECT(next == n, "wave parts cover n=%lld nw=%d", (long long)n, nw);
        ++cases;
    }
    // StripParts
    for (int it = 0; it < 2000; ++it) {
        const int64_t n = 1 + (int64_t)(rnd() % 5000);
        const StripParts p{n};
        int64_t next = 0;
        const int S = (int)((n + 15) / 16);
        for (int c = 0; c < S + 2; ++c) {
            int64_t lo; int cnt;
            p.range(c, lo, cnt);
            if (c < S) { EXPECT(lo == next && cnt >= 1 && cnt <= 16, "strip parts n=%lld c=%d", (long long)n, c); next = lo + cnt; }
            else EXPECT(cnt == 0, "strip parts past the end n=%lld c=%d", (long long)n, c);
        }
        EXPECT(next == n, "strip parts cover n=%lld", (long long)n);
        ++cases;
    }
    // PanelParts: main launch (n_blocks blocks x p_max rounds x nw waves, panels of <= ns strips) + K-split strips
    for (int it = 0; it < 6000; ++it) {
        const int nw = (rnd() & 1) ? 4 : 8;
        const int ns = 3 + (int)(rnd() % 6);
        const int n_blocks = 1 + (int)(rnd() % 256);
        int64_t n_main = (rnd() % 5 == 0) ? 0 : (int64_t)(rnd() % 200000);
        int64_t tail_rows = (rnd() % 3 == 0) ? 0 : 1 + (int64_t)(rnd() % 40000);
        if (tail_rows > 0) n_main = n_main / 16 * 16;          // a tail exists only behind whole strips of the main launch
        if (n_main == 0 && tail_rows == 0) tail_rows = 17;
        const int64_t n = n_main + tail_rows;
        const int64_t S = (n_main + 15) / 16;
        // what mq_plan computes: blocks take an even share of the strips; p_max = rounds of the block with the most
        const int64_t strips_max = (S + n_blocks - 1) / n_blocks;
        const int p_max = n_main > 0 ? (int)((strips_max + nw * ns - 1) / (nw * ns)) : 0;
        const int n_main_parts = n_blocks * p_max * nw;
        const int tail_strips = (int)((tail_rows + 15) / 16);
        const PanelParts p{n_main, n, n_blocks, p_max, nw, ns, n_main_parts};
        std::vector<char> seen((size_t)n, 0);
        // the scan's own arithmetic, block by block
        const int64_t bq = n_blocks ? S / n_blocks : 0, brem = S - bq * n_blocks;
        for (int b = 0; b < n_blocks && n_main > 0; ++b) {
            const int64_t bs_lo = (int64_t)b * bq + (b < brem ? b : brem);
            const int bs = (int)(bq + (b < brem ? 1 : 0));
            const int P = bs ? (bs + nw * ns - 1) / (nw * ns) : 0;
            EXPECT(P <= p_max, "rounds of block %d exceed p_max", b);
            for (int rd = 0; rd < p_max; ++rd)
                for (int w = 0; w < nw; ++w) {
                    int64_t lo; int cnt;
                    p.range((b * p_max + rd) * nw + w, lo, cnt);
                    int64_t want_lo = 0, want_hi = 0;
                    if (bs > 0 && rd < P) {
                        const int npan = nw * P, pq = bs / npan, prem = bs - pq * npan, q = nw * rd + w;
                        want_lo = (bs_lo + (int64_t)q * pq + (q < prem ? q : prem)) * 16;
                        want_hi = (bs_lo + (int64_t)(q + 1) * pq + (q + 1 < prem ? q + 1 : prem)) * 16;
                        if (want_hi > n_main) want_hi = n_main;
                        EXPECT((want_hi - want_lo + 15) / 16 <= ns, "a panel of more than ns strips");
                    }
                    const int want = want_hi > want_lo ? (int)(want_hi - want_lo) : 0;
                    EXPECT(cnt == want && (cnt == 0 || lo == want_lo), "panel part b=%d rd=%d w=%d: %lld+%d, want %lld+%d", b, rd, w,
                           (long long)lo, cnt, (long long)want_lo, want);
                    for (int i = 0; i < cnt; ++i) { EXPECT(!seen[(size_t)(lo + i)], "row %lld twice", (long long)(lo + i)); seen[(size_t)(lo + i)] = 1; }
                }
        }
        for (int s = 0; s < tail_strips; ++s) {
            int64_t lo; int cnt;
            p.range(n_main_parts + s, lo, cnt);
            EXPECT(lo == n_main + 16 * (int64_t)s && cnt == (n - lo < 16 ? (int)(n - lo) : 16), "tail strip %d", s);
            for (int i = 0; i < cnt; ++i) { EXPECT(!seen[(size_t)(lo + i)], "row %lld twice", (long long)(lo + i)); seen[(size_t)(lo + i)] = 1; }
        }
        long covered = 0;
        for (char c : seen) covered += c;
        EXPECT(covered == n, "panel parts cover %ld of %lld rows (n_main %lld, blocks %d, p_max %d, nw %d, ns %d)", covered, (long long)n,
               (long long)n_main, n_blocks, p_max, nw, ns);
        ++cases;
    }
    printf("%s %ld cases, %d failures\n", fails ? "FAILED" : "ok", cases, fails);
    return fails ? 1 : 0;
}
