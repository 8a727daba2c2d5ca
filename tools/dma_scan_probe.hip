// dma_scan_probe.hip — prototype of the X-through-LDS-DMA scan (round 2).
//
// X rows are streamed HBM -> LDS with global_load_lds_dwordx4 in 1-KiB linear pieces (one row's
// 256-float slice per instruction), per-wave private rings, no barriers; the wave reads the A
// fragments of the 16x16x4 MFMA back with ds_read_b128 (row pitch 1024+32 B: conflict free) and
// re-issues the slot at once.  The queries (B operand) stay in registers for a whole 256-float
// slice and are shared by all the strips of the wave's row panel (k-slice-major order), loaded
// by inline-asm global loads so the compiler never drains the DMA queue for them.
// Chain order and 512-float folds are those of k_scan -> bit-identical dots;
// checked here against a scalar fmaf reference kernel.
//
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/dma_scan_probe.hip -o tools/dma_scan_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int kPieceB = 1024 + 32;       // bytes per row piece in LDS (256 floats + pad)
constexpr int kStageB = 16 * kPieceB;    // one stage = 16 rows x 256 floats

__device__ inline void wait_vm_dyn(int n) {
#define W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
    switch (n) {
        W(0) W(1) W(2) W(3) W(4) W(5) W(6) W(7) W(8) W(9) W(10) W(11) W(12) W(13) W(14) W(15)
        W(16) W(17) W(18) W(19) W(20) W(21) W(22) W(23) W(24) W(25) W(26) W(27) W(28) W(29) W(30) W(31)
        W(32) W(33) W(34) W(35) W(36) W(37) W(38) W(39) W(40) W(41) W(42) W(43) W(44) W(45) W(46) W(47)
        W(48) W(49) W(50) W(51) W(52) W(53) W(54) W(55) W(56) W(57) W(58) W(59) W(60) W(61) W(62)
        default: asm volatile("s_waitcnt vmcnt(63)" ::: "memory"); break;
    }
#undef W
}

// B fragment loads (L2-resident staged queries) as inline asm: invisible to the compiler's
// waitcnt pass, counted by hand together with the DMA pieces
__device__ inline void load_b16(f32x4 (&b)[16], const float* qrow) {
#define LB(I) asm volatile("global_load_dwordx4 %0, %1, off offset:" #I "*64" : "=v"(b[I]) : "v"(qrow) : "memory");
    LB(0) LB(1) LB(2) LB(3) LB(4) LB(5) LB(6) LB(7) LB(8) LB(9) LB(10) LB(11) LB(12) LB(13) LB(14) LB(15)
#undef LB
}
__device__ inline void tie16(f32x4 (&b)[16]) {
    asm volatile("" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]));
    asm volatile("" : "+v"(b[8]), "+v"(b[9]), "+v"(b[10]), "+v"(b[11]), "+v"(b[12]), "+v"(b[13]), "+v"(b[14]), "+v"(b[15]));
}

struct Cursor {      // position of a stage in the wave's work list
    int64_t lo;      // first row of the panel
    int rows;        // rows of the panel (<= 16*NS)
    int ns;          // strips of the panel
    int p;           // panel index (global)
    int kc;          // 256-float slice
    int s;           // strip inside the panel
};

template <int NS, int NW, int MODE, int NST = 2, int AUX = 0>
__global__ __launch_bounds__(NW * 64) void k_scan_dma(const float* __restrict__ X, int64_t xs, int64_t n, int dp,
                                                      const float* __restrict__ Q, float* __restrict__ dots,
                                                      int64_t dots_stride, int n_panels, int ppw) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int r = lane & 15, g = lane >> 4;
    const int w = blockIdx.x * NW + wave;
    char* ring = lds + wave * NST * kStageB;
    const int nkc = dp >> 8;
    const int p_begin = w * ppw;
    int p_end = p_begin + ppw;
    if (p_end > n_panels) p_end = n_panels;
    if (p_begin >= p_end) return;

    auto panel = [&](Cursor& c, int p) {
        c.p = p;
        c.lo = (int64_t)p * n / n_panels;
        const int64_t hi = (int64_t)(p + 1) * n / n_panels;
        c.rows = (int)(hi - c.lo);
        c.ns = (c.rows + 15) >> 4;
        c.kc = 0;
        c.s = 0;
    };
    auto advance = [&](Cursor& c) {   // returns false past the end
        if (++c.s < c.ns) return;
        c.s = 0;
        if (++c.kc < nkc) return;
        panel(c, c.p + 1);            // may be p_end: caller checks
    };
    auto count = [&](const Cursor& c) { int k = c.rows - 16 * c.s; return k > 16 ? 16 : k; };
    auto issue = [&](const Cursor& c, char* slot) {
        const int cnt = count(c);
        const float* src = X + (c.lo + 16 * c.s) * xs + c.kc * 256 + lane * 4;
#pragma unroll
        for (int rr = 0; rr < 16; ++rr)
            if (rr < cnt)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + rr * xs),
                                                 (__attribute__((address_space(3))) void*)(slot + rr * kPieceB), 16, 0, AUX);
        return cnt;
    };

    const float* qrow = Q + (int64_t)r * dp + 4 * g;   // B fragment base of this lane
    f32x4 bc[16], bn[16];
    Cursor ic;                 // issue cursor (runs 2 stages ahead of the consumer)
    panel(ic, p_begin);
    int y[NST];                // ops issued after the stage in slot i
    int yq = 0;                // ops issued after the B loads
#pragma unroll
    for (int i = 0; i < NST; ++i) y[i] = 0;
    auto bump = [&](int k) {
#pragma unroll
        for (int i = 0; i < NST; ++i) y[i] += k;
        yq += k;
    };
    // prologue: B(0) (waited for at once: registers are only trusted behind a tie), stage 0, 1
    load_b16(bn, qrow);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tie16(bn);
#pragma unroll
    for (int i = 0; i < NST; ++i)
        if (ic.p < p_end) { const int c = issue(ic, ring + i * kStageB); bump(c); y[i] = 0; advance(ic); }

    int t = 0;   // ring slot of the consumer
    for (int p = p_begin; p < p_end; ++p) {
        const int64_t lo = (int64_t)p * n / n_panels;
        const int64_t hi = (int64_t)(p + 1) * n / n_panels;
        const int ns = (int)((hi - lo + 15) >> 4);
        f32x4 acc[NS], tot[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) { acc[s] = f32x4{0, 0, 0, 0}; tot[s] = f32x4{0, 0, 0, 0}; }
        for (int kc = 0; kc < nkc; ++kc) {
            // bn holds B(kc), landed and tied.  Move it to bc and start B(kc+1) (wraps to slice 0
            // for the next panel); it is waited for at the end of this slice.
#pragma unroll
            for (int i = 0; i < 16; ++i) bc[i] = bn[i];
            if (MODE & 1) {
                const int kn = kc + 1 < nkc ? kc + 1 : 0;
                load_b16(bn, qrow + kn * 256);
                bump(16); yq = 0;
            }
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                if (s < ns) {
                    const int slot = t;
                    char* sp = ring + slot * kStageB;
                    int yc = y[0];
#pragma unroll
                    for (int i = 1; i < NST; ++i) yc = slot == i ? y[i] : yc;
                    wait_vm_dyn(yc);
                    f32x4 a[16];
                    const char* ap = sp + r * kPieceB + g * 16;
#pragma unroll
                    for (int i = 0; i < 16; ++i) a[i] = (MODE & 4) ? f32x4{1, 2, 3, 4} : *reinterpret_cast<const f32x4*>(ap + i * 64);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    if (ic.p < p_end) {
                        const int c = issue(ic, sp);
                        advance(ic);
                        bump(c);
#pragma unroll
                        for (int i = 0; i < NST; ++i) y[i] = slot == i ? 0 : y[i];
                    }
                    if (MODE & 2) {
#pragma unroll
                        for (int i = 0; i < 16; ++i)
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][j], bc[i][j], acc[s], 0, 0, 0);
                    } else {
#pragma unroll
                        for (int i = 0; i < 16; ++i) acc[s] += a[i] * bc[i];
                    }
                    if ((kc & 1) || kc + 1 == nkc) { tot[s] = tot[s] + acc[s]; acc[s] = f32x4{0, 0, 0, 0}; }
                    t = t + 1 == NST ? 0 : t + 1;
                }
            }
            if (MODE & 1) { wait_vm_dyn(yq); tie16(bn); }
        }
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

// ---------------------------------------------------------------------------------------------
// Half stages: the ring slot is 16 rows x 128 floats (8 KiB + pad), FOUR slots per wave in the
// same LDS.  One DMA instruction moves TWO rows x 512 B (lanes 0-31: the even row of the pair,
// lanes 32-63: the odd one); the odd row's 16-B segments are XOR-1 swizzled on the SOURCE side
// (LDS position j holds global segment j ^ 1), so that the 16 lanes of a ds_read_b128 pass — the
// 16 rows at one k offset — hit 16 different 16-B slots: pair p sits at p * 1056 B (32-B steps
// mod 256), the odd row 512 B behind the even one and one segment aside.
// A slot is in the consumer's hands for 8 ds_read_b128 + 32 MFMAs instead of 16 + 64.
// Work order: (panel, slice kc, strip s, half h); chain order unchanged -> bit-identical.
// ---------------------------------------------------------------------------------------------
constexpr int kHalfB = 8 * kPieceB;

struct CursorH {
    int64_t lo;
    int rows, ns, p, kc, s, h;
};

template <int NS, int NW, int NST, int AUX, int MODE = 3>
__global__ __launch_bounds__(NW * 64) void k_scan_half(const float* __restrict__ X, int64_t xs, int64_t n, int dp,
                                                       const float* __restrict__ Q, float* __restrict__ dots,
                                                       int64_t dots_stride, int n_panels, int ppw) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int r = lane & 15, g = lane >> 4;
    const int w = blockIdx.x * NW + wave;
    char* ring = lds + wave * NST * kHalfB;
    const int nkc = dp >> 8;
    const int p_begin = w * ppw;
    int p_end = p_begin + ppw;
    if (p_end > n_panels) p_end = n_panels;
    if (p_begin >= p_end) return;

    auto panel = [&](CursorH& c, int p) {
        c.p = p;
        c.lo = (int64_t)p * n / n_panels;
        const int64_t hi = (int64_t)(p + 1) * n / n_panels;
        c.rows = (int)(hi - c.lo);
        c.ns = (c.rows + 15) >> 4;
        c.kc = 0;
        c.s = 0;
        c.h = 0;
    };
    auto advance = [&](CursorH& c) {
        if (++c.h < 2) return;
        c.h = 0;
        if (++c.s < c.ns) return;
        c.s = 0;
        if (++c.kc < nkc) return;
        panel(c, c.p + 1);
    };
    // DMA side: lane -> (row of the pair, swizzled 16-B segment)
    const int sub = lane >> 5;
    const int seg = (lane & 31) ^ sub;
    const int64_t sub_off = (int64_t)sub * xs;
    auto issue = [&](const CursorH& c, char* slot) {
        int cnt = c.rows - 16 * c.s;          // valid rows of the strip (>= 1)
        cnt = cnt > 16 ? 16 : cnt;
        const int npair = (cnt + 1) >> 1;
        // even-row address of pair 0 (per lane only through the segment); pair j is 2 j rows on
        // (scalar), the odd row one row further (per-lane constant sub_off) — except for an odd
        // tail, where the missing row repeats the even one
        const float* base = X + (c.lo + 16 * c.s) * xs + c.kc * 256 + c.h * 128 + seg * 4;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (j < npair) {
                const float* a = base + (int64_t)(2 * j) * xs;
                if (2 * j + 1 < cnt) a += sub_off;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)a,
                                                 (__attribute__((address_space(3))) void*)(slot + j * kPieceB), 16, 0, AUX);
            }
        return npair;
    };
    // reader side: row r of the strip, k offset g within a 16-float block
    const int rd_off = (r >> 1) * kPieceB + (r & 1) * 512 + ((g ^ (r & 1)) * 16);

    const float* qrow = Q + (int64_t)r * dp + 4 * g;
    f32x4 bc[16], bn[16];
    CursorH ic;
    panel(ic, p_begin);
    int y[NST];
    int yq = 0;
#pragma unroll
    for (int i = 0; i < NST; ++i) y[i] = 0;
    auto bump = [&](int k) {
#pragma unroll
        for (int i = 0; i < NST; ++i) y[i] += k;
        yq += k;
    };
    load_b16(bn, qrow);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tie16(bn);
#pragma unroll
    for (int i = 0; i < NST; ++i)
        if (ic.p < p_end) { const int c = issue(ic, ring + i * kHalfB); bump(c); y[i] = 0; advance(ic); }

    int t = 0;
    for (int p = p_begin; p < p_end; ++p) {
        const int64_t lo = (int64_t)p * n / n_panels;
        const int64_t hi = (int64_t)(p + 1) * n / n_panels;
        const int ns = (int)((hi - lo + 15) >> 4);
        f32x4 acc[NS], tot[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) { acc[s] = f32x4{0, 0, 0, 0}; tot[s] = f32x4{0, 0, 0, 0}; }
        for (int kc = 0; kc < nkc; ++kc) {
#pragma unroll
            for (int i = 0; i < 16; ++i) bc[i] = bn[i];
            {
                const int kn = kc + 1 < nkc ? kc + 1 : 0;
                load_b16(bn, qrow + kn * 256);
                bump(16); yq = 0;
            }
            // one half stage: wait, read, hand the slot back, 32 MFMAs on bc[8 h ..]
            auto half = [&](f32x4& accs, const f32x4* bh) __attribute__((always_inline)) {
                const int slot = t;
                char* sp = ring + slot * kHalfB;
                int yc = y[0];
#pragma unroll
                for (int i = 1; i < NST; ++i) yc = slot == i ? y[i] : yc;
                wait_vm_dyn(yc);
                f32x4 a[8];
                const char* ap = sp + rd_off;
#pragma unroll
                for (int i = 0; i < 8; ++i) a[i] = (MODE & 4) ? f32x4{1, 2, 3, 4} : *reinterpret_cast<const f32x4*>(ap + i * 64);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                if (ic.p < p_end) {
                    const int c = issue(ic, sp);
                    advance(ic);
                    bump(c);
#pragma unroll
                    for (int i = 0; i < NST; ++i) y[i] = slot == i ? 0 : y[i];
                }
                if (MODE & 2) {
#pragma unroll
                    for (int i = 0; i < 8; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            accs = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][j], bh[i][j], accs, 0, 0, 0);
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) accs += a[i] * bh[i];
                }
                t = t + 1 == NST ? 0 : t + 1;
            };
#pragma clang loop unroll(full)
            for (int s = 0; s < NS; ++s) {
                if (s < ns) {
                    half(acc[s], &bc[0]);
                    half(acc[s], &bc[8]);
                    if ((kc & 1) || kc + 1 == nkc) { tot[s] = tot[s] + acc[s]; acc[s] = f32x4{0, 0, 0, 0}; }
                }
            }
            wait_vm_dyn(yq); tie16(bn);
        }
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

// ---------------------------------------------------------------------------------------------
// Piece-granular ring: the wave's LDS ring is RP one-row pieces (not whole stages); a stage's 16
// slots are refilled with the next pieces in line as soon as it has been read, so RP-16 .. RP
// KiB stay in flight per wave (stage ring of 2: 16 .. 32).
// ---------------------------------------------------------------------------------------------
template <int NS, int NW, int RP, int AUX>
__global__ __launch_bounds__(NW * 64) void k_scan_pr(const float* __restrict__ X, int64_t xs, int64_t n, int dp,
                                                     const float* __restrict__ Q, float* __restrict__ dots,
                                                     int64_t dots_stride, int n_panels, int ppw) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int r = lane & 15, g = lane >> 4;
    const int w = blockIdx.x * NW + wave;
    char* ring = lds + wave * RP * kPieceB;
    const int nkc = dp >> 8;
    const int p_begin = w * ppw;
    int p_end = p_begin + ppw;
    if (p_end > n_panels) p_end = n_panels;
    if (p_begin >= p_end) return;

    auto panel = [&](Cursor& c, int p) {
        c.p = p;
        c.lo = (int64_t)p * n / n_panels;
        const int64_t hi = (int64_t)(p + 1) * n / n_panels;
        c.rows = (int)(hi - c.lo);
        c.ns = (c.rows + 15) >> 4;
        c.kc = 0;
        c.s = 0;
    };
    auto advance = [&](Cursor& c) {
        if (++c.s < c.ns) return;
        c.s = 0;
        if (++c.kc < nkc) return;
        panel(c, c.p + 1);
    };
    auto count = [&](const Cursor& c) { int k = c.rows - 16 * c.s; return k > 16 ? 16 : k; };

    int mark0 = 0, mark1 = 0, mark2 = 0, mark3 = 0;      // `issued` right after a stage's last piece
    int start0 = 0, start1 = 0, start2 = 0, start3 = 0;  // ring position of a stage's first piece
    auto set4 = [](int& a0, int& a1, int& a2, int& a3, int i, int v) {
        a0 = i == 0 ? v : a0; a1 = i == 1 ? v : a1; a2 = i == 2 ? v : a2; a3 = i == 3 ? v : a3;
    };
    auto get4 = [](int a0, int a1, int a2, int a3, int i) { return i == 0 ? a0 : (i == 1 ? a1 : (i == 2 ? a2 : a3)); };

    Cursor ic;
    panel(ic, p_begin);
    int ipiece = 0;      // next piece of the issue cursor's stage
    int head = 0;        // ring position of the next piece
    int nfree = RP;
    int issued = 0;      // vector-memory operations issued so far (pieces and B loads)
    int istage = 0, cstage = 0;   // stages completely issued / consumed
    auto pump = [&]() {
        while (nfree > 0 && ic.p < p_end && istage - cstage < 4) {
            const int cnt = count(ic);
            if (ipiece == 0) set4(start0, start1, start2, start3, istage & 3, head);
            const float* src = X + (ic.lo + 16 * ic.s + ipiece) * xs + ic.kc * 256 + lane * 4;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(ring + head * kPieceB), 16, 0, AUX);
            head = head + 1 == RP ? 0 : head + 1;
            --nfree;
            ++issued;
            if (++ipiece == cnt) {
                set4(mark0, mark1, mark2, mark3, istage & 3, issued);
                ++istage;
                ipiece = 0;
                advance(ic);
            }
        }
    };

    const float* qrow = Q + (int64_t)r * dp + 4 * g;
    f32x4 bc[16], bn[16];
    load_b16(bn, qrow);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tie16(bn);
    pump();
    int qmark = 0;

    for (int p = p_begin; p < p_end; ++p) {
        const int64_t lo = (int64_t)p * n / n_panels;
        const int64_t hi = (int64_t)(p + 1) * n / n_panels;
        const int rows = (int)(hi - lo);
        const int ns = (rows + 15) >> 4;
        f32x4 acc[NS], tot[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) { acc[s] = f32x4{0, 0, 0, 0}; tot[s] = f32x4{0, 0, 0, 0}; }
        for (int kc = 0; kc < nkc; ++kc) {
#pragma unroll
            for (int i = 0; i < 16; ++i) bc[i] = bn[i];
            load_b16(bn, qrow + (kc + 1 < nkc ? kc + 1 : 0) * 256);
            issued += 16;
            qmark = issued;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                if (s < ns) {
                    const int ci = cstage & 3;
                    wait_vm_dyn(issued - get4(mark0, mark1, mark2, mark3, ci));
                    int pos = get4(start0, start1, start2, start3, ci) + r;
                    pos = pos >= RP ? pos - RP : pos;
                    const char* ap = ring + pos * kPieceB + g * 16;
                    f32x4 a[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) a[i] = *reinterpret_cast<const f32x4*>(ap + i * 64);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    int cnt = rows - 16 * s;
                    cnt = cnt > 16 ? 16 : cnt;
                    nfree += cnt;
                    ++cstage;
                    pump();
#pragma unroll
                    for (int i = 0; i < 16; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][j], bc[i][j], acc[s], 0, 0, 0);
                    if ((kc & 1) || kc + 1 == nkc) { tot[s] = tot[s] + acc[s]; acc[s] = f32x4{0, 0, 0, 0}; }
                }
            }
            wait_vm_dyn(issued - qmark);
            tie16(bn);
        }
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

template <int NS, int NW, int RP, int AUX>
void run_pr(const char* name, const float* X, int64_t n, int dp, const float* Q, float* dots, const float* ref, int64_t ds,
            int cus) {
    const int n_waves = cus * NW;
    int ppw = (int)((n + (int64_t)n_waves * 16 * NS - 1) / ((int64_t)n_waves * 16 * NS));
    if (ppw < 1) ppw = 1;
    int n_panels = n_waves * ppw;
    if ((int64_t)n_panels > n) n_panels = (int)n;
    const size_t ldsb = (size_t)NW * RP * kPieceB;
    CK(hipFuncSetAttribute((const void*)k_scan_pr<NS, NW, RP, AUX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    CK(hipMemset(dots, 0xFF, (size_t)16 * ds * 4));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> ts;
    const int blocks = (n_panels + ppw * NW - 1) / (ppw * NW);
    for (int rep = 0; rep < 22; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_scan_pr<NS, NW, RP, AUX>), dim3(blocks), dim3(NW * 64), ldsb, 0, X, (int64_t)dp, n, dp, Q, dots, ds, n_panels, ppw);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2) ts.push_back(ms);
    }
    CK(hipGetLastError());
    std::sort(ts.begin(), ts.end());
    std::vector<float> a((size_t)16 * ds), b((size_t)16 * ds);
    CK(hipMemcpy(a.data(), dots, a.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), ref, b.size() * 4, hipMemcpyDeviceToHost));
    long bad = 0;
    for (int q = 0; q < 16; ++q) for (int64_t i = 0; i < n; ++i) if (memcmp(&a[q * ds + i], &b[q * ds + i], 4)) { if (bad < 3) printf("   mismatch q%d row%lld: %g vs %g\n", q, (long long)i, a[q * ds + i], b[q * ds + i]); ++bad; }
    const double bytes = (double)n * dp * 4;
    printf("%-28s RP=%d AUX=%d NS=%d NW=%d lds=%zu  min %.4f med %.4f ms  %.0f / %.0f GB/s  mismatches=%ld\n", name, RP, AUX, NS, NW, ldsb,
           ts.front(), ts[ts.size() / 2], bytes / ts.front() / 1e6, bytes / ts[ts.size() / 2] / 1e6, bad);
    fflush(stdout);
}

// scalar reference in the canonical chain order
__global__ void k_ref(const float* X, int64_t xs, int64_t n, int dp, const float* Q, float* dots, int64_t ds) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int q = blockIdx.y;
    if (i >= n) return;
    const float* x = X + i * xs;
    const float* qq = Q + (int64_t)q * dp;
    float tot = 0.f, acc = 0.f;
    int nkb = dp / 16;
    for (int kb = 0; kb < nkb; ++kb) {
        for (int j = 0; j < 4; ++j)
            for (int g = 0; g < 4; ++g) { int k = 16 * kb + 4 * g + j; acc = __builtin_fmaf(x[k], qq[k], acc); }
        if ((kb + 1) % 32 == 0 || kb + 1 == nkb) { tot = tot + acc; acc = 0.f; }
    }
    dots[(int64_t)q * ds + i] = tot;
}
__global__ void k_fill_random(float* X, size_t n, unsigned long long seed) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * 256) {
        unsigned long long h = (i + seed) * 0x9E3779B97F4A7C15ull; h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
        X[i] = (float)(h & 0xFFFFFF) / 16777216.0f - 0.25f;
    }
}


template <int NS, int NW, int MODE, int NST = 2, int AUX = 0>
void run(const char* name, const float* X, int64_t n, int dp, const float* Q, float* dots, const float* ref, int64_t ds,
         int cus, bool check) {
    const int n_waves = cus * NW;
    int ppw = (int)((n + (int64_t)n_waves * 16 * NS - 1) / ((int64_t)n_waves * 16 * NS));
    if (ppw < 1) ppw = 1;
    int n_panels = n_waves * ppw;
    if ((int64_t)n_panels > n) { n_panels = (int)n; }
    const size_t ldsb = (size_t)NW * NST * kStageB;
    CK(hipFuncSetAttribute((const void*)k_scan_dma<NS, NW, MODE, NST, AUX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    CK(hipMemset(dots, 0xFF, (size_t)16 * ds * 4));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9, sum = 0; int reps = 12;
    const int blocks = (n_panels + ppw * NW - 1) / (ppw * NW);
    for (int rep = 0; rep < reps; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_scan_dma<NS, NW, MODE, NST, AUX>), dim3(blocks), dim3(NW * 64), ldsb, 0, X, (int64_t)dp, n, dp, Q, dots, ds, n_panels, ppw);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2) { sum += ms; if (ms < best) best = ms; }
    }
    CK(hipGetLastError());
    double bytes = (double)n * dp * 4;
    long bad = -1;
    if (check) {
        std::vector<float> a((size_t)16 * ds), b((size_t)16 * ds);
        CK(hipMemcpy(a.data(), dots, a.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(b.data(), ref, b.size() * 4, hipMemcpyDeviceToHost));
        bad = 0;
        for (int q = 0; q < 16; ++q) for (int64_t i = 0; i < n; ++i) if (memcmp(&a[q * ds + i], &b[q * ds + i], 4)) { if (bad < 5) printf("   mismatch q%d row%lld: %g vs %g\n", q, (long long)i, a[q * ds + i], b[q * ds + i]); ++bad; }
    }
    printf("%-34s NST=%d AUX=%d NS=%d NW=%d MODE=%d blocks=%d ppw=%d  best %.4f ms avg %.4f ms  %.0f GB/s (best) mismatches=%ld\n", name, NST, AUX, NS, NW, MODE, blocks, ppw, best,
           sum / (reps - 2), bytes / best / 1e6, bad);
    fflush(stdout);
}

template <int NS, int NW, int NST, int AUX, int MODE = 3>
void run_half(const char* name, const float* X, int64_t n, int dp, const float* Q, float* dots, const float* ref, int64_t ds,
              int cus, bool check) {
    const int n_waves = cus * NW;
    int ppw = (int)((n + (int64_t)n_waves * 16 * NS - 1) / ((int64_t)n_waves * 16 * NS));
    if (ppw < 1) ppw = 1;
    int n_panels = n_waves * ppw;
    if ((int64_t)n_panels > n) { n_panels = (int)n; }
    const size_t ldsb = (size_t)NW * NST * kHalfB;
    CK(hipFuncSetAttribute((const void*)k_scan_half<NS, NW, NST, AUX, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    CK(hipMemset(dots, 0xFF, (size_t)16 * ds * 4));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9, sum = 0; int reps = check ? 3 : 12;
    const int blocks = (n_panels + ppw * NW - 1) / (ppw * NW);
    for (int rep = 0; rep < reps; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_scan_half<NS, NW, NST, AUX, MODE>), dim3(blocks), dim3(NW * 64), ldsb, 0, X, (int64_t)dp, n, dp, Q, dots, ds, n_panels, ppw);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2 || reps < 4) { sum += ms; if (ms < best) best = ms; }
    }
    CK(hipGetLastError());
    double bytes = (double)n * dp * 4;
    long bad = -1;
    if (check) {
        std::vector<float> a((size_t)16 * ds), b((size_t)16 * ds);
        CK(hipMemcpy(a.data(), dots, a.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(b.data(), ref, b.size() * 4, hipMemcpyDeviceToHost));
        bad = 0;
        for (int q = 0; q < 16; ++q) for (int64_t i = 0; i < n; ++i) if (memcmp(&a[q * ds + i], &b[q * ds + i], 4)) { if (bad < 5) printf("   mismatch q%d row%lld: %g vs %g\n", q, (long long)i, a[q * ds + i], b[q * ds + i]); ++bad; }
    }
    printf("%-34s HALF NST=%d AUX=%d NS=%d NW=%d blocks=%d ppw=%d  best %.4f ms avg %.4f ms  %.0f GB/s (best) mismatches=%ld\n", name, NST, AUX, NS, NW, blocks, ppw, best,
           sum / (reps < 4 ? reps : reps - 2), bytes / best / 1e6, bad);
    fflush(stdout);
}

int main(int argc, char** argv) {
    int64_t n = argc > 1 ? atoll(argv[1]) : 100000;
    int dp = argc > 2 ? atoi(argv[2]) : 4096;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    float *X, *Q, *dots, *ref;
    int64_t ds = (n + 63) / 64 * 64;
    CK(hipMalloc(&X, (size_t)n * dp * 4)); CK(hipMalloc(&Q, (size_t)16 * dp * 4));
    CK(hipMalloc(&dots, (size_t)16 * ds * 4)); CK(hipMalloc(&ref, (size_t)16 * ds * 4));
    hipLaunchKernelGGL(k_fill_random, dim3(4096), dim3(256), 0, 0, X, (size_t)n * dp, 1ull);
    hipLaunchKernelGGL(k_fill_random, dim3(64), dim3(256), 0, 0, Q, (size_t)16 * dp, 77ull);
    CK(hipMemset(ref, 0, (size_t)16 * ds * 4));
    hipLaunchKernelGGL(k_ref, dim3((unsigned)((n + 255) / 256), 16), dim3(256), 0, 0, X, (int64_t)dp, n, dp, Q, ref, ds);
    CK(hipDeviceSynchronize());
    printf("n=%lld dp=%d cus=%d\n", (long long)n, dp, cus);
    if (argc > 3 && atoi(argv[3]) == 1) {
        // half-stage ring vs the shipped whole-stage ring, alternating, bit-exact check first
        run_half<8, 4, 4, 2>("half stages 4w x 4 nt (check)", X, n, dp, Q, dots, ref, ds, cus, true);
        for (int round = 0; round < 2; ++round) {
            run<8, 4, 3, 2, 2>("whole stages 4w x 2 nt", X, n, dp, Q, dots, ref, ds, cus, false);
            run_half<8, 4, 4, 2>("half stages 4w x 4 nt", X, n, dp, Q, dots, ref, ds, cus, false);
            run_half<8, 4, 4, 2, 1>("half: VALU instead of MFMA", X, n, dp, Q, dots, ref, ds, cus, false);
            run_half<8, 4, 4, 2, 5>("half: VALU, no LDS reads", X, n, dp, Q, dots, ref, ds, cus, false);
            run_half<8, 4, 4, 2, 7>("half: MFMA, no LDS reads", X, n, dp, Q, dots, ref, ds, cus, false);
            run_half<8, 4, 2, 2>("half stages 4w x 2 slots nt", X, n, dp, Q, dots, ref, ds, cus, false);
        }
        return 0;
    }
    // cost decomposition of the shipped structure (NS = 8, 4 waves x 2 stages, nt); MODE bits:
    // 1 = B loads (queries from L2), 2 = MFMA (else 64 VALU fma), 4 = NO LDS reads of the stage
    for (int round = 0; round < 2; ++round) {
        run<8, 4, 3, 2, 2>("full: B + LDS reads + MFMA", X, n, dp, Q, dots, ref, ds, cus, round == 0);
        run<8, 4, 2, 2, 2>("no B loads", X, n, dp, Q, dots, ref, ds, cus, false);
        run<8, 4, 1, 2, 2>("B + LDS reads + VALU (no MFMA)", X, n, dp, Q, dots, ref, ds, cus, false);
        run<8, 4, 0, 2, 2>("LDS reads + VALU", X, n, dp, Q, dots, ref, ds, cus, false);
        run<8, 4, 7, 2, 2>("B + MFMA, no LDS reads", X, n, dp, Q, dots, ref, ds, cus, false);
        run<8, 4, 6, 2, 2>("MFMA only", X, n, dp, Q, dots, ref, ds, cus, false);
        run<8, 4, 5, 2, 2>("B + VALU, no LDS reads", X, n, dp, Q, dots, ref, ds, cus, false);
        run<8, 4, 4, 2, 2>("control structure only (VALU)", X, n, dp, Q, dots, ref, ds, cus, false);
    }
    return 0;
}
