// mall_reuse_probe.hip (round 5) - can back-to-back scans of the same resident pool share bytes through the
// 256-MiB Infinity Cache?  The pool (100 000 x 4096 fp32 = 1.64 GB) is far larger than the cache, every scan
// streams it once with non-temporal LDS-DMA reads, and the next scan starts at the other end of what the
// cache last saw.  A scan's 512-float chunks are independent chains (only the FOLD of the
// chunk partials is ordered), so a launch may multiply ONE chunk out of turn - first - and hold its partial
// until its turn in the fold.  Scheme under test: launch t ends with a chunk read with the DEFAULT cache policy
// (205 MB: it stays in the Infinity Cache), launch t + 1 starts with that chunk (hits) and ends with another:
//     A: 0 1 2 3 4 5 6 [7]    B: 7 0 1 2 3 4 5 [6]    C: 6 0 1 2 3 4 5 [7]    D = B, E = C, ...
// This probe streams exactly those bytes in exactly that order (nothing is consumed): every wave walks its
// own rows (a panel of its ~6 strips, slice-major inside a chunk, 16 rows x 1 KiB per stage, ring of two
// stages - the shipped k_scan_dma's X side) and reports the average launch time of
//   (a) every launch in order, all non-temporal                       - what ships
//   (b) the scheme above: the last chunk of a launch read with the default policy
//   (c) the scheme's ORDERS, every chunk non-temporal                 - do non-temporal lines stay anyway?
//   (d) order 0..7 with the last chunk read with the default policy   - what the policy alone costs
// build: hipcc -O3 --offload-arch=gfx950 tools/mall_reuse_probe.hip -o tools/mall_reuse_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kPieceB = 1024 + 32;
constexpr int kStageB = 16 * kPieceB;
constexpr int NW = 4, NST = 2;

struct Order {
    int chunk[16];
    int n;        // chunks per row
    int keep;     // chunk read with the default policy (-1: none)
    int keep_lo, keep_hi;   // ... or the slices [keep_lo, keep_hi) (part 2 of the probe: a FIXED resident set, no reorder)
};

__global__ __launch_bounds__(NW * 64) void k_stream(const float* __restrict__ X, int64_t xs, int64_t n, Order o,
                                                    int total_waves, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * NW + wave;
    char* ring = lds + wave * NST * kStageB;
    const int64_t strips = n / 16;
    int slot = 0, issued = 0;
    auto stage = [&](int64_t strip, int kc, bool keep) {
        const float* src = X + strip * 16 * xs + kc * 256 + lane * 4;
        char* sp = ring + slot * kStageB;
        if (issued >= NST) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        if (keep) {
#pragma unroll
            for (int rr = 0; rr < 16; ++rr)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + rr * xs),
                                                 (__attribute__((address_space(3))) void*)(sp + rr * kPieceB), 16, 0, 0);
        } else {
#pragma unroll
            for (int rr = 0; rr < 16; ++rr)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + rr * xs),
                                                 (__attribute__((address_space(3))) void*)(sp + rr * kPieceB), 16, 0, 2);
        }
        slot = slot + 1 == NST ? 0 : slot + 1;
        ++issued;
    };
    const int64_t s_lo = strips * w / total_waves, s_hi = strips * (w + 1) / total_waves;
    for (int ci = 0; ci < o.n; ++ci) {
        const int c = o.chunk[ci];
        for (int kc = 2 * c; kc < 2 * c + 2; ++kc) {        // a chunk = two 256-float slices
            const bool keep = c == o.keep || (kc >= o.keep_lo && kc < o.keep_hi);
            for (int64_t s = s_lo; s < s_hi; ++s) stage(s, kc, keep);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (sink && lane == 0 && w == 0) sink[0] = *reinterpret_cast<float*>(ring);
}

__global__ void k_fill(float* X, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) X[i] = (float)((i * 2654435761ull) & 0xFFFF) * 1e-5f;
}

static Order make(int first, int keep, int nch) {   // first < 0: in order
    Order o{};
    o.n = nch;
    o.keep = keep;
    o.keep_lo = o.keep_hi = 0;
    int k = 0;
    if (first >= 0) o.chunk[k++] = first;
    for (int c = 0; c < nch; ++c)
        if (c != first) o.chunk[k++] = c;
    return o;
}

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 100000;
    const int dp = 4096, nch = dp / 512;
    const int64_t xs = dp + 128;
    int dev = 0;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, dev));
    const int cus = prop.multiProcessorCount;
    float* X;
    CK(hipMalloc(&X, (size_t)n * xs * 4));
    hipLaunchKernelGGL(k_fill, dim3((unsigned)(((size_t)n * xs + 255) / 256)), dim3(256), 0, 0, X, (size_t)n * xs);
    float* sink;
    CK(hipMalloc(&sink, 64));
    const size_t lds = (size_t)NW * NST * kStageB;
    CK(hipFuncSetAttribute((const void*)k_stream, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int total_waves = cus * NW;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto run = [&](const char* name, int variant) {
        // variant 0: in order, nt; 1: the scheme; 2: the scheme's orders, all nt; 3: in order, last chunk kept
        auto order_of = [&](int t) {
            if (variant == 0) return make(-1, -1, nch);
            if (variant == 3) return make(-1, nch - 1, nch);
            Order o = t == 0 ? make(-1, nch - 1, nch) : (t % 2 == 1 ? make(nch - 1, nch - 2, nch) : make(nch - 2, nch - 1, nch));
            if (variant == 2) o.keep = -1;
            return o;
        };
        for (int t = 0; t < 10; ++t)
            hipLaunchKernelGGL(k_stream, dim3(cus), dim3(NW * 64), lds, 0, X, xs, n, order_of(t), total_waves, sink);
        CK(hipDeviceSynchronize());
        const int reps = 40;
        CK(hipEventRecord(e0, 0));
        for (int t = 10; t < 10 + reps; ++t)
            hipLaunchKernelGGL(k_stream, dim3(cus), dim3(NW * 64), lds, 0, X, xs, n, order_of(t), total_waves, sink);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double per = ms / reps;
        printf("%-72s %.4f ms per launch  %.0f GB/s\n", name, per, (double)n * dp * 4 / (per * 1e-3) / 1e9);
    };
    printf("n=%lld dp=%d cus=%d: pool %.0f MB, one chunk of every row %.0f MB\n", (long long)n, dp, cus, n * dp * 4 / 1e6,
           n * 512 * 4 / 1e6);
    // part 2: no reorder at all - a FIXED set of slices is always read with the default policy, everything else
    // non-temporal (non-temporal reads do not allocate: does the fixed set stay resident from launch to launch?)
    auto run_fixed = [&](int lo, int hi) {
        Order o = make(-1, -1, nch);
        o.keep_lo = lo;
        o.keep_hi = hi;
        for (int t = 0; t < 10; ++t)
            hipLaunchKernelGGL(k_stream, dim3(cus), dim3(NW * 64), lds, 0, X, xs, n, o, total_waves, sink);
        CK(hipDeviceSynchronize());
        const int reps = 40;
        CK(hipEventRecord(e0, 0));
        for (int t = 0; t < reps; ++t)
            hipLaunchKernelGGL(k_stream, dim3(cus), dim3(NW * 64), lds, 0, X, xs, n, o, total_waves, sink);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double per = ms / reps;
        printf("fixed set: slices [%2d, %2d) of 16 with the default policy (%4.0f MB)    %.4f ms per launch  %.0f GB/s\n", lo, hi,
               (hi - lo) * n * 1024.0 / 1e6, per, (double)n * dp * 4 / (per * 1e-3) / 1e9);
    };
    if (argc > 2) {   // part 3: a mid-size pool - how much of it should stay resident?
        for (int rep = 0; rep < 2; ++rep)
            for (int hi = 0; hi <= 16; hi += 2) run_fixed(0, hi);
        return 0;
    }
    for (int rep = 0; rep < 2; ++rep) {
        run_fixed(0, 0);
        run_fixed(15, 16);
        run_fixed(14, 16);
        run_fixed(13, 16);
        run_fixed(12, 16);
        run_fixed(0, 2);
        run_fixed(7, 9);
        run_fixed(0, 16);
    }
    for (int rep = 0; rep < 3; ++rep) {
        run("(a) in order, non-temporal", 0);
        run("(b) last chunk kept (default policy), next launch starts with it", 1);
        run("(c) the same orders, every chunk non-temporal", 2);
        run("(d) in order, last chunk with the default policy", 3);
    }
    return 0;
}
