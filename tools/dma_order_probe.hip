// dma_order_probe.hip — how much does the ORDER in which a wave walks its rows matter to a pure
// LDS-DMA read stream?  (round 2: the shipped k_scan_dma walks a <= 8-strip panel slice-major,
// 1 KiB of a row per visit, 128 row streams per wave; is there head-room in longer contiguous
// visits / fewer concurrent row streams?)
//
// Every wave streams its share of X[n][dp] HBM -> LDS with global_load_lds_dwordx4, 16 rows x 1 KiB
// per stage, ring of NST stages, nothing is consumed.  Order: for each panel of S strips:
//   for kb in blocks of P slices:  for strip in panel:  for kc in the P slices:  stage(strip, kc)
// P = 1, S = 8 is the shipped kernel's order; P = dp/256 is row-contiguous.  MODE 1 = "march":
// all waves walk one linear window together (copy-kernel order; no row ownership).
//
// build: hipcc -O3 --offload-arch=gfx950 tools/dma_order_probe.hip -o tools/dma_order_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kPieceB = 1024 + 32;
constexpr int kStageB = 16 * kPieceB;

template <int NW, int NST, int AUX>
__global__ __launch_bounds__(NW * 64) void k_stream(const float* __restrict__ X, int64_t xs, int64_t n, int dp,
                                                    int S, int P, int mode, int total_waves, float* sink, int d1, int d2) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * NW + wave;
    char* ring = lds + wave * NST * kStageB;
    const int nkc = dp >> 8;
    const int64_t strips = n / 16;
    int slot = 0;
    int issued = 0;
    auto stage = [&](int64_t strip, int kc) {
        const float* src = X + strip * 16 * xs + kc * 256 + lane * 4;
        char* sp = ring + slot * kStageB;
        // the slot's previous stage must have landed before it is overwritten: all but the
        // youngest (NST-1) stages
        if (issued >= NST) {
            if (NST == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if (NST == 3) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
        }
        // d1: idle time between "landed" and the re-issue (the real kernel's 16 ds_read_b128),
        // d2: idle time after the re-issue (its 64 MFMAs), in units of 64 clocks
        for (int t = 0; t < d1; ++t) __builtin_amdgcn_s_sleep(1);
#pragma unroll
        for (int rr = 0; rr < 16; ++rr)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + rr * xs),
                                             (__attribute__((address_space(3))) void*)(sp + rr * kPieceB), 16, 0, AUX);
        for (int t = 0; t < d2; ++t) __builtin_amdgcn_s_sleep(1);
        slot = slot + 1 == NST ? 0 : slot + 1;
        ++issued;
    };
    if (mode == 1) {
        // march: stage index g = it * total_waves + w over (strip, kc) in row-major order
        const int64_t total = strips * nkc;
        for (int64_t g = w; g < total; g += total_waves) stage(g / nkc, (int)(g % nkc));
    } else {
        const int64_t s_lo = strips * w / total_waves, s_hi = strips * (w + 1) / total_waves;
        for (int64_t p0 = s_lo; p0 < s_hi; p0 += S) {
            const int64_t p1 = p0 + S < s_hi ? p0 + S : s_hi;
            for (int kb = 0; kb < nkc; kb += P)
                for (int64_t s = p0; s < p1; ++s)
                    for (int kc = kb; kc < kb + P && kc < nkc; ++kc) stage(s, kc);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (sink && lane == 0 && w == 0) sink[0] = *reinterpret_cast<float*>(ring);
}

// Half stages: 16 rows x 128 floats per stage = 8 DMA instructions of TWO rows x 512 B each
// (lanes 0-31 one row, lanes 32-63 the next; the odd row's 16-B segments XOR-1 swizzled on the
// source side as a conflict-free LDS image would need), ring of NST slots of 8 x 1056 B.  Same
// bytes in the LDS as 2 whole stages with NST = 4, but a slot is half as long in the consumer's
// hands.  Order: strip-major inside a panel slice: (kc, strip, half).
template <int NW, int NST, int AUX>
__global__ __launch_bounds__(NW * 64) void k_stream_half(const float* __restrict__ X, int64_t xs, int64_t n, int dp,
                                                         int S, int total_waves, float* sink, int d1, int d2) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int kHalfB = 8 * kPieceB;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * NW + wave;
    char* ring = lds + wave * NST * kHalfB;
    const int nkc = dp >> 8;
    const int64_t strips = n / 16;
    const int sub = lane >> 5;                       // which row of the pair
    const int seg = (lane & 31) ^ sub;               // XOR-1 swizzle for the odd row
    int slot = 0, issued = 0;
    auto stage = [&](int64_t strip, int kc, int h) {
        const float* src = X + (strip * 16 + sub) * xs + kc * 256 + h * 128 + seg * 4;
        char* sp = ring + slot * kHalfB;
        if (issued >= NST) {
            if (NST == 4) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
        for (int t = 0; t < d1; ++t) __builtin_amdgcn_s_sleep(1);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 2 * j * xs),
                                             (__attribute__((address_space(3))) void*)(sp + j * kPieceB), 16, 0, AUX);
        for (int t = 0; t < d2; ++t) __builtin_amdgcn_s_sleep(1);
        slot = slot + 1 == NST ? 0 : slot + 1;
        ++issued;
    };
    const int64_t s_lo = strips * w / total_waves, s_hi = strips * (w + 1) / total_waves;
    for (int64_t p0 = s_lo; p0 < s_hi; p0 += S) {
        const int64_t p1 = p0 + S < s_hi ? p0 + S : s_hi;
        for (int kc = 0; kc < nkc; ++kc)
            for (int64_t s = p0; s < p1; ++s)
                for (int h = 0; h < 2; ++h) stage(s, kc, h);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (sink && lane == 0 && w == 0) sink[0] = *reinterpret_cast<float*>(ring);
}

__global__ void k_fill(float* X, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) X[i] = (float)((i * 2654435761ull) & 0xFFFF) * 1e-5f;
}

static int64_t g_xs = 4096;   // row stride in floats (argv[1] = padding floats per row)

template <int NW, int NST, int AUX>
void run(const char* name, const float* X, int64_t n, int dp, int S, int P, int mode, int cus, float* sink, int d1 = 0, int d2 = 0) {
    const size_t lds = (size_t)NW * NST * kStageB;
    CK(hipFuncSetAttribute((const void*)k_stream<NW, NST, AUX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e9f, sum = 0.f;
    const int reps = 12;
    for (int i = 0; i < reps + 3; ++i) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_stream<NW, NST, AUX>), dim3(cus), dim3(NW * 64), lds, 0, X, g_xs, n, dp, S, P, mode,
                           cus * NW, sink, d1, d2);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (i >= 3) { best = ms < best ? ms : best; sum += ms; }
    }
    const double bytes = (double)n * dp * 4;
    printf("%-44s d1=%d d2=%d NW=%d NST=%d AUX=%d S=%d P=%d mode=%d  best %.4f ms avg %.4f ms  %.0f GB/s (best) %.0f (avg)\n", name, d1, d2, NW, NST,
           AUX, S, P, mode, best, sum / reps, bytes / best / 1e6, bytes / (sum / reps) / 1e6);
    fflush(stdout);
}

template <int NW, int NST, int AUX>
void run_half(const char* name, const float* X, int64_t n, int dp, int S, int cus, float* sink, int d1, int d2) {
    const size_t lds = (size_t)NW * NST * 8 * kPieceB;
    CK(hipFuncSetAttribute((const void*)k_stream_half<NW, NST, AUX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e9f, sum = 0.f;
    const int reps = 12;
    for (int i = 0; i < reps + 3; ++i) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_stream_half<NW, NST, AUX>), dim3(cus), dim3(NW * 64), lds, 0, X, g_xs, n, dp, S,
                           cus * NW, sink, d1, d2);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (i >= 3) { best = ms < best ? ms : best; sum += ms; }
    }
    const double bytes = (double)n * dp * 4;
    printf("%-44s d1=%d d2=%d NW=%d NST=%d AUX=%d S=%d half stages  best %.4f ms avg %.4f ms  %.0f GB/s (best) %.0f (avg)\n", name,
           d1, d2, NW, NST, AUX, S, best, sum / reps, bytes / best / 1e6, bytes / (sum / reps) / 1e6);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int64_t n = 100000 / 16 * 16;
    const int dp = 4096;
    const int pad = argc > 1 ? atoi(argv[1]) : 0;
    g_xs = dp + pad;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float *X, *sink;
    CK(hipMalloc(&X, (size_t)n * g_xs * 4));
    CK(hipMalloc(&sink, 64));
    k_fill<<<(unsigned)(((size_t)n * g_xs + 255) / 256), 256>>>(X, (size_t)n * g_xs);
    CK(hipDeviceSynchronize());
    printf("n=%lld dp=%d row stride=%lld floats cus=%d\n", (long long)n, dp, (long long)g_xs, cus);
    if (argc > 1) {   // padded rows: the short list
        for (int rep = 0; rep < 3; ++rep) {
            run<4, 2, 2>("shipped order (P=1,S=8) nt", X, n, dp, 8, 1, 0, cus, sink);
            run<4, 2, 2>("row-contiguous (P=16) nt", X, n, dp, 8, 16, 0, cus, sink);
            run<4, 2, 2>("nt, gap 512 + mfma 2048", X, n, dp, 8, 1, 0, cus, sink, 8, 32);
        }
        return 0;
    }
    run<4, 2, 2>("shipped order (P=1,S=8) nt", X, n, dp, 8, 1, 0, cus, sink);
    run<4, 2, 0>("shipped order (P=1,S=8)", X, n, dp, 8, 1, 0, cus, sink);
    run<4, 2, 2>("P=2 S=8 nt", X, n, dp, 8, 2, 0, cus, sink);
    run<4, 2, 2>("P=4 S=8 nt", X, n, dp, 8, 4, 0, cus, sink);
    run<4, 2, 2>("P=1 S=4 nt", X, n, dp, 4, 1, 0, cus, sink);
    run<4, 2, 2>("P=1 S=2 nt", X, n, dp, 2, 1, 0, cus, sink);
    run<4, 2, 2>("P=2 S=4 nt", X, n, dp, 4, 2, 0, cus, sink);
    run<4, 2, 2>("row-contiguous (P=16) nt", X, n, dp, 8, 16, 0, cus, sink);
    run<4, 2, 0>("row-contiguous (P=16)", X, n, dp, 8, 16, 0, cus, sink);
    run<4, 2, 2>("march (copy order) nt", X, n, dp, 8, 1, 1, cus, sink);
    run<4, 2, 0>("march (copy order)", X, n, dp, 8, 1, 1, cus, sink);
    run<3, 3, 2>("3 waves x 3 stages, shipped order nt", X, n, dp, 8, 1, 0, cus, sink);
    run<3, 3, 2>("3 waves x 3 stages, row-contiguous nt", X, n, dp, 8, 16, 0, cus, sink);
    run<3, 3, 2>("3 waves x 3 stages, march nt", X, n, dp, 8, 1, 1, cus, sink);
    run<2, 4, 2>("2 waves x 4 stages, row-contiguous nt", X, n, dp, 8, 16, 0, cus, sink);
    run<2, 4, 2>("2 waves x 4 stages, march nt", X, n, dp, 8, 1, 1, cus, sink);
    run<1, 4, 2>("1 wave x 4 stages, march nt", X, n, dp, 8, 1, 1, cus, sink);
    run<4, 2, 2>("shipped order again nt", X, n, dp, 8, 1, 0, cus, sink);
    // emulate the consumer's timing: d1 x 64 clocks before the re-issue, d2 x 64 after it
    run<4, 2, 2>("nt, ds_read gap 256 clk", X, n, dp, 8, 1, 0, cus, sink, 4, 0);
    run<4, 2, 2>("nt, ds_read gap 512 clk", X, n, dp, 8, 1, 0, cus, sink, 8, 0);
    run<4, 2, 2>("nt, ds_read gap 1024 clk", X, n, dp, 8, 1, 0, cus, sink, 16, 0);
    run<4, 2, 2>("nt, mfma 2048 clk after issue", X, n, dp, 8, 1, 0, cus, sink, 0, 32);
    run<4, 2, 2>("nt, gap 512 + mfma 2048", X, n, dp, 8, 1, 0, cus, sink, 8, 32);
    run<4, 2, 2>("nt, gap 512 + mfma 3072", X, n, dp, 8, 1, 0, cus, sink, 8, 48);
    run<4, 2, 2>("nt, gap 1024 + mfma 3072", X, n, dp, 8, 1, 0, cus, sink, 16, 48);
    run<3, 3, 2>("3w3s nt, gap 512 + mfma 2048", X, n, dp, 8, 1, 0, cus, sink, 8, 32);
    run<3, 3, 2>("3w3s nt, gap 1024 + mfma 3072", X, n, dp, 8, 1, 0, cus, sink, 16, 48);
    // half stages (two rows per DMA instruction), 4 slots per wave: same LDS, finer hand-over
    run_half<4, 4, 2>("half stages x4 nt", X, n, dp, 8, cus, sink, 0, 0);
    run_half<4, 4, 2>("half stages x4 nt, gap 256 + mfma 1024", X, n, dp, 8, cus, sink, 4, 16);
    run_half<4, 4, 2>("half stages x4 nt, gap 384 + mfma 1024", X, n, dp, 8, cus, sink, 6, 16);
    run_half<4, 2, 2>("half stages x2 nt, gap 256 + mfma 1024", X, n, dp, 8, cus, sink, 4, 16);
    run<4, 2, 2>("whole stages x2 nt, gap 512 + mfma 2048", X, n, dp, 8, 1, 0, cus, sink, 8, 32);
    run_half<4, 4, 2>("half stages x4 nt, gap 256 + mfma 1024", X, n, dp, 8, cus, sink, 4, 16);
    run<4, 2, 2>("whole stages x2 nt, gap 512 + mfma 2048", X, n, dp, 8, 1, 0, cus, sink, 8, 32);
    return 0;
}
