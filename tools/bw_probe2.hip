// bw_probe2.hip — does the shape of a wave's load instruction matter?  Same bytes, three shapes:
//   rows16x64 : 16 rows x 64 B per instruction  (MFMA A-operand shape, what k_scan v3 issues)
//   rows8x128 : 8 rows x 128 B per instruction  (full 128-B lines)
//   glds1k    : global_load_lds, 1 KiB contiguous of ONE row per instruction, 16 rows x 1 KiB
//               per stage into a per-wave LDS ring, then ds_read_b128
// each wave owns a contiguous row range; rows are 16 KB apart (D = 4096 fp32)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr size_t D = 4096;

template <int S, int U>
__global__ __launch_bounds__(256) void k_rows16(const float* X, size_t n_rows, size_t n_waves, float* out) {
    size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= n_waves) return;
    int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    size_t lo = w * n_rows / n_waves, hi = (w + 1) * n_rows / n_waves;
    f32x4 acc = {0, 0, 0, 0};
    for (size_t r0 = lo; r0 < hi; r0 += 16 * S) {
        const f32x4* p[S];
        for (int s = 0; s < S; ++s) { size_t row = r0 + 16 * s + r; if (row >= hi) row = hi - 1; p[s] = (const f32x4*)(X + row * D + 4 * g); }
        for (int kb = 0; kb < (int)D / 16; kb += U) {
            f32x4 v[U][S];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int s = 0; s < S; ++s) v[u][s] = p[s][(kb + u) * 4];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int s = 0; s < S; ++s) acc += v[u][s];
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}
template <int S, int U>
__global__ __launch_bounds__(256) void k_rows8(const float* X, size_t n_rows, size_t n_waves, float* out) {
    size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= n_waves) return;
    int lane = threadIdx.x & 63, r = lane >> 3, c = lane & 7;   // 8 rows x 8 lanes(128 B)
    size_t lo = w * n_rows / n_waves, hi = (w + 1) * n_rows / n_waves;
    f32x4 acc = {0, 0, 0, 0};
    for (size_t r0 = lo; r0 < hi; r0 += 8 * S) {
        const f32x4* p[S];
        for (int s = 0; s < S; ++s) { size_t row = r0 + 8 * s + r; if (row >= hi) row = hi - 1; p[s] = (const f32x4*)(X + row * D) + c; }
        for (int kb = 0; kb < (int)D / 32; kb += U) {      // 32 floats = 128 B per step
            f32x4 v[U][S];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int s = 0; s < S; ++s) v[u][s] = p[s][(kb + u) * 8];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int s = 0; s < S; ++s) acc += v[u][s];
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}
// LDS-DMA: stage = 16 rows x 1 KiB (256 floats); NST stages ring per wave
template <int NST>
__global__ __launch_bounds__(256) void k_glds(const float* X, size_t n_rows, size_t n_waves, float* out) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= n_waves) return;
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    char* my = lds + (size_t)wave * NST * 16384;
    size_t lo = w * n_rows / n_waves, hi = (w + 1) * n_rows / n_waves;
    const int nchunk = D / 256;                 // 16 chunks of 1 KiB per row
    size_t nstrips = (hi - lo + 15) / 16;
    size_t total = nstrips * nchunk;            // stages to process
    auto issue = [&](size_t st) {
        size_t strip = st / nchunk; int ch = (int)(st % nchunk);
        char* dst = my + (st % NST) * 16384;
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            size_t row = lo + strip * 16 + rr; if (row >= hi) row = hi - 1;
            const float* src = X + row * D + ch * 256 + lane * 4;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(dst + rr * 1024), 16, 0, 0);
        }
    };
    f32x4 acc = {0, 0, 0, 0};
    for (int s = 0; s < NST - 1 && (size_t)s < total; ++s) issue(s);
    for (size_t st = 0; st < total; ++st) {
        if (st + NST - 1 < total) issue(st + NST - 1);
        // wait until stage st has landed: at most (NST-1)*16 younger loads may remain
        if (st + NST - 1 < total) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 1) * 16) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const f32x4* src = (const f32x4*)(my + (st % NST) * 16384);
#pragma unroll
        for (int i = 0; i < 16; ++i) acc += src[i * 64 + lane];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}
__global__ void k_fill_random(float* X, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * 256) { unsigned long long h = i * 0x9E3779B97F4A7C15ull; h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32; X[i] = (float)(h & 0xFFFFFF) / 16777216.0f; }
}
template <class F> void timeit(const char* name, F&& launch, size_t bytes) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms;
    }
    hipError_t e = hipGetLastError();
    printf("%-34s %.3f ms  %.0f GB/s %s\n", name, best, bytes / best / 1e6, e == hipSuccess ? "" : hipGetErrorString(e));
}
int main() {
    const size_t N = 100000;
    float *X, *out; hipMalloc(&X, N * D * 4); hipMalloc(&out, 4); hipMemset(X, 0, N * D * 4);
    if (getenv("PROBE_RANDOM")) { hipLaunchKernelGGL(k_fill_random, dim3(4096), dim3(256), 0, 0, X, N * D); hipDeviceSynchronize(); printf("X filled with random floats\n"); }
    size_t bytes = N * D * 4;
#define RUN16(S, U, WPC) timeit("rows16x64 S=" #S " U=" #U " wpc=" #WPC, [&] { size_t nw = 256 * WPC; hipLaunchKernelGGL((k_rows16<S, U>), dim3(nw / 4), dim3(256), 0, 0, X, N, nw, out); }, bytes)
#define RUN8(S, U, WPC) timeit("rows8x128 S=" #S " U=" #U " wpc=" #WPC, [&] { size_t nw = 256 * WPC; hipLaunchKernelGGL((k_rows8<S, U>), dim3(nw / 4), dim3(256), 0, 0, X, N, nw, out); }, bytes)
#define RUNG(NST, WPC) timeit("glds1k NST=" #NST " wpc=" #WPC, [&] { size_t nw = 256 * WPC; hipLaunchKernelGGL((k_glds<NST>), dim3(nw / 4), dim3(256), 4 * NST * 16384, 0, X, N, nw, out); }, bytes)
    RUN16(1, 8, 8); RUN16(1, 8, 16); RUN16(4, 4, 4); RUN16(8, 4, 4); RUN16(4, 4, 8); RUN16(2, 8, 8);
    RUN8(1, 8, 8); RUN8(2, 4, 8); RUN8(4, 4, 4); RUN8(8, 4, 4); RUN8(8, 2, 8); RUN8(16, 2, 4);
    hipFuncSetAttribute((const void*)k_glds<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)k_glds<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    RUNG(2, 4); RUNG(2, 8); RUNG(3, 4);
    return 0;
}
