// mfma_rate.hip — issue rate of the fp32-input MFMAs on this part (no memory traffic).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float seed) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a = seed + threadIdx.x * 1e-3f, b = seed * 0.5f + threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    f32x4 s = {0, 0, 0, 0};
    for (int i = 0; i < NACC; ++i) s += acc[i];
    if (s.x == 12345.678f) out[0] = s.y;
}
template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float seed) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    float a = seed + threadIdx.x * 1e-3f, b = seed * 0.5f + threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    if (s == 12345.678f) out[0] = s;
}
template <class F> float timeit(F&& launch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms; }
    return best;
}
int main() {
    float* out; hipMalloc(&out, 4);
    const int iters = 20000;
    for (int wpc : {4, 8, 16}) {
        int blocks = 256 * wpc / 4;
        float ms = timeit([&] { hipLaunchKernelGGL((k16<4>), dim3(blocks), dim3(256), 0, 0, out, iters, 0.37f); });
        double n = (double)blocks * 4 * iters * 4;
        printf("16x16x4 f32, 4 acc, %2d waves/CU: %.3f ms  %.1f TFLOP/s  %.1f ns per MFMA per SIMD\n", wpc, ms, n * 2048 / ms / 1e9, ms * 1e6 / (n / 1024));
        ms = timeit([&] { hipLaunchKernelGGL((k16<1>), dim3(blocks), dim3(256), 0, 0, out, iters, 0.37f); });
        n = (double)blocks * 4 * iters;
        printf("16x16x4 f32, 1 acc, %2d waves/CU: %.3f ms  %.1f TFLOP/s  %.1f ns per MFMA per SIMD\n", wpc, ms, n * 2048 / ms / 1e9, ms * 1e6 / (n / 1024));
        ms = timeit([&] { hipLaunchKernelGGL((k32<2>), dim3(blocks), dim3(256), 0, 0, out, iters, 0.37f); });
        n = (double)blocks * 4 * iters * 2;
        printf("32x32x2 f32, 2 acc, %2d waves/CU: %.3f ms  %.1f TFLOP/s  %.1f ns per MFMA per SIMD\n", wpc, ms, n * 4096 / ms / 1e9, ms * 1e6 / (n / 1024));
    }
    return 0;
}
