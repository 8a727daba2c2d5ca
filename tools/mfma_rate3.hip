// mfma_rate3.hip — the k-block of k_scan_mq in isolation: T chains of four v_mfma_f32_16x16x4_f32, B
// fragments read from LDS one k-block ahead (ds_read_b128), optionally one LDS-DMA issue per k-block.
// Which ingredient takes the matrix pipe from 34 to 40 clocks per MFMA?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int T, int LDSB, int DMA, int AFRESH>
__global__ __launch_bounds__(256) void kb(float* out, const float* src, int iters, float seed) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    f32x4 acc[T];
    for (int i = 0; i < T; ++i) { acc[i] = f32x4{0, 0, 0, 0}; asm volatile("" : "+a"(acc[i])); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, g = lane >> 4;
    const char* bs = lds + (r / 2) * 1056 + (r % 2) * 512 + ((g ^ (r % 2)) * 16);
    f32x4 a[8];
    for (int i = 0; i < 8; ++i) a[i] = f32x4{seed + lane, seed * 2, seed * 3, seed * 4 + i};
    f32x4 bcur[T], bnx[T];
    for (int t = 0; t < T; ++t) bcur[t] = f32x4{seed, seed + t, 1.f, 2.f};
    char* ring = lds + 5 * 8448 * 2 + wave * 2 * 8448;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < T; ++t) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][j], bcur[t][j], acc[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (LDSB && t == 0) {
#pragma unroll
                    for (int t2 = 0; t2 < T; ++t2) bnx[t2] = *reinterpret_cast<const f32x4*>(bs + t2 * 8448 + ((i + 1) & 7) * 64);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (DMA && t < DMA && i < 8 / DMA) {
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + ((size_t)blockIdx.x * 64 + lane) * 4 + (it & 7) * 4096),
                                                     (__attribute__((address_space(3))) void*)(ring + (it & 1) * 8448 + (DMA * i + t) * 1056), 16, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (AFRESH && i >= 6) { a[i] = *reinterpret_cast<const f32x4*>(ring + r * 64 + g * 16 + i * 1056); __builtin_amdgcn_sched_barrier(0); }
            if (LDSB) {
#pragma unroll
                for (int t = 0; t < T; ++t) bcur[t] = bnx[t];
            }
        }
    }
    f32x4 s = {0, 0, 0, 0};
    for (int i = 0; i < T; ++i) s += acc[i];
    if (s.x == 12345.678f) out[0] = s.y;
}
template <class F> float timeit(F&& launch) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) { (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms; }
    return best;
}
template <int T, int LDSB, int DMA, int AFRESH> void run(float* out, const float* src) {
    const int iters = 2000;
    const size_t lds = 5 * 8448 * 2 + 4 * 2 * 8448;
    (void)hipFuncSetAttribute((const void*)kb<T, LDSB, DMA, AFRESH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    float ms = timeit([&] { hipLaunchKernelGGL((kb<T, LDSB, DMA, AFRESH>), dim3(256), dim3(256), lds, 0, out, src, iters, 0.37f); });
    double n = 256.0 * 4 * iters * 8 * T * 4;
    printf("T=%d  B from LDS %d  DMA pieces per k-block %d  A refresh %d: %.1f TFLOP/s  %.1f clocks per MFMA per SIMD at 2.4 GHz\n", T, LDSB, DMA, AFRESH,
           n * 2048 / ms / 1e9, ms * 1e-3 * 2.4e9 / (n / 1024));
}
int main() {
    float *out, *src; (void)hipMalloc(&out, 4); (void)hipMalloc(&src, 64 << 20);
    run<5, 0, 0, 0>(out, src); run<5, 1, 0, 0>(out, src); run<5, 1, 1, 0>(out, src); run<5, 1, 2, 0>(out, src); run<5, 1, 2, 1>(out, src);
    run<5, 0, 2, 0>(out, src); run<2, 1, 1, 1>(out, src); run<3, 1, 1, 1>(out, src); run<4, 1, 2, 1>(out, src);
    return 0;
}
