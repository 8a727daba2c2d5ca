// mfma_rate2.hip — v_mfma_f32_16x16x4_f32 issue rate by ACCUMULATOR PATTERN (no memory traffic):
// round-robin over NACC accumulators vs runs of RUN consecutive MFMAs on one accumulator before
// moving to the next, at 1 and 2 waves per SIMD.  (round 3: why k_scan_mq sat at 40 clocks per MFMA)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC, int RUN>
__global__ __launch_bounds__(512) void k16(float* out, int iters, float seed) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) { acc[i] = f32x4{0, 0, 0, 0}; asm volatile("" : "+a"(acc[i])); }
    float a = seed + threadIdx.x * 1e-3f, b = seed * 0.5f + threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
#pragma unroll
            for (int r = 0; r < RUN; ++r) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    f32x4 s = {0, 0, 0, 0};
    for (int i = 0; i < NACC; ++i) s += acc[i];
    if (s.x == 12345.678f) out[0] = s.y;
}
template <class F> float timeit(F&& launch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms; }
    return best;
}
template <int NACC, int RUN> void run(float* out, int wpc) {
    const int iters = 4000;
    const int threads = wpc == 4 ? 256 : 512;
    float ms = timeit([&] { hipLaunchKernelGGL((k16<NACC, RUN>), dim3(256), dim3(threads), 0, 0, out, iters, 0.37f); });
    double n = 256.0 * (threads / 64) * iters * NACC * RUN;
    printf("16x16x4 f32, %d acc x runs of %d, %d waves/CU: %.1f TFLOP/s  %.1f clocks per MFMA per SIMD at 2.4 GHz\n", NACC, RUN, wpc,
           n * 2048 / ms / 1e9, ms * 1e-3 * 2.4e9 / (n / 1024));
}
int main() {
    float* out; hipMalloc(&out, 4);
    for (int wpc : {4, 8}) {
        run<1, 1>(out, wpc); run<5, 1>(out, wpc); run<5, 2>(out, wpc); run<5, 4>(out, wpc); run<5, 8>(out, wpc);
        run<2, 4>(out, wpc); run<3, 4>(out, wpc); run<4, 4>(out, wpc); run<5, 32>(out, wpc);
    }
    return 0;
}
