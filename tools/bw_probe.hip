// bw_probe.hip — read-bandwidth ceilings for the access patterns of the kNN scan.
//   linear : wave reads 1 KiB contiguous per instruction, each wave owns a contiguous chunk
//   rows16 : wave reads 16 rows x 64 B per instruction (the MFMA 16x16x4 A-operand pattern),
//            rows `stride` floats apart; each wave owns a contiguous range of rows
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int U>
__global__ __launch_bounds__(256) void k_linear(const float* X, size_t floats_per_wave, float* out) {
    size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    const f32x4* p = (const f32x4*)(X + w * floats_per_wave) + lane;
    size_t n = floats_per_wave / 256;  // 1 KiB steps
    f32x4 acc = {0, 0, 0, 0};
    for (size_t i = 0; i < n; i += U) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = p[(i + u) * 64];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}
template <int U>
__global__ __launch_bounds__(256) void k_rows16(const float* X, size_t stride, size_t n_rows, int d,
                                                size_t n_waves, float* out) {
    size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= n_waves) return;
    int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    size_t lo = w * n_rows / n_waves, hi = (w + 1) * n_rows / n_waves;
    f32x4 acc = {0, 0, 0, 0};
    for (size_t r0 = lo; r0 < hi; r0 += 16) {
        size_t row = r0 + r < hi ? r0 + r : hi - 1;
        const f32x4* p = (const f32x4*)(X + row * stride + 4 * g);
        for (int kb = 0; kb < d / 16; kb += U) {
            f32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = p[(kb + u) * 4];
#pragma unroll
            for (int u = 0; u < U; ++u) acc += v[u];
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}
int main() {
    const size_t N = 100000, D = 4096;
    const size_t strides[] = {4096, 4096 + 64, 4096 + 128, 4096 + 1024 + 64};
    float *X, *out;
    hipMalloc(&X, N * (D + 2048) * 4);
    hipMalloc(&out, 4);
    hipMemset(X, 0, N * (D + 2048) * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t waves = 2048;
    for (int wpc = 4; wpc <= 16; wpc *= 2) {
        size_t nw = 256 * wpc;
        size_t fpw = N * D / nw / 2048 * 2048;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_linear<8>, dim3(nw / 4), dim3(256), 0, 0, X, fpw, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("linear  waves/CU=%2d U=8: %.3f ms  %.0f GB/s\n", wpc, ms, fpw * nw * 4 / ms / 1e6);
        }
    }
    (void)waves;
    for (size_t st : strides)
        for (int wpc = 8; wpc <= 16; wpc *= 2) {
            size_t nw = 256 * wpc;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(k_rows16<8>, dim3(nw / 4), dim3(256), 0, 0, X, st, N, (int)D, nw, out);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep) printf("rows16  stride=%zu waves/CU=%2d U=8: %.3f ms  %.0f GB/s\n", st, wpc, ms, N * D * 4 / ms / 1e6);
            }
        }
    for (size_t st : {(size_t)4096, (size_t)4160}) {
        size_t nw = 2048;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_rows16<16>, dim3(nw / 4), dim3(256), 0, 0, X, st, N, (int)D, nw, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("rows16  stride=%zu waves/CU= 8 U=16: %.3f ms  %.0f GB/s\n", st, ms, N * D * 4 / ms / 1e6);
        }
    }
    return 0;
}
