// bw_probe3.hip — does consuming the stream with MFMAs (instead of a VALU add) cost bandwidth?
// Same loads as bw_probe2's rows16x64 S=1 U=8; three consumers: add, 4 dependent MFMA 16x16x4
// per loaded float4 (the k_scan chain), 16 MFMAs on 4 independent accumulators.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr size_t D = 4096;
template <int MODE, int U>
__global__ __launch_bounds__(256) void k(const float* X, size_t n_rows, size_t n_waves, float* out) {
    size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= n_waves) return;
    int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    size_t lo = w * n_rows / n_waves, hi = (w + 1) * n_rows / n_waves;
    f32x4 acc = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0}, acc3 = {0, 0, 0, 0}, acc4 = {0, 0, 0, 0};
    for (size_t r0 = lo; r0 < hi; r0 += 16) {
        size_t row = r0 + r; if (row >= hi) row = hi - 1;
        const f32x4* p = (const f32x4*)(X + row * D + 4 * g);
        for (int kb = 0; kb < (int)D / 16; kb += U) {
            f32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = p[(kb + u) * 4];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (MODE == 0) acc += v[u];
                if (MODE == 1) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(v[u][j], v[u][j], acc, 0, 0, 0);
                }
                if (MODE == 2) {
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(v[u][0], v[u][0], acc, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(v[u][1], v[u][1], acc2, 0, 0, 0);
                    acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(v[u][2], v[u][2], acc3, 0, 0, 0);
                    acc4 = __builtin_amdgcn_mfma_f32_16x16x4f32(v[u][3], v[u][3], acc4, 0, 0, 0);
                }
                if (MODE == 4) {  // accumulator in AccVGPRs
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %1, %0" : "+a"(acc) : "v"(v[u][j]));
                }
                if (MODE == 5) {  // 4x4x1 (16 blocks), 8 cycles each, same flops per loaded value? (2 passes)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(v[u][j], v[u][j], acc, 0, 0, 0);
                }
                if (MODE == 3) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] = __builtin_fmaf(v[u][j], v[u][j], acc[j]);
                }
            }
        }
    }
    acc += acc2 + acc3 + acc4;
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}
__global__ void k_fill_random(float* X, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * 256) { unsigned long long h = i * 0x9E3779B97F4A7C15ull; h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32; X[i] = (float)(h & 0xFFFFFF) / 16777216.0f; }
}
template <class F> void timeit(const char* name, F&& launch, size_t bytes) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms; }
    printf("%-40s %.3f ms  %.0f GB/s\n", name, best, bytes / best / 1e6);
}
int main() {
    const size_t N = 100000; float *X, *out; hipMalloc(&X, N * D * 4); hipMalloc(&out, 4);
    hipLaunchKernelGGL(k_fill_random, dim3(4096), dim3(256), 0, 0, X, N * D); hipDeviceSynchronize();
    size_t bytes = N * D * 4;
#define RUN(MODE, U, WPC, label) timeit(label " U=" #U " wpc=" #WPC, [&] { size_t nw = 256 * WPC; hipLaunchKernelGGL((k<MODE, U>), dim3(nw / 4), dim3(256), 0, 0, X, N, nw, out); }, bytes)
    RUN(4, 8, 16, "mfma acc in AGPR"); RUN(4, 8, 8, "mfma acc in AGPR"); RUN(4, 8, 4, "mfma acc in AGPR");
    RUN(5, 8, 16, "mfma 4x4x1 (1/4 the MFMA time)"); RUN(5, 8, 8, "mfma 4x4x1 (1/4 the MFMA time)"); RUN(5, 8, 4, "mfma 4x4x1 (1/4 the MFMA time)");
    RUN(0, 8, 16, "add"); RUN(1, 8, 16, "mfma dependent chain"); RUN(2, 8, 16, "mfma 4 independent"); RUN(3, 8, 16, "valu fma");
    RUN(0, 8, 8, "add"); RUN(1, 8, 8, "mfma dependent chain"); RUN(2, 8, 8, "mfma 4 independent"); RUN(3, 8, 8, "valu fma");
    RUN(0, 8, 4, "add"); RUN(1, 8, 4, "mfma dependent chain"); RUN(2, 8, 4, "mfma 4 independent");
    return 0;
}
