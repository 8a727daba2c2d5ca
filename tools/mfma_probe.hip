// mfma_probe.hip — which fp32 summation order does v_mfma_f32_16x16x4_f32 implement?
// Prints the hypothesis (if any) that reproduces the hardware result bit for bit.
// Build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/mfma_probe.hip -o gpurun_out/mfma_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* A, const float* B, const float* C, float* D) {
    int l = threadIdx.x;
    float a = A[(l & 15) * 4 + (l >> 4)];
    float b = B[(l >> 4) * 16 + (l & 15)];
    f32x4 c;
    for (int e = 0; e < 4; ++e) c[e] = C[((l >> 4) * 4 + e) * 16 + (l & 15)];
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    for (int e = 0; e < 4; ++e) D[((l >> 4) * 4 + e) * 16 + (l & 15)] = c[e];
}
int main() {
    float hA[64], hB[64], hC[256], hD[256];
    srand(1);
    int ok_seq = 1, ok_rev = 1, ok_pair = 1, ok_unfused = 1, ok_dbl = 1;
    for (int trial = 0; trial < 200; ++trial) {
        for (int i = 0; i < 64; ++i) { hA[i] = (float)rand() / RAND_MAX * 2 - 1; hB[i] = (float)rand() / RAND_MAX * 2 - 1; }
        for (int i = 0; i < 256; ++i) hC[i] = (float)rand() / RAND_MAX * 2 - 1;
        float *dA, *dB, *dC, *dD;
        hipMalloc(&dA, 256); hipMalloc(&dB, 256); hipMalloc(&dC, 1024); hipMalloc(&dD, 1024);
        hipMemcpy(dA, hA, 256, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 256, hipMemcpyHostToDevice);
        hipMemcpy(dC, hC, 1024, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
        hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dD);
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            float c = hC[i * 16 + j], d = hD[i * 16 + j];
            float s = c; for (int kk = 0; kk < 4; ++kk) s = fmaf(hA[i * 4 + kk], hB[kk * 16 + j], s);
            float r = c; for (int kk = 3; kk >= 0; --kk) r = fmaf(hA[i * 4 + kk], hB[kk * 16 + j], r);
            float p = fmaf(hA[i*4+1], hB[16+j], hA[i*4]*hB[j]) + fmaf(hA[i*4+3], hB[48+j], hA[i*4+2]*hB[32+j]) + c;
            float u = c; for (int kk = 0; kk < 4; ++kk) u = u + hA[i * 4 + kk] * hB[kk * 16 + j];
            double dd = c; for (int kk = 0; kk < 4; ++kk) dd += (double)hA[i * 4 + kk] * hB[kk * 16 + j];
            if (memcmp(&s, &d, 4)) ok_seq = 0;
            if (memcmp(&r, &d, 4)) ok_rev = 0;
            if (memcmp(&p, &d, 4)) ok_pair = 0;
            if (memcmp(&u, &d, 4)) ok_unfused = 0;
            float df = (float)dd; if (memcmp(&df, &d, 4)) ok_dbl = 0;
        }
    }
    printf("mfma_f32_16x16x4f32 order: sequential_fma_k0123=%d reverse_fma=%d pairwise=%d unfused=%d exact_then_round=%d\n",
           ok_seq, ok_rev, ok_pair, ok_unfused, ok_dbl);
    return ok_seq ? 0 : 1;
}
