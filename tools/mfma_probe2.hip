// mfma_probe2.hip — find the first MFMA step whose result differs from a sequential fmaf chain
// on a long, positive, un-normalised accumulation, and print its operands.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
// A[16][K] row-major, B[16][K] (column j's vector), trace[step][16][16]
__global__ void k(const float* A, const float* B, int K, float* trace) {
    int l = threadIdx.x, r = l & 15, g = l >> 4;
    f32x4 c = {0, 0, 0, 0};
    for (int s = 0; s < K / 4; ++s) {
        float a = A[r * K + 4 * s + g];
        float b = B[r * K + 4 * s + g];
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
        for (int e = 0; e < 4; ++e) trace[(size_t)s * 256 + (g * 4 + e) * 16 + r] = c[e];
    }
}
static float bits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
int main(int argc, char** argv) {
    int K = 4096;
    float scaleA = argc > 1 ? atof(argv[1]) : 1.0f;
    std::vector<float> hA(16 * K), hB(16 * K), tr((size_t)K / 4 * 256);
    srand(3);
    for (auto& v : hA) v = (float)rand() / RAND_MAX * scaleA;
    for (auto& v : hB) v = (float)rand() / RAND_MAX;
    float *dA, *dB, *dT;
    hipMalloc(&dA, hA.size() * 4); hipMalloc(&dB, hB.size() * 4); hipMalloc(&dT, tr.size() * 4);
    hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, K, dT);
    hipMemcpy(tr.data(), dT, tr.size() * 4, hipMemcpyDeviceToHost);
    int nbad = 0;
    for (int i = 0; i < 16 && nbad < 5; ++i) for (int j = 0; j < 16 && nbad < 5; ++j) {
        float c = 0;
        for (int s = 0; s < K / 4; ++s) {
            float cin = c;
            for (int kk = 0; kk < 4; ++kk) c = fmaf(hA[i * K + 4 * s + kk], hB[j * K + 4 * s + kk], c);
            float d = tr[(size_t)s * 256 + i * 16 + j];
            if (memcmp(&c, &d, 4)) {
                printf("first mismatch i=%d j=%d step=%d cin=%a seq=%a hw=%a\n", i, j, s, cin, c, d);
                for (int kk = 0; kk < 4; ++kk) printf("   a=%a b=%a\n", hA[i * K + 4 * s + kk], hB[j * K + 4 * s + kk]);
                // hypotheses from cin
                double dd = cin; for (int kk = 0; kk < 4; ++kk) dd += (double)hA[i*K+4*s+kk] * hB[j*K+4*s+kk];
                float h1 = (float)dd;
                float r = cin; for (int kk = 3; kk >= 0; --kk) r = fmaf(hA[i*K+4*s+kk], hB[j*K+4*s+kk], r);
                float p01 = fmaf(hA[i*K+4*s+1], hB[j*K+4*s+1], hA[i*K+4*s]*hB[j*K+4*s]);
                printf("   exact_round=%a reverse=%a\n", h1, r); (void)p01;
                ++nbad; c = d; // resync
                break;
            }
        }
    }
    printf("K=%d scaleA=%g mismatching outputs (first 5 shown): %d\n", K, scaleA, nbad);
    return 0;
}
