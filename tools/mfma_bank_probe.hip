// mfma_bank_probe.hip - does v_mfma_f32_16x16x4_f32 issue slower when srcA and srcB sit in the SAME VGPR bank
// (register index mod 4)?  k_scan_mq's operands are the j-th components of two float4 tuples: always the same bank.
// Real shader cycles by s_memtime, one wave per SIMD, chains of 16 on 5 accumulators (as the stage loop runs them).
#include <hip/hip_runtime.h>
#include <cstdio>
#define MF(ACC, A, B) "v_mfma_f32_16x16x4_f32 a[" ACC "], " A ", " B ", a[" ACC "]\n"
#define CH16(ACC, A, B) MF(ACC, A, B) MF(ACC, A, B) MF(ACC, A, B) MF(ACC, A, B) MF(ACC, A, B) MF(ACC, A, B) MF(ACC, A, B) MF(ACC, A, B) \
                        MF(ACC, A, B) MF(ACC, A, B) MF(ACC, A, B) MF(ACC, A, B) MF(ACC, A, B) MF(ACC, A, B) MF(ACC, A, B) MF(ACC, A, B)
// operands rotate through 4 components like a[i][j], b[ii][j], j = 0..3
#define K4(ACC, A0, A1, A2, A3, B0, B1, B2, B3) MF(ACC, A0, B0) MF(ACC, A1, B1) MF(ACC, A2, B2) MF(ACC, A3, B3)
template <int VARIANT>
__global__ __launch_bounds__(256) void k(unsigned long long* out, int iters) {
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (VARIANT == 0)   // same bank: A = v8..v11, B = v12..v15 (component j with component j)
            asm volatile(K4("0:3", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15") K4("0:3", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15")
                         K4("0:3", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15") K4("0:3", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15")
                         K4("4:7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15") K4("4:7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15")
                         K4("4:7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15") K4("4:7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15")
                         ::: "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");
        else if (VARIANT == 1)   // different banks: B shifted by one register (v13..v16)
            asm volatile(K4("0:3", "v8", "v9", "v10", "v11", "v13", "v14", "v15", "v16") K4("0:3", "v8", "v9", "v10", "v11", "v13", "v14", "v15", "v16")
                         K4("0:3", "v8", "v9", "v10", "v11", "v13", "v14", "v15", "v16") K4("0:3", "v8", "v9", "v10", "v11", "v13", "v14", "v15", "v16")
                         K4("4:7", "v8", "v9", "v10", "v11", "v13", "v14", "v15", "v16") K4("4:7", "v8", "v9", "v10", "v11", "v13", "v14", "v15", "v16")
                         K4("4:7", "v8", "v9", "v10", "v11", "v13", "v14", "v15", "v16") K4("4:7", "v8", "v9", "v10", "v11", "v13", "v14", "v15", "v16")
                         ::: "v8", "v9", "v10", "v11", "v13", "v14", "v15", "v16", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");
        else   // one operand pair only (no rotation), same bank
            asm volatile(CH16("0:3", "v8", "v12") CH16("4:7", "v8", "v12") ::: "v8", "v12", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
int main() {
    unsigned long long* out; hipMalloc(&out, 256 * 4 * 8);
    unsigned long long h[1024];
    const int iters = 2000;
    const char* names[3] = {"same bank, components rotate (a[i][j], b[ii][j])", "B shifted by one register (different banks)", "one operand pair, same bank"};
    for (int rep = 0; rep < 2; ++rep)
        for (int v = 0; v < 3; ++v) {
            if (v == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, out, iters);
            if (v == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, out, iters);
            if (v == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, out, iters);
            hipDeviceSynchronize();
            hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
            double s = 0; for (int i = 0; i < 1024; ++i) s += (double)h[i];
            printf("%-55s %.2f shader cycles per MFMA (mean over 1024 waves, %d MFMAs each)\n", names[v], s / 1024 / (iters * 32.0), iters * 32);
        }
    return 0;
}
