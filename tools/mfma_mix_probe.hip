// mfma_mix_probe.hip - what slows v_mfma_f32_16x16x4_f32 below its 32.4 clocks in k_scan_mq's stage loop?
// Chains of 16 on two accumulators (32 MFMAs per iteration), one wave per SIMD, whole chip, real shader cycles;
// optionally, per iteration: DS = 10 ds_read_b128 (B fragments + A refresh at the stage loop's density),
// DMA = 2 global_load_lds_dwordx4 (1 KiB each: 8 X pieces + ~2.5 B pieces per 160 MFMAs), waited for a ring's depth behind.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <bool DS, bool DMA, bool SPREAD>
__global__ __launch_bounds__(256) void k(unsigned long long* out, const float* __restrict__ src, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* mine = lds + wave * 16384;
    f32x4 acc0{0, 0, 0, 0}, acc1{0, 0, 0, 0};
    f32x4 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = f32x4{1.f + lane, 2.f, 3.f, 4.f}; b[i] = f32x4{0.5f, 0.25f, lane * 1.f, 1.f}; }
    const char* g = reinterpret_cast<const char*>(src) + ((size_t)blockIdx.x * 4 + wave) * (1 << 20) + lane * 16;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        // 16 MFMAs on acc0, work items placed between the MFMAs (SPREAD) or behind the chain
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][j], b[i][j], acc0, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (SPREAD && j == 1) {
                    if (DS && i < 3) { b[(i + 1) & 3] = *reinterpret_cast<const f32x4*>(mine + ((it + i) & 7) * 1024 + lane * 16); __builtin_amdgcn_sched_barrier(0); }
                    if (DMA && i == 0) {
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + (size_t)((it * 2) & 1023) * 1024),
                                                         (__attribute__((address_space(3))) void*)(mine + 8192 + ((it * 2) & 7) * 1024), 16, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
        if (!SPREAD) {
            if (DS) { for (int i = 0; i < 3; ++i) b[(i + 1) & 3] = *reinterpret_cast<const f32x4*>(mine + ((it + i) & 7) * 1024 + lane * 16); __builtin_amdgcn_sched_barrier(0); }
            if (DMA) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + (size_t)((it * 2) & 1023) * 1024),
                                                 (__attribute__((address_space(3))) void*)(mine + 8192 + ((it * 2) & 7) * 1024), 16, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][j], b[i][j], acc1, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (SPREAD && j == 1) {
                    if (DS && i < 2) { a[(i + 1) & 3] = *reinterpret_cast<const f32x4*>(mine + ((it + i + 3) & 7) * 1024 + lane * 16); __builtin_amdgcn_sched_barrier(0); }
                    if (DMA && i == 0) {
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + (size_t)((it * 2 + 1) & 1023) * 1024),
                                                         (__attribute__((address_space(3))) void*)(mine + 8192 + ((it * 2 + 1) & 7) * 1024), 16, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
        if (!SPREAD) {
            if (DS) { for (int i = 0; i < 2; ++i) a[(i + 1) & 3] = *reinterpret_cast<const f32x4*>(mine + ((it + i + 3) & 7) * 1024 + lane * 16); __builtin_amdgcn_sched_barrier(0); }
            if (DMA) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + (size_t)((it * 2 + 1) & 1023) * 1024),
                                                 (__attribute__((address_space(3))) void*)(mine + 8192 + ((it * 2 + 1) & 7) * 1024), 16, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (DMA) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    f32x4 s = acc0 + acc1;
    if (s.x == 12345.678f) out[0] = (unsigned long long)s.y;
    if (lane == 0) out[8 + blockIdx.x * 4 + wave] = t1 - t0;
}
template <bool DS, bool DMA, bool SPREAD> void run(unsigned long long* out, const float* src, const char* name) {
    const int iters = 1500;
    unsigned long long h[1032];
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<DS, DMA, SPREAD>), dim3(256), dim3(256), 65536, 0, out, src, iters);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    }
    double s = 0, mx = 0; for (int i = 0; i < 1024; ++i) { s += (double)h[8 + i]; if ((double)h[8 + i] > mx) mx = (double)h[8 + i]; }
    printf("%-62s %.2f cycles per MFMA (mean), %.2f (slowest wave)\n", name, s / 1024 / (iters * 32.0), mx / (iters * 32.0));
}
int main() {
    unsigned long long* out; (void)hipMalloc(&out, 1032 * 8);
    float* src; (void)hipMalloc(&src, (size_t)1 << 30); (void)hipMemset(src, 0, (size_t)1 << 30);
    (void)hipFuncSetAttribute((const void*)k<false, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    run<false, false, false>(out, src, "MFMA chains only");
    run<true, false, false>(out, src, "+ 5 ds_read_b128 per 32 MFMAs, behind the chains");
    run<true, false, true>(out, src, "+ 5 ds_read_b128 per 32 MFMAs, between the MFMAs");
    run<false, true, false>(out, src, "+ 2 LDS-DMA pieces per 32 MFMAs, behind the chains");
    run<false, true, true>(out, src, "+ 2 LDS-DMA pieces per 32 MFMAs, between the MFMAs");
    run<true, true, false>(out, src, "+ both, behind the chains");
    run<true, true, true>(out, src, "+ both, between the MFMAs");
    return 0;
}
