// k_scan_dma with time stamps (tools/build_mq_probes.sh, DMA_CLOCK=1): when a wave's queries and its
// first stage have landed and when it is done.  profiles/r03_dma_probe_clock.txt.
#pragma once
#define CSLAM_DMA_HOOK_KERNEL_BEGIN                           \
    const uint64_t pc_r0 = __builtin_amdgcn_s_memrealtime(); \
    uint64_t pc_r1 = 0, pc_r2 = 0;
#define CSLAM_DMA_HOOK_QUERIES_LANDED pc_r1 = __builtin_amdgcn_s_memrealtime();
#define CSLAM_DMA_HOOK_STAGE_LANDED \
    if (pc_r2 == 0) pc_r2 = __builtin_amdgcn_s_memrealtime();
#define CSLAM_DMA_HOOK_KERNEL_END                                                                                        \
    if (lane == 0 && (w % 97) == 0) {                                                                                    \
        const uint64_t r3 = __builtin_amdgcn_s_memrealtime();                                                            \
        printf("dma probe: wave %4d rows %lld: entry %llu (x10 ns, absolute), queries landed +%llu, first stage landed " \
               "+%llu, done +%llu\n",                                                                                    \
               w, (long long)(whi - wlo), (unsigned long long)(pc_r0 % 100000000ull),                                    \
               (unsigned long long)(pc_r1 - pc_r0), (unsigned long long)(pc_r2 - pc_r0),                                 \
               (unsigned long long)(r3 - pc_r0));                                                                        \
    }
