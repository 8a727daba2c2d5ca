// k_scan_mq with per-stage clock stamps (PROBES=STAGES): shader cycles of each of a wave's first 60 stage bodies
// (the k-block loops) and of the gap in front of it, printed by wave 0 of two blocks.
#pragma once
#define CSLAM_MQ_HOOK_KERNEL_BEGIN                                     \
    const uint64_t pc_t0 = __builtin_amdgcn_s_memtime();               \
    uint64_t pc_s0 = pc_t0, pc_e0 = pc_t0;                             \
    __shared__ uint32_t pc_body[60], pc_gap[60];                       \
    int pc_n = 0;
#define CSLAM_MQ_HOOK_STAGE_BEGIN pc_s0 = __builtin_amdgcn_s_memtime();
#define CSLAM_MQ_HOOK_STAGE_END                                                  \
    {                                                                            \
        const uint64_t pc_now = __builtin_amdgcn_s_memtime();                    \
        if (pc_n < 60 && wave == 0 && lane == 0) {                               \
            pc_body[pc_n] = (uint32_t)(pc_now - pc_s0);                          \
            pc_gap[pc_n] = (uint32_t)(pc_s0 - pc_e0);                            \
        }                                                                        \
        ++pc_n;                                                                  \
        pc_e0 = pc_now;                                                          \
    }
#define CSLAM_MQ_HOOK_KERNEL_END                                                                                    \
    if ((blockIdx.x == 3 || blockIdx.x == 100) && wave == 0 && lane == 0) {                                         \
        printf("mq stage probe: block %d mode %d T=%d NW=%d stages %d, total %llu cycles; body/gap per stage:",      \
               (int)blockIdx.x, MODE, T, NW, J, (unsigned long long)(__builtin_amdgcn_s_memtime() - pc_t0));        \
        for (int q = 0; q < (pc_n < 60 ? pc_n : 60); ++q) printf(" %u/%u", pc_body[q], pc_gap[q]);                  \
        printf("\n");                                                                                               \
    }
