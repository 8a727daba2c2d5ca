// k_scan_mq with in-kernel clock stamps (tools/build_mq_probes.sh, PROBES=CLOCK): shader cycles of the
// stage loop, of the waits for the B pieces and at the barrier, and inside the k-block loops; printed
// by two blocks.  profiles/r03_mq_probes.txt, r04_mq_clock_probe_1.txt.
#pragma once
#define CSLAM_MQ_HOOK_KERNEL_BEGIN                                                                  \
    const uint64_t pc_t0 = __builtin_amdgcn_s_memtime(), pc_r0 = __builtin_amdgcn_s_memrealtime();  \
    uint64_t pc_wait_b = 0, pc_wait_bar = 0, pc_body = 0, pc_b0 = 0, pc_b1 = 0, pc_s0 = 0;
#define CSLAM_MQ_HOOK_STEP_BEGIN pc_b0 = __builtin_amdgcn_s_memtime();
#define CSLAM_MQ_HOOK_B_LANDED pc_b1 = __builtin_amdgcn_s_memtime();
#define CSLAM_MQ_HOOK_BARRIER_PASSED \
    pc_wait_b += pc_b1 - pc_b0;      \
    pc_wait_bar += __builtin_amdgcn_s_memtime() - pc_b1;
#define CSLAM_MQ_HOOK_STAGE_BEGIN pc_s0 = __builtin_amdgcn_s_memtime();
#define CSLAM_MQ_HOOK_STAGE_END pc_body += __builtin_amdgcn_s_memtime() - pc_s0;
#define CSLAM_MQ_HOOK_KERNEL_END                                                                                           \
    {                                                                                                                      \
        const uint64_t dt = __builtin_amdgcn_s_memtime() - pc_t0, dr = __builtin_amdgcn_s_memrealtime() - pc_r0;           \
        if ((blockIdx.x == 3 || blockIdx.x == 100) && (threadIdx.x & 63) == 0 && wave == 0)                                \
            printf("mq clock probe: block %d mode %d stages %d T=%d NW=%d: %llu shader cycles in %llu x 10 ns = %.3f GHz, " \
                   "%.1f cycles per MFMA; of them waiting for B %llu, at the barrier %llu (%d steps), inside the k-block "  \
                   "loops of the stages %llu\n",                                                                           \
                   (int)blockIdx.x, MODE, J, T, NW, (unsigned long long)dt, (unsigned long long)dr,                        \
                   (double)dt / (10.0 * dr), (double)dt / ((double)J * NB * 4 * T), (unsigned long long)pc_wait_b,         \
                   (unsigned long long)pc_wait_bar, gstep, (unsigned long long)pc_body);                                   \
    }
