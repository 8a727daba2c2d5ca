// Hook points of the scan kernels for TIMING PROBES and the ADDRESS AUDIT.  The product build defines every hook empty (and
// the two functional macros as the real operation): no probe code lives in the shipped translation
// units.  A probe build (tools/build_mq_probes.sh) compiles the same sources with
// -DCSLAM_PROBE_HOOKS_FILE='"../../tools/probes/<probe>.hpp"', which defines the hooks it needs before
// the defaults below are filled in (in-kernel clock stamps, the stage loop with one ingredient
// compiled out - results of such builds are wrong, time only).
#pragma once
#ifdef CSLAM_PROBE_HOOKS_FILE
#include CSLAM_PROBE_HOOKS_FILE
#endif

// ---- k_scan_mq (scan_mq.hip) ---------------------------------------------------------------------
#ifndef CSLAM_MQ_HOOK_KERNEL_BEGIN
#define CSLAM_MQ_HOOK_KERNEL_BEGIN      // in front of the round loop
#endif
#ifndef CSLAM_MQ_HOOK_STEP_BEGIN
#define CSLAM_MQ_HOOK_STEP_BEGIN        // top of a slice step, in front of the wait for the B pieces
#endif
#ifndef CSLAM_MQ_HOOK_B_LANDED
#define CSLAM_MQ_HOOK_B_LANDED          // this wave's B pieces have landed, in front of the barrier
#endif
#ifndef CSLAM_MQ_HOOK_BARRIER_PASSED
#define CSLAM_MQ_HOOK_BARRIER_PASSED
#endif
#ifndef CSLAM_MQ_HOOK_STAGE_BEGIN
#define CSLAM_MQ_HOOK_STAGE_BEGIN       // in front of a stage's k-block loop
#endif
#ifndef CSLAM_MQ_HOOK_STAGE_END
#define CSLAM_MQ_HOOK_STAGE_END
#endif
#ifndef CSLAM_MQ_HOOK_KERNEL_END
#define CSLAM_MQ_HOOK_KERNEL_END        // behind the round loop
#endif
#ifndef CSLAM_MQ_MFMA                   // one step of a (row, query) chain
#define CSLAM_MQ_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#endif
#ifndef CSLAM_MQ_REFILL                 // false: the ring is never refilled
#define CSLAM_MQ_REFILL true
#endif

// Address audit (tools/probes/mq_probe_audit.hpp): every global address the kernel is about to use is compared
// with what MqArgs says the operand holds; a violation is logged and the access redirected (loads: AUDIT_SRC
// may replace the pointer variable) or left out (stores: AUDIT_DST / AUDIT_IDX guard the block behind them).
// kinds: 0 pool rows, 1 staged queries, 2 query norms, 3 row norms, 4 chunk partials, 5 composites, 6 dots
#ifndef CSLAM_MQ_AUDIT_SRC
#define CSLAM_MQ_AUDIT_SRC(kind, ptr, bytes)
#endif
#ifndef CSLAM_MQ_AUDIT_DST
#define CSLAM_MQ_AUDIT_DST(kind, ptr, bytes)
#endif
#ifndef CSLAM_MQ_AUDIT_IDX
#define CSLAM_MQ_AUDIT_IDX(kind, idx, bound)
#endif
#ifndef CSLAM_MQ_AUDIT_COMBINE
#define CSLAM_MQ_AUDIT_COMBINE
#endif

// ---- k_scan_dma (knn.hip) ------------------------------------------------------------------------
#ifndef CSLAM_DMA_HOOK_KERNEL_BEGIN
#define CSLAM_DMA_HOOK_KERNEL_BEGIN
#endif
#ifndef CSLAM_DMA_HOOK_QUERIES_LANDED
#define CSLAM_DMA_HOOK_QUERIES_LANDED
#endif
#ifndef CSLAM_DMA_HOOK_STAGE_LANDED
#define CSLAM_DMA_HOOK_STAGE_LANDED
#endif
#ifndef CSLAM_DMA_HOOK_KERNEL_END
#define CSLAM_DMA_HOOK_KERNEL_END
#endif
