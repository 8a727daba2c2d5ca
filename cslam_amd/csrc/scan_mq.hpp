// Launch interface of the multi-tile LDS-DMA scan (scan_mq.hip) for knn.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "dma_scan.hpp"

namespace cslam {

struct MqArgs {
    const float* X;        // first row of the window
    int64_t xs;            // row stride (floats)
    int64_t n;             // rows of the window
    int dp;                // padded row length (multiple of the slice width)
    const float* Q;        // staged queries [16 tiles][dp], rows past m_valid zero
    const float* rnq;      // [16 tiles] 1 / ||q||   (best-match mode)
    int m_valid;
    float* dots;           // dots mode: [m][dots_stride]
    int64_t dots_stride;
    int n_blocks;          // grid: rows are split evenly over the blocks
    int p_max;             // rounds of the block with the most rows (candidate layout)
    int device;
    const float* rnx;      // 1 / ||x|| of the window's rows (best-match mode)
    uint32_t row0;         // pool row of the window's first row
    uint64_t* cand;        // best-match mode: [m][cand_stride] composites, n_blocks * p_max * waves per query
    int64_t cand_stride;
    int64_t split;         // window-relative row from which composites go to cand_c (INT64_MAX: one window)
    uint64_t* cand_c;
    f32x4* part;           // K-split mode: [strip][chunks][tiles][64] chunk partials (chunk = 512 floats)
    int part_ks;           // K-split mode: K parts per strip group (grid = n_blocks x part_ks)
    int part_per_cu;       // K-split mode: 2 = the two-blocks-per-CU form (four waves, rings of two 64-float stages)
};

constexpr int kMqModeDots = 0, kMqModeBest = 1, kMqModePart = 2;

int mq_max_tiles();                          // query tiles (of 16) one pass takes
int mq_waves_available(int waves);            // 4 always; 8 only in builds with -DCSLAM_MQ_WAVES8
int mq_panel_strips(int tiles, int waves);   // strips (of 16 rows) per wave and round
int launch_scan_mq(const MqArgs& a, int tiles, int kw, int waves, int mode, bool nt, hipStream_t st);
// fold the K-split pass's chunk partials of window `a` (dots, or composites at cand[..][cand_off + strip])
int launch_mq_combine(const MqArgs& a, int tiles, int nchunks, bool best, int64_t cand_off, hipStream_t st);

}  // namespace cslam
