// Device-side helpers shared by knn.hip and shard.hip: (similarity, row) composites and their
// wave / block maxima.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace cslam {

// ---------------------------------------------------------------------------------------------
// composites: (orderable(sim) << 32) | (0xFFFFFFFF - row)  → max = best similarity, lowest row
// ---------------------------------------------------------------------------------------------
__device__ inline uint32_t f2ord(float s) {
    if (s == 0.0f) s = 0.0f;  // -0 → +0
    uint32_t u = __float_as_uint(s);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ inline float ord2f(uint32_t o) {
    uint32_t u = (o & 0x80000000u) ? (o ^ 0x80000000u) : ~o;
    return __uint_as_float(u);
}
__device__ inline uint64_t make_comp(float s, uint32_t row) {
    return ((uint64_t)f2ord(s) << 32) | (uint64_t)(0xFFFFFFFFu - row);
}

// 64-lane maximum of a 64-bit composite on the VALU (DPP) — `__shfl_xor` of a 64-bit value is two
// LDS-crossbar ds_bpermute per step, 12 dependent ones per round of the select.  0 is the
// identity (an empty composite).  Result in every lane.
template <int CTRL, int ROW_MASK = 0xF>
__device__ inline uint64_t dpp_mov_u64(uint64_t v) {
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, CTRL, ROW_MASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), CTRL, ROW_MASK, 0xF, false);
    return ((uint64_t)(uint32_t)hi << 32) | (uint64_t)(uint32_t)lo;
}
__device__ inline uint64_t wave_max_u64(uint64_t m) {
    uint64_t o;
    o = dpp_mov_u64<0xB1>(m);        m = o > m ? o : m;   // quad_perm [1,0,3,2]
    o = dpp_mov_u64<0x4E>(m);        m = o > m ? o : m;   // quad_perm [2,3,0,1]
    o = dpp_mov_u64<0x141>(m);       m = o > m ? o : m;   // row_half_mirror
    o = dpp_mov_u64<0x140>(m);       m = o > m ? o : m;   // row_mirror: every lane = max of its row
    o = dpp_mov_u64<0x142, 0xA>(m);  m = o > m ? o : m;   // row_bcast15: rows 1, 3 take rows 0, 2
    o = dpp_mov_u64<0x143, 0xC>(m);  m = o > m ? o : m;   // row_bcast31: rows 2, 3 take rows 0..1
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)m, 63);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(m >> 32), 63);
    return ((uint64_t)hi << 32) | lo;
}


// best composite of c[0..n) over a block of 256 threads (4 waves); result in every thread.
// `scratch` = 4 uint64_t of LDS; two barriers.
__device__ inline uint64_t block_max_u64(const uint64_t* __restrict__ c, int64_t n, uint64_t* scratch) {
    uint64_t m = 0;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const uint64_t v = c[i];
        m = v > m ? v : m;
    }
    m = wave_max_u64(m);
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = m;
    __syncthreads();
    m = scratch[0];
#pragma unroll
    for (int x = 1; x < 4; ++x) m = scratch[x] > m ? scratch[x] : m;
    __syncthreads();
    return m;
}
__device__ inline int32_t comp_row(uint64_t c) {
    return c ? (int32_t)(0xFFFFFFFFu - (uint32_t)(c & 0xFFFFFFFFu)) : -1;
}
__device__ inline float comp_sim(uint64_t c) { return c ? ord2f((uint32_t)(c >> 32)) : -INFINITY; }

}  // namespace cslam
