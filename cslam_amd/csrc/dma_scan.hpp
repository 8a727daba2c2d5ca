// Device-side pieces shared by the LDS-DMA scans (knn.hip: k_scan_dma / k_scan_dma_ks, scan_mq.hip:
// k_scan_mq): the geometry of a DMA piece in LDS and the run-time `s_waitcnt vmcnt`.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace cslam {

// COVERS = true: the caller's count says that every INLINE-ASM load issued so far has landed when
// the wait returns (k_scan_dma's query fragments; tie_b() follows).  The region is bracketed by
// assembly comments that tools/check_asm_load_hazard.py reads: it cannot follow a run-time count,
// so it takes the kernel's word at a COVERS selector (the bit-exact tests check that word) and
// keeps every asm load owed across the others.
template <bool COVERS = false>
__device__ inline void wait_vm_dyn(int n) {
    // s_waitcnt takes an immediate: wave-uniform switch over the exact count (vmcnt is 6 bits;
    // more than 48 younger operations cannot hide anything that is still needed here)
    asm volatile("; cslam.wait_dyn.begin" ::: "memory");
#define CSLAM_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
    switch (n) {
        CSLAM_W(0) CSLAM_W(1) CSLAM_W(2) CSLAM_W(3) CSLAM_W(4) CSLAM_W(5) CSLAM_W(6) CSLAM_W(7)
        CSLAM_W(8) CSLAM_W(9) CSLAM_W(10) CSLAM_W(11) CSLAM_W(12) CSLAM_W(13) CSLAM_W(14)
        CSLAM_W(15) CSLAM_W(16) CSLAM_W(17) CSLAM_W(18) CSLAM_W(19) CSLAM_W(20) CSLAM_W(21)
        CSLAM_W(22) CSLAM_W(23) CSLAM_W(24) CSLAM_W(25) CSLAM_W(26) CSLAM_W(27) CSLAM_W(28)
        CSLAM_W(29) CSLAM_W(30) CSLAM_W(31) CSLAM_W(32) CSLAM_W(33) CSLAM_W(34) CSLAM_W(35)
        CSLAM_W(36) CSLAM_W(37) CSLAM_W(38) CSLAM_W(39) CSLAM_W(40) CSLAM_W(41) CSLAM_W(42)
        CSLAM_W(43) CSLAM_W(44) CSLAM_W(45) CSLAM_W(46) CSLAM_W(47) CSLAM_W(48)
        default: asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); break;
    }
#undef CSLAM_W
    if (COVERS)
        asm volatile("; cslam.wait_dyn.end covers-asm-loads" ::: "memory");
    else
        asm volatile("; cslam.wait_dyn.end" ::: "memory");
}

// Slice width KW (floats): 256 = one row per DMA instruction.  KW = 128 / 64: one DMA instruction
// moves R = 256 / KW ROWS x KW floats (lanes [k 64/R, (k+1) 64/R) fetch row k of the piece).  The
// 16-B segments of row k of a piece are XOR-k swizzled on the SOURCE side (LDS position j holds
// global segment j ^ k) and the piece pitch is 1024 + 32 B (R <= 2) or 1024 + 64 B (R = 4), so
// that the 16 lanes of a ds_read_b128 pass - 16 rows at one k offset - hit 16 different 16-B slots.
template <int KW>
struct DmaGeom {
    static constexpr int R = 256 / KW;                       // rows per DMA instruction
    static constexpr int NB = KW / 16;                       // B fragments per slice
    static constexpr int kPiece = 1024 + (R == 4 ? 64 : 32); // LDS bytes per piece
    static constexpr int kStage = 16 * kPiece;               // 16 R rows x KW floats
    static constexpr int kFold = 512 / KW;                   // slices per 512-float chunk
};


}  // namespace cslam
