// shard.hip — rank-local half of the multi-robot matching step, entirely on the device.
//
// rank r = robot r holds ONLY its own descriptor pool (the reference mirrors every peer pool on
// every robot, cslam/loop_closure_sparse_matching.py:26-32).  One swarm step:
//   publish  own new block [b][dim] + keyframe ids  → RCCL all-gather (descriptor communicator)
//   match    append the own block, ONE scan of the pool for all gathered descriptors,
//            records on the device:
//              foreign descriptor x best OLD row  → add_other_robot_global_descriptor (lcsm.py:58-76)
//              foreign descriptor x best NEW row  → the peer's add_local_global_descriptor
//                                                  (lcsm.py:37-56), emitted on its behalf
//              own descriptor x local pool        → match_local_loop_closures (lcsm.py:78-97)
//            → RCCL all-gather of the 20-byte records (record communicator) → pinned host copy
//   fetch    the host reads the records of a finished step (rank order, slot order: every rank
//            applies them to its selector replica in the same order)
// Nothing in publish/match synchronises the host; steps are pipelined over kSlots buffers.
//
// Equivalence with the reference's sequential processing: "everybody receives the step's
// descriptors, then everybody inserts its own" — tests/test_multi_robot.py replays that order
// through the reference semantics and compares the union of candidate edges.
#include <cstring>
#include <vector>

#include "common.hpp"
#include "composite.hpp"
#include "topk_parts.hpp"

using namespace cslam;

namespace {

constexpr int kSlots = 4;        // steps in flight
constexpr int kFusedMaxK = 64;   // b_max + k_local up to here: one scan, one top-K per descriptor

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_block(const float* __restrict__ src, int64_t src_stride,
                                                    int b, int b_max, int dim,
                                                    float* __restrict__ dst) {
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (w >= b_max) return;
    float* d = dst + w * dim;
    if (w < b) {
        const float* s = src + w * src_stride;
        for (int k = lane; k < dim; k += 64) d[k] = s[k];
    } else {
        for (int k = lane; k < dim; k += 64) d[k] = 0.f;
    }
}

// One thread per gathered descriptor i = s * b_max + j (owner s, position j).
// FUSED: topk_idx/topk_sim hold the sorted top-K of the pool AFTER the own block was appended;
//        the best old row (< n0) and the best new row (>= n0) are both inside it (K > b).
// else : idx_a/sim_a = best in [0, n0), idx_c/sim_c = best in [n0, n0 + b).
// Record slots 3i (received descriptor vs old rows), 3i+1 (vs new rows, on the peer's behalf),
// 3i+2 (own descriptor: intra-robot match); robot0_id = -1 marks an empty slot.
// The record slots of ONE gathered descriptor i (one thread): see k_make_records.
template <bool FUSED>
__device__ inline void make_records_one(
    int i, int me, int b_max, const int64_t* __restrict__ all_meta, int64_t n0,
    const int32_t* kf_of_row, double thr, const int32_t* topk_idx, const float* topk_sim, int K, int k_local,
    int min_inbetween, const int32_t* idx_a, const float* sim_a, const int32_t* idx_c, const float* sim_c,
    cslam_match_record_t* __restrict__ recs) {
    const int64_t* own_meta = all_meta + (int64_t)me * (1 + b_max);
    auto kf_at = [&](int32_t row) {
        return row >= n0 ? (int32_t)own_meta[1 + (row - n0)] : kf_of_row[row];
    };
    const int s = i / b_max, j = i % b_max;
    const int64_t* meta = all_meta + (int64_t)s * (1 + b_max);
    cslam_match_record_t none;
    none.robot0_id = -1;
    none.robot0_keyframe_id = 0;
    none.robot1_id = 0;
    none.robot1_keyframe_id = 0;
    none.weight = 0.f;
    cslam_match_record_t r0 = none, r1 = none, r2 = none;
    if (j < (int)meta[0]) {
        const int32_t kf_s = (int32_t)meta[1 + j];
        if (s != me) {
            int32_t ia = -1, ic = -1;
            float sa = 0.f, sc = 0.f;
            if (FUSED) {
                for (int t = 0; t < K; ++t) {
                    const int32_t row = topk_idx[(int64_t)i * K + t];
                    if (row < 0) break;
                    if (row < n0) {
                        if (ia < 0) { ia = row; sa = topk_sim[(int64_t)i * K + t]; }
                    } else if (ic < 0) {
                        ic = row; sc = topk_sim[(int64_t)i * K + t];
                    }
                    if (ia >= 0 && ic >= 0) break;
                }
            } else {
                ia = idx_a[i]; sa = sim_a[i];
                ic = idx_c[i]; sc = sim_c[i];
            }
            const bool a_ok = ia >= 0 && (double)sa >= thr;
            if (a_ok) {  // this robot's add_other_robot_global_descriptor (lcsm.py:69-75)
                r0.robot0_id = me;
                r0.robot0_keyframe_id = kf_at(ia);
                r0.robot1_id = s;
                r0.robot1_keyframe_id = kf_s;
                r0.weight = sa;
            }
            // the peer's add_local_global_descriptor would have met this newer keyframe in its
            // mirror of our pool (lcsm.py:47-55): only when it beats every older row
            if (ic >= 0 && (double)sc >= thr && !(ia >= 0 && sa >= sc)) {
                r1.robot0_id = s;
                r1.robot0_keyframe_id = kf_s;
                r1.robot1_id = me;
                r1.robot1_keyframe_id = kf_at(ic);
                r1.weight = sc;
            }
        } else if (FUSED && k_local > 0) {
            // match_local_loop_closures (lcsm.py:78-97) for own keyframe j, whose pool row is
            // n0 + j: the reference searched the pool as it was right after inserting it, i.e.
            // rows <= n0 + j; keep the first k_local of those, drop the keyframe itself when it
            // ranks first, then take the first far-enough, similar-enough one
            const int64_t self_row = n0 + j;
            int kept = 0;
            for (int t = 0; t < K && kept < k_local; ++t) {
                const int32_t row = topk_idx[(int64_t)i * K + t];
                if (row < 0) break;
                if (row > self_row) continue;  // a later keyframe of this block: not there yet
                ++kept;
                if (kept == 1 && row == self_row) continue;  // lcsm.py:83-84
                const int32_t kf = kf_at(row);
                const int d = kf > kf_s ? kf - kf_s : kf_s - kf;
                if (d < min_inbetween) continue;
                const float sv = topk_sim[(int64_t)i * K + t];
                if ((double)sv < thr) continue;
                r2.robot0_id = me;
                r2.robot0_keyframe_id = kf_s;
                r2.robot1_id = me;
                r2.robot1_keyframe_id = kf;
                r2.weight = sv;
                break;
            }
        }
    }
    recs[3 * (int64_t)i + 0] = r0;
    recs[3 * (int64_t)i + 1] = r1;
    recs[3 * (int64_t)i + 2] = r2;
}

template <bool FUSED>
__global__ __launch_bounds__(256) void k_make_records(
    int me, int nb_robots, int b_max, const int64_t* __restrict__ all_meta, int64_t n0,
    int32_t* __restrict__ kf_of_row, int b_own, double thr, const int32_t* __restrict__ topk_idx,
    const float* __restrict__ topk_sim, int K, int k_local, int min_inbetween,
    const int32_t* __restrict__ idx_a, const float* __restrict__ sim_a,
    const int32_t* __restrict__ idx_c, const float* __restrict__ sim_c,
    cslam_match_record_t* __restrict__ recs) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    // keyframe id of the pool rows appended by this step: rows >= n0 are looked up in the own
    // meta block (their table entries are written here, for the steps to come)
    const int64_t* own_meta = all_meta + (int64_t)me * (1 + b_max);
    if (i < b_own) kf_of_row[n0 + i] = (int32_t)own_meta[1 + i];
    if (i >= nb_robots * b_max) return;
    make_records_one<FUSED>(i, me, b_max, all_meta, n0, kf_of_row, thr, topk_idx, topk_sim, K, k_local, min_inbetween,
                            idx_a, sim_a, idx_c, sim_c, recs);
}

// Fused intra + inter search, rank select and records in ONE launch behind the scan (round 4): block i runs the
// rank select of topk_parts.hpp for gathered descriptor i over what the best-match scan with dots left
// (pool_topk_candidates) and its first thread goes on to the descriptor's record slots - instead of
// k_topk_from_parts + k_make_records<true> (one kernel boundary of ~6 us and a 4-us launch less per step).
template <class Parts>
__global__ __launch_bounds__(256) void k_topk_records(
    const uint64_t* cand, int64_t cand_stride, int n_parts, const Parts parts, const float* dots,
    int64_t dots_stride, const float* __restrict__ rnx, const float* __restrict__ rnq, uint32_t row0, int64_t n,
    int me, int b_max, const int64_t* __restrict__ all_meta, int64_t n0, int32_t* kf_of_row, int b_own, double thr,
    int32_t* topk_idx, float* topk_sim, int K, int k_local, int min_inbetween,
    cslam_match_record_t* __restrict__ recs) {
    const int i = blockIdx.x;
    topk_from_parts_body(cand + (int64_t)i * cand_stride, n_parts, parts, dots + (int64_t)i * dots_stride, rnx, rnq[i],
                         row0, n, K, topk_idx + (int64_t)i * K, topk_sim + (int64_t)i * K);
    __syncthreads();   // the block's list is complete and visible to its first thread
    if (threadIdx.x != 0) return;
    const int64_t* own_meta = all_meta + (int64_t)me * (1 + b_max);
    if (i < b_own) kf_of_row[n0 + i] = (int32_t)own_meta[1 + i];
    make_records_one<true>(i, me, b_max, all_meta, n0, kf_of_row, thr, topk_idx, topk_sim, K, k_local, min_inbetween,
                           nullptr, nullptr, nullptr, nullptr, recs);
}

// Split best-match mode, final level and records in ONE launch behind the scan: block i reduces
// the waves' composites of gathered descriptor i (best old row, best row of the new block) and
// its first thread writes the two inter-robot record slots exactly like k_make_records<false>.
struct RecordArgs {
    int me, nb_robots, b_max;
    const int64_t* all_meta;
    int64_t n0;
    int32_t* kf_of_row;
    int b_own;
    double thr;
    cslam_match_record_t* recs;
};
// the record slots of gathered descriptor i from its two winners (best old row, best row of the new block)
__device__ inline void records_of_descriptor(const RecordArgs& R, int i, uint64_t ba, uint64_t bc) {
    const int64_t* own_meta = R.all_meta + (int64_t)R.me * (1 + R.b_max);
    if (i < R.b_own) R.kf_of_row[R.n0 + i] = (int32_t)own_meta[1 + i];
    auto kf_at = [&](int32_t row) {
        return row >= R.n0 ? (int32_t)own_meta[1 + (row - R.n0)] : R.kf_of_row[row];
    };
    const int s = i / R.b_max, j = i % R.b_max;
    const int64_t* meta = R.all_meta + (int64_t)s * (1 + R.b_max);
    cslam_match_record_t none;
    none.robot0_id = -1;
    none.robot0_keyframe_id = 0;
    none.robot1_id = 0;
    none.robot1_keyframe_id = 0;
    none.weight = 0.f;
    cslam_match_record_t r0 = none, r1 = none;
    if (j < (int)meta[0] && s != R.me) {
        const int32_t kf_s = (int32_t)meta[1 + j];
        const int32_t ia = comp_row(ba), ic = comp_row(bc);
        const float sa = comp_sim(ba), sc = comp_sim(bc);
        if (ia >= 0 && (double)sa >= R.thr) {  // lcsm.py:69-75
            r0.robot0_id = R.me;
            r0.robot0_keyframe_id = kf_at(ia);
            r0.robot1_id = s;
            r0.robot1_keyframe_id = kf_s;
            r0.weight = sa;
        }
        if (ic >= 0 && (double)sc >= R.thr && !(ia >= 0 && sa >= sc)) {  // lcsm.py:47-55, on the peer's behalf
            r1.robot0_id = s;
            r1.robot0_keyframe_id = kf_s;
            r1.robot1_id = R.me;
            r1.robot1_keyframe_id = kf_at(ic);
            r1.weight = sc;
        }
    }
    R.recs[3 * (int64_t)i + 0] = r0;
    R.recs[3 * (int64_t)i + 1] = r1;
    R.recs[3 * (int64_t)i + 2] = none;
}
__global__ __launch_bounds__(256) void k_records_from_candidates(
    const RecordArgs R, const uint64_t* __restrict__ cand_a, const uint64_t* __restrict__ cand_c,
    int64_t cand_stride, int64_t n_cand) {
    __shared__ uint64_t scratch[4];
    const int i = blockIdx.x;
    const uint64_t ba = block_max_u64(cand_a + (int64_t)i * cand_stride, n_cand, scratch);
    const uint64_t bc = block_max_u64(cand_c + (int64_t)i * cand_stride, n_cand, scratch);
    if (threadIdx.x != 0) return;
    records_of_descriptor(R, i, ba, bc);
}
// The gathered exchange buffer holds, per rank, [b_max][dim] fp32 rows followed by the (count,
// keyframe ids) header: ONE all-gather per step instead of two.  This kernel copies the rows to the
// dense [R * b_max][dim] block the scan reads its queries from and the headers to all_meta.
__global__ __launch_bounds__(256) void k_unpack_exchange(const char* __restrict__ pack, int64_t per_rank_bytes,
                                                         int64_t hdr_off, int R, int b_max, int dim,
                                                         float* __restrict__ all_rows,
                                                         int64_t* __restrict__ all_meta, int rows_too) {
    const int s = blockIdx.y;
    const char* src = pack + (int64_t)s * per_rank_bytes;
    if (blockIdx.x == 0) {
        const int64_t* meta = reinterpret_cast<const int64_t*>(src + hdr_off);
        for (int i = threadIdx.x; i < 1 + b_max; i += 256) all_meta[(int64_t)s * (1 + b_max) + i] = meta[i];
    }
    if (!rows_too) return;   // the scan's query block is staged straight from `pack` (publish)
    const int64_t n4 = (int64_t)b_max * dim / 4;   // dim is a multiple of 4 here (16-byte copies), else scalar
    const float* rows = reinterpret_cast<const float*>(src);
    float* dst = all_rows + (int64_t)s * b_max * dim;
    if (((int64_t)b_max * dim) % 4 == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(rows) & 15) == 0) {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256)
            reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(rows)[i];
    } else {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)b_max * dim; i += (int64_t)gridDim.x * 256)
            dst[i] = rows[i];
    }
}

__global__ void k_fill_i32(int32_t* p, int64_t n, int32_t v) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}

struct Slot {
    char* pack = nullptr;             // [R][per_rank_bytes]: rows [b_max][dim] + header, own part at index me
                                      // (in-place all-gather: ONE collective for descriptors and ids)
    float* all_rows = nullptr;        // [R][b_max][dim]   unpacked: the scan's query block
    int64_t* all_meta = nullptr;      // [R][1 + b_max]    count, keyframe ids
    int64_t* meta_host = nullptr;     // pinned [1 + b_max]
    int32_t* topk_idx = nullptr;      // [R*b_max][K]
    float* topk_sim = nullptr;
    int32_t *idx_a = nullptr, *idx_c = nullptr;   // bulk mode
    float *sim_a = nullptr, *sim_c = nullptr;
    float* staged = nullptr;          // best2, 17..80 gathered descriptors: the multi-tile scan's query block
    float* staged_rnq = nullptr;      // ... and inverse norms, staged at publish time on the exchange stream
    bool staged_valid = false;        // this step's block was staged at publish time
    bool rows_valid = true;           // all_rows holds this step's gathered rows (false: staged from `pack`, rows not unpacked)
    cslam_match_record_t* recs_all = nullptr;     // [R][3*R*b_max]  (own part at index me)
    cslam_match_record_t* recs_host = nullptr;    // pinned copy
    hipEvent_t ev_src = nullptr, ev_pub = nullptr, ev_match = nullptr, ev_done = nullptr;
    int b = 0;
    int64_t n0 = 0;   // pool rows before this step's own block (appended at publish time)
    int64_t step = -1;
    int state = 0;  // 0 free, 1 published, 2 matched
};

}  // namespace

struct cslam_shard {
    cslam_pool_t* pool = nullptr;
    cslam_comm_t* comm_desc = nullptr;
    cslam_comm_t* comm_rec = nullptr;
    int me = 0, R = 1, b_max = 0, dim = 0, device = 0;
    int k_local = 0, min_inbetween = 0, K = 1;
    bool fused = true;
    bool best2 = false;  // no intra-robot matching, <= 80 gathered descriptors: ONE split best-match scan pass
    double thr = 0.0;
    int32_t* kf_of_row = nullptr;
    int64_t kf_cap = 0;
    hipStream_t s_own = nullptr;  // used when there is no communicator (single robot)
    Slot slot[kSlots];
    int64_t pub_step = 0, match_step = 0, fetch_step = 0;
    int64_t rec_slots = 0;        // 3 * R * b_max
    int64_t per_rank_bytes = 0;   // exchange payload of one rank: rows + header, 16-byte multiple
    int64_t hdr_off = 0;          // byte offset of the header inside a rank's part (8-byte aligned)
};

namespace {

template <class T>
int dalloc(T** p, size_t n) {
    CSLAM_HIP_TRY(hipMalloc(reinterpret_cast<void**>(p), n * sizeof(T)));
    return CSLAM_OK;
}

int ensure_kf_capacity(cslam_shard* sh, int64_t want, hipStream_t st) {
    if (want <= sh->kf_cap) return CSLAM_OK;
    int64_t ncap = sh->kf_cap > 0 ? sh->kf_cap : 1024;
    while (ncap < want) ncap *= 2;
    int32_t* nk = nullptr;
    CSLAM_TRY(dalloc(&nk, (size_t)ncap));
    // Record kernels of earlier steps, queued on the MATCHING stream behind a scan, still write
    // kf_of_row[n0 + i] into the old table (and read it): `st` (the exchange stream) is not ordered
    // behind them, so the whole device drains BEFORE the old entries are copied — a copy enqueued
    // first would lose those writes and later records would carry keyframe id -1
    // (tests/test_shard_gpu.py::test_pipelined_keyframe_table_growth).
    if (sh->kf_of_row) CSLAM_HIP_TRY(hipDeviceSynchronize());
    CSLAM_HIP_TRY(hipMemsetAsync(nk + sh->kf_cap, 0xFF, (size_t)(ncap - sh->kf_cap) * sizeof(int32_t), st));
    if (sh->kf_of_row)
        CSLAM_HIP_TRY(hipMemcpyAsync(nk, sh->kf_of_row, (size_t)sh->kf_cap * sizeof(int32_t),
                                     hipMemcpyDeviceToDevice, st));
    CSLAM_HIP_TRY(hipStreamSynchronize(st));
    if (sh->kf_of_row) CSLAM_HIP_TRY(hipFree(sh->kf_of_row));
    sh->kf_of_row = nk;
    sh->kf_cap = ncap;
    return CSLAM_OK;
}

}  // namespace

extern "C" {

int cslam_shard_create(cslam_pool_t* pool, cslam_comm_t* comm_desc, cslam_comm_t* comm_rec,
                       int robot_id, int nb_robots, int max_block, double similarity_threshold,
                       int nb_best_matches, int intra_min_inbetween, cslam_shard_t** out) {
    if (!pool || !out || nb_robots < 1 || robot_id < 0 || robot_id >= nb_robots || max_block < 1 ||
        nb_best_matches < 0)
        return fail(CSLAM_E_INVALID, "shard_create: bad args");
    if (nb_robots > 1 && (!comm_desc || !comm_rec))
        return fail(CSLAM_E_INVALID, "shard_create: %d robots need both communicators", nb_robots);
    if (comm_desc && (cslam_comm_size(comm_desc) != nb_robots || cslam_comm_rank(comm_desc) != robot_id))
        return fail(CSLAM_E_INVALID, "shard_create: communicator rank/size != robot id/count");
    if (comm_rec && (cslam_comm_size(comm_rec) != nb_robots || cslam_comm_rank(comm_rec) != robot_id))
        return fail(CSLAM_E_INVALID, "shard_create: record communicator rank/size mismatch");
    // comm_desc == comm_rec is allowed: both exchanges then run on ONE communicator and ONE stream, in
    // the program order publish(t+1), match(t) of every rank - the simplest thing RCCL can be asked
    // to do (the record exchange of step t waits for scan t, the descriptor exchange of step t+2
    // queues behind it and still has the whole scan t+1 to complete in).
    if (max_block + (nb_best_matches > 1 ? nb_best_matches : 1) > kFusedMaxK && nb_best_matches > 0)
        return fail(CSLAM_E_INVALID,
                    "shard_create: intra-robot matching (nb_best_matches = %d) needs max_block + nb_best_matches <= %d; "
                    "a bulk block of %d is matched inter-robot only - pass nb_best_matches = 0",
                    nb_best_matches, kFusedMaxK, max_block);
    cslam_shard* sh = new cslam_shard();
    sh->pool = pool;
    sh->comm_desc = comm_desc;
    sh->comm_rec = comm_rec;
    sh->me = robot_id;
    sh->R = nb_robots;
    sh->b_max = max_block;
    sh->dim = cslam_pool_dim(pool);
    sh->device = cslam_pool_device(pool);
    sh->thr = similarity_threshold;
    sh->fused = max_block + (nb_best_matches > 1 ? nb_best_matches : 1) <= kFusedMaxK;
    sh->k_local = sh->fused ? nb_best_matches : 0;   // bulk blocks: inter-robot matching only
    sh->min_inbetween = intra_min_inbetween;
    sh->K = max_block + (sh->k_local > 1 ? sh->k_local : 1);
    // inter-robot matching only (frontend.enable_intra_robot_loop_closures = false, the node's
    // default, loop_closure_detection_node.py:36) asks two questions per gathered descriptor — best
    // old row, best row of the new block: one split best-match scan, no top-K lists
    sh->best2 = nb_best_matches == 0 && (int64_t)nb_robots * max_block <= 80;   // one scan pass (scan_mq.hip)
    sh->rec_slots = 3 * (int64_t)nb_robots * max_block;
    sh->hdr_off = ((int64_t)max_block * sh->dim * (int64_t)sizeof(float) + 7) / 8 * 8;
    sh->per_rank_bytes = (sh->hdr_off + (int64_t)(1 + max_block) * (int64_t)sizeof(int64_t) + 15) / 16 * 16;
    int rc = CSLAM_OK;
    hipError_t e = hipSetDevice(sh->device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&sh->s_own, hipStreamNonBlocking);
    if (e != hipSuccess) rc = fail(CSLAM_E_HIP, "shard_create: %s", hipGetErrorString(e));
    const size_t M = (size_t)nb_robots * max_block;
    for (int i = 0; i < kSlots && rc == CSLAM_OK; ++i) {
        Slot& s = sh->slot[i];
        auto A = [&](int r) { if (rc == CSLAM_OK) rc = r; };
        A(dalloc(&s.pack, (size_t)nb_robots * sh->per_rank_bytes));
        A(dalloc(&s.all_rows, M * sh->dim));
        A(dalloc(&s.all_meta, (size_t)nb_robots * (1 + max_block)));
        A(dalloc(&s.recs_all, (size_t)nb_robots * sh->rec_slots));
        static Knob k_prestage{"CSLAM_SHARD_PRESTAGE", 1};
        const bool prestage = knob(k_prestage) != 0;
        if (sh->best2 && M > 16 && prestage) {
            // the gathered block is staged for the multi-tile scan as soon as it has landed (publish), on
            // the exchange stream: the staging launch leaves the matching stream's critical path
            A(dalloc(&s.staged, (size_t)pool_staged_floats(pool, (int)M)));
            A(dalloc(&s.staged_rnq, (size_t)((M + 15) / 16 * 16)));
        }
        if (sh->fused && !sh->best2) {
            A(dalloc(&s.topk_idx, M * sh->K));
            A(dalloc(&s.topk_sim, M * sh->K));
        } else {
            A(dalloc(&s.idx_a, M));
            A(dalloc(&s.idx_c, M));
            A(dalloc(&s.sim_a, M));
            A(dalloc(&s.sim_c, M));
        }
        if (rc != CSLAM_OK) break;
        hipError_t h = hipHostMalloc(reinterpret_cast<void**>(&s.meta_host),
                                     (size_t)(1 + max_block) * sizeof(int64_t), hipHostMallocDefault);
        if (h == hipSuccess)
            h = hipHostMalloc(reinterpret_cast<void**>(&s.recs_host),
                              (size_t)nb_robots * sh->rec_slots * sizeof(cslam_match_record_t),
                              hipHostMallocDefault);
        for (hipEvent_t* ev : {&s.ev_src, &s.ev_pub, &s.ev_match, &s.ev_done})
            if (h == hipSuccess) h = hipEventCreateWithFlags(ev, hipEventDisableTiming);
        if (h != hipSuccess) rc = fail(CSLAM_E_HIP, "shard_create: %s", hipGetErrorString(h));
    }
    if (rc == CSLAM_OK) rc = ensure_kf_capacity(sh, cslam_pool_capacity(pool), sh->s_own);
    if (rc != CSLAM_OK) {
        cslam_shard_destroy(sh);
        return rc;
    }
    *out = sh;
    return CSLAM_OK;
}

int cslam_shard_destroy(cslam_shard_t* sh) {
    if (!sh) return CSLAM_OK;
    (void)hipSetDevice(sh->device);
    (void)hipDeviceSynchronize();
    for (int i = 0; i < kSlots; ++i) {
        Slot& s = sh->slot[i];
        for (void* p : {(void*)s.pack, (void*)s.all_rows, (void*)s.all_meta, (void*)s.staged, (void*)s.staged_rnq,
                        (void*)s.topk_idx, (void*)s.topk_sim,
                        (void*)s.idx_a, (void*)s.idx_c, (void*)s.sim_a, (void*)s.sim_c,
                        (void*)s.recs_all})
            if (p) (void)hipFree(p);
        if (s.meta_host) (void)hipHostFree(s.meta_host);
        if (s.recs_host) (void)hipHostFree(s.recs_host);
        for (hipEvent_t ev : {s.ev_src, s.ev_pub, s.ev_match, s.ev_done})
            if (ev) (void)hipEventDestroy(ev);
    }
    if (sh->kf_of_row) (void)hipFree(sh->kf_of_row);
    if (sh->s_own) (void)hipStreamDestroy(sh->s_own);
    delete sh;
    return CSLAM_OK;
}

/* Register rows that were put into the pool before the shard existed (or behind its back):
 * keyframe ids of pool rows [first, first + n). */
int cslam_shard_set_keyframes(cslam_shard_t* sh, int64_t first, const int32_t* keyframe_ids,
                              int64_t n) {
    if (!sh || first < 0 || n < 0 || (n > 0 && !keyframe_ids))
        return fail(CSLAM_E_INVALID, "shard_set_keyframes: bad args");
    if (n == 0) return CSLAM_OK;
    CSLAM_HIP_TRY(hipSetDevice(sh->device));
    CSLAM_TRY(ensure_kf_capacity(sh, first + n, sh->s_own));
    CSLAM_HIP_TRY(hipMemcpy(sh->kf_of_row + first, keyframe_ids, (size_t)n * sizeof(int32_t),
                            hipMemcpyHostToDevice));
    return CSLAM_OK;
}

int cslam_shard_publish(cslam_shard_t* sh, const float* new_rows_dev, int64_t row_stride,
                        const int64_t* keyframe_ids, int b, void* stream) {
    if (!sh || b < 0 || b > (sh ? sh->b_max : 0) || (b > 0 && (!new_rows_dev || !keyframe_ids || row_stride < sh->dim)))
        return fail(CSLAM_E_INVALID, "shard_publish: bad args (block of %d, max %d)", b,
                    sh ? sh->b_max : 0);
    Slot& s = sh->slot[sh->pub_step % kSlots];
    if (s.state != 0)
        return fail(CSLAM_E_INVALID, "shard_publish: %d steps in flight — fetch the oldest first", kSlots);
    CSLAM_HIP_TRY(hipSetDevice(sh->device));
    hipStream_t st = (hipStream_t)stream;
    const int R = sh->R, me = sh->me, bm = sh->b_max, dim = sh->dim;
    s.meta_host[0] = b;
    for (int i = 0; i < bm; ++i) s.meta_host[1 + i] = i < b ? keyframe_ids[i] : -1;
    // this rank's part of the exchange buffer: rows, then the header (in-place all-gather)
    char* my_pack = s.pack + (int64_t)me * sh->per_rank_bytes;
    float* my_rows = reinterpret_cast<float*>(my_pack);
    int64_t* my_meta = reinterpret_cast<int64_t*>(my_pack + sh->hdr_off);
    // With peers, packing and the gather run on the descriptor communicator's own stream, behind
    // an event on the producer's stream: the compute stream (the previous step's scan) is not
    // touched by a publish at all.
    hipStream_t ps = st;
    if (R > 1) {
        ps = (hipStream_t)cslam_comm_stream(sh->comm_desc);
        CSLAM_HIP_TRY(hipEventRecord(s.ev_src, st));
        CSLAM_HIP_TRY(hipStreamWaitEvent(ps, s.ev_src, 0));
    }
    CSLAM_HIP_TRY(hipMemcpyAsync(my_meta, s.meta_host, (size_t)(1 + bm) * sizeof(int64_t),
                                 hipMemcpyHostToDevice, ps));
    hipLaunchKernelGGL(k_pack_block, dim3((unsigned)((bm + 3) / 4)), dim3(256), 0, ps, new_rows_dev,
                       row_stride, b, bm, dim, my_rows);
    CSLAM_HIP_TRY(hipGetLastError());
    // The own block enters the pool HERE, on the exchange stream, not in front of the scan on the
    // compute stream: scans in flight read rows below their own window only, and match() works on
    // the explicit windows [0, n0) / [n0, n0 + b) recorded now.
    s.n0 = cslam_pool_size(sh->pool);
    if (b > 0) {
        CSLAM_TRY(pool_append_on(sh->pool, my_rows, b, dim, ps));
        CSLAM_TRY(ensure_kf_capacity(sh, s.n0 + b, ps));
    }
    // ONE collective: rows and (count, keyframe ids) of every rank
    if (R > 1)
        CSLAM_TRY(cslam_comm_allgather(sh->comm_desc, my_pack, s.pack, sh->per_rank_bytes, CSLAM_COMM_OWN_STREAM));
    // Stage the gathered block for the multi-tile scan here, on the exchange stream - but only when that
    // scan leaves CUs free: a scan block fills a CU (registers and LDS), so beside a scan on every CU the
    // staging kernel would only queue up in front of the NEXT scan, where it costs more than on the
    // matching stream itself (cfg4, 25 000 rows: 0.216 ms per step with 60 CUs free, 0.267-0.297 without
    // them against 0.245; profiles/r03_shard_pack_prestage_ab.txt).  A staged block is read STRAIGHT from the
    // exchange buffer (one part per rank: rows, then the header) and only the headers are unpacked: the
    // dense copy of the rows is made later, by match(), in the rare case a path without the staged block
    // runs (ensure_rows) - one 1.3-MB copy and its launch less per step on the exchange stream.
    s.staged_valid = s.staged != nullptr && (sh->per_rank_bytes & 15) == 0 &&
                     pool_mq_free_cus(sh->pool, R * bm, s.n0 + b) >= 16;
    s.rows_valid = !s.staged_valid;
    {
        const int64_t n4 = (int64_t)bm * dim / 4;
        unsigned gx = (unsigned)((n4 + 255) / 256);
        gx = !s.rows_valid ? 1 : (gx < 1 ? 1 : (gx > 64 ? 64 : gx));
        hipLaunchKernelGGL(k_unpack_exchange, dim3(gx, R), dim3(256), 0, ps, s.pack, sh->per_rank_bytes, sh->hdr_off, R, bm,
                           dim, s.all_rows, s.all_meta, s.rows_valid ? 1 : 0);
        CSLAM_HIP_TRY(hipGetLastError());
    }
    if (s.staged_valid)
        CSLAM_TRY(pool_stage_queries_on(sh->pool, reinterpret_cast<const float*>(s.pack), dim, R * bm, s.staged,
                                        s.staged_rnq, ps, bm, sh->per_rank_bytes / 4));
    CSLAM_HIP_TRY(hipEventRecord(s.ev_pub, ps));
    s.b = b;
    s.step = sh->pub_step++;
    s.state = 1;
    return CSLAM_OK;
}

int cslam_shard_match(cslam_shard_t* sh, void* stream) {
    if (!sh) return fail(CSLAM_E_INVALID, "shard_match: null");
    if (sh->match_step >= sh->pub_step) return fail(CSLAM_E_INVALID, "shard_match: nothing published");
    Slot& s = sh->slot[sh->match_step % kSlots];
    if (s.state != 1) return fail(CSLAM_E_INVALID, "shard_match: slot not in published state");
    CSLAM_HIP_TRY(hipSetDevice(sh->device));
    hipStream_t st = (hipStream_t)stream;
    const int R = sh->R, me = sh->me, bm = sh->b_max, dim = sh->dim;
    const int M = R * bm;
    // the exchange was issued a whole step ago: normally it is complete, and then the compute
    // stream does not need the device-side wait (a cross-stream barrier packet in front of the
    // scan cost ~10 us of the 17-us gap between two steps, profiles/r02_shard_step_timeline.txt)
    if (hipEventQuery(s.ev_pub) != hipSuccess) CSLAM_HIP_TRY(hipStreamWaitEvent(st, s.ev_pub, 0));
    const int64_t n0 = s.n0, hi = s.n0 + s.b;   // the own block was appended by publish()
    // this step's searches read rows [0, hi) only, complete since ev_pub: they do not wait for the
    // appends of later steps that the exchange stream has already queued
    PoolInternalScope internal(sh->pool);
    // the dense copy of the gathered rows, for the paths that read their queries from all_rows (publish left it
    // out when it staged the block for the multi-tile scan)
    auto ensure_rows = [&]() -> int {
        if (s.rows_valid) return CSLAM_OK;
        const int64_t n4 = (int64_t)bm * dim / 4;
        unsigned gx = (unsigned)((n4 + 255) / 256);
        gx = gx < 1 ? 1 : (gx > 64 ? 64 : gx);
        hipLaunchKernelGGL(k_unpack_exchange, dim3(gx, R), dim3(256), 0, st, s.pack, sh->per_rank_bytes, sh->hdr_off, R,
                           bm, dim, s.all_rows, s.all_meta, 1);
        CSLAM_HIP_TRY(hipGetLastError());
        s.rows_valid = true;
        return CSLAM_OK;
    };
    cslam_match_record_t* my_recs = s.recs_all + (int64_t)me * sh->rec_slots;
    const unsigned gm = (unsigned)(((M > s.b ? M : s.b) + 255) / 256);
    if (R == 1 && sh->k_local == 0) {
        // a single robot without intra-robot matching: nothing to search, keyframe table only
        hipLaunchKernelGGL((k_make_records<false>), dim3(gm), dim3(256), 0, st, me, R, bm, s.all_meta,
                           n0, sh->kf_of_row, s.b, sh->thr, nullptr, nullptr, 1, 0, 0, s.idx_a, s.sim_a,
                           s.idx_c, s.sim_c, my_recs);
        CSLAM_TRY(pool_order_on(sh->pool, st));  // later pool calls order behind the append
    } else if (sh->best2) {
        // ONE scan launch: best row before n0 and best row of the new block per gathered
        // descriptor; its candidates are reduced by the record kernel itself
        BestCandidates bc;
        bool applied = false;
        const RecordArgs rargs{me, R, bm, s.all_meta, n0, sh->kf_of_row, s.b, sh->thr, my_recs};
        CSLAM_TRY(pool_best_split_candidates(sh->pool, s.all_rows, dim, M, n0, hi, st, &bc, &applied,
                                             s.staged_valid ? s.staged : nullptr,
                                             s.staged_valid ? s.staged_rnq : nullptr));
        if (applied) {
            hipLaunchKernelGGL(k_records_from_candidates, dim3(M), dim3(256), 0, st, rargs, bc.a, bc.c, bc.stride,
                               bc.n);
        } else {
            CSLAM_TRY(ensure_rows());
            CSLAM_TRY(cslam_pool_search_best_split_dev(sh->pool, s.all_rows, dim, M, n0, hi, s.idx_a,
                                                       s.sim_a, s.idx_c, s.sim_c, st));
            hipLaunchKernelGGL((k_make_records<false>), dim3(gm), dim3(256), 0, st, me, R, bm,
                               s.all_meta, n0, sh->kf_of_row, s.b, sh->thr, nullptr, nullptr, 1, 0, 0,
                               s.idx_a, s.sim_a, s.idx_c, s.sim_c, my_recs);
        }
    } else if (sh->fused) {
        // ONE scan of rows [0, n0 + b): the sorted top-K (K > b) of every gathered descriptor
        // holds both its best older row and its best row of the new block
        // (round 4) the scan leaves the parts' composites and the dots; rank select AND records in one launch
        // behind it.  Where that scan does not run for this shape (tiny pools, K > 32, several passes):
        // the pool's own search, then the records.  CSLAM_SHARD_FUSE_TOPK=0: always the latter (A/B).
        CSLAM_TRY(ensure_rows());
        static Knob k_fuse{"CSLAM_SHARD_FUSE_TOPK", 1};
        const bool fuse_env = knob(k_fuse) != 0;
        TopkCandidates tc;
        bool applied = false;
        if (fuse_env)
            CSLAM_TRY(pool_topk_candidates(sh->pool, s.all_rows, dim, M, sh->K, 0, hi, st, &tc, &applied));
        if (applied && tc.kind == 0) {
            const WaveParts parts{tc.wq, tc.wrem};
            hipLaunchKernelGGL((k_topk_records<WaveParts>), dim3(M), dim3(256), 0, st, tc.cand, tc.cand_stride,
                               tc.n_parts, parts, tc.dots, tc.dots_stride, tc.rnx, tc.rnq, tc.row0, tc.n, me, bm,
                               s.all_meta, n0, sh->kf_of_row, s.b, sh->thr, s.topk_idx, s.topk_sim, sh->K,
                               sh->k_local, sh->min_inbetween, my_recs);
        } else if (applied) {
            const PanelParts parts{tc.n_main, tc.n, tc.n_blocks, tc.p_max, tc.nw, tc.ns, tc.n_main_parts};
            hipLaunchKernelGGL((k_topk_records<PanelParts>), dim3(M), dim3(256), 0, st, tc.cand, tc.cand_stride,
                               tc.n_parts, parts, tc.dots, tc.dots_stride, tc.rnx, tc.rnq, tc.row0, tc.n, me, bm,
                               s.all_meta, n0, sh->kf_of_row, s.b, sh->thr, s.topk_idx, s.topk_sim, sh->K,
                               sh->k_local, sh->min_inbetween, my_recs);
        } else {
            CSLAM_TRY(cslam_pool_search_range_dev(sh->pool, s.all_rows, dim, M, sh->K, 0, hi, s.topk_idx,
                                                  s.topk_sim, st));
            hipLaunchKernelGGL((k_make_records<true>), dim3(gm), dim3(256), 0, st, me, R, bm, s.all_meta,
                               n0, sh->kf_of_row, s.b, sh->thr, s.topk_idx, s.topk_sim, sh->K, sh->k_local,
                               sh->min_inbetween, nullptr, nullptr, nullptr, nullptr, my_recs);
        }
    } else {
        // bulk blocks: best old row of every FOREIGN descriptor (the own block is skipped: two
        // calls around it), best new row through the row-range search
        CSLAM_TRY(ensure_rows());
        hipLaunchKernelGGL(k_fill_i32, dim3(gm), dim3(256), 0, st, s.idx_a, (int64_t)M, -1);
        hipLaunchKernelGGL(k_fill_i32, dim3(gm), dim3(256), 0, st, s.idx_c, (int64_t)M, -1);
        const int lo_n = me * bm, hi_first = (me + 1) * bm, hi_n = M - hi_first;
        if (lo_n > 0)
            CSLAM_TRY(cslam_pool_search_range_dev(sh->pool, s.all_rows, dim, lo_n, 1, 0, n0, s.idx_a,
                                                  s.sim_a, st));
        if (hi_n > 0)
            CSLAM_TRY(cslam_pool_search_range_dev(sh->pool, s.all_rows + (int64_t)hi_first * dim, dim,
                                                  hi_n, 1, 0, n0, s.idx_a + hi_first,
                                                  s.sim_a + hi_first, st));
        if (s.b > 0) {
            if (lo_n > 0)
                CSLAM_TRY(cslam_pool_search_range_dev(sh->pool, s.all_rows, dim, lo_n, 1, n0, hi,
                                                      s.idx_c, s.sim_c, st));
            if (hi_n > 0)
                CSLAM_TRY(cslam_pool_search_range_dev(sh->pool, s.all_rows + (int64_t)hi_first * dim,
                                                      dim, hi_n, 1, n0, hi, s.idx_c + hi_first,
                                                      s.sim_c + hi_first, st));
        }
        hipLaunchKernelGGL((k_make_records<false>), dim3(gm), dim3(256), 0, st, me, R, bm, s.all_meta,
                           n0, sh->kf_of_row, s.b, sh->thr, nullptr, nullptr, 1, 0, 0, s.idx_a, s.sim_a,
                           s.idx_c, s.sim_c, my_recs);
    }
    CSLAM_HIP_TRY(hipGetLastError());
    // record exchange + host copy on the record communicator's stream: off the critical path of
    // the next step's scan
    const size_t rec_bytes = (size_t)sh->rec_slots * sizeof(cslam_match_record_t);
    hipStream_t rs = R > 1 ? (hipStream_t)cslam_comm_stream(sh->comm_rec) : sh->s_own;
    CSLAM_HIP_TRY(hipEventRecord(s.ev_match, st));
    CSLAM_HIP_TRY(hipStreamWaitEvent(rs, s.ev_match, 0));
    if (R > 1)
        CSLAM_TRY(cslam_comm_allgather(sh->comm_rec, my_recs, s.recs_all, (int64_t)rec_bytes,
                                       CSLAM_COMM_OWN_STREAM));
    CSLAM_HIP_TRY(hipMemcpyAsync(s.recs_host, s.recs_all, rec_bytes * R, hipMemcpyDeviceToHost, rs));
    CSLAM_HIP_TRY(hipEventRecord(s.ev_done, rs));
    s.state = 2;
    sh->match_step++;
    return CSLAM_OK;
}

int cslam_shard_fetch(cslam_shard_t* sh, cslam_match_record_t* out, int64_t cap, int64_t* n_out,
                      int64_t* step_index) {
    if (!sh || !n_out || cap < 0 || (cap > 0 && !out)) return fail(CSLAM_E_INVALID, "shard_fetch: bad args");
    if (sh->fetch_step >= sh->match_step) return fail(CSLAM_E_INVALID, "shard_fetch: no matched step pending");
    Slot& s = sh->slot[sh->fetch_step % kSlots];
    CSLAM_HIP_TRY(hipSetDevice(sh->device));
    CSLAM_HIP_TRY(hipEventSynchronize(s.ev_done));
    int64_t n = 0;
    const int64_t total = (int64_t)sh->R * sh->rec_slots;
    for (int64_t i = 0; i < total; ++i) {
        if (s.recs_host[i].robot0_id < 0) continue;
        if (n < cap) out[n] = s.recs_host[i];
        ++n;
    }
    if (n > cap) return fail(CSLAM_E_INVALID, "shard_fetch: %lld records, room for %lld", (long long)n, (long long)cap);
    *n_out = n;
    if (step_index) *step_index = s.step;
    s.state = 0;
    sh->fetch_step++;
    return CSLAM_OK;
}

int64_t cslam_shard_max_records(const cslam_shard_t* sh) {
    return sh ? (int64_t)sh->R * sh->rec_slots : 0;
}

int cslam_shard_pending(const cslam_shard_t* sh, int64_t* published, int64_t* matched,
                        int64_t* fetched) {
    if (!sh) return fail(CSLAM_E_INVALID, "shard_pending: null");
    if (published) *published = sh->pub_step;
    if (matched) *matched = sh->match_step;
    if (fetched) *fetched = sh->fetch_step;
    return CSLAM_OK;
}

}  // extern "C"
