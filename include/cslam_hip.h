/*
 * cslam_hip.h — C ABI of libcslam_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the inter-robot loop-closure hot path of nhahn/cslam:
 *   - descriptor pool + brute-force cosine kNN   (replaces cslam/nns_matching.py:5-78)
 *   - MAC candidate selection numerics           (replaces cslam/mac/mac.py:19-233,
 *                                                 cslam/mac/utils.py:44-122 and the networkx
 *                                                 `tracemin_lu` Fiedler solve called at mac.py:52-58)
 *
 * The reference has no FFI for this path (plain Python classes); the entry
 * points below are what a ctypes binding inside those classes would call.  INTEGRATION.md
 * shows the reference-side stub.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error (CSLAM_E_*); the message of the last
 *     error on the calling thread is returned by cslam_last_error().
 *   - `*_dev` arguments are device (HBM) pointers, everything else is host memory owned by
 *     the caller.  `stream` is a hipStream_t passed as void*; NULL is the default (null) stream,
 *     i.e. what `torch.cuda.current_stream().cuda_stream` is unless the caller changed it.
 *     Host-pointer entry points run on a private stream and return synchronised.
 *   - one caller thread per handle (the reference node is a single-threaded rclpy executor).
 *   - no torch types, no C++ types.
 */
#ifndef CSLAM_HIP_H
#define CSLAM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CSLAM_OK 0
#define CSLAM_E_INVALID (-1)      /* bad argument */
#define CSLAM_E_HIP (-2)          /* HIP runtime error (message has the hipError string) */
#define CSLAM_E_NODEVICE (-3)     /* no gfx950 device visible */
#define CSLAM_E_DISCONNECTED (-4) /* Laplacian of a disconnected graph: the reference's SuperLU
                                     raises "Factor is exactly singular" here (acm.py:455 catches it) */
#define CSLAM_E_NOCONV (-5)       /* eigen-solver did not reach the tolerance */
#define CSLAM_E_COMM (-6)         /* RCCL missing or a collective failed (message has the reason) */

const char* cslam_last_error(void);
/* ABI version of this header (bumped on any signature change). */
int cslam_abi_version(void);
/* Number of visible HIP devices (does not create a context). */
int cslam_device_count(void);

/* ------------------------------------------------------------------------------------------
 * Descriptor pool  — NearestNeighborsMatching state (nns_matching.py:9-21)
 * Rows are stored un-normalised, fp32, row-major, zero-padded to a multiple of 64 floats (16 for
 * dim <= 16) plus 128 never-read floats when that stride is a multiple of 4 KiB (see
 * cslam_pool_device_view); 1/max(||x||,1e-8) is cached per row at add time.
 * ------------------------------------------------------------------------------------------ */
typedef struct cslam_pool cslam_pool_t;

/* dim > 0; capacity_hint rows are reserved up front (0 = 1000 like nns_matching.py:21). */
int cslam_pool_create(int dim, int64_t capacity_hint, int device_id, cslam_pool_t** out);
int cslam_pool_destroy(cslam_pool_t* pool);

/* add_item (nns_matching.py:23-42), batched: append n_rows dense rows [n_rows][dim].  `rows` is
 * copied before the call returns (the caller may reuse it at once); small appends are applied on the
 * handle's stream without a host synchronisation — every later call on the handle is ordered behind
 * them, and an error of the append itself surfaces with the next synchronising call. */
int cslam_pool_add(cslam_pool_t* pool, const float* rows, int64_t n_rows);
/* same, rows already in HBM; row_stride in floats (>= dim). Async on `stream`. */
int cslam_pool_add_dev(cslam_pool_t* pool, const float* rows_dev, int64_t n_rows,
                       int64_t row_stride, void* stream);

int64_t cslam_pool_size(const cslam_pool_t* pool);     /* self.n */
int64_t cslam_pool_capacity(const cslam_pool_t* pool); /* len(self.data) */
int cslam_pool_dim(const cslam_pool_t* pool);
int cslam_pool_device(const cslam_pool_t* pool);       /* HIP device the pool lives on */
/* `.data[first:first+n]` → host [n][dim] (tests/test_sparse_matching.py:36,49,66). */
int cslam_pool_read_rows(const cslam_pool_t* pool, int64_t first, int64_t n, float* out);
/* cached 1/max(||row||,1e-8) for rows [first, first+n) → host. */
int cslam_pool_read_inv_norms(const cslam_pool_t* pool, int64_t first, int64_t n, float* out);
/* device address of the row storage and its stride in floats (for zero-copy consumers).
 * The stride is NOT the dimension in general: rows are zero-padded to a multiple of 64 floats and,
 * when that makes the stride a multiple of 4 KiB, followed by another 128 floats that are never
 * read — always index rows with the stride returned here. */
int cslam_pool_device_view(const cslam_pool_t* pool, const float** rows_dev, int64_t* row_stride);

/* search (nns_matching.py:44-63) for n_queries queries at once.
 *   queries  [n_queries][dim] fp32 (host)
 *   out_idx  [n_queries][k] int32 row numbers, best first; ties → lowest row first
 *   out_sim  [n_queries][k] fp32 cosine similarities
 *   out_count[n_queries]    number of valid results = min(k, size)
 * Entries past out_count are -1 / -inf.  Empty pool → all counts 0 (nns_matching.py:54-55).
 * Synchronous: the results are in out_* on return (small calls wait on a pinned word the last kernel
 * writes, larger ones on the pool's stream). */
int cslam_pool_search(cslam_pool_t* pool, const float* queries, int n_queries, int k,
                      int32_t* out_idx, float* out_sim, int32_t* out_count);
/* same with queries and outputs resident in HBM; asynchronous on `stream`.
 * query_stride in floats (>= dim). out_count may be NULL. */
int cslam_pool_search_dev(cslam_pool_t* pool, const float* queries_dev, int64_t query_stride,
                          int n_queries, int k, int32_t* out_idx_dev, float* out_sim_dev,
                          void* stream);
/* search restricted to pool rows [row_lo, row_hi) (returned indices are absolute row numbers).
 * Used by the sharded multi-robot step: "rows appended since the step started". */
int cslam_pool_search_range_dev(cslam_pool_t* pool, const float* queries_dev,
                                int64_t query_stride, int n_queries, int k, int64_t row_lo,
                                int64_t row_hi, int32_t* out_idx_dev, float* out_sim_dev,
                                void* stream);
/* Best match on both sides of a row boundary in ONE pass over the pool: for every query the best
 * row in [0, split_row) → idx_a / sim_a and the best row in [split_row, row_hi) → idx_c / sim_c
 * (-1 / -inf where a side is empty; row_hi < 0 = the pool size).  The sharded step's two questions — the best OLD row for
 * `add_other_robot_global_descriptor` (cslam/loop_closure_sparse_matching.py:58-76) and the best row
 * of the block appended in this step, which the peer's `add_local_global_descriptor` would have met
 * in its mirror (lcsm.py:37-56).  Results equal two cslam_pool_search_range_dev calls with k = 1. */
int cslam_pool_search_best_split_dev(cslam_pool_t* pool, const float* queries_dev,
                                     int64_t query_stride, int n_queries, int64_t split_row,
                                     int64_t row_hi, int32_t* idx_a_dev, float* sim_a_dev,
                                     int32_t* idx_c_dev, float* sim_c_dev, void* stream);
/* Similarities only: sims_dev[q * sims_stride + row] for row < size (all rows, unsorted). */
int cslam_pool_similarities_dev(cslam_pool_t* pool, const float* queries_dev,
                                int64_t query_stride, int n_queries, float* sims_dev,
                                int64_t sims_stride, void* stream);

/* Introspection used by bench.py to time the dominant kernel with HIP events on the launch
 * stream: average duration in ms of the last `n` scan-kernel launches recorded since
 * cslam_pool_profile_begin().  */
int cslam_pool_profile_begin(cslam_pool_t* pool, int max_launches);
/* same, bracketing only every `every`-th launch (an event pair costs ~8 us of stream time). */
int cslam_pool_profile_begin_sampled(cslam_pool_t* pool, int max_samples, int every);
int cslam_pool_profile_end(cslam_pool_t* pool, double* avg_scan_ms, int* n_launches);
/* Ordering of the *_dev entry points against calls on OTHER streams (the host-pointer API runs on the pool's
 * own stream): an event behind the last *_dev call, which the next call on another stream waits for.  The
 * event is recorded LAZILY - by that next call, on the stream the *_dev work went to, i.e. behind everything
 * the caller has queued on that stream since.  Consequences for the caller: (1) the stream handed to a *_dev
 * call must stay alive until the pool's next call on any other stream (or cslam_pool_flush_dev /
 * cslam_pool_destroy); (2) unrelated work queued on that stream behind the *_dev call delays that next
 * call.  cslam_pool_flush_dev records the event NOW: call it behind the last *_dev call of a batch before
 * queueing unrelated work on the stream or destroying it.  No-op when nothing is pending. */
int cslam_pool_flush_dev(cslam_pool_t* pool);
/* Measurement knobs (CSLAM_* environment variables) are read once per process; after this
 * call every knob re-reads its variable at its next use.  For tests and A/B tools, not for the node. */
int cslam_knobs_reload(void);

/* Name (static string) of the dot-product kernel variant the last search launched, e.g.
 * "k_scan_dma<NS=8,NW=4,NST=2,nt>": what the roofline line of bench.py is about. */
const char* cslam_pool_scan_kernel(const cslam_pool_t* pool);

/* ------------------------------------------------------------------------------------------
 * Multi-robot exchange: rank = robot = GPU, RCCL over xGMI.
 * Replaces the ROS 2 fan-out of GlobalDescriptors (cslam/global_descriptor_loop_closure_
 * detection.py:196-209) and InterRobotMatches (:223-283) between robots that share a node.
 * RCCL is resolved with dlopen at the first call; CSLAM_E_COMM if it is missing.
 * ------------------------------------------------------------------------------------------ */
typedef struct cslam_comm cslam_comm_t;
#define CSLAM_COMM_ID_BYTES 128
/* pass as `stream` to run a collective on the communicator's own (non-blocking) stream */
#define CSLAM_COMM_OWN_STREAM ((void*)(intptr_t)-1)

/* Rank 0 makes the id (ncclGetUniqueId) and hands the 128 bytes to every rank out of band. */
int cslam_comm_unique_id(uint8_t* id_out);
/* Collective over all ranks: every process calls it with the same id and its own rank. */
int cslam_comm_init(int rank, int nranks, const uint8_t* unique_id, int device_id,
                    cslam_comm_t** out);
/* Same communicator object over a caller-supplied transport instead of RCCL: `allgather` must
 * leave rank r's bytes at recv_dev + r * bytes_per_rank on every rank before work enqueued on
 * `stream` after it runs (it may synchronise the stream).  Used by the tests (robot shards as
 * threads of one process on one GPU), by bench.py's one-GPU rehearsal (gloo through host memory)
 * and for fabrics other than xGMI (robots on different hosts).  Returns 0 on success. */
typedef int (*cslam_allgather_fn)(void* user, const void* send_dev, void* recv_dev,
                                  int64_t bytes_per_rank, void* stream);
int cslam_comm_init_custom(int rank, int nranks, int device_id, cslam_allgather_fn allgather,
                           void* user, cslam_comm_t** out);
/* Loopback transport (measurement on ONE GPU, tools/shard_step_time.py): every all-gather copies
 * the caller's block into all nranks slots with one small kernel — rank `rank` of `nranks` sees
 * its own descriptors as everybody's.  No peer, no RCCL: the compute side of the sharded step
 * with collectives of realistic launch count but no fabric. */
int cslam_comm_init_loopback(int rank, int nranks, int device_id, cslam_comm_t** out);
int cslam_comm_destroy(cslam_comm_t* comm);
int cslam_comm_rank(const cslam_comm_t* comm);
int cslam_comm_size(const cslam_comm_t* comm);
void* cslam_comm_stream(const cslam_comm_t* comm); /* hipStream_t of the communicator */
/* payload bytes received by all-gathers so far, number of collectives, RCCL version code */
int cslam_comm_stats(const cslam_comm_t* comm, int64_t* bytes_gathered, int64_t* n_collectives,
                     int32_t* rccl_version);
/* recv_dev[r * bytes_per_rank ...] = rank r's send_dev; in place when
 * send_dev == recv_dev + rank * bytes_per_rank.  Asynchronous on `stream`. */
int cslam_comm_allgather(cslam_comm_t* comm, const void* send_dev, void* recv_dev,
                         int64_t bytes_per_rank, void* stream);
/* The descriptor exchange of one step: this rank contributes b rows (row_stride floats apart);
 * all_dev [nranks][b_max][dim] receives every rank's block (rows past its count zeroed),
 * counts_dev [nranks] the counts. */
int cslam_allgather_descriptors(cslam_comm_t* comm, const float* local_dev, int64_t row_stride,
                                int32_t b, int32_t b_max, int32_t dim, float* all_dev,
                                int32_t* counts_dev, void* stream);

/* Plain copy helper for transports written in Python: kind 0 = host→device, 1 = device→host,
 * 2 = device→device; asynchronous on `stream` unless sync != 0. */
int cslam_device_copy(void* dst, const void* src, int64_t bytes, int kind, void* stream, int sync);

/* The 20-byte inter-robot match record of gdlcd.py:276 (InterRobotMatch message fields).
 * robot0_id == robot1_id marks an intra-robot match (LocalKeyframeMatch, gdlcd.py:145-168). */
typedef struct {
    int32_t robot0_id;
    int32_t robot0_keyframe_id;
    int32_t robot1_id;
    int32_t robot1_keyframe_id;
    float weight;
} cslam_match_record_t;

/* Rank-local half of the sharded matching step (csrc/shard.hip): robot `robot_id` owns `pool`
 * and nothing of the other robots' (the reference mirrors every peer pool,
 * loop_closure_sparse_matching.py:26-32).  comm_desc / comm_rec: the communicators of the descriptor
 * and of the record exchange over the same ranks - two (each exchange on a stream of its own), or
 * the SAME one passed twice (both on one stream: every rank then issues exactly one sequence of
 * collectives, publish(t+1), match(t), publish(t+2), ... - the form bench.py uses); both NULL for a
 * single robot.  Collectives per step: ONE all-gather of [max_block][dim] rows + (count, keyframe
 * ids) per rank at publish time, ONE all-gather of the match records at match time; every rank must
 * call publish / match in the same order.
 * nb_best_matches > 0 enables the intra-robot match of own keyframes (match_local_loop_closures,
 * lcsm.py:78-97) and needs max_block + nb_best_matches <= 64 (CSLAM_E_INVALID otherwise: bulk
 * blocks are matched inter-robot only, pass 0). */
typedef struct cslam_shard cslam_shard_t;
int cslam_shard_create(cslam_pool_t* pool, cslam_comm_t* comm_desc, cslam_comm_t* comm_rec,
                       int robot_id, int nb_robots, int max_block, double similarity_threshold,
                       int nb_best_matches, int intra_min_inbetween, cslam_shard_t** out);
int cslam_shard_destroy(cslam_shard_t* shard);
/* keyframe ids of pool rows [first, first+n) that were added outside the shard */
int cslam_shard_set_keyframes(cslam_shard_t* shard, int64_t first, const int32_t* keyframe_ids,
                              int64_t n);
/* Step part 1 — publish this robot's new block (b <= max_block rows in HBM, keyframe ids on
 * the host): the block is APPENDED TO THE POOL here (on the exchange stream, behind an event on
 * `stream`, the stream that produced the rows) and the all-gather of everybody's blocks is started.
 * Asynchronous.  Ordering contract: scans already in flight read rows below their own window only;
 * every later call on the pool (any stream) is ordered behind the append; a growth of the pool or
 * of the keyframe table drains the device first. */
int cslam_shard_publish(cslam_shard_t* shard, const float* new_rows_dev, int64_t row_stride,
                        const int64_t* keyframe_ids, int b, void* stream);
/* Step part 2 — for the oldest published step: scan the pool (rows before / of the block that
 * publish() appended) for all gathered descriptors, build the match records on the device, start
 * their all-gather and the copy to pinned host memory.  Asynchronous on `stream`. */
int cslam_shard_match(cslam_shard_t* shard, void* stream);
/* Step part 3 — records of ALL ranks for the oldest matched step, rank order then descriptor
 * order (identical on every rank).  Blocks until that step has finished on the device. */
int cslam_shard_fetch(cslam_shard_t* shard, cslam_match_record_t* out, int64_t cap, int64_t* n_out,
                      int64_t* step_index);
int64_t cslam_shard_max_records(const cslam_shard_t* shard);
int cslam_shard_pending(const cslam_shard_t* shard, int64_t* published, int64_t* matched,
                        int64_t* fetched);

/* ------------------------------------------------------------------------------------------
 * MAC problem  — cslam/mac/mac.py:19-233
 * Nodes 0..n-1; fixed edges (i,j,weight) form L_odom (mac.py:22); candidates (i,j,weight) are
 * weighted by w in [0,1]^m (mac.py:61-77).  All arithmetic fp64.
 * ------------------------------------------------------------------------------------------ */
typedef struct cslam_mac cslam_mac_t;

/* MAC.__init__ (mac.py:21-33) + weight_graph_lap_from_edge_list (utils.py:44-81). */
int cslam_mac_create(int32_t num_poses, const int32_t* fixed_i, const int32_t* fixed_j,
                     const double* fixed_w, int64_t n_fixed, const int32_t* cand_i,
                     const int32_t* cand_j, const double* cand_w, int64_t n_cand, int device_id,
                     cslam_mac_t** out);
int cslam_mac_destroy(cslam_mac_t* mac);

/* combined_laplacian(w, tol=1e-10) (mac.py:61-77) as CSR with duplicates summed
 * (utils.py:121-122).  Two-call protocol: indptr/indices/data may be NULL to query *nnz. */
int cslam_mac_laplacian(cslam_mac_t* mac, const double* w, double tol, int64_t* nnz,
                        int32_t* indptr, int32_t* indices, double* data);

/* evaluate_fiedler_pair(w) (mac.py:79-96 → networkx tracemin_lu, tol 1e-8).
 *   lambda2: Fiedler value; v[n]: unit Fiedler vector, sign fixed so that the first entry
 *   of largest magnitude is positive; iters: solver iterations used.
 * Returns CSLAM_E_DISCONNECTED if L(w) is the Laplacian of a disconnected graph. */
int cslam_mac_fiedler(cslam_mac_t* mac, const double* w, double tol, double* lambda2,
                      double* v, int32_t* iters);

/* grad_from_fiedler (mac.py:112-130): g[k] = weight_k (v_i - v_j)^2. */
int cslam_mac_grad(cslam_mac_t* mac, const double* v, double* g);

/* round_solution (mac.py:132-147): 0/1 mask of the k largest x; ties → lowest index kept. */
int cslam_topk_mask(const double* x, int64_t m, int64_t k, double* mask);
/* round_solution_tiebreaker (mac.py:168-189): k largest by (round(w,decimals), weight). */
int cslam_mac_round_tiebreak(cslam_mac_t* mac, const double* w, int64_t k, int decimals,
                             double* mask);

/* fw_subset (mac.py:191-233): whole Frank-Wolfe loop on the device.
 *   w_init[m] in, w_round[m] (0/1), w_relaxed[m], *upper out; *iters_done = FW iterations run;
 *   lambda_trace (may be NULL) receives f_i of every iteration (max_iters entries). */
int cslam_mac_fw_subset(cslam_mac_t* mac, const double* w_init, int64_t k, int max_iters,
                        double duality_gap_tol, double* w_round, double* w_relaxed,
                        double* upper, int32_t* iters_done, double* lambda_trace);

/* Diagnostics of fw_subset (mac.py:205-230), for parity tests and bench.py: with tracing enabled
 * every later cslam_mac_fw_subset call records, per evaluated iteration i, the direction s_i
 * (mac.py:215: the k indices round_solution picked from the gradient, ascending).  f_i (mac.py:211),
 * the dual bound u_i (mac.py:219) and the Fiedler solver's iteration count are always recorded.
 *   trace_read: it = iteration of the LAST fw_subset call; s_idx[k] may be NULL; *n_idx = k when
 *   directions were recorded, else 0. */
int cslam_mac_fw_trace_enable(cslam_mac_t* mac, int on);
int cslam_mac_fw_trace_read(cslam_mac_t* mac, int32_t it, int32_t* s_idx, int64_t* n_idx,
                            double* f, double* u, int32_t* solver_iters);

/* Solver statistics of the last cslam_mac_fiedler / cslam_mac_fw_subset call. */
int cslam_mac_stats(const cslam_mac_t* mac, int64_t* spmv_count, double* device_ms);

/* ------------------------------------------------------------------------------------------
 * Geometric verification stand-in.  The reference hands matched 3D keypoints
 * to rtabmap's RegistrationVis (src/front_end/rgbd_handler.cpp:684-706); this is a functional
 * replacement, parity unpinned by construction (third-party code outside the tree).
 * RANSAC over `iterations` 3-correspondence samples of src[n][3] -> dst[n][3] (host pointers):
 *   sample_transform[12]  rotation (row-major 3x3) then translation of the best sample
 *   inlier_sums[16]       over its inliers: sum src (3), sum dst (3), sum src_i*dst_j (9), count
 *                         — enough for the caller's least-squares (Kabsch) refinement
 *   inlier_mask[n]        optional, 1 = inlier of the best sample
 *   best_hypothesis       optional, index of the best sample (-1: none valid)
 * Deterministic for a given seed: most inliers wins, ties go to the lowest sample index. */
int cslam_ransac_rigid3d(const float* src, const float* dst, int64_t n, int iterations,
                         double inlier_distance, uint64_t seed, int device_id,
                         double* sample_transform, double* inlier_sums, uint8_t* inlier_mask,
                         int32_t* best_hypothesis);

#ifdef __cplusplus
}
#endif
#endif /* CSLAM_HIP_H */
