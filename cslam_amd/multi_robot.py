"""Robot-per-rank sharding of the inter-robot matching (BASELINE.json configs[3]).

One process per GPU; rank r holds robot r's descriptor pool and NOTHING of the other robots'
(the reference mirrors every peer pool on every robot, loop_closure_sparse_matching.py:26-32).
The ROS fan-out of `GlobalDescriptors` messages (global_descriptor_loop_closure_detection.py:
200-209) becomes one all-gather of the step's new descriptors over RCCL/xGMI; the
`InterRobotMatches` exchange (gdlcd.py:223-283) becomes a second, tiny all-gather of 20-byte
match records.  There is no other collective: the scan itself never leaves the GPU.

Step protocol (equivalent to the reference replayed in the order "everybody receives the
step's descriptors, then everybody inserts its own" — tests/test_multi_robot.py proves the
union of candidate edges is identical):
  A. all-gather the new blocks (padded to the largest block) + keyframe ids + counts
  B. every rank matches the OTHER ranks' new descriptors against its pool as it was when the
     step began                      → the `add_other_robot_global_descriptor` edges (lcsm.py:58-76)
  C. every rank appends its own block, then matches the same descriptors against just the
     appended rows; where that beats (B) the peer's `add_local_global_descriptor`
     (lcsm.py:37-56) would have found this newer keyframe → extra edge emitted on the
     peer's behalf.  (Scanning only the new rows is why cslam_pool_search_range_dev exists.)
  D. all-gather the match records; every rank applies them in rank order to its replica of
     the candidate selector; the broker (lowest rank, neighbors_manager.py:48-64) selects.
"""
import ctypes

import numpy as np

from .algebraic_connectivity_maximization import (AlgebraicConnectivityMaximization,
                                                  EdgeInterRobot)

# r0, kf0, r1, kf1, similarity — the 20-byte record of gdlcd.py:276
RECORD_FIELDS = 5


class HipPoolMatcher(object):
    """Device pool behind the step protocol (queries and results stay in HBM)."""

    def __init__(self, dim, device_id=0, capacity_hint=0):
        import torch
        from . import _lib
        from .nns_matching import NearestNeighborsMatching
        self._torch = torch
        self._lib = _lib
        self.pool = NearestNeighborsMatching(dim=dim, device_id=device_id,
                                             capacity_hint=capacity_hint)
        self.dim = dim
        self.device = torch.device("cuda", device_id)
        self.keyframes = []

    @property
    def size(self):
        return self.pool.n

    def add_rows(self, block, keyframe_ids):
        """block: CUDA float32 tensor [b, dim] (contiguous)."""
        if block.shape[0] == 0:
            return
        st = self._torch.cuda.current_stream().cuda_stream
        self.pool.add_items_device(block.data_ptr(), block.shape[0], block.stride(0),
                                   list(keyframe_ids), stream=st)
        self.keyframes.extend(int(k) for k in keyframe_ids)

    def search_best(self, queries, row_lo, row_hi):
        """Best match of every query among pool rows [row_lo,row_hi) → (rows int32[M] with -1
        for an empty range, sims float32[M]) as numpy."""
        torch = self._torch
        m = queries.shape[0]
        if m == 0 or row_hi <= row_lo:
            return np.full(m, -1, np.int32), np.full(m, -np.inf, np.float32)
        idx = torch.empty((m, 1), dtype=torch.int32, device=self.device)
        sim = torch.empty((m, 1), dtype=torch.float32, device=self.device)
        st = torch.cuda.current_stream().cuda_stream
        L = self._lib.load()
        self._lib.check(L.cslam_pool_search_range_dev(
            self.pool.handle, ctypes.c_void_p(queries.data_ptr()), queries.stride(0), m, 1,
            int(row_lo), int(row_hi), ctypes.c_void_p(idx.data_ptr()),
            ctypes.c_void_p(sim.data_ptr()), ctypes.c_void_p(st)))
        return idx[:, 0].cpu().numpy(), sim[:, 0].cpu().numpy()


class TorchDistComm(object):
    """all-gathers over torch.distributed (backend 'nccl' = RCCL over xGMI on GPUs, 'gloo' on
    CPU for the tests).  Works on whatever device the tensors live on."""

    def __init__(self, group=None):
        import torch
        import torch.distributed as dist
        self.torch = torch
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    def all_gather(self, t):
        """t: tensor [n, ...] with the same shape on every rank → [world, n, ...]."""
        if self.world == 1:
            return t.unsqueeze(0)
        shape = tuple(t.shape)
        out = self.torch.empty((self.world * shape[0],) + shape[1:], dtype=t.dtype,
                               device=t.device)
        self.dist.all_gather_into_tensor(out, t.contiguous(), group=self.group)
        return out.view((self.world,) + shape)


class ShardedLoopClosureMatching(object):
    """Rank-local half of the multi-robot matching step."""

    def __init__(self, comm, matcher, similarity_threshold, max_block=16, selector=None):
        """
        Args:
            comm: TorchDistComm (rank = robot id)
            matcher: HipPoolMatcher (or any object with size/add_rows/search_best — the CPU
                     tests plug the oracle in)
            similarity_threshold: frontend.similarity_threshold
            max_block: largest number of new descriptors a rank contributes per step
        """
        self.comm = comm
        self.matcher = matcher
        self.thr = float(similarity_threshold)
        self.max_block = int(max_block)
        self.robot_id = comm.rank
        self.nb_robots = comm.world
        self.candidate_selector = selector or AlgebraicConnectivityMaximization(
            self.robot_id, self.nb_robots)
        self.bytes_gathered = 0

    # -- helpers ---------------------------------------------------------------------------
    def _pad(self, block, kfs):
        torch = self.comm.torch
        b = block.shape[0]
        if b > self.max_block:
            raise ValueError(f"block of {b} descriptors exceeds max_block={self.max_block}")
        padded = torch.zeros((self.max_block, block.shape[1]), dtype=torch.float32,
                             device=block.device)
        padded[:b] = block
        meta = torch.full((self.max_block + 1,), -1, dtype=torch.int64, device=block.device)
        meta[0] = b
        if b:
            meta[1:b + 1] = torch.as_tensor(list(kfs), dtype=torch.int64, device=block.device)
        return padded, meta

    def step(self, new_block, keyframe_ids):
        """One swarm step.  new_block: float32 tensor [b, D] on the matcher's device.

        Returns the list of EdgeInterRobot records this rank produced (also applied, together
        with every other rank's, to `candidate_selector`)."""
        torch = self.comm.torch
        me = self.robot_id
        padded, meta = self._pad(new_block, keyframe_ids)
        blocks = self.comm.all_gather(padded)       # [R, max_block, D]   (A)
        metas = self.comm.all_gather(meta)          # [R, max_block + 1]
        self.bytes_gathered += blocks.numel() * 4 + metas.numel() * 8
        metas_h = metas.cpu().numpy()
        counts = metas_h[:, 0]

        rows_q, owner, kf_q = [], [], []
        for s in range(self.nb_robots):
            if s == me:
                continue
            c = int(counts[s])
            if c:
                rows_q.append(blocks[s, :c])
                owner.extend([s] * c)
                kf_q.extend(int(k) for k in metas_h[s, 1:c + 1])
        n0 = self.matcher.size
        records = []
        if rows_q:
            queries = torch.cat(rows_q, dim=0).contiguous()
            row_a, sim_a = self.matcher.search_best(queries, 0, n0)            # (B)
        self.matcher.add_rows(new_block, keyframe_ids)                          # (C)
        if rows_q:
            row_c, sim_c = self.matcher.search_best(queries, n0, self.matcher.size)
            kfs_local = self.matcher.keyframes
            for i in range(len(owner)):
                s, kf_s = owner[i], kf_q[i]
                if row_a[i] >= 0 and sim_a[i] >= self.thr:
                    # what this robot's add_other_robot_global_descriptor creates (lcsm.py:73)
                    records.append((me, kfs_local[row_a[i]], s, kf_s, float(sim_a[i])))
                if row_c[i] >= 0 and sim_c[i] >= self.thr and not (
                        row_a[i] >= 0 and sim_a[i] >= sim_c[i]):
                    # the peer's add_local_global_descriptor would see this newer keyframe
                    # (lcsm.py:52): emitted with the peer as robot0
                    records.append((s, kf_s, me, kfs_local[row_c[i]], float(sim_c[i])))

        self._exchange_and_apply(records, new_block.device)                     # (D)
        return [EdgeInterRobot(int(r[0]), int(r[1]), int(r[2]), int(r[3]), r[4]) for r in records]

    def _exchange_and_apply(self, records, device):
        torch = self.comm.torch
        cap = 2 * self.max_block * max(self.nb_robots - 1, 1)
        buf = torch.zeros((cap + 1, RECORD_FIELDS), dtype=torch.float64, device=device)
        buf[0, 0] = len(records)
        if records:
            buf[1:len(records) + 1] = torch.as_tensor(records, dtype=torch.float64, device=device)
        allrec = self.comm.all_gather(buf).cpu().numpy()
        self.bytes_gathered += allrec.size * 8
        for s in range(self.nb_robots):  # fixed rank order → identical replicas everywhere
            c = int(allrec[s, 0, 0])
            for r in allrec[s, 1:c + 1]:
                self.candidate_selector.add_match(
                    EdgeInterRobot(int(r[0]), int(r[1]), int(r[2]), int(r[3]),
                                   float(np.float32(r[4]))))

    def is_broker(self):
        """Lowest robot id in range is the broker (neighbors_manager.py:48-64)."""
        return self.robot_id == 0

    def select_candidates(self, budget, is_neighbor_in_range=None):
        """MAC selection on the broker replica (single GPU; the graph is small)."""
        if is_neighbor_in_range is None:
            is_neighbor_in_range = {r: True for r in range(self.nb_robots)}
        return self.candidate_selector.select_candidates(int(budget), dict(is_neighbor_in_range))
