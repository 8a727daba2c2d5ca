"""Robot-per-rank sharding of the inter-robot matching (BASELINE.json configs[3]).

One process per GPU; rank r holds robot r's descriptor pool and NOTHING of the other robots'
(the reference mirrors every peer pool on every robot, loop_closure_sparse_matching.py:26-32).
The ROS fan-out of `GlobalDescriptors` messages (global_descriptor_loop_closure_detection.py:
200-209) becomes one all-gather of the step's new descriptors over RCCL/xGMI; the
`InterRobotMatches` exchange (gdlcd.py:223-283) becomes a second, tiny all-gather of 20-byte
match records.  There is no other collective: the scan itself never leaves the GPU.

Two implementations of the same protocol live here:
  * `DeviceShard` — the product path: the whole step is enqueued on the GPU through the C ABI
    (`cslam_shard_*`, csrc/shard.hip), the collectives are RCCL calls made inside the library
    (`cslam_comm_*`, csrc/comm.hip); the host only publishes blocks and fetches 20-byte records.
  * `ShardedLoopClosureMatching` — the protocol written out on the host with a pluggable matcher
    and `torch.distributed` collectives: what the CPU tests (gloo, world 2/3) run with the oracle
    as the matcher, and the yardstick the device path is compared with on the GPU.

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


# ------------------------------------------------------------------------------------------------
# Product path: communicators and the device-side step behind the C ABI
# ------------------------------------------------------------------------------------------------
RECORD_DTYPE = np.dtype([("robot0_id", np.int32), ("robot0_keyframe_id", np.int32),
                         ("robot1_id", np.int32), ("robot1_keyframe_id", np.int32),
                         ("weight", np.float32)])   # cslam_match_record_t, 20 bytes


class RcclComm(object):
    """cslam_comm_t over RCCL (xGMI between the GPUs of a node).  Rank 0 calls `unique_id()` and
    hands the 128 bytes to the other ranks out of band (bench.py: torch.distributed store)."""

    def __init__(self, rank, nranks, unique_id, device_id):
        from . import _lib
        self._lib = _lib
        self._L = _lib.load()
        if len(unique_id) != _lib.CSLAM_COMM_ID_BYTES:
            raise ValueError("unique_id must be %d bytes" % _lib.CSLAM_COMM_ID_BYTES)
        self._id = (ctypes.c_uint8 * _lib.CSLAM_COMM_ID_BYTES).from_buffer_copy(bytes(unique_id))
        h = ctypes.c_void_p()
        _lib.check(self._L.cslam_comm_init(int(rank), int(nranks), self._id, int(device_id),
                                           ctypes.byref(h)))
        self.handle = h
        self.rank, self.world = int(rank), int(nranks)
        self.transport = "rccl"

    @staticmethod
    def unique_id():
        from . import _lib
        buf = (ctypes.c_uint8 * _lib.CSLAM_COMM_ID_BYTES)()
        _lib.check(_lib.load().cslam_comm_unique_id(buf))
        return bytes(buf)

    def size(self):
        """Ranks the communicator itself counts (cslam_comm_size: what RCCL was initialised with)."""
        return int(self._L.cslam_comm_size(self.handle))

    def stats(self):
        b, n, v = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int32()
        self._lib.check(self._L.cslam_comm_stats(self.handle, ctypes.byref(b), ctypes.byref(n),
                                                 ctypes.byref(v)))
        return {"bytes_gathered": b.value, "collectives": n.value, "rccl_version": v.value}

    def close(self):
        if getattr(self, "handle", None):
            self._L.cslam_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CallbackComm(RcclComm):
    """cslam_comm_t over a transport written in Python: `transport(send_ptr, recv_ptr,
    bytes_per_rank, stream)` must leave every rank's bytes in `recv_ptr` (device memory)."""

    def __init__(self, rank, nranks, device_id, transport, name="callback"):
        from . import _lib
        self._lib = _lib
        self._L = _lib.load()

        def _cb(user, send, recv, nbytes, stream):
            try:
                transport(send, recv, int(nbytes), stream)
                return 0
            except Exception as e:  # surfaces as CSLAM_E_COMM in the caller
                self.last_exception = e
                return 1

        self._cb = _lib.ALLGATHER_FN(_cb)   # keep the thunk alive as long as the communicator
        self.last_exception = None
        h = ctypes.c_void_p()
        _lib.check(self._L.cslam_comm_init_custom(int(rank), int(nranks), int(device_id), self._cb,
                                                  None, ctypes.byref(h)))
        self.handle = h
        self.rank, self.world = int(rank), int(nranks)
        self.transport = name


class LoopbackComm(RcclComm):
    """cslam_comm_t whose all-gather replicates the caller's block into every slot with one kernel
    (cslam_comm_init_loopback): the compute side of the sharded step on ONE GPU, no peers."""

    def __init__(self, rank, nranks, device_id):
        from . import _lib
        self._lib = _lib
        self._L = _lib.load()
        h = ctypes.c_void_p()
        _lib.check(self._L.cslam_comm_init_loopback(int(rank), int(nranks), int(device_id),
                                                    ctypes.byref(h)))
        self.handle = h
        self.rank, self.world = int(rank), int(nranks)
        self.transport = "loopback (one GPU, no peers)"


class LocalGroup(object):
    """Transport between robot shards that are THREADS of one process sharing one GPU (tests,
    or several small robots per GPU): device-to-device copies around a thread barrier."""

    def __init__(self, world):
        import threading
        self.world = world
        self.barrier = threading.Barrier(world)
        self.send = [None] * world

    def transport(self, rank):
        from . import _lib
        L = _lib.load()

        def fn(send, recv, nbytes, stream):
            _lib.check(L.cslam_device_copy(None, None, 0, 2, stream, 1))   # producer work done
            self.send[rank] = send
            self.barrier.wait()
            for r in range(self.world):
                if self.send[r] != recv + r * nbytes:                       # in-place own block
                    _lib.check(L.cslam_device_copy(recv + r * nbytes, self.send[r], nbytes, 2,
                                                   stream, 0))
            _lib.check(L.cslam_device_copy(None, None, 0, 2, stream, 1))
            self.barrier.wait()                                             # send buffers free again
        return fn


def torch_dist_transport(group=None):
    """Transport through host memory and torch.distributed (gloo): bench.py's rehearsal of
    `--gpus N` on a one-GPU box, where RCCL cannot place two ranks on one device."""
    import torch
    import torch.distributed as dist
    from . import _lib
    L = _lib.load()
    world = dist.get_world_size(group)

    def fn(send, recv, nbytes, stream):
        mine = torch.empty(nbytes, dtype=torch.uint8)
        _lib.check(L.cslam_device_copy(mine.data_ptr(), send, nbytes, 1, stream, 1))
        everybody = torch.empty(world * nbytes, dtype=torch.uint8)
        dist.all_gather_into_tensor(everybody, mine, group=group)
        _lib.check(L.cslam_device_copy(recv, everybody.data_ptr(), world * nbytes, 0, stream, 1))
    return fn


class DeviceShard(object):
    """Rank-local half of the sharded matching step on the device (csrc/shard.hip).

    publish(block, keyframe_ids) → match() → fetch_records(); `step()` does all three.  Several
    steps may be in flight (publish t+1 before match t: the exchange overlaps the scan)."""

    def __init__(self, pool, robot_id, nb_robots, max_block, similarity_threshold,
                 comm_desc=None, comm_rec=None, nb_best_matches=0, intra_min_inbetween=0,
                 selector=None):
        """
        Args:
            pool: NearestNeighborsMatching (device pool of THIS robot; dim fixed)
            comm_desc, comm_rec: RcclComm/CallbackComm for the descriptor and the record exchange
                                 (None for a single robot); the SAME communicator may serve both
                                 (one stream, one sequence of collectives per rank)
            nb_best_matches, intra_min_inbetween: frontend.nb_best_matches /
                frontend.intra_loop_min_inbetween_keyframes; 0 disables the intra-robot match
        """
        from . import _lib
        self._lib = _lib
        self._L = _lib.load()
        self.pool = pool
        self.robot_id, self.nb_robots = int(robot_id), int(nb_robots)
        self.max_block = int(max_block)
        self.comm_desc, self.comm_rec = comm_desc, comm_rec
        self.candidate_selector = selector or AlgebraicConnectivityMaximization(
            self.robot_id, self.nb_robots)
        h = ctypes.c_void_p()
        _lib.check(self._L.cslam_shard_create(
            pool.handle, comm_desc.handle if comm_desc else None,
            comm_rec.handle if comm_rec else None, self.robot_id, self.nb_robots, self.max_block,
            float(similarity_threshold), int(nb_best_matches), int(intra_min_inbetween),
            ctypes.byref(h)))
        self.handle = h
        self._cap = int(self._L.cslam_shard_max_records(h))
        self._buf = np.empty(self._cap, dtype=RECORD_DTYPE)
        if pool.n:
            kfs = np.asarray([pool.items[i] for i in range(pool.n)], dtype=np.int32)
            _lib.check(self._L.cslam_shard_set_keyframes(h, 0, _lib.ptr(kfs), len(kfs)))

    def set_keyframes(self, first, keyframe_ids):
        kfs = np.ascontiguousarray(keyframe_ids, dtype=np.int32)
        self._lib.check(self._L.cslam_shard_set_keyframes(self.handle, int(first),
                                                          self._lib.ptr(kfs), len(kfs)))

    def publish(self, block, keyframe_ids, stream=None):
        """block: CUDA float32 tensor [b, dim] (unit inner stride) or (device_ptr, row_stride)."""
        if isinstance(block, tuple):
            ptr, stride = block
            b = len(keyframe_ids)
        else:
            ptr, stride, b = block.data_ptr(), (block.stride(0) if block.shape[0] else self.pool.dim), block.shape[0]
            if b and (block.stride(1) != 1 or block.shape[1] != self.pool.dim):
                raise ValueError("publish expects float32 [b, dim] with unit inner stride")
        if b != len(keyframe_ids):
            raise ValueError("publish: len(keyframe_ids) != rows of the block")
        kfs = np.ascontiguousarray(keyframe_ids, dtype=np.int64)
        self._lib.check(self._L.cslam_shard_publish(self.handle, ctypes.c_void_p(ptr), int(stride),
                                                    self._lib.ptr(kfs) if b else None, int(b),
                                                    ctypes.c_void_p(stream or 0)))
        for kf in keyframe_ids:                     # the library appended the own block (publish time)
            self.pool.items[self.pool.n] = int(kf)
            self.pool.n += 1

    def match(self, stream=None):
        self._lib.check(self._L.cslam_shard_match(self.handle, ctypes.c_void_p(stream or 0)))

    def fetch_records(self):
        """All ranks' records of the oldest matched step as a structured array (rank order, then
        descriptor order; identical on every rank).  Blocks until that step is done."""
        n, step = ctypes.c_int64(), ctypes.c_int64()
        self._lib.check(self._L.cslam_shard_fetch(self.handle, self._lib.ptr(self._buf.view(np.uint8)),
                                                  self._cap, ctypes.byref(n), ctypes.byref(step)))
        return self._buf[:n.value].copy()

    def apply(self, records):
        """Feed fetched records to the selector replica; returns (the inter-robot edges of the step
        - EVERY rank's, in rank order then descriptor order: all replicas apply the same list -,
        this robot's intra-robot matches [(keyframe, matched keyframe, similarity)])."""
        mine, local = [], []
        for r in records:
            r0, k0, r1, k1, w = int(r[0]), int(r[1]), int(r[2]), int(r[3]), float(np.float32(r[4]))
            if r0 == r1:
                if r0 == self.robot_id:
                    local.append((k0, k1, w))
                continue
            e = EdgeInterRobot(r0, k0, r1, k1, w)
            self.candidate_selector.add_match(e)
            mine.append(e)
        return mine, local

    def step(self, block, keyframe_ids, stream=None):
        self.publish(block, keyframe_ids, stream)
        self.match(stream)
        return self.apply(self.fetch_records())

    def is_broker(self):
        return self.robot_id == 0

    def select_candidates(self, budget, is_neighbor_in_range=None):
        if is_neighbor_in_range is None:
            is_neighbor_in_range = {r: True for r in range(self.nb_robots)}
        return self.candidate_selector.select_candidates(int(budget), dict(is_neighbor_in_range))

    def close(self):
        if getattr(self, "handle", None):
            self._L.cslam_shard_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
