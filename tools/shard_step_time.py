"""Device-side cost of the sharded step (csrc/shard.hip) on ONE GPU: robot 0 of R with the
library's loopback transport (cslam_comm_init_loopback: one small kernel per all-gather fills the
other ranks' slots with copies of its own block; no real peer, no RCCL) — what append + one scan (top-(b+k)) + record building cost next to the plain
16-query search_best step.  Usage: python tools/shard_step_time.py [R] [steps] [nb_best_matches]
(nb_best_matches = 0, the default: inter-robot matching only → split best-match scan)"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cslam_amd import _lib  # noqa: E402
if os.environ.get("CSLAM_PROBE_LIB"):   # another build of the library (A/B of two builds in one call)
    _lib.LIB_PATH = os.environ["CSLAM_PROBE_LIB"]
from cslam_amd.multi_robot import DeviceShard, LoopbackComm  # noqa: E402
from cslam_amd.nns_matching import NearestNeighborsMatching  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
nb_best = int(sys.argv[3]) if len(sys.argv) > 3 else 0
N, D = int(os.environ.get("SHARD_ROWS", 100_000)), 4096
per_rank = int(os.environ.get("SHARD_PER_RANK", 16 // R))   # cfg4: SHARD_ROWS=25000 SHARD_PER_RANK=10, R = 8
L = _lib.load()
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
X = torch.rand((N, D), device=dev, generator=g)
pool = NearestNeighborsMatching(dim=D, capacity_hint=N + (steps + 40) * per_rank)
st = torch.cuda.current_stream().cuda_stream
pool.add_items_device(X.data_ptr(), N, D, items=None, stream=st)


cd = LoopbackComm(0, R, 0)
cr = cd if os.environ.get("SHARD_ONE_COMM", "1") == "1" else LoopbackComm(0, R, 0)
sh = DeviceShard(pool, 0, R, per_rank, 0.0, cd, cr, nb_best_matches=nb_best, intra_min_inbetween=10)
blocks = torch.rand((8, per_rank, D), device=dev, generator=g)
kf = [N]
pub = [0]


# new descriptors come from the front end's stream (VPR), not from the matching stream
front = torch.cuda.Stream()
pst = front.cuda_stream if os.environ.get("SHARD_PUBLISH_SAME_STREAM") != "1" else st


def publish():
    sh.publish(blocks[pub[0] % 8], list(range(kf[0], kf[0] + per_rank)), pst)
    kf[0] += per_rank
    pub[0] += 1


host = [0.0]


def run(n):
    matched = fetched = recs = 0
    publish()
    for t in range(n):
        h0 = time.perf_counter()
        publish()
        sh.match(st)
        host[0] += time.perf_counter() - h0      # host time of enqueueing one step (no waiting)
        matched += 1
        if matched - fetched > 1:
            recs += len(sh.fetch_records())
            fetched += 1
    sh.match(st)            # the block published last
    matched += 1
    while fetched < matched:
        recs += len(sh.fetch_records())
        fetched += 1
    torch.cuda.synchronize()
    return recs


run(10)
host[0] = 0.0
t0 = time.perf_counter()
recs = run(steps)
dt = (time.perf_counter() - t0) / (steps + 1)
print(f"R={R} per_rank={per_rank} nb_best_matches={nb_best}: {dt * 1e3:.4f} ms per step ({host[0] / steps * 1e3:.4f} ms of host enqueue time), "
      f"{recs / (steps + 1):.1f} records per step, kernel {L.cslam_pool_scan_kernel(pool.handle).decode()}")

q = torch.rand((16, D), device=dev, generator=g)
oi = torch.empty((16, 1), dtype=torch.int32, device=dev)
os_ = torch.empty((16, 1), dtype=torch.float32, device=dev)
for _ in range(5):
    _lib.check(L.cslam_pool_search_dev(pool.handle, ctypes.c_void_p(q.data_ptr()), D, 16, 1,
                                       ctypes.c_void_p(oi.data_ptr()), ctypes.c_void_p(os_.data_ptr()),
                                       ctypes.c_void_p(st)))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    _lib.check(L.cslam_pool_search_dev(pool.handle, ctypes.c_void_p(q.data_ptr()), D, 16, 1,
                                       ctypes.c_void_p(oi.data_ptr()), ctypes.c_void_p(os_.data_ptr()),
                                       ctypes.c_void_p(st)))
torch.cuda.synchronize()
print(f"plain search_best step on the same pool: {(time.perf_counter() - t0) / steps * 1e3:.4f} ms")
