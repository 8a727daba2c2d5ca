"""Is the persistent Fiedler solver (64 of 256 CUs, latency-bound) sensitive to what the rest of the chip does?
Cold solve of the cfg3 graph (device ms) alone, and while another stream keeps the other CUs busy with bulk
1024-query searches (fp32 MFMA at ~125 TFLOP/s: power up, clocks down? or the power state up?).  A measurement,
not a product feature."""
import ctypes, json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cslam_amd import _lib
from cslam_amd.nns_matching import NearestNeighborsMatching
from mac_bench_util import fiedler_bench

L = _lib.load()


def solve():
    r = fiedler_bench()
    return r["fiedler_cold_device_ms"], r["fiedler_cold_iters"]


print("alone:", [solve() for _ in range(3)], flush=True)
N, D = 60000, 4096
g = torch.Generator(device="cuda").manual_seed(0)
X = torch.rand((N, D), device="cuda", generator=g)
pool = NearestNeighborsMatching(dim=D, capacity_hint=N)
pool.add_items_device(X.data_ptr(), N, D, items=None, stream=None)
Q = torch.rand((1024, D), device="cuda", generator=g)
idx = torch.empty((1024, 1), dtype=torch.int32, device="cuda"); sim = torch.empty((1024, 1), dtype=torch.float32, device="cuda")
side = torch.cuda.Stream()
stop = False


def heater():
    while not stop:
        for _ in range(4):
            _lib.check(L.cslam_pool_search_dev(pool.handle, ctypes.c_void_p(Q.data_ptr()), D, 1024, 1,
                                               ctypes.c_void_p(idx.data_ptr()), ctypes.c_void_p(sim.data_ptr()),
                                               ctypes.c_void_p(side.cuda_stream)))
        side.synchronize()


th = threading.Thread(target=heater); th.start()
time.sleep(0.5)
print("beside bulk GEMM searches on another stream:", [solve() for _ in range(3)], flush=True)
stop = True; th.join()
time.sleep(0.5)
print("alone again:", [solve() for _ in range(3)], flush=True)
