"""search (k = 1) for mid-size query batches against 100k x 4096: which dots kernel runs, ms."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cslam_amd import _lib
from cslam_amd.nns_matching import NearestNeighborsMatching
L = _lib.load(); N, D = 100000, 4096
g = torch.Generator(device="cuda").manual_seed(0)
X = torch.rand((N, D), device="cuda", generator=g)
m = NearestNeighborsMatching(dim=D, capacity_hint=N)
m.add_items_device(X.data_ptr(), N, D, items=None, stream=None)
for M in (32, 33, 48, 64, 80, 96, 128, 160, 192, 256):
    Q = torch.rand((M, D), device="cuda", generator=g)
    idx = torch.empty((M, 1), dtype=torch.int32, device="cuda"); sim = torch.empty((M, 1), dtype=torch.float32, device="cuda")
    def run():
        _lib.check(L.cslam_pool_search_dev(m.handle, ctypes.c_void_p(Q.data_ptr()), D, M, 1, ctypes.c_void_p(idx.data_ptr()), ctypes.c_void_p(sim.data_ptr()), None))
    for _ in range(3): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): run()
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 10
    print(f"M={M}: {t*1e3:.3f} ms  {2.0*N*D*M/t/1e12:.1f} TFLOP/s  {L.cslam_pool_scan_kernel(m.handle).decode()}", flush=True)
