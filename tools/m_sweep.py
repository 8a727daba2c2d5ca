"""Time of one search vs number of queries (scan T=1/2/4 variants vs the bulk GEMM path)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cslam_amd import _lib
from cslam_amd.nns_matching import NearestNeighborsMatching
import torch
L = _lib.load(); N, D = 100000, 4096
g = torch.Generator(device="cuda").manual_seed(0)
X = torch.rand((N, D), device="cuda", generator=g)
m = NearestNeighborsMatching(dim=D, capacity_hint=N)
_lib.check(L.cslam_pool_add_dev(m.handle, ctypes.c_void_p(X.data_ptr()), N, D, None)); torch.cuda.synchronize(); m.n = N
for M in (8, 16, 17, 32, 33, 48, 64, 65, 80, 95, 96, 128, 192, 256):
    Q = torch.rand((M, D), device="cuda", generator=g)
    idx = torch.empty((M, 1), dtype=torch.int32, device="cuda"); sim = torch.empty((M, 1), dtype=torch.float32, device="cuda")
    def run():
        _lib.check(L.cslam_pool_search_dev(m.handle, ctypes.c_void_p(Q.data_ptr()), D, M, 1, ctypes.c_void_p(idx.data_ptr()), ctypes.c_void_p(sim.data_ptr()), None))
    for _ in range(3): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): run()
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 10
    print(f"M={M:4d}: {t*1e3:.3f} ms  {t*1e6/M:.1f} us/query", flush=True)
