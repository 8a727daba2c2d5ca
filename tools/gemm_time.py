"""Time the bulk (>= 96 queries) path: 1024 queries against 100k x 4096."""
import ctypes, sys, time
sys.path.insert(0, "/root/repo")
from cslam_amd import _lib
from cslam_amd.nns_matching import NearestNeighborsMatching
import torch
L = _lib.load(); N, D = 100000, 4096
g = torch.Generator(device="cuda").manual_seed(0)
X = torch.rand((N, D), device="cuda", generator=g)
m = NearestNeighborsMatching(dim=D, capacity_hint=N)
_lib.check(L.cslam_pool_add_dev(m.handle, ctypes.c_void_p(X.data_ptr()), N, D, None)); torch.cuda.synchronize(); m.n = N
for M in (256, 1024):
    Q = torch.rand((M, D), device="cuda", generator=g)
    idx = torch.empty((M, 1), dtype=torch.int32, device="cuda"); sim = torch.empty((M, 1), dtype=torch.float32, device="cuda")
    def run():
        _lib.check(L.cslam_pool_search_dev(m.handle, ctypes.c_void_p(Q.data_ptr()), D, M, 1, ctypes.c_void_p(idx.data_ptr()), ctypes.c_void_p(sim.data_ptr()), None))
    for _ in range(2): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(4): run()
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 4
    print(f"M={M}: {t*1e3:.3f} ms  {2.0*N*D*M/t/1e12:.1f} TFLOP/s", flush=True)
