"""First-contact GPU script: parity on one cfg2-shaped slice + raw scan timing."""
import ctypes, sys, time
import numpy as np
sys.path.insert(0, ".")
from cslam_amd import _lib
from cslam_amd.nns_matching import NearestNeighborsMatching
from oracle import knn as o
import torch
L = _lib.load()
N, D = int(sys.argv[1]) if len(sys.argv) > 1 else 100000, 4096
g = torch.Generator(device="cuda").manual_seed(0)
X = torch.rand((N, D), device="cuda", generator=g)
m = NearestNeighborsMatching(dim=D, capacity_hint=N)
_lib.check(L.cslam_pool_add_dev(m.handle, ctypes.c_void_p(X.data_ptr()), N, D, None))
torch.cuda.synchronize(); m.n = N
for M in (1, 16, 32, 64, 256):
    Q = torch.rand((M, D), device="cuda", generator=g)
    idx = torch.empty((M, 10), dtype=torch.int32, device="cuda")
    sim = torch.empty((M, 10), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    def run():
        _lib.check(L.cslam_pool_search_dev(m.handle, ctypes.c_void_p(Q.data_ptr()), D, M, 10,
                   ctypes.c_void_p(idx.data_ptr()), ctypes.c_void_p(sim.data_ptr()), ctypes.c_void_p(st)))
    for _ in range(3): run()
    torch.cuda.synchronize()
    _lib.check(L.cslam_pool_profile_begin(m.handle, 64))
    t = time.time(); 
    for _ in range(10): run()
    torch.cuda.synchronize(); dt = (time.time() - t) / 10
    ms = ctypes.c_double(); nl = ctypes.c_int()
    _lib.check(L.cslam_pool_profile_end(m.handle, ctypes.byref(ms), ctypes.byref(nl)))
    passes = (M + 63) // 64
    print(f"M={M:4d} search {dt*1e3:8.3f} ms  scan kernel {ms.value:8.4f} ms x{passes}  "
          f"scan GB/s {N*D*4/ms.value/1e6:8.1f}  cand/s {M/dt:10.0f}", flush=True)
# parity spot check vs oracle on the first 4096 rows
Xs = X[:4096].cpu().numpy(); Qs = torch.rand((16, D), device="cuda", generator=g).cpu().numpy()
m2 = NearestNeighborsMatching(); m2.add_items(Xs, list(range(4096)))
r, s = m2.search_batch(Qs, 10); oi, os_, _ = o.search(Xs, Qs, 10)
print("parity idx", np.array_equal(r, oi), "sims bitexact", s.tobytes() == os_.tobytes(), "maxdiff", np.abs(s-os_).max())
