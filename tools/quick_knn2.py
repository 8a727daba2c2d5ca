"""scan timing only, for layout experiments (env CSLAM_ROW_PAD)."""
import ctypes, sys, os
sys.path.insert(0, ".")
from cslam_amd import _lib
from cslam_amd.nns_matching import NearestNeighborsMatching
import torch
L = _lib.load()
N, D = 100000, 4096
g = torch.Generator(device="cuda").manual_seed(0)
X = torch.rand((N, D), device="cuda", generator=g)
m = NearestNeighborsMatching(dim=D, capacity_hint=N)
_lib.check(L.cslam_pool_add_dev(m.handle, ctypes.c_void_p(X.data_ptr()), N, D, None))
torch.cuda.synchronize(); m.n = N
for M in (16,):
    Q = torch.rand((M, D), device="cuda", generator=g)
    idx = torch.empty((M, 10), dtype=torch.int32, device="cuda"); sim = torch.empty((M, 10), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    def run():
        _lib.check(L.cslam_pool_search_dev(m.handle, ctypes.c_void_p(Q.data_ptr()), D, M, 10, ctypes.c_void_p(idx.data_ptr()), ctypes.c_void_p(sim.data_ptr()), ctypes.c_void_p(st)))
    for _ in range(3): run()
    torch.cuda.synchronize()
    _lib.check(L.cslam_pool_profile_begin(m.handle, 64))
    for _ in range(20): run()
    torch.cuda.synchronize()
    ms = ctypes.c_double(); nl = ctypes.c_int()
    _lib.check(L.cslam_pool_profile_end(m.handle, ctypes.byref(ms), ctypes.byref(nl)))
    print(f"ROW_PAD={os.environ.get('CSLAM_ROW_PAD','0')} M={M} scan {ms.value:.4f} ms  {N*D*4/ms.value/1e6:.0f} GB/s", flush=True)
