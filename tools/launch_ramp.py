"""How the scan kernel's duration settles over the first launches of a process (HIP events per
chunk of 10 launches): decides how many launches a run needs before it shows steady state."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cslam_amd import _lib
from cslam_amd.nns_matching import NearestNeighborsMatching
L = _lib.load()
N, D = 100_000, 4096
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
X = torch.rand((N, D), device=dev, generator=g)
pool = NearestNeighborsMatching(dim=D, capacity_hint=N)
st = torch.cuda.current_stream().cuda_stream
pool.add_items_device(X.data_ptr(), N, D, items=None, stream=st)
q = torch.rand((16, D), device=dev, generator=g)
oi = torch.empty((16, 1), dtype=torch.int32, device=dev)
os_ = torch.empty((16, 1), dtype=torch.float32, device=dev)
torch.cuda.synchronize()
out = []
for chunk in range(30):
    _lib.check(L.cslam_pool_profile_begin(pool.handle, 10))
    for _ in range(10):
        _lib.check(L.cslam_pool_search_dev(pool.handle, ctypes.c_void_p(q.data_ptr()), D, 16, 1,
                                           ctypes.c_void_p(oi.data_ptr()), ctypes.c_void_p(os_.data_ptr()),
                                           ctypes.c_void_p(st)))
    torch.cuda.synchronize()
    ms, n = ctypes.c_double(), ctypes.c_int()
    _lib.check(L.cslam_pool_profile_end(pool.handle, ctypes.byref(ms), ctypes.byref(n)))
    out.append(round(ms.value * 1e3, 1))
print("avg kernel us per chunk of 10 launches:", out)
