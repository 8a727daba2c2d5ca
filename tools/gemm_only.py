import ctypes, sys
sys.path.insert(0, "/root/repo")
from cslam_amd import _lib
from cslam_amd.nns_matching import NearestNeighborsMatching
import torch
L = _lib.load(); N, D, M = 100000, 4096, 1024
g = torch.Generator(device="cuda").manual_seed(0)
X = torch.rand((N, D), device="cuda", generator=g)
m = NearestNeighborsMatching(dim=D, capacity_hint=N)
_lib.check(L.cslam_pool_add_dev(m.handle, ctypes.c_void_p(X.data_ptr()), N, D, None)); torch.cuda.synchronize(); m.n = N
Q = torch.rand((M, D), device="cuda", generator=g)
idx = torch.empty((M, 1), dtype=torch.int32, device="cuda"); sim = torch.empty((M, 1), dtype=torch.float32, device="cuda")
for _ in range(4):
    _lib.check(L.cslam_pool_search_dev(m.handle, ctypes.c_void_p(Q.data_ptr()), D, M, 1, ctypes.c_void_p(idx.data_ptr()), ctypes.c_void_p(sim.data_ptr()), None))
torch.cuda.synchronize()
