"""Does the in-library HIP-event bracketing of the scan launch cost step time?  Same process,
alternating rounds of 200 search_best steps with and without cslam_pool_profile_begin."""
import ctypes, os, sys, time
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
def run(n):
    for _ in range(n):
        _lib.check(L.cslam_pool_search_dev(pool.handle, ctypes.c_void_p(q.data_ptr()), D, 16, 1,
                                           ctypes.c_void_p(oi.data_ptr()), ctypes.c_void_p(os_.data_ptr()),
                                           ctypes.c_void_p(st)))
run(100); torch.cuda.synchronize()
res = {True: [], False: []}
for rnd in range(6):
    for prof in (True, False):
        if prof:
            _lib.check(L.cslam_pool_profile_begin(pool.handle, 200))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        run(200)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
        if prof:
            ms, n = ctypes.c_double(), ctypes.c_int()
            _lib.check(L.cslam_pool_profile_end(pool.handle, ctypes.byref(ms), ctypes.byref(n)))
            res[prof].append((round(dt * 1e6, 2), round(ms.value * 1e3, 2)))
        else:
            res[prof].append(round(dt * 1e6, 2))
print("with events (step us, kernel us):", res[True])
print("without events (step us):      ", res[False])
