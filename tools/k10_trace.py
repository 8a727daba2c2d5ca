"""search(k = 10) of 16 queries on 100k x 4096, for rocprofv3 kernel traces (which kernels, how long)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cslam_amd import _lib
from cslam_amd.nns_matching import NearestNeighborsMatching
L = _lib.load()
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
N, D, K = 100_000, 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 10
st = torch.cuda.current_stream().cuda_stream
X = torch.rand((N, D), device=dev, generator=g)
pool = NearestNeighborsMatching(dim=D, capacity_hint=N)
pool.add_items_device(X.data_ptr(), N, D, items=None, stream=st)
q = torch.rand((16, D), device=dev, generator=g)
oi = torch.empty((16, K), dtype=torch.int32, device=dev)
os_ = torch.empty((16, K), dtype=torch.float32, device=dev)
def run():
    _lib.check(L.cslam_pool_search_dev(pool.handle, ctypes.c_void_p(q.data_ptr()), D, 16, K,
                                       ctypes.c_void_p(oi.data_ptr()), ctypes.c_void_p(os_.data_ptr()), ctypes.c_void_p(st)))
for _ in range(60): run()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): run()
torch.cuda.synchronize()
print(f"k={K}: {(time.perf_counter() - t0) / 200 * 1e3:.4f} ms per call", flush=True)
