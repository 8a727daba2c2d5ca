"""search_best (16 queries, device API) on a ~1.6 GB pool at every descriptor dimension the VPR
heads of the reference produce (64…4096): which scan kernel runs and what read rate it reaches."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cslam_amd import _lib
from cslam_amd.nns_matching import NearestNeighborsMatching
L = _lib.load()
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
st = torch.cuda.current_stream().cuda_stream
BYTES = int(sys.argv[1]) if len(sys.argv) > 1 else 1_638_400_000
DIMS = [int(a) for a in sys.argv[2:]] or [64, 128, 192, 256, 384, 512, 768, 1024, 2048, 4096, 8192]
for D in DIMS:
    n = BYTES // (D * 4)
    X = torch.rand((n, D), device=dev, generator=g)
    pool = NearestNeighborsMatching(dim=D, capacity_hint=n)
    pool.add_items_device(X.data_ptr(), n, D, items=None, stream=st)
    del X
    q = torch.rand((16, D), device=dev, generator=g)
    oi = torch.empty((16, 1), dtype=torch.int32, device=dev)
    os_ = torch.empty((16, 1), dtype=torch.float32, device=dev)
    def run():
        _lib.check(L.cslam_pool_search_dev(pool.handle, ctypes.c_void_p(q.data_ptr()), D, 16, 1,
                                           ctypes.c_void_p(oi.data_ptr()), ctypes.c_void_p(os_.data_ptr()),
                                           ctypes.c_void_p(st)))
    for _ in range(60): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): run()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
    print(f"D={D:5d} n={n:8d}  {dt * 1e3:.4f} ms  {n * D * 4 / dt / 1e12:.2f} TB/s  {16 / dt / 1e3:.1f} k candidates/s  "
          f"{L.cslam_pool_scan_kernel(pool.handle).decode()}", flush=True)
    pool.close()
    torch.cuda.empty_cache()
