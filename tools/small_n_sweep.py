"""search_best (16 queries, device API) across pool sizes: where should the LDS-DMA scan take over
from the register-ring scan?  Run once per CSLAM_SCAN_KERNEL setting (the knob is read once)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cslam_amd import _lib
from cslam_amd.nns_matching import NearestNeighborsMatching
L = _lib.load()
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
st = torch.cuda.current_stream().cuda_stream
for D in (4096, 512):
    out = []
    for n in (64, 256, 1000, 2000, 4000, 8000, 16000, 32000):
        X = torch.rand((n, D), device=dev, generator=g)
        pool = NearestNeighborsMatching(dim=D, capacity_hint=n)
        pool.add_items_device(X.data_ptr(), n, D, items=None, stream=st)
        q = torch.rand((16, D), device=dev, generator=g)
        oi = torch.empty((16, 1), dtype=torch.int32, device=dev)
        os_ = torch.empty((16, 1), dtype=torch.float32, device=dev)
        def run(k=1):
            _lib.check(L.cslam_pool_search_dev(pool.handle, ctypes.c_void_p(q.data_ptr()), D, 16, k,
                                               ctypes.c_void_p(oi.data_ptr()), ctypes.c_void_p(os_.data_ptr()),
                                               ctypes.c_void_p(st)))
        for _ in range(10): run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100): run()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
        out.append((n, round(dt * 1e6, 1), L.cslam_pool_scan_kernel(pool.handle).decode()[:12]))
        pool.close()
    print(f"D={D} kernel={os.environ.get('CSLAM_SCAN_KERNEL', 'default')}:", out, flush=True)
