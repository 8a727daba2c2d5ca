"""16 best-match queries (k_scan_dma) over pools of 5k..100k rows x 4096-d: kernel time by HIP events and
TB/s of the rows read, alternating CSLAM_SCAN_SPREAD (waves dealt round-robin over the CUs when the pool has fewer
strips than wave slots) off / on in one process."""
import ctypes, os, sys, time
os.environ["CSLAM_KNOBS_LIVE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cslam_amd import _lib
from cslam_amd.nns_matching import NearestNeighborsMatching
L = _lib.load(); D = 4096
g = torch.Generator(device="cuda").manual_seed(0)
for N in (4200, 5000, 7000, 10000, 14000, 16000, 20000, 25000, 40000, 60000, 100000):
    X = torch.rand((N, D), device="cuda", generator=g)
    m = NearestNeighborsMatching(dim=D, capacity_hint=N)
    m.add_items_device(X.data_ptr(), N, D, items=None, stream=None)
    Q = torch.rand((16, D), device="cuda", generator=g)
    idx = torch.empty((16, 1), dtype=torch.int32, device="cuda"); sim = torch.empty((16, 1), dtype=torch.float32, device="cuda")
    def run():
        _lib.check(L.cslam_pool_search_dev(m.handle, ctypes.c_void_p(Q.data_ptr()), D, 16, 1, ctypes.c_void_p(idx.data_ptr()), ctypes.c_void_p(sim.data_ptr()), None))
    for _ in range(30): run()
    out = []
    for rep in range(3):
        for early in ("0", "1"):
            os.environ["CSLAM_SCAN_SPREAD"] = early
            for _ in range(5): run()
            torch.cuda.synchronize()
            _lib.check(L.cslam_pool_profile_begin(m.handle, 40))
            t0 = time.perf_counter()
            for _ in range(40): run()
            torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 40
            ms = ctypes.c_double(); nl = ctypes.c_int()
            _lib.check(L.cslam_pool_profile_end(m.handle, ctypes.byref(ms), ctypes.byref(nl)))
            out.append(f"spread={early}: kernel {ms.value*1e3:.1f} us = {N*D*4/ms.value/1e9:.2f} TB/s, step {t*1e6:.1f} us")
    print(f"N={N}: " + " | ".join(out), flush=True)
    m.close(); del X
