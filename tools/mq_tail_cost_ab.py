"""k_scan_mq tail plan: the K-split tail launch is taken when its modelled cost is below
CSLAM_MQ_TAIL_MAXCOST percent of a strip time (default 80).  80 queries over pools of 20k..60k rows x
4096-d, alternating thresholds in one process."""
import ctypes, os, sys, time
os.environ["CSLAM_KNOBS_LIVE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cslam_amd import _lib
from cslam_amd.nns_matching import NearestNeighborsMatching
L = _lib.load(); D = 4096
g = torch.Generator(device="cuda").manual_seed(0)
for N in (20000, 25000, 30000, 40000, 50000, 60000):
    X = torch.rand((N, D), device="cuda", generator=g)
    m = NearestNeighborsMatching(dim=D, capacity_hint=N)
    m.add_items_device(X.data_ptr(), N, D, items=None, stream=None)
    for M in (32, 80):
        Q = torch.rand((M, D), device="cuda", generator=g)
        idx = torch.empty((M, 1), dtype=torch.int32, device="cuda"); sim = torch.empty((M, 1), dtype=torch.float32, device="cuda")
        def run():
            _lib.check(L.cslam_pool_search_dev(m.handle, ctypes.c_void_p(Q.data_ptr()), D, M, 1, ctypes.c_void_p(idx.data_ptr()), ctypes.c_void_p(sim.data_ptr()), None))
        out = []
        for rep in range(2):
            for thr in ("80", "90", "100"):
                os.environ["CSLAM_MQ_TAIL_MAXCOST"] = thr
                for _ in range(10): run()
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(40): run()
                torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 40
                out.append(f"<{thr}%: {t*1e6:.1f} us")
        print(f"N={N} M={M}: " + " | ".join(out), flush=True)
    m.close(); del X
