"""search with k > 1 at 17+ queries, A/B (round 4): k_scan_mq<best> + dots + k_topk_from_parts (default)
against the dots pass + two select levels over every row (CSLAM_TOPK_WAVES=0).  One process, settings
alternating (CSLAM_KNOBS_LIVE), results compared bit for bit.  Step = one cslam_pool_search_dev call,
wall clock over back-to-back calls.
usage: topk_mq_ab.py [rows ...]      env MQ_M=80,48  MQ_K=26,10"""
import ctypes, os, sys, time
os.environ["CSLAM_KNOBS_LIVE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cslam_amd import _lib
from cslam_amd.nns_matching import NearestNeighborsMatching
L = _lib.load(); D = int(os.environ.get("MQ_D", "4096"))
MS = [int(x) for x in os.environ.get("MQ_M", "80,48").split(",")]
KS = [int(x) for x in os.environ.get("MQ_K", "26,10").split(",")]
sizes = [int(a) for a in sys.argv[1:]] or [100000, 25000]
SETTINGS = [("parts (default)", {}), ("dots + two-level select", {"CSLAM_TOPK_WAVES": "0"})]
g = torch.Generator(device="cuda").manual_seed(0)
for N in sizes:
    X = torch.rand((N, D), device="cuda", generator=g)
    m = NearestNeighborsMatching(dim=D, capacity_hint=N)
    m.add_items_device(X.data_ptr(), N, D, items=None, stream=None)
    for M in MS:
        Q = torch.rand((M, D), device="cuda", generator=g)
        for K in KS:
            idx = torch.empty((M, K), dtype=torch.int32, device="cuda"); sim = torch.empty((M, K), dtype=torch.float32, device="cuda")
            def run():
                _lib.check(L.cslam_pool_search_dev(m.handle, ctypes.c_void_p(Q.data_ptr()), D, M, K, ctypes.c_void_p(idx.data_ptr()), ctypes.c_void_p(sim.data_ptr()), None))
            ref = None
            res = {name: [] for name, _ in SETTINGS}
            for rep in range(3):
                for name, env in SETTINGS:
                    os.environ.pop("CSLAM_TOPK_WAVES", None)
                    os.environ.update(env)
                    for _ in range(10): run()
                    torch.cuda.synchronize()
                    reps = 30
                    t0 = time.perf_counter()
                    for _ in range(reps): run()
                    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / reps
                    out = (idx.cpu().numpy().tobytes(), sim.cpu().numpy().tobytes())
                    if ref is None: ref = out
                    res[name].append((t * 1e3, out == ref))
            os.environ.pop("CSLAM_TOPK_WAVES", None)
            for name, _ in SETTINGS:
                r = res[name]
                print(f"N={N} M={M:3d} k={K:2d} {name:26s} step " + " ".join(f"{a:.4f}" for a, _ in r) + " ms  "
                      f"bits {'same' if all(c for _, c in r) else 'DIFFER'}", flush=True)
    m.close(); del X
