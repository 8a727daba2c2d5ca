"""Per-call latency of the reference-style host API (the drop-in usage: add_item + search_best
per keyframe, cslam/loop_closure_sparse_matching.py:45-49) on small and large pools."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cslam_amd.nns_matching import NearestNeighborsMatching
rs = np.random.RandomState(0)
for n, d in ((1000, 128), (1000, 4096), (20000, 4096)):
    x = rs.rand(n + 300, d).astype(np.float32)
    m = NearestNeighborsMatching()
    m.add_items(x[:n], list(range(n)))
    for i in range(20):
        m.add_item(x[n + i], n + i)
        m.search_best(x[n + i])
    t0 = time.perf_counter()
    for i in range(20, 120):
        m.add_item(x[n + i], n + i)
    ta = (time.perf_counter() - t0) / 100
    t0 = time.perf_counter()
    for i in range(100):
        m.search_best(x[n + 120 + i])
    ts = (time.perf_counter() - t0) / 100
    t0 = time.perf_counter()
    for i in range(100):
        m.search(x[n + 120 + i], 10)
    tk = (time.perf_counter() - t0) / 100
    print(f"pool {n} x {d}: add_item {ta * 1e6:.1f} us, search_best {ts * 1e6:.1f} us, search k=10 {tk * 1e6:.1f} us", flush=True)
    m.close()
