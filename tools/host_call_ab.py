"""The host-pointer search call beyond k_small_search's limits (round 4): completion by the polled pinned word
(the call's final kernel signals: k_best_pair / k_topk_from_parts) against stream synchronisation
(CSLAM_SMALL_BEST_POLL=0), and the queries' way into HBM: a copy kernel reading the pinned buffer against
hipMemcpyAsync (CSLAM_HOST_Q_COPY=1).  Prebuilt ctypes arguments, settings alternating in one process."""
import os, sys, time
os.environ["CSLAM_KNOBS_LIVE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cslam_amd import _lib
from cslam_amd.nns_matching import NearestNeighborsMatching
L = _lib.load()
rs = np.random.RandomState(0)
for n, d, m, K in [(10000, 128, 1, 1), (25000, 128, 1, 1), (100000, 128, 1, 1), (10000, 128, 1, 10), (100000, 128, 16, 1),
                   (25000, 4096, 1, 1), (25000, 4096, 4, 1), (100000, 4096, 16, 1), (100000, 4096, 16, 10), (25000, 4096, 80, 1), (25000, 4096, 80, 26)]:
    X = rs.rand(n, d).astype(np.float32)
    Q = rs.rand(m, d).astype(np.float32)
    p = NearestNeighborsMatching(dim=d, capacity_hint=n)
    p.add_items(X, list(range(n)))
    del X
    idx = np.empty((m, K), np.int32); sim = np.empty((m, K), np.float32); cnt = np.empty(m, np.int32)
    args = (p.handle, _lib.ptr(Q), m, K, _lib.ptr(idx), _lib.ptr(sim), _lib.ptr(cnt))
    SETTINGS = [("default", {}), ("stream sync", {"CSLAM_SMALL_BEST_POLL": "0"}), ("polled, any size", {"CSLAM_SMALL_BEST_POLL": "2"}),
                ("hipMemcpyAsync, default", {"CSLAM_HOST_Q_COPY": "1"}), ("hipMemcpyAsync, sync", {"CSLAM_HOST_Q_COPY": "1", "CSLAM_SMALL_BEST_POLL": "0"}),
                ("hipMemcpyAsync, polled", {"CSLAM_HOST_Q_COPY": "1", "CSLAM_SMALL_BEST_POLL": "2"})]
    KEYS = ("CSLAM_SMALL_BEST_POLL", "CSLAM_HOST_Q_COPY")
    res = {name: [] for name, _ in SETTINGS}; ref = None
    reps = 2000 if n * d < 50_000_000 else 300
    for rep in range(3):
        for name, env in SETTINGS:
            for key in KEYS: os.environ.pop(key, None)
            os.environ.update(env)
            for _ in range(50): L.cslam_pool_search(*args)
            t0 = time.perf_counter()
            for _ in range(reps): L.cslam_pool_search(*args)
            t = (time.perf_counter() - t0) / reps
            b = (idx.tobytes(), sim.tobytes())
            if ref is None: ref = b
            res[name].append((t * 1e6, b == ref))
    for key in KEYS: os.environ.pop(key, None)
    for name, _ in SETTINGS:
        print(f"n={n:6d} d={d:4d} m={m:2d} k={K:2d} {name:24s} " + " ".join(f"{a:7.1f}" for a, _ in res[name]) + " us per C call  bits " +
              ("same" if all(c for _, c in res[name]) else "DIFFER") + "  " + L.cslam_pool_scan_kernel(p.handle).decode()[:40], flush=True)
    p.close()
