"""search_best on a small pool, A/B (round 4): k_small_best (one launch; query read from the pinned bounce
buffer or copied first; completion by polling a pinned word or by stream synchronisation) against the
general path (query copy + stage + scan + final level + synchronisation; CSLAM_SMALL_BEST=0).
Host-pointer C ABI called with prebuilt ctypes arguments (the Python wrapper's own cost is the last
column: NearestNeighborsMatching.search_best per call); settings alternating in one process."""
import ctypes, os, sys, time
os.environ["CSLAM_KNOBS_LIVE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cslam_amd import _lib
from cslam_amd.nns_matching import NearestNeighborsMatching
L = _lib.load()
SETTINGS = [
    ("fused, default", {}),
    ("fused, pinned query, sync", {"CSLAM_SMALL_BEST_Q": "1", "CSLAM_SMALL_BEST_POLL": "0"}),
    ("fused, pinned query, poll", {"CSLAM_SMALL_BEST_Q": "1", "CSLAM_SMALL_BEST_POLL": "1"}),
    ("fused, copied query, sync", {"CSLAM_SMALL_BEST_Q": "0", "CSLAM_SMALL_BEST_POLL": "0"}),
    ("fused, copied query, poll", {"CSLAM_SMALL_BEST_Q": "0", "CSLAM_SMALL_BEST_POLL": "1"}),
    ("general path", {"CSLAM_SMALL_BEST": "0"}),
]
KEYS = sorted({k for _, e in SETTINGS for k in e})
rs = np.random.RandomState(0)
shapes = [(1000, 128, 1, 1), (200, 128, 1, 1), (4000, 128, 1, 1), (1000, 128, 8, 1), (1000, 512, 1, 1), (1000, 4096, 1, 1), (4000, 4096, 1, 1),
          (1000, 128, 1, 10), (200, 128, 1, 10), (4000, 128, 1, 10), (1000, 4096, 1, 10)]   # k = 10: the intra-robot match (lcsm.py:85)
for n, d, m, K in shapes:
    X = rs.rand(n, d).astype(np.float32)
    Q = rs.rand(m, d).astype(np.float32)
    p = NearestNeighborsMatching(dim=d, capacity_hint=n)
    p.add_items(X, list(range(n)))
    idx = np.empty((m, K), np.int32); sim = np.empty((m, K), np.float32); cnt = np.empty(m, np.int32)
    args = (p.handle, _lib.ptr(Q), m, K, _lib.ptr(idx), _lib.ptr(sim), _lib.ptr(cnt))
    fn = L.cslam_pool_search
    res = {name: [] for name, _ in SETTINGS}
    wrap = {name: [] for name, _ in SETTINGS}
    ref = None
    for rep in range(3):
        for name, env in SETTINGS:
            for k in KEYS: os.environ.pop(k, None)
            os.environ.update(env)
            for _ in range(300): fn(*args)
            t0 = time.perf_counter()
            for _ in range(3000): fn(*args)
            t = (time.perf_counter() - t0) / 3000
            b = (idx.tobytes(), sim.tobytes())
            if ref is None: ref = b
            res[name].append((t * 1e6, b == ref))
            if m == 1:
                t0 = time.perf_counter()
                for _ in range(1000): p.search_best(Q[0]) if K == 1 else p.search(Q[0], K)
                wrap[name].append((time.perf_counter() - t0) / 1000 * 1e6)
    for k in KEYS: os.environ.pop(k, None)
    for name, _ in SETTINGS:
        print(f"n={n:5d} d={d:4d} m={m:2d} k={K:2d} {name:28s} " + " ".join(f"{a:6.1f}" for a, _ in res[name]) + " us per C call  bits " +
              ("same" if all(c for _, c in res[name]) else "DIFFER") +
              ("   search_best() / search() " + " ".join(f"{a:6.1f}" for a in wrap[name]) if wrap[name] else ""), flush=True)
    p.close()
