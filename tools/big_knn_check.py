"""One-off sanity at a large row count: 3,000,000 x 128 pool, 16 and 200 queries, k = 1 and 10,
bit-exact against the C oracle."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cslam_amd.nns_matching import NearestNeighborsMatching
from oracle import knn as oracle_knn
rs = np.random.RandomState(0)
n, d = 3_000_000, 128
x = rs.rand(n, d).astype(np.float32)
pool = NearestNeighborsMatching(dim=d)
t0 = time.time()
for lo in range(0, n, 500_000):
    pool.add_items(x[lo:lo + 500_000], list(range(lo, lo + 500_000)))
print("added", pool.n, "rows in", round(time.time() - t0, 2), "s", flush=True)
for m, k in ((16, 1), (16, 10), (200, 1), (200, 10)):
    q = rs.rand(m, d).astype(np.float32)
    q[0] = x[n - 1]
    t0 = time.time(); rows, sims = pool.search_batch(q, k); t1 = time.time() - t0
    oi, os_, _ = oracle_knn.search(x, q, k)
    ok = np.array_equal(rows, oi) and np.array_equal(sims.view(np.uint32), os_.view(np.uint32))
    print(f"m={m} k={k}: {t1*1e3:.2f} ms  bit-exact={ok}  first hit row {rows[0,0]}", flush=True)
    assert ok and rows[0, 0] == n - 1
