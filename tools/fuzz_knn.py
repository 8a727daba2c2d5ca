"""Randomised parity sweep of the HIP kNN against the C oracle (bit-exact similarities and
indices) over ragged shapes, k values, duplicated / zero / huge-norm rows and incremental adds.
Every case also runs the split best-match search (cslam_pool_search_best_split_dev: best row before
/ from a random split, random upper row bound) against the oracle on the two row ranges, and (round 4)
a top-k search on a random row window (cslam_pool_search_range_dev).
Not part of the test suite (minutes); run by hand:  python tools/fuzz_knn.py [cases] [seed]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cslam_amd.nns_matching import NearestNeighborsMatching  # noqa: E402
from cslam_amd import _lib  # noqa: E402
from oracle import knn as oracle_knn  # noqa: E402

L = _lib.load()

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for c in range(cases):
    dim = int(rs.choice([1, 3, 16, 17, 64, 100, 120, 128, 192, 255, 256, 320, 384, 511, 512, 640, 1000, 1024, 2048, 4096]))
    n = int(rs.choice([1, 2, 15, 16, 17, 127, 128, 129, 1000, 4097, 8192, 12000, 16383, 16384, 20000, 33001, 70000]))
    if n * dim > 40_000_000:
        n = 40_000_000 // dim
    m = int(rs.choice([1, 2, 15, 16, 17, 31, 33, 40, 48, 64, 65, 80, 81, 96, 130, 160, 300]))
    while n * dim * m > 3_000_000_000 and m > 1:   # keep the strict-chain oracle to seconds per case
        m = max(1, m // 2)
    k = int(rs.choice([1, 1, 1, 2, 3, 5, 10, 16, 17, 26, 32, 128, 129, 1000]))
    scale = float(rs.choice([1.0, 1e-3, 1e3]))
    x = (rs.rand(n, dim).astype(np.float32) - (0.5 if rs.rand() < 0.5 else 0.0)) * scale
    if n > 4:
        x[rs.randint(n)] = 0.0                       # zero row
        x[rs.randint(n)] = x[rs.randint(n)]          # duplicate
        x[rs.randint(n)] *= 1e4
    q = rs.rand(m, dim).astype(np.float32) - (0.5 if rs.rand() < 0.5 else 0.0)
    if m > 2:
        q[1] = x[rs.randint(n)]
        q[2] = 0.0
    pool = NearestNeighborsMatching(dim=dim)
    cut = int(rs.randint(0, n + 1))
    if cut:
        pool.add_items(x[:cut], list(range(cut)))
    for i in range(cut, min(n, cut + 3)):
        pool.add_item(x[i], i)
    if cut + 3 < n:
        pool.add_items(x[cut + 3:], list(range(cut + 3, n)))
    rows, sims = pool.search_batch(q, k)
    oi, os_, _ = oracle_knn.search(x, q, k)
    kk = min(k, n)
    ok = rows.shape == (m, kk) and np.array_equal(rows, oi[:, :kk]) and \
        np.array_equal(sims.view(np.uint32), os_[:, :kk].view(np.uint32))
    if not ok:
        bad += 1
        print("MISMATCH", dict(dim=dim, n=n, m=m, k=k, scale=scale, cut=cut), flush=True)
    if ok and n >= 3:
        # a row window through the device-pointer API (the shard's searches; k_small_best and
        # k_topk_from_parts on windows that do not start at row 0)
        lo = int(rs.randint(0, n - 1))
        hi_w = int(rs.randint(lo + 1, n + 1))
        kw = min(k, 64)
        qd = torch.from_numpy(q).cuda()
        wi = torch.empty((m, kw), dtype=torch.int32, device="cuda")
        ws = torch.empty((m, kw), dtype=torch.float32, device="cuda")
        _lib.check(L.cslam_pool_search_range_dev(pool.handle, ctypes.c_void_p(qd.data_ptr()), dim, m, kw, lo, hi_w,
                                                 ctypes.c_void_p(wi.data_ptr()), ctypes.c_void_p(ws.data_ptr()), None))
        torch.cuda.synchronize()
        oi_w, os_w, oc_w = oracle_knn.search(x[lo:hi_w], q, kw)
        kk_w = int(oc_w[0])
        ok3 = np.array_equal(wi.cpu().numpy()[:, :kk_w], oi_w[:, :kk_w] + lo) and \
            np.array_equal(ws.cpu().numpy()[:, :kk_w].view(np.uint32), os_w[:, :kk_w].view(np.uint32)) and \
            bool(np.all(wi.cpu().numpy()[:, kk_w:] == -1))
        if not ok3:
            bad += 1
            print("WINDOW MISMATCH", dict(dim=dim, n=n, m=m, k=kw, lo=lo, hi=hi_w, scale=scale), flush=True)
    if ok and n >= 2:
        ms = min(m, 100)
        hi = int(rs.randint(2, n + 1))
        split = int(rs.randint(1, hi))
        qd = torch.from_numpy(q[:ms]).cuda()
        out = [torch.empty(ms, dtype=torch.int32, device="cuda"), torch.empty(ms, device="cuda"),
               torch.empty(ms, dtype=torch.int32, device="cuda"), torch.empty(ms, device="cuda")]
        _lib.check(L.cslam_pool_search_best_split_dev(pool.handle, ctypes.c_void_p(qd.data_ptr()), dim, ms,
                                                      split, hi, *[ctypes.c_void_p(t.data_ptr()) for t in out],
                                                      None))
        torch.cuda.synchronize()
        ia, sa, ic, sc = [t.cpu().numpy() for t in out]
        oa_i, oa_s, _ = oracle_knn.search(x[:split], q[:ms], 1)
        oc_i, oc_s, _ = oracle_knn.search(x[split:hi], q[:ms], 1)
        ok2 = np.array_equal(ia, oa_i[:, 0]) and np.array_equal(ic, oc_i[:, 0] + split) and \
            np.array_equal(sa.view(np.uint32), oa_s[:, 0].view(np.uint32)) and \
            np.array_equal(sc.view(np.uint32), oc_s[:, 0].view(np.uint32))
        if not ok2:
            bad += 1
            print("SPLIT MISMATCH", dict(dim=dim, n=n, m=ms, split=split, hi=hi, scale=scale), flush=True)
    pool.close()
    if c % 25 == 0:
        print(f"case {c}/{cases} ok so far: {c + 1 - bad}", flush=True)
print("cases", cases, "mismatches", bad)
sys.exit(1 if bad else 0)
