import sys, ctypes
import numpy as np
sys.path.insert(0, ".")
import torch
from cslam_amd.multi_robot import HipPoolMatcher
from oracle import knn as o
rs = np.random.RandomState(4)
X = rs.rand(700, 96).astype(np.float32); Q = rs.rand(9, 96).astype(np.float32)
hm = HipPoolMatcher(96)
hm.add_rows(torch.from_numpy(X[:500]).cuda(), list(range(500)))
hm.add_rows(torch.from_numpy(X[500:]).cuda(), list(range(500, 700)))
q = torch.from_numpy(Q).cuda()
for lo, hi in ((0, 700), (0, 500), (500, 700), (500, 700), (123, 124), (650, 700), (500, 700)):
    rows, sims = hm.search_best(q, lo, hi)
    oi, os_, _ = o.search(X[lo:hi], Q, 1)
    print(lo, hi, "ok" if np.array_equal(rows, oi[:, 0] + lo) else f"BAD got {rows} want {oi[:,0]+lo}", flush=True)
