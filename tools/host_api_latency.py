"""Latency of the reference-style host API (numpy in, numpy out) at cfg2 size."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cslam_amd.nns_matching import NearestNeighborsMatching
import torch, ctypes
from cslam_amd import _lib
L = _lib.load()
N, D = 100_000, 4096
g = torch.Generator(device="cuda").manual_seed(0)
X = torch.rand((N, D), device="cuda", generator=g)
m = NearestNeighborsMatching(dim=D, capacity_hint=N + 1000)
_lib.check(L.cslam_pool_add_dev(m.handle, ctypes.c_void_p(X.data_ptr()), N, D, None)); torch.cuda.synchronize(); m.n = N
m.items = {i: i for i in range(N)}
rs = np.random.RandomState(0)
for M, k in ((1, 1), (1, 10), (16, 1)):
    q = rs.rand(M, D).astype(np.float32)
    f = (lambda: m.search_best(q[0])) if (M, k) == (1, 1) else (lambda: m.search(q[0], k)) if M == 1 else (lambda: m.search_best_batch(q))
    for _ in range(5): f()
    t0 = time.perf_counter()
    for _ in range(100): f()
    print(f"host API M={M} k={k}: {(time.perf_counter()-t0)/100*1e3:.3f} ms per call", flush=True)
# add_item latency
v = rs.rand(D).astype(np.float32)
t0 = time.perf_counter()
for i in range(200): m.add_item(v, N + i)
print(f"add_item: {(time.perf_counter()-t0)/200*1e6:.1f} us per row", flush=True)
