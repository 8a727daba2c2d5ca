import sys, ctypes
import numpy as np
sys.path.insert(0, ".")
from cslam_amd import _lib
from cslam_amd.nns_matching import NearestNeighborsMatching
from oracle import knn as o
L = _lib.load()
def run(name, X, Q):
    m = NearestNeighborsMatching(); m.add_items(X, list(range(len(X))))
    got = np.empty(len(X), np.float32)
    _lib.check(L.cslam_pool_read_inv_norms(m.handle, 0, len(X), _lib.ptr(got)))
    on = o.inv_norms(X)
    r, s = m.search_batch(Q, min(10, len(X))); oi, os_, _ = o.search(X, Q, min(10, len(X)))
    print(f"{name:28s} invnorm_ok={got.tobytes()==on.tobytes()} idx_ok={np.array_equal(r,oi)} sims_ok={s.tobytes()==os_.tobytes()} maxulp={np.abs(s.view(np.int32)-os_.view(np.int32)).max()}", flush=True)
    m.close()
rs = np.random.RandomState(5)
for D in (16, 64, 128, 200, 208, 256, 4096):
    X = rs.rand(600, D).astype(np.float32); Q = rs.rand(5, D).astype(np.float32)
    Xn = X / np.linalg.norm(X, axis=1, keepdims=True)
    run(f"D={D} Xnorm Qraw", Xn, Q)
    run(f"D={D} Xraw Qraw", X, Q)
    run(f"D={D} Xraw*3 Qraw", X * 3, Q)
    run(f"D={D} Xnorm*4(pow2) Qraw", Xn * 4, Q)
    run(f"D={D} Xraw Qnorm", X, Q / np.linalg.norm(Q, axis=1, keepdims=True))
    run(f"D={D} Xrandn Qrandn", rs.randn(600, D).astype(np.float32), rs.randn(5, D).astype(np.float32))
