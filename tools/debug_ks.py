import sys, os, ctypes, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cslam_amd import _lib
from cslam_amd.nns_matching import NearestNeighborsMatching
L = _lib.load()
n, d, m = 32, 2048, 2
rs = np.random.RandomState(0)
X = rs.rand(n, d).astype(np.float32); Q = rs.rand(m, d).astype(np.float32)
p = NearestNeighborsMatching(dim=d, capacity_hint=n); p.add_items(X, list(range(n)))
qd = torch.from_numpy(Q).cuda()
sims = torch.zeros((m, n), device="cuda")
_lib.check(L.cslam_pool_similarities_dev(p.handle, ctypes.c_void_p(qd.data_ptr()), d, m, ctypes.c_void_p(sims.data_ptr()), n, None))
torch.cuda.synchronize()
s = sims.cpu().numpy()
xn = X / np.linalg.norm(X, axis=1, keepdims=True); qn = Q / np.linalg.norm(Q, axis=1, keepdims=True)
ref = qn @ xn.T
print(L.cslam_pool_scan_kernel(p.handle).decode()[:24], "KSPLIT env", os.environ.get("CSLAM_SCAN_KSPLIT"))
print("max abs err vs fp64-ish cosine:", np.abs(s - ref).max())
# which chunk pattern explains it?  per-chunk partial cosines
nch = d // 512
parts = np.stack([(qn[:, c*512:(c+1)*512] @ xn[:, c*512:(c+1)*512].T) for c in range(nch)])   # [c][m][n]
q, r = 0, 5
print("got", s[q, r], "ref", ref[q, r], "chunk partials", parts[:, q, r])
for mask in range(1, 1 << nch):
    v = sum(parts[c, q, r] for c in range(nch) if mask >> c & 1)
    if abs(v - s[q, r]) < 2e-4: print("  matches subset", [c for c in range(nch) if mask >> c & 1])
print("row pattern of error:", np.round((s - ref)[0, :20], 4))
