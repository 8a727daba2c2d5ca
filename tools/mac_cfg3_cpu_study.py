"""CPU side of the cfg3 selection-parity study (profiles/r04_mac_cfg3_contract.txt).
Build container only (oracle = test infrastructure; minutes of CPU time):
  python tools/mac_cfg3_cpu_study.py trace 1e-8  /tmp/tr8.npz    oracle fw_subset, TraceMIN-LU at tol
  python tools/mac_cfg3_cpu_study.py trace 1e-12 /tmp/tr12.npz
  python tools/mac_cfg3_cpu_study.py teacher /tmp/tr8.npz        iterates of a trace re-solved at tol 1e-13
  python tools/mac_cfg3_cpu_study.py eigsh /tmp/tr12.npz         free run with scipy eigsh (tol 0) vs a trace
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import mac_oracle as mo                      # noqa: E402
from tools.mac_bench_util import cfg3_graph, greedy_w    # noqa: E402

K = 500


def problem(cls=mo.MacOracle):
    n, fixed, cand = cfg3_graph()
    f = np.array([(e.i, e.j, e.weight) for e in fixed])
    c = np.array([(e.i, e.j, e.weight) for e in cand])
    o = cls(f[:, 0].astype(int), f[:, 1].astype(int), f[:, 2], c[:, 0].astype(int), c[:, 1].astype(int),
            c[:, 2], n)
    return o, greedy_w(cand, K)


def save(path, tr, r, w, u):
    np.savez_compressed(path, f=np.array([t["f"] for t in tr]), u=np.array([t["u"] for t in tr]),
                        w=np.array([t["w"] for t in tr]), s=np.array([np.nonzero(t["s"])[0] for t in tr]),
                        v=np.array([t["v"] for t in tr]), g=np.array([t["g"] for t in tr]),
                        rounded=np.nonzero(r)[0], w_final=w, upper=u)


def cmd_trace(tol, out):
    class O(mo.MacOracle):
        def evaluate_fiedler_pair(self, w, tol_=None):
            return self.find_fiedler_pair(self.combined_laplacian(w), tol)
    o, w0 = problem(O)
    tr = []
    t0 = time.time()
    r, w, u = o.fw_subset(w0, K, max_iters=20, trace=tr)
    print("time", time.time() - t0)
    save(out, tr, r, w, u)
    print(np.array([t["f"] for t in tr]))


def cmd_teacher(path):
    o, _ = problem()
    a = np.load(path)
    for i in [0, 1, 2, 5, 8, 13, 16, 18, 19]:
        w = a["w"][i]
        lam, v = o.find_fiedler_pair(o.combined_laplacian(w), 1e-13)
        g = o.grad_from_fiedler(v)
        s = set(np.argpartition(g, -K)[-K:])
        sr = set(a["s"][i])
        gr = a["g"][i]
        kth_r = np.sort(gr)[-K]
        kth = np.sort(g)[-K]
        mism = list(s ^ sr)
        band_r = max([abs(gr[j] - kth_r) / kth_r for j in mism], default=0)
        band_e = max([abs(g[j] - kth) / kth for j in mism], default=0)
        vv = mo.sign_normalise(v)
        print(i, "lam_exact=%.10e f_ref=%.10e d=%.1e" % (lam, a["f"][i], abs(lam - a["f"][i])),
              "mism=%d band_ref=%.2e band_exact=%.2e" % (len(mism) // 2, band_r, band_e),
              "dv=%.1e relg=%.1e" % (np.abs(vv - a["v"][i]).max() / np.abs(vv).max(),
                                     (np.abs(g - gr) / kth).max()))


def cmd_eigsh(path):
    import scipy.sparse.linalg as spla

    class O(mo.MacOracle):
        def evaluate_fiedler_pair(self, w, tol_=None):
            L = self.combined_laplacian(w)
            vals, vecs = spla.eigsh(L.tocsc(), k=3, sigma=-1e-7, which="LM", tol=0)
            o = np.argsort(vals)
            return vals[o[1]], vecs[:, o[1]]
    o, w0 = problem(O)
    tr = []
    t0 = time.time()
    r, w, u = o.fw_subset(w0, K, max_iters=20, trace=tr)
    print("time", time.time() - t0)
    b = np.load(path)
    for i in range(20):
        print(i, "%.10e %.10e" % (tr[i]["f"], b["f"][i]), len(set(np.nonzero(tr[i]["s"])[0]) & set(b["s"][i])))
    print("final overlap", len(set(np.nonzero(r)[0]) & set(b["rounded"])))


if __name__ == "__main__":
    if sys.argv[1] == "trace":
        cmd_trace(float(sys.argv[2]), sys.argv[3])
    elif sys.argv[1] == "teacher":
        cmd_teacher(sys.argv[2])
    elif sys.argv[1] == "eigsh":
        cmd_eigsh(sys.argv[2])
