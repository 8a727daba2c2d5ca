"""Does a better start vector cut the Fiedler solve's iterations?  (round 3, DESIGN.md section 5 MAC.)  The cfg3
graph of tools/fiedler_precond_proto.py, single-vector LOBPCG with the chain-tridiagonal preconditioner, started from
the prolonged Fiedler vector of a coarse graph (aggregates of 16..256 consecutive poses).  numpy/scipy on the CPU;
output in profiles/r03_fiedler_start_proto.txt."""
import sys, time
import numpy as np, scipy.sparse as sp
from scipy.linalg import eigh, solve_banded
import os
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'fiedler_precond_proto.py')).read().split("def lobpcg(b")[0])
def lobpcg1(x0, tol=1e-9, maxit=600, hist=False):
    X = proj(x0.reshape(-1,1)); X /= np.linalg.norm(X)
    Pm = None; h=[]
    for it in range(maxit):
        AX = L @ X; th = (X.T @ AX).item(); R = AX - X*th
        res = np.abs(R).sum()/Linf; h.append(res)
        if res < tol: return it, th, h
        W = proj(Msolve(R))
        S = np.hstack([X, W] + ([Pm] if Pm is not None else [])); S, _ = np.linalg.qr(S)
        th2, C2 = eigh(S.T @ (L @ S)); Xn = S @ C2[:, :1]
        Pm = Xn - X @ (X.T @ Xn); X = Xn
    return maxit, th, h
rs = np.random.RandomState(7)
it, th, h = lobpcg1(rs.normal(size=n))
print("random start:", it, th, "residual every 20:", ["%.1e" % v for v in h[::20]])
for seg in (16, 32, 64, 128, 256):
    nc = (n + seg - 1)//seg
    # piecewise-constant aggregation for the coarse operator, linear interpolation for the prolongation
    P = sp.csr_matrix((np.ones(n), (np.arange(n), np.arange(n)//seg)), shape=(n, nc))
    Lc = (P.T @ L @ P).toarray(); Mc = np.asarray((P.T @ P).todense())
    wv, V = eigh(Lc, Mc)
    vc = V[:, 1]
    x_pc = P @ vc
    # linear interpolation between aggregate centres along each chain
    centres = (np.arange(nc) + 0.5) * seg
    x_lin = np.empty(n)
    for r in range(2):
        lo, hi = r*25000, (r+1)*25000
        cs = [c for c in range(nc) if lo <= centres[c] < hi]
        x_lin[lo:hi] = np.interp(np.arange(lo, hi), centres[cs], vc[cs])
    for name, x0 in (("piecewise-constant", x_pc), ("linear", x_lin)):
        it, th, h = lobpcg1(x0)
        print(f"coarse {nc} dofs (seg {seg}) {name}: start residual {h[0]:.2e}, {it} iterations")
    # smoothed start: a few preconditioned steps are already what LOBPCG does
