"""Iteration counts of the Fiedler solve on the cfg3 graph (n = 50 000, 500 active candidates) by solver
variant, in numpy/scipy on the CPU - the experiment behind DESIGN.md section 5 (MAC), round 3:
does a block of 2-4 vectors, a coarse-space correction or an exact (Woodbury) inverse cut the 234
iterations of the single-vector LOBPCG with the chain-tridiagonal preconditioner?
Output: profiles/r03_fiedler_precond_proto.txt.  Same stopping rule as the HIP solver
(||r||_1 / ||L||_inf < 1e-9).  Runs without the HIP library."""
import sys, time

import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as sla
from scipy.linalg import eigh, solve_banded
# cfg3 graph without the HIP library
def graph(nb_robots=2, P=25000, nb_candidates=200000, seed=0):
    rs = np.random.RandomState(seed); n = nb_robots*P
    fi, fj = [], []
    for r in range(nb_robots):
        base = r*P; fi += list(range(base, base+P-1)); fj += list(range(base+1, base+P))
    for r in range(nb_robots-1):
        fi.append(r*P+P-1); fj.append((r+1)*P+P-1)
    r0 = rs.randint(nb_robots, size=nb_candidates*2)
    r1 = (r0+1+rs.randint(max(nb_robots-1,1), size=nb_candidates*2)) % nb_robots
    a = r0*P + rs.randint(P, size=nb_candidates*2); b = r1*P + rs.randint(P, size=nb_candidates*2)
    lo, hi = np.minimum(a,b), np.maximum(a,b)
    _, first = np.unique(lo.astype(np.int64)*n+hi, return_index=True)
    keep = np.sort(first)[:nb_candidates]; w = rs.rand(len(keep))
    return n, np.array(fi), np.array(fj), a[keep], b[keep], w
n, fi, fj, ci, cj, cw = graph()
k = 500
sel = np.argsort(cw)[-k:]
ei = np.concatenate([fi, ci[sel]]); ej = np.concatenate([fj, cj[sel]]); ew = np.concatenate([np.ones(len(fi)), cw[sel]])
A = sp.coo_matrix((np.concatenate([ew, ew]), (np.concatenate([ei, ej]), np.concatenate([ej, ei]))), shape=(n, n)).tocsr()
d = np.asarray(A.sum(1)).ravel(); L = (sp.diags(d) - A).tocsr()
# preconditioner: tridiagonal part of L on the chains (index neighbours)
off = np.zeros(n-1)
for r in range(2): off[r*25000:(r+1)*25000-1] = -1.0
ab = np.zeros((3, n)); ab[1] = d; ab[0,1:] = off; ab[2,:-1] = off
def Msolve(R): return solve_banded((1,1), ab, R)
ones = np.ones((n,1))/np.sqrt(n)
def proj(X): return X - ones @ (ones.T @ X)
Linf = abs(L).sum(1).max()
def lobpcg(b, tol=1e-9, maxit=600, seed=7):
    rs = np.random.RandomState(seed)
    X = proj(rs.normal(size=(n, b))); X, _ = np.linalg.qr(X)
    Pm = None
    for it in range(maxit):
        AX = L @ X
        th, C = eigh(X.T @ AX); X = X @ C; AX = AX @ C
        R = AX - X * th
        res = np.abs(R[:,0]).sum() / Linf     # residual of the wanted pair, the solver's norm
        if res < tol: return it, th[0]
        W = proj(Msolve(R))
        S = np.hstack([X, W] + ([Pm] if Pm is not None else []))
        S, _ = np.linalg.qr(S)
        G = S.T @ (L @ S)
        th2, C2 = eigh(G)
        Xn = S @ C2[:, :b]
        Pm = Xn - X @ (X.T @ Xn)
        X = Xn
    return maxit, th[0]
for b in (1, 2, 3, 4):
    t0 = time.time(); it, lam = lobpcg(b); print(f"block {b}: {it} iterations, lambda2 = {lam:.12e}  ({time.time()-t0:.1f} s)", flush=True)

print("--- preconditioner variants, block 1")
def lobpcg1(prec, tol=1e-9, maxit=600, seed=7):
    rs = np.random.RandomState(seed)
    X = proj(rs.normal(size=(n, 1))); X /= np.linalg.norm(X)
    Pm = None
    for it in range(maxit):
        AX = L @ X; th = float(X.T @ AX); R = AX - X*th
        if np.abs(R).sum()/Linf < tol: return it, th
        W = proj(prec(R))
        S = np.hstack([X, W] + ([Pm] if Pm is not None else [])); S, _ = np.linalg.qr(S)
        th2, C2 = eigh(S.T @ (L @ S)); Xn = S @ C2[:, :1]
        Pm = Xn - X @ (X.T @ Xn); X = Xn
    return maxit, th
lu = sla.splu((L + 1e-9*sp.eye(n)).tocsc())
print("exact inverse:", lobpcg1(lambda R: lu.solve(R)))
for seg in (2048, 1024, 512, 256, 128):
    nc = n // seg + (1 if n % seg else 0)
    P = sp.csr_matrix((np.ones(n), (np.arange(n), np.arange(n)//seg)), shape=(n, nc))
    Lc = (P.T @ L @ P).toarray(); Lci = np.linalg.pinv(Lc)
    add = lambda R, P=P, Lci=Lci: Msolve(R) + P @ (Lci @ (P.T @ R))
    print(f"tridiag + piecewise-constant coarse space, segments of {seg} ({nc} dofs):", lobpcg1(add))
    # multiplicative (coarse first, then smooth the remainder)
    def mult(R, P=P, Lci=Lci):
        z = P @ (Lci @ (P.T @ R)); return z + Msolve(R - L @ z)
    print(f"   multiplicative:", lobpcg1(mult))
