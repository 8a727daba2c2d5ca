"""MAC — maximise algebraic connectivity; host mirror of cslam/mac/mac.py:19-233.

Same class name, constructor and methods as the reference.  The numerics run on the GPU through
the C ABI (include/cslam_hip.h, csrc/mac.hip): edge weights, L(w) mat-vec, the Fiedler pair
(LOBPCG with a chain-tridiagonal preconditioner instead of networkx TraceMIN + SuperLU), the
gradient, the top-k direction and the whole Frank-Wolfe loop.  There is no CPU fallback.

Error behaviour kept from the reference: a disconnected L(w) raises (the reference's SuperLU
raises RuntimeError('Factor is exactly singular'); here `GraphDisconnectedError`, also a
RuntimeError) so the retry ladder of
cslam/algebraic_connectivity_maximization.py:449-465 behaves the same.
"""
import ctypes
import itertools
from collections import namedtuple

import numpy as np

from .. import _lib
from .utils import Edge, weight_graph_lap_from_edge_list  # noqa: F401  (re-exported like mac.py:14)

MACResult = namedtuple("MACResult", ["w", "F_unrounded", "objective_values", "duality_gaps"])


def _edge_rows(edges):
    """list of Edge(i, j, weight) → float64 [n, 3] in one pass (np.array on a list of namedtuples
    is ~10x slower than this)."""
    n = len(edges)
    if n == 0:
        return np.zeros((0, 3))
    return np.fromiter(itertools.chain.from_iterable(edges), dtype=np.float64, count=3 * n).reshape(n, 3)


class MAC:

    def __init__(self, fixed_measurements, candidate_measurements, num_poses, device_id=0):
        self._lib = _lib.load()
        if self._lib.cslam_device_count() <= 0:
            raise _lib.CslamHipError(_lib.CSLAM_E_NODEVICE,
                                     "no HIP device visible; cslam_amd has no CPU path")
        self.num_poses = int(num_poses)
        f = _edge_rows(fixed_measurements)
        c = _edge_rows(candidate_measurements)
        self._init_arrays(f[:, 0], f[:, 1], f[:, 2], c[:, 0], c[:, 1], c[:, 2], device_id)

    @classmethod
    def from_arrays(cls, fixed_i, fixed_j, fixed_w, cand_i, cand_j, cand_w, num_poses, device_id=0):
        """Same object from edge arrays (no per-edge Python objects): what
        AlgebraicConnectivityMaximization.select_candidates hands over at cfg3 scale."""
        self = cls.__new__(cls)
        self._lib = _lib.load()
        if self._lib.cslam_device_count() <= 0:
            raise _lib.CslamHipError(_lib.CSLAM_E_NODEVICE,
                                     "no HIP device visible; cslam_amd has no CPU path")
        self.num_poses = int(num_poses)
        self._init_arrays(fixed_i, fixed_j, fixed_w, cand_i, cand_j, cand_w, device_id)
        return self

    def _init_arrays(self, fi, fj, fw, ci, cj, cw, device_id):
        self._fi = _lib.as_i32(fi)
        self._fj = _lib.as_i32(fj)
        self._fw = _lib.as_f64(fw)
        self.weights = _lib.as_f64(cw)
        self._ci = _lib.as_i32(ci)
        self._cj = _lib.as_i32(cj)
        self.edge_list = np.stack([self._ci.astype(np.int64), self._cj.astype(np.int64)], axis=1) \
            if len(self._ci) else np.zeros((0, 2), np.int64)
        self._handle = ctypes.c_void_p()
        _lib.check(self._lib.cslam_mac_create(
            self.num_poses, _lib.ptr(self._fi), _lib.ptr(self._fj), _lib.ptr(self._fw),
            len(self._fw), _lib.ptr(self._ci), _lib.ptr(self._cj), _lib.ptr(self.weights),
            len(self.weights), int(device_id), ctypes.byref(self._handle)))
        self._L_odom = None
        self.last_iters = 0
        self.trace_directions = False   # set True: fw_subset also records every direction s_i
        self._direction_trace = []
        self._upper_trace = np.zeros(0)
        self._solver_iters_trace = np.zeros(0, np.int32)
        self._trace_pending = None       # (iterations, k) of the last fw_subset whose trace was not read yet

    # ---- lifetime --------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_handle", None):
            self._lib.cslam_mac_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def L_odom(self):
        """Laplacian of the fixed edges (mac.py:22), as scipy CSR."""
        if self._L_odom is None:
            self._L_odom = self.combined_laplacian(np.zeros(len(self.weights)))
        return self._L_odom

    # ---- L(w) ------------------------------------------------------------------------------
    def combined_laplacian(self, w, tol=1e-10):
        """L(w) = L_fixed + sum_k w_k c_k L_k over w_k > tol (mac.py:61-77), scipy CSR."""
        from scipy.sparse import csr_matrix
        w = _lib.as_f64(w)
        nnz = ctypes.c_int64(0)
        _lib.check(self._lib.cslam_mac_laplacian(self._handle, _lib.ptr(w), float(tol),
                                                 ctypes.byref(nnz), None, None, None))
        indptr = np.empty(self.num_poses + 1, np.int32)
        indices = np.empty(max(nnz.value, 1), np.int32)
        data = np.empty(max(nnz.value, 1), np.float64)
        _lib.check(self._lib.cslam_mac_laplacian(self._handle, _lib.ptr(w), float(tol),
                                                 ctypes.byref(nnz), _lib.ptr(indptr),
                                                 _lib.ptr(indices), _lib.ptr(data)))
        return csr_matrix((data[:nnz.value], indices[:nnz.value], indptr),
                          shape=(self.num_poses, self.num_poses))

    # ---- Fiedler pair ----------------------------------------------------------------------
    def evaluate_fiedler_pair(self, w, method="tracemin_lu", tol=1e-8):
        """(lambda_2(L(w)), v_2(L(w))) — mac.py:79-96.  `method` is accepted for signature
        compatibility; the device solver is always used."""
        w = _lib.as_f64(w)
        if len(w) != len(self.weights):
            raise ValueError("w has the wrong length")
        lam = ctypes.c_double()
        v = np.empty(self.num_poses, np.float64)
        it = ctypes.c_int32()
        _lib.check(self._lib.cslam_mac_fiedler(self._handle, _lib.ptr(w), float(tol),
                                               ctypes.byref(lam), _lib.ptr(v), ctypes.byref(it)))
        self.last_iters = it.value
        return lam.value, v

    def find_fiedler_pair(self, L, method="tracemin_lu", tol=1e-8):
        """Fiedler pair of an arbitrary graph Laplacian given as a scipy sparse matrix
        (mac.py:35-59).  The off-diagonal entries define the edges."""
        assert method != "lobpcg"  # mac.py:51
        from scipy.sparse import triu
        U = triu(L, k=1).tocoo()
        edges = [Edge(int(i), int(j), float(-v)) for i, j, v in zip(U.row, U.col, U.data) if v != 0]
        tmp = MAC(edges, [], L.shape[0])
        try:
            return tmp.evaluate_fiedler_pair(np.zeros(0), method, tol)
        finally:
            tmp.close()

    def evaluate_objective(self, w):
        """F(w) = lambda_2(L(w)) (mac.py:98-110)."""
        return self.evaluate_fiedler_pair(w)[0]

    # ---- gradient and rounding -------------------------------------------------------------
    def grad_from_fiedler(self, fiedler_vec):
        """grad_k = weight_k (v_i - v_j)^2 (mac.py:112-130)."""
        v = _lib.as_f64(fiedler_vec)
        g = np.empty(len(self.weights), np.float64)
        _lib.check(self._lib.cslam_mac_grad(self._handle, _lib.ptr(v), _lib.ptr(g)))
        return g

    def round_solution(self, w, k):
        """0/1 vector of the k largest entries (mac.py:132-147); ties → lowest index."""
        w = _lib.as_f64(w)
        out = np.zeros(len(w), np.float64)
        if k > 0 and len(w) > 0:
            _lib.check(self._lib.cslam_topk_mask(_lib.ptr(w), len(w), int(k), _lib.ptr(out)))
        return out

    def round_solution_tiebreaker(self, w, k, decimal_tol=10):
        """k largest by (round(w, decimal_tol), weight) (mac.py:168-189).

        The selection runs in the native library (cslam_mac_round_tiebreak).  Only when the
        k-th and (k+1)-th keys are EXACTLY equal — the outcome then depends on numpy's
        introselect, e.g. when all candidate weights are 1 — is the very numpy call of the
        reference evaluated instead, so ties resolve to the same edges as the reference."""
        w = _lib.as_f64(w)
        m = len(w)
        if k <= 0 or m == 0:
            return np.zeros(m)
        if k >= m:
            return np.ones(m)
        rounded = np.empty(m, np.float64)
        _lib.check(self._lib.cslam_mac_round_tiebreak(self._handle, _lib.ptr(w), int(k),
                                                      int(decimal_tol), _lib.ptr(rounded)))
        return self._resolve_boundary_tie(w, rounded, k, decimal_tol)

    def _resolve_boundary_tie(self, w, rounded, k, decimal_tol=10):
        """`rounded` is the native selection (ties at the boundary → higher weight, then lower
        index).  If the last selected and the first rejected key are EXACTLY equal the reference's
        outcome is whatever numpy's introselect leaves in the last k slots: evaluate that very
        call then; otherwise the set is unique and `rounded` stands."""
        m = len(w)
        tw = w.round(decimals=decimal_tol)
        sel = rounded > 0.5
        lo_tw = tw[sel].min()
        lo_wt = self.weights[sel & (tw == lo_tw)].min()
        hi_tw = tw[~sel].max()
        hi_wt = self.weights[~sel & (tw == hi_tw)].max()
        if (lo_tw, lo_wt) != (hi_tw, hi_wt):
            return rounded  # boundary is tie-free: the set is unique
        keys = np.empty(m, dtype=[("w", "float"), ("weight", "float")])
        keys["w"] = tw
        keys["weight"] = self.weights
        rounded = np.zeros(m)
        rounded[np.argpartition(keys, -k, order=["w", "weight"])[-k:]] = 1.0
        return rounded

    def simple_random_round(self, w, k):
        """Bernoulli rounding (mac.py:149-166)."""
        return (np.asarray(w) > np.random.rand(len(w))).astype(np.float64)

    # ---- Frank-Wolfe -----------------------------------------------------------------------
    def fw_subset(self, w_init, k, max_iters=5, duality_gap_tol=1e-8):
        """Frank-Wolfe on the relaxed subset-selection problem (mac.py:191-233), whole loop on
        the device.  Returns (rounded, unrounded, upper_bound)."""
        w_init = _lib.as_f64(w_init)
        m = len(self.weights)
        if len(w_init) != m:
            raise ValueError("w_init has the wrong length")
        w_round = np.empty(m, np.float64)
        w_relaxed = np.empty(m, np.float64)
        upper = ctypes.c_double()
        its = ctypes.c_int32()
        self.lambda_trace = np.full(max(int(max_iters), 1), np.nan)
        _lib.check(self._lib.cslam_mac_fw_trace_enable(self._handle, int(bool(self.trace_directions))))
        _lib.check(self._lib.cslam_mac_fw_subset(
            self._handle, _lib.ptr(w_init), int(k), int(max_iters), float(duality_gap_tol),
            _lib.ptr(w_round), _lib.ptr(w_relaxed), ctypes.byref(upper), ctypes.byref(its),
            _lib.ptr(self.lambda_trace)))
        self.fw_iters = its.value
        self.lambda_trace = self.lambda_trace[:its.value]
        # per-iteration diagnostics (dual bounds, solver iterations, directions) stay on the device until somebody
        # asks: 20 small reads cost a selection round 1 ms of its 27
        self._trace_pending = (its.value, int(k))
        if self.trace_directions:
            self._read_fw_trace()
        if its.value < max_iters or (its.value > 0 and max_iters > 0 and
                                     upper.value - self.lambda_trace[-1] < duality_gap_tol):
            print("Duality gap tolerance reached, found optimal solution")  # mac.py:224
        else:
            print("Reached maximum iterations")  # mac.py:232
        if 0 < k < m:   # the library already rounded w_relaxed; only a boundary tie needs numpy
            w_round = self._resolve_boundary_tie(w_relaxed, w_round, k)
        return w_round, w_relaxed, upper.value

    @property
    def upper_trace(self):
        """Dual bounds u_i of the last fw_subset call (mac.py:219)."""
        self._read_fw_trace()
        return self._upper_trace

    @property
    def solver_iters_trace(self):
        """Eigen-solver iterations per Frank-Wolfe iteration of the last fw_subset call."""
        self._read_fw_trace()
        return self._solver_iters_trace

    @property
    def direction_trace(self):
        """With `trace_directions`: the directions s_i (mac.py:215) of the last fw_subset call, ascending indices."""
        self._read_fw_trace()
        return self._direction_trace

    def _read_fw_trace(self):
        """Fetch the per-iteration diagnostics of the last fw_subset call from the device (once)."""
        if self._trace_pending is None or not self._handle:
            return
        iters, k = self._trace_pending
        self._trace_pending = None
        self._upper_trace = np.empty(iters)
        self._solver_iters_trace = np.empty(iters, np.int32)
        self._direction_trace = []
        u = ctypes.c_double()
        f = ctypes.c_double()
        si = ctypes.c_int32()
        n_idx = ctypes.c_int64()
        for it in range(iters):
            idx = np.empty(max(k, 1), np.int32) if self.trace_directions else None
            _lib.check(self._lib.cslam_mac_fw_trace_read(
                self._handle, it, _lib.ptr(idx) if idx is not None else None, ctypes.byref(n_idx),
                ctypes.byref(f), ctypes.byref(u), ctypes.byref(si)))
            self._upper_trace[it] = u.value
            self._solver_iters_trace[it] = si.value
            if idx is not None:
                self._direction_trace.append(idx[:n_idx.value].copy())

    def stats(self):
        """(solver iterations so far, device ms of the last fiedler / fw_subset call)."""
        n = ctypes.c_int64()
        ms = ctypes.c_double()
        _lib.check(self._lib.cslam_mac_stats(self._handle, ctypes.byref(n), ctypes.byref(ms)))
        return n.value, ms.value

