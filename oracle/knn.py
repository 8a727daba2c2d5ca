"""ctypes front of oracle/knn_oracle.c (CPU restatement of cslam/nns_matching.py:44-78).

TEST INFRASTRUCTURE ONLY — see oracle/__init__.py.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle_knn.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle_knn.so")
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        fp = ctypes.POINTER(ctypes.c_float)
        ip = ctypes.POINTER(ctypes.c_int32)
        i64 = ctypes.c_int64
        L.oracle_inv_norms.argtypes = [fp, i64, i64, fp]
        L.oracle_similarities.argtypes = [fp, i64, i64, fp, i64, fp]
        L.oracle_search.argtypes = [fp, i64, i64, fp, i64, i64, ip, fp, ip]
        L.oracle_num_threads.restype = ctypes.c_int
        L.oracle_set_threads.argtypes = [ctypes.c_int]
        # a 1-GPU box grants 16 cores of CPU share however many the host shows
        try:
            avail = len(os.sched_getaffinity(0))
        except AttributeError:
            avail = os.cpu_count() or 1
        L.oracle_set_threads(max(1, min(avail, 16)))
        _LIB = L
    return _LIB


def _f32(a, ndim):
    a = np.ascontiguousarray(a, dtype=np.float32)
    assert a.ndim == ndim
    return a


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def inv_norms(rows):
    rows = _f32(rows, 2)
    out = np.empty(rows.shape[0], np.float32)
    lib().oracle_inv_norms(_p(rows, ctypes.c_float), rows.shape[0], rows.shape[1],
                           _p(out, ctypes.c_float))
    return out


def similarities(rows, queries):
    rows = _f32(rows, 2)
    queries = _f32(np.atleast_2d(queries), 2)
    n, d = rows.shape
    m = queries.shape[0]
    assert queries.shape[1] == d
    out = np.empty((m, n), np.float32)
    lib().oracle_similarities(_p(rows, ctypes.c_float), n, d, _p(queries, ctypes.c_float), m,
                              _p(out, ctypes.c_float))
    return out


def search(rows, queries, k):
    """Returns (idx[m,k] int32, sim[m,k] float32, count[m] int32)."""
    rows = _f32(rows, 2)
    queries = _f32(np.atleast_2d(queries), 2)
    n, d = rows.shape
    m = queries.shape[0]
    idx = np.empty((m, max(k, 1)), np.int32)
    sim = np.empty((m, max(k, 1)), np.float32)
    cnt = np.empty(m, np.int32)
    lib().oracle_search(_p(rows, ctypes.c_float), n, d, _p(queries, ctypes.c_float), m, k,
                        _p(idx, ctypes.c_int32), _p(sim, ctypes.c_float),
                        _p(cnt, ctypes.c_int32))
    return idx[:, :k], sim[:, :k], cnt


def num_threads():
    return lib().oracle_num_threads()


def search_numpy(rows, query, k, eps=1e-8):
    """The reference's OWN per-query arithmetic with BLAS-backed numpy in place of its torch-CPU calls
    (cslam/nns_matching.py:57-63: `cosine_similarity(view, q)` over the whole pool - the row norms are
    recomputed at every call, as there - then a full descending `argsort`, first k).  Not the numeric
    contract of the HIP path (that is knn_oracle.c's pinned summation order): this is the CPU path a user
    of the reference runs, timed by bench.py's cpu_baseline beside the strict-chain port, and pinned to
    the same goldens within the reference's 1e-6 (tests/test_oracle_knn.py).
    Returns (idx[k] int64, sim[k] float32)."""
    rows = _f32(rows, 2)
    q = _f32(np.asarray(query).reshape(-1), 1)
    dots = rows @ q                                                  # sgemv
    xn = np.sqrt(np.einsum("ij,ij->i", rows, rows))                  # a second pass over the pool, per call
    qn = np.sqrt(np.dot(q, q))
    sims = (dots / np.maximum(xn, eps)) / max(float(qn), eps)        # torch >= 1.12: each norm clamped by eps
    order = np.argsort(-sims, kind="stable")[:k]
    return order, sims[order].astype(np.float32)
