"""The HIP kNN path checked DIRECTLY against outputs of the reference (no oracle in between).

tests/golden/knn_ref.npz holds what cslam/nns_matching.py:44-63 (`search`) returned on torch-CPU
for five seeded cases (tests/golden/gen_golden.py::gen_knn): the reference's own test_similarity
shape, cfg1's 128-d unit rows, a 4096-d slice of cfg2, raw/negative data, ties and zero rows.
Contract, the tolerance of the reference's own test
(tests/test_sparse_matching.py:75-78): similarities within 1e-6 of the reference's; indices
identical wherever the similarities involved differ by more than 1e-6 (torch's argsort is
unstable, so inside a near-tie any order is a valid reference output).

(The 64-d descriptor stream of lcsm_ref.json / lcsm_desc.npz and cfg1_ref.json meet the HIP path in
tests/test_mac_gpu.py::test_loop_closure_sparse_matching_end_to_end and tests/test_cfg1_gpu.py.)"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "knn_ref.npz"))
TOL = 1e-6


def _cosines64(X, q):
    """Exact (float64) cosine similarities with the reference's eps clamp (nns_matching.py:61)."""
    X = X.astype(np.float64)
    q = q.astype(np.float64)
    return (X @ q) / (np.maximum(np.linalg.norm(X, axis=1), 1e-8) * max(np.linalg.norm(q), 1e-8))


def _check_against_reference(case, rows, sims, q_index):
    X, Q, ref_idx, ref_sim, k = (G[f"{case}_{s}"] for s in ("X", "Q", "idx", "sim", "k"))
    kk = min(int(k), X.shape[0])
    q = q_index
    exact = _cosines64(X, Q[q])
    assert len(rows) == kk and len(sims) == kk
    np.testing.assert_allclose(sims, ref_sim[q, :kk], atol=TOL, rtol=0)
    assert np.all(sims[:-1] >= sims[1:])
    for j in range(kk):
        if rows[j] != ref_idx[q, j]:
            # only allowed inside a near-tie of the reference
            assert abs(exact[rows[j]] - exact[ref_idx[q, j]]) <= TOL, (case, q, j)
    # same SET outside near-ties at the cut
    for r in set(ref_idx[q, :kk].tolist()) - set(int(x) for x in rows):
        assert abs(exact[r] - exact[int(rows[-1])]) <= TOL, (case, q, r)


@pytest.mark.parametrize("case", [str(c) for c in G["cases"]])
def test_search_equals_the_reference_per_query(gpu, case):
    """One `search(query, k)` call per query — the reference's call pattern (nns_matching.py:44)."""
    from cslam_amd.nns_matching import NearestNeighborsMatching
    X, Q, k = G[f"{case}_X"], G[f"{case}_Q"], int(G[f"{case}_k"])
    m = NearestNeighborsMatching()
    for i in range(X.shape[0]):
        m.add_item(X[i], i)                     # nns_matching.py:23-42, one row at a time
    for q in range(Q.shape[0]):
        items, sims = m.search(Q[q], k)
        assert sims.dtype == np.float32
        _check_against_reference(case, [int(i) for i in items], np.asarray(sims), q)
        item, sim = m.search_best(Q[q])         # nns_matching.py:65-78
        exact = _cosines64(X, Q[q])
        ref_best = int(G[f"{case}_idx"][q, 0])
        assert abs(float(sim) - float(G[f"{case}_sim"][q, 0])) <= TOL
        assert item == ref_best or abs(exact[item] - exact[ref_best]) <= TOL
    m.close()


@pytest.mark.parametrize("case", [str(c) for c in G["cases"]])
def test_batched_search_equals_the_reference(gpu, case):
    """All queries of a case in one pass (`search_batch`: the path the sharded step takes)."""
    from cslam_amd.nns_matching import NearestNeighborsMatching
    X, Q, k = G[f"{case}_X"], G[f"{case}_Q"], int(G[f"{case}_k"])
    m = NearestNeighborsMatching()
    m.add_items(X, list(range(X.shape[0])))
    rows, sims = m.search_batch(Q, k)
    for q in range(Q.shape[0]):
        _check_against_reference(case, rows[q].tolist(), sims[q], q)
    m.close()


def test_raw_c_abi_equals_the_reference(gpu):
    """The same five cases through the bare C ABI (cslam_pool_create / _add / _search): what a
    maintainer's ctypes stub would call."""
    import ctypes
    from cslam_amd import _lib
    lib = gpu
    for case in [str(c) for c in G["cases"]]:
        X, Q, k = G[f"{case}_X"], G[f"{case}_Q"], int(G[f"{case}_k"])
        n, d = X.shape
        kk = min(k, n)
        h = ctypes.c_void_p()
        _lib.check(lib.cslam_pool_create(d, 0, 0, ctypes.byref(h)))
        Xc = np.ascontiguousarray(X, np.float32)
        _lib.check(lib.cslam_pool_add(h, _lib.ptr(Xc), n))
        Qc = np.ascontiguousarray(Q, np.float32)
        idx = np.full((Q.shape[0], k), -1, np.int32)
        sim = np.zeros((Q.shape[0], k), np.float32)
        cnt = np.zeros(Q.shape[0], np.int32)
        _lib.check(lib.cslam_pool_search(h, _lib.ptr(Qc), Q.shape[0], k, _lib.ptr(idx), _lib.ptr(sim),
                                         _lib.ptr(cnt)))
        assert np.all(cnt == kk)
        for q in range(Q.shape[0]):
            _check_against_reference(case, idx[q, :kk].tolist(), sim[q, :kk], q)
        lib.cslam_pool_destroy(h)
