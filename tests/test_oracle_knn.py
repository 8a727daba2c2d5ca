"""Pins oracle/knn_oracle.c against outputs of the reference (cslam/nns_matching.py run on
torch-CPU by tests/golden/gen_golden.py → tests/golden/knn_ref.npz).

Contract: similarities within 1e-6 of the reference's, indices identical
wherever the reference's neighbouring similarities differ by more than 1e-6 — the tolerance the
reference's own test grants (tests/test_sparse_matching.py:75-78)."""
import os

import numpy as np
import pytest

from oracle import knn as oracle_knn

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "knn_ref.npz"))


@pytest.mark.parametrize("case", [str(c) for c in G["cases"]])
def test_oracle_matches_reference(case):
    X, Q, ref_idx, ref_sim, k = (G[f"{case}_{s}"] for s in ("X", "Q", "idx", "sim", "k"))
    k = int(k)
    idx, sim, cnt = oracle_knn.search(X, Q, k)
    kk = min(k, X.shape[0])
    assert np.all(cnt == kk)
    all_sims = oracle_knn.similarities(X, Q)
    for q in range(Q.shape[0]):
        # every reference similarity is reproduced to 1e-6 at the reference's own index
        np.testing.assert_allclose(all_sims[q, ref_idx[q, :kk]], ref_sim[q, :kk], atol=1e-6, rtol=0)
        np.testing.assert_allclose(sim[q, :kk], ref_sim[q, :kk], atol=1e-6, rtol=0)
        assert np.all(sim[q, :kk - 1] >= sim[q, 1:kk])
        for j in range(kk):
            if idx[q, j] != ref_idx[q, j]:
                # only allowed inside a near-tie of the reference
                assert abs(all_sims[q, idx[q, j]] - all_sims[q, ref_idx[q, j]]) <= 1e-6, (case, q, j)


def test_ties_lowest_row_first_and_zero_rows():
    X, Q = G["ties_zero16_X"], G["ties_zero16_Q"]
    idx, sim, _ = oracle_knn.search(X, Q, 5)
    # rows 3 and 9 are bit-identical → exact tie → lower row first
    assert list(idx[0, :2]) == [3, 9] and sim[0, 0] == sim[0, 1]
    s = oracle_knn.similarities(X, Q)
    assert np.all(s[:, 5] == 0.0)  # zero row: clamped norm → similarity exactly 0


def test_empty_and_k_larger_than_n():
    X = np.random.RandomState(0).rand(3, 8).astype(np.float32)
    idx, sim, cnt = oracle_knn.search(X, X[:1], 5)
    assert cnt[0] == 3 and idx.shape == (1, 5) and list(idx[0, 3:]) == [-1, -1]
    idx, sim, cnt = oracle_knn.search(np.zeros((0, 8), np.float32), X[:1], 2)
    assert cnt[0] == 0


def test_euclidean_order_property():
    """tests/test_sparse_matching.py:51-81 on the oracle."""
    rs = np.random.RandomState(0)
    X = rs.rand(100, 100)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    for _ in range(20):
        q = rs.rand(100)
        q /= np.linalg.norm(q)
        ds = np.linalg.norm(q[None, :] - X.astype(np.float32), axis=1)
        nd = np.argsort(ds)
        idx, sims, _ = oracle_knn.search(X, q[None, :], 100)
        for j in range(100):
            if nd[j] != idx[0, j]:
                assert abs(ds[nd[j]] - ds[idx[0, j]]) < 1e-6


@pytest.mark.parametrize("case", [str(c) for c in G["cases"]])
def test_numpy_restatement_of_the_reference_path_matches_reference(case):
    """oracle.knn.search_numpy (what bench.py times as the reference-like CPU path) against the same goldens."""
    X, Q, ref_idx, ref_sim, k = (G[f"{case}_{s}"] for s in ("X", "Q", "idx", "sim", "k"))
    kk = min(int(k), X.shape[0])
    all_sims = oracle_knn.similarities(X, Q)
    for q in range(Q.shape[0]):
        idx, sim = oracle_knn.search_numpy(X, Q[q], int(k))
        np.testing.assert_allclose(sim[:kk], ref_sim[q, :kk], atol=1e-6, rtol=0)
        for j in range(kk):
            if idx[j] != ref_idx[q, j]:
                assert abs(all_sims[q, idx[j]] - all_sims[q, ref_idx[q, j]]) <= 2e-6, (case, q, j)
