"""Pins oracle/mac_oracle.py (numpy/scipy restatement of cslam/mac + networkx tracemin_lu)
against outputs of the reference recorded in tests/golden/mac_ref.npz."""
import os

import numpy as np
import pytest

from oracle import mac_oracle as mo

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "mac_ref.npz"))
CASES = [str(c) for c in G["cases"]]


def _mac(case):
    g = lambda s: G[f"{case}_{s}"]
    total = int(g("nb")[4])
    return mo.MacOracle(g("fi"), g("fj"), g("fw"), g("ci"), g("cj"), g("cw"), total), g


@pytest.mark.parametrize("case", CASES)
def test_laplacian_matches_reference(case):
    mac, g = _mac(case)
    L = mac.combined_laplacian(g("w_init")).tocsr()
    L.sum_duplicates()
    L.sort_indices()
    np.testing.assert_array_equal(L.indptr, g("L0_indptr"))
    np.testing.assert_array_equal(L.indices, g("L0_indices"))
    np.testing.assert_array_equal(L.data, g("L0_data"))


@pytest.mark.parametrize("case", CASES)
def test_fiedler_and_gradient_match_reference(case):
    mac, g = _mac(case)
    lam, v = mac.evaluate_fiedler_pair(g("w_init"))
    assert abs(lam - g("lambda_trace")[0]) < 1e-12
    np.testing.assert_allclose(mo.sign_normalise(v), g("v0"), atol=1e-9)
    np.testing.assert_allclose(mac.grad_from_fiedler(v), g("g0"), atol=1e-12)


@pytest.mark.parametrize("case", CASES)
def test_fw_subset_matches_reference(case):
    mac, g = _mac(case)
    k = int(g("nb")[3])
    trace = []
    rounded, relaxed, upper = mac.fw_subset(g("w_init"), k, max_iters=20, trace=trace)
    np.testing.assert_allclose([t["f"] for t in trace], g("lambda_trace"), atol=1e-12)
    np.testing.assert_array_equal(rounded, g("rounded"))
    np.testing.assert_allclose(relaxed, g("relaxed"), atol=1e-12)
    assert abs(upper - float(g("upper"))) < 1e-12
    # BASELINE.json cfg3 comparator: scipy eigsh on the last iterate agrees to 1e-10
    assert abs(trace[-1]["f"] - float(g("eigsh_lambda_last"))) < 1e-10


def test_disconnected_graph_raises_like_superlu():
    # two chains, no link: the reference's splu raises RuntimeError (acm.py:455 catches it)
    fi = np.array([0, 1, 3, 4]); fj = np.array([1, 2, 4, 5]); fw = np.ones(4)
    mac = mo.MacOracle(fi, fj, fw, np.array([0]), np.array([5]), np.array([0.5]), 6)
    with pytest.raises(RuntimeError):
        mac.evaluate_fiedler_pair(np.zeros(1))
    lam, _ = mac.evaluate_fiedler_pair(np.ones(1))
    assert lam > 0
