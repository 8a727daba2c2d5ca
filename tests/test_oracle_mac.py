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


# ---- BASELINE configs[2] size: the oracle against the reference's cfg3 runs -----------------------
def _cfg3():
    from mac_bench_util import cfg3_graph
    ref = np.load(os.path.join(os.path.dirname(__file__), "golden", "mac_cfg3_ref.npz"))
    n, fixed, cand = cfg3_graph()
    f = np.array([(e.i, e.j, e.weight) for e in fixed])
    c = np.array([(e.i, e.j, e.weight) for e in cand])
    assert [int(c[:, 0].sum()), int(c[:, 1].sum())] == ref["graph_checksum"].tolist()
    mac = mo.MacOracle(f[:, 0].astype(int), f[:, 1].astype(int), f[:, 2],
                       c[:, 0].astype(int), c[:, 1].astype(int), c[:, 2], n)
    w0 = np.zeros(len(cand))
    w0[ref["w_init_idx"]] = 1.0
    return ref, mac, int(ref["nb"][3]), w0


def test_cfg3_oracle_reproduces_the_reference_trajectory():
    """At n = 50 000, m = 200 000 the oracle IS the reference: the first iterations of its free run
    (default tol, mac.py:35) reproduce f_i to the last bit, the directions s_i and the dual bounds."""
    ref, mac, k, w0 = _cfg3()
    trace = []
    mac.fw_subset(w0, k, max_iters=4, trace=trace)
    for it, t in enumerate(trace):
        assert t["f"] == ref["ref_f"][it], it
        np.testing.assert_array_equal(np.nonzero(t["s"])[0], ref["ref_s"][it])
        assert abs(t["u"] - ref["ref_u"][it]) <= 1e-15
    # where the reference's default-tolerance run parts from its converged self: iteration 1
    assert np.array_equal(ref["ref_s"][0], ref["tight_s"][0])
    assert len(set(ref["ref_s"][1].tolist()) & set(ref["tight_s"][1].tolist())) == 498


def test_cfg3_oracle_teacher_forced_late_iterate_and_converged_solver():
    """A late iterate of the reference's run (w_16, 8000 non-zeros) through the oracle at the default
    tol (bit-equal f, same top of the gradient) and at tol = 1e-12 (the converged solver: lambda_2
    moves by < 1e-12, the gradient at the cut by less than the band the GPU test grants)."""
    ref, mac, k, _ = _cfg3()
    w = np.zeros(len(mac.weights))
    w[ref["ref_w16_idx"]] = ref["ref_w16_val"]
    lam, v = mac.evaluate_fiedler_pair(w)
    assert lam == ref["ref_f"][16]
    g = mac.grad_from_fiedler(v)
    gi, gv = ref["ref_g16_idx"], ref["ref_g16_val"]
    np.testing.assert_array_equal(g[gi], gv)
    lam12, v12 = mac.find_fiedler_pair(mac.combined_laplacian(w), tol=1e-12)
    assert abs(lam12 - lam) < 1e-12
    g12 = mac.grad_from_fiedler(v12)
    assert np.max(np.abs(g12[gi] - gv)) <= 5e-3 * gv[k - 1]
