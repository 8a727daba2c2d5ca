"""Geometric verification stand-in.  Parity is unpinned by construction (the
reference calls rtabmap, which is not in its tree): functional properties only — a known rigid
motion is recovered from noisy correspondences with outliers, garbage is rejected, the outcome
drives candidate → fixed / dropped in AlgebraicConnectivityMaximization like
`receive_inter_robot_loop_closure` does."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rigid(rs):
    q = rs.randn(4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    r = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    return r, rs.uniform(-2, 2, 3)


@pytest.mark.parametrize("n,outliers,seed", [(400, 0.4, 0), (60, 0.2, 1), (2000, 0.7, 2)])
def test_recovers_known_motion_with_outliers(gpu, n, outliers, seed):
    from cslam_amd.geometric_verification import GeometricVerifier
    rs = np.random.RandomState(seed)
    r, t = _rigid(rs)
    src = rs.uniform(-5, 5, (n, 3))
    dst = src @ r.T + t + rs.normal(0, 0.01, (n, 3))
    bad = rs.rand(n) < outliers
    dst[bad] = rs.uniform(-8, 8, (int(bad.sum()), 3))
    reg = GeometricVerifier(iterations=600 if outliers > 0.5 else 300, seed=seed).compute_transformation(src, dst)
    assert reg.success, reg.rejected_msg
    assert abs(reg.inliers - int((~bad).sum())) <= max(3, 0.02 * n)
    np.testing.assert_allclose(reg.transform[:3, :3], r, atol=5e-3)
    np.testing.assert_allclose(reg.transform[:3, 3], t, atol=2e-2)
    assert abs(np.linalg.det(reg.transform[:3, :3]) - 1.0) < 1e-9
    # deterministic for a seed
    again = GeometricVerifier(iterations=600 if outliers > 0.5 else 300, seed=seed).compute_transformation(src, dst)
    np.testing.assert_array_equal(again.transform, reg.transform)


def test_rejects_unrelated_point_sets_and_tiny_inputs(gpu):
    from cslam_amd.geometric_verification import GeometricVerifier
    rs = np.random.RandomState(3)
    v = GeometricVerifier()
    reg = v.compute_transformation(rs.uniform(-5, 5, (300, 3)), rs.uniform(-5, 5, (300, 3)))
    assert not reg.success and reg.transform is None and "inliers" in reg.rejected_msg
    reg = v.compute_transformation(np.zeros((5, 3)), np.zeros((5, 3)))
    assert not reg.success and "Not enough matches" in reg.rejected_msg
    # degenerate geometry (all points on a line): no valid sample
    line = np.outer(np.linspace(0, 1, 50), [1.0, 2.0, 3.0])
    reg = v.compute_transformation(line, line + 1.0)
    assert not reg.success
    with pytest.raises(ValueError):
        v.compute_transformation(np.zeros((4, 3)), np.zeros((5, 3)))


def test_verification_outcome_moves_candidates_like_the_reference(gpu):
    """success → the candidate becomes a fixed edge, failure → it is dropped
    (algebraic_connectivity_maximization.py:191-254 via gdlcd.receive_inter_robot_loop_closure)."""
    from cslam_amd.algebraic_connectivity_maximization import (AlgebraicConnectivityMaximization,
                                                               EdgeInterRobot)
    from cslam_amd.geometric_verification import GeometricVerifier
    rs = np.random.RandomState(4)
    acm = AlgebraicConnectivityMaximization(robot_id=0, max_nb_robots=2)
    good, bad = EdgeInterRobot(0, 3, 1, 7, 0.9), EdgeInterRobot(0, 5, 1, 2, 0.8)
    acm.add_match(good)
    acm.add_match(bad)
    r, t = _rigid(rs)
    src = rs.uniform(-3, 3, (200, 3))
    v = GeometricVerifier()
    res_good = v.verify_candidate(good, src, src @ r.T + t + rs.normal(0, 0.005, (200, 3)))
    res_bad = v.verify_candidate(bad, src, rs.uniform(-3, 3, (200, 3)))
    assert res_good.success and not res_bad.success
    for res in (res_good, res_bad):        # what receive_inter_robot_loop_closure does (gdlcd.py:432-467)
        edge = EdgeInterRobot(res.robot0_id, res.robot0_keyframe_id, res.robot1_id, res.robot1_keyframe_id,
                              acm.fixed_weight)
        if res.success:
            acm.candidate_edges_to_fixed([edge])
        else:
            acm.remove_candidate_edges([edge], failed=True)
    assert [acm.edge_key(e) for e in acm.fixed_edges] == [(0, 3, 1, 7)]
    assert acm.fixed_edges[0].weight == acm.fixed_weight
    assert not acm.candidate_edges
    # both are blacklisted: a later match of the same pair is ignored
    acm.add_match(EdgeInterRobot(0, 5, 1, 2, 0.99))
    assert not acm.candidate_edges
