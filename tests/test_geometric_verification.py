"""Geometric verification stand-in.  Parity with the REFERENCE is unpinned by
construction (it calls rtabmap, which is not in its tree).  Checked here: functional properties —
a known rigid motion is recovered from noisy correspondences with outliers, garbage is rejected,
the outcome drives candidate → fixed / dropped in AlgebraicConnectivityMaximization like
`receive_inter_robot_loop_closure` does — and, round 4, the HIP kernels against an independent
restatement of the same RANSAC (oracle/geom_oracle.py): winning hypothesis, inlier masks, sample
transform, accept / reject and refined pose on planted, unrelated and degenerate inputs."""
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


# ---- against the oracle (oracle/geom_oracle.py: the same RANSAC restated in numpy) ------------------
def _cases():
    """Three planted motions (different sizes / outlier rates / seeds), a collinear cloud, unrelated
    clouds, and a cloud whose inliers all sit on one line (degenerate refinement)."""
    out = []
    for n, outliers, seed in [(400, 0.4, 0), (60, 0.2, 1), (2000, 0.7, 2)]:
        rs = np.random.RandomState(seed)
        r, t = _rigid(rs)
        src = rs.uniform(-5, 5, (n, 3))
        dst = src @ r.T + t + rs.normal(0, 0.01, (n, 3))
        bad = rs.rand(n) < outliers
        dst[bad] = rs.uniform(-8, 8, (int(bad.sum()), 3))
        out.append((f"planted{n}", src, dst, 600 if outliers > 0.5 else 300, seed))
    line = np.outer(np.linspace(0, 1, 50), [1.0, 2.0, 3.0])
    out.append(("collinear", line, line + 1.0, 300, 0))
    rs = np.random.RandomState(3)
    out.append(("unrelated", rs.uniform(-5, 5, (300, 3)), rs.uniform(-5, 5, (300, 3)), 300, 0))
    # 40 exact inliers on a line + unrelated rest: samples are found, the refinement is degenerate
    src = rs.uniform(-5, 5, (200, 3))
    dst = rs.uniform(-5, 5, (200, 3))
    src[:40] = np.outer(np.linspace(-3, 3, 40), [1.0, 0.5, -0.25])
    dst[:40] = src[:40] + [0.5, -1.0, 2.0]
    out.append(("line_inliers", src, dst, 300, 5))
    return out


@pytest.mark.parametrize("case", range(6))
def test_ransac_equals_the_oracle(gpu, case):
    """The raw C ABI (`cslam_ransac_rigid3d`) against oracle/geom_oracle.py: the same winning
    hypothesis, IDENTICAL inlier masks, the sample transform and the inlier sums to 1e-9 relative;
    then the host ladder: accept / reject identical, refined pose within 1e-6.  (Parity with the
    reference itself is unpinned for this row: rtabmap is third-party, rgbd_handler.cpp:684-706.)"""
    import ctypes
    from cslam_amd import _lib
    from cslam_amd.geometric_verification import GeometricVerifier
    from oracle import geom_oracle as go
    name, src, dst, iters, seed = _cases()[case]
    s32 = np.ascontiguousarray(src, np.float32)
    d32 = np.ascontiguousarray(dst, np.float32)
    n = len(s32)
    sample, sums, mask, best = np.zeros(12), np.zeros(16), np.zeros(n, np.uint8), ctypes.c_int32(-1)
    _lib.check(gpu.cslam_ransac_rigid3d(_lib.ptr(s32), _lib.ptr(d32), n, iters, 0.1, ctypes.c_uint64(seed), 0,
                                        _lib.ptr(sample), _lib.ptr(sums), _lib.ptr(mask), ctypes.byref(best)))
    o_best, (o_r, o_t), o_mask = go.ransac_rigid3d(s32, d32, iters, 0.1, seed)
    assert best.value == o_best, name
    np.testing.assert_array_equal(mask.astype(bool), o_mask)
    if o_best >= 0:
        np.testing.assert_allclose(sample[:9].reshape(3, 3), o_r, rtol=0, atol=1e-12)
        np.testing.assert_allclose(sample[9:], o_t, rtol=0, atol=1e-11)
        a, b = s32[o_mask].astype(np.float64), d32[o_mask].astype(np.float64)
        want = np.concatenate([a.sum(0), b.sum(0), (a.T @ b).reshape(-1), [o_mask.sum()]])
        np.testing.assert_allclose(sums, want, rtol=1e-9, atol=1e-9)
    reg = GeometricVerifier(iterations=iters, seed=seed).compute_transformation(src, dst)
    ok, tf, inl, msg = go.compute_transformation(src, dst, iterations=iters, seed=seed)
    assert reg.success == ok, (name, reg.rejected_msg, msg)
    assert reg.inliers == inl and reg.rejected_msg == msg
    if ok:
        np.testing.assert_allclose(reg.transform, tf, rtol=0, atol=1e-6)
    assert {"collinear": False, "unrelated": False, "line_inliers": False}.get(name, True) == ok
