"""Geometric verification stand-in.

The reference verifies a candidate by registering the matched 3D keypoints of the two keyframes
with rtabmap (`RegistrationVis::computeTransformation` after LightGlue matching,
src/front_end/rgbd_handler.cpp:317-412, 657-726) and publishes an `InterRobotLoopClosure` with
`success` and the relative pose; `AlgebraicConnectivityMaximization` then turns the candidate into
a fixed edge or drops it (algebraic_connectivity_maximization.py:191-254).  rtabmap is third-party
code outside the reference tree, so parity is UNPINNED by construction: this module is a
functional replacement with rtabmap's parameter vocabulary (Vis/Iterations = 300,
Vis/InlierDistance = 0.1 m, Vis/MinInliers = 20), RANSAC on the GPU + Kabsch refinement."""
import ctypes
from collections import namedtuple

import numpy as np

from . import _lib

Registration = namedtuple("Registration", ["success", "transform", "inliers", "inliers_ratio", "rejected_msg"])

InterRobotLoopClosureResult = namedtuple(
    "InterRobotLoopClosureResult",
    ["robot0_id", "robot0_keyframe_id", "robot1_id", "robot1_keyframe_id", "success", "transform"])


def _kabsch_from_sums(sums):
    """Least-squares rigid transform dst ~ R src + t from sum src, sum dst, sum src_i dst_j, count."""
    cnt = sums[15]
    ca, cb = sums[0:3] / cnt, sums[3:6] / cnt
    h = sums[6:15].reshape(3, 3) - cnt * np.outer(ca, cb)       # sum (a - ca)(b - cb)^T
    u, sv, vt = np.linalg.svd(h)
    d = np.sign(np.linalg.det(vt.T @ u.T))
    r = vt.T @ np.diag([1.0, 1.0, d]) @ u.T
    return r, cb - r @ ca, sv


class GeometricVerifier(object):
    """3D-3D RANSAC registration of matched keypoints on the GPU."""

    def __init__(self, iterations=300, inlier_distance=0.1, min_inliers=20, seed=0, device_id=0):
        self.iterations = int(iterations)
        self.inlier_distance = float(inlier_distance)
        self.min_inliers = int(min_inliers)
        self.seed = int(seed)
        self.device_id = int(device_id)
        self._lib = _lib.load()

    def compute_transformation(self, points_from, points_to):
        """points_from[i] in keyframe `from` matches points_to[i] in keyframe `to` (both [N, 3],
        metres).  Returns Registration(success, 4x4 transform from→to or None, inliers, ratio, msg)."""
        src = np.ascontiguousarray(points_from, dtype=np.float32).reshape(-1, 3)
        dst = np.ascontiguousarray(points_to, dtype=np.float32).reshape(-1, 3)
        if src.shape != dst.shape:
            raise ValueError("points_from and points_to must have the same shape [N, 3]")
        n = src.shape[0]
        if n < self.min_inliers or n < 3:
            return Registration(False, None, 0, 0.0, f"Not enough matches ({n} < {self.min_inliers})")
        sample = np.zeros(12)
        sums = np.zeros(16)
        mask = np.zeros(n, np.uint8)
        best = ctypes.c_int32(-1)
        _lib.check(self._lib.cslam_ransac_rigid3d(
            _lib.ptr(src), _lib.ptr(dst), n, self.iterations, self.inlier_distance,
            ctypes.c_uint64(self.seed), self.device_id, _lib.ptr(sample), _lib.ptr(sums), _lib.ptr(mask),
            ctypes.byref(best)))
        inliers = int(round(sums[15]))
        if best.value < 0:
            return Registration(False, None, 0, 0.0, "No valid hypothesis (degenerate geometry)")
        if inliers < max(3, self.min_inliers):
            return Registration(False, None, inliers, inliers / n,
                                f"Not enough inliers {inliers}/{self.min_inliers}")
        r, t, sv = _kabsch_from_sums(sums)
        if not sv[1] > 1e-6 * sv[0]:   # inliers (nearly) on a line: the rotation about it is free
            return Registration(False, None, inliers, inliers / n, "Degenerate geometry (collinear inliers)")
        # inliers of the refined transform
        err = np.linalg.norm(src.astype(np.float64) @ r.T + t - dst, axis=1)
        inliers = int((err <= self.inlier_distance).sum())
        if inliers < self.min_inliers:
            return Registration(False, None, inliers, inliers / n,
                                f"Not enough inliers {inliers}/{self.min_inliers}")
        tf = np.eye(4)
        tf[:3, :3], tf[:3, 3] = r, t
        return Registration(True, tf, inliers, inliers / n, "")

    def verify_candidate(self, edge, points_from, points_to):
        """Candidate edge (EdgeInterRobot) + its matched keypoints → the record the reference
        publishes on `/cslam/inter_robot_loop_closure` (rgbd_handler.cpp:687-706)."""
        reg = self.compute_transformation(points_from, points_to)
        return InterRobotLoopClosureResult(edge.robot0_id, edge.robot0_keyframe_id, edge.robot1_id,
                                           edge.robot1_keyframe_id, reg.success, reg.transform)
