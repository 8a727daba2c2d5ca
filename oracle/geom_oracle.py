"""CPU restatement (numpy, fp64) of the geometric-verification stand-in of SURVEY.md §8f row N4.

TEST INFRASTRUCTURE ONLY — see oracle/__init__.py.  Nothing under cslam_amd/ imports this.

PARITY UNPINNED against the reference, by construction: the reference delegates this step to
third-party code that is not under /root/reference — rtabmap `RegistrationVis::computeTransformation`
behind LightGlue matches (call site src/front_end/rgbd_handler.cpp:684-706; rtabmap is a ROS
dependency, no version pinned in the tree) — and holds no test or fixture for it.  What this file
pins instead is the HIP stand-in (cslam_amd/csrc/geom.hip) against an independent restatement of
the SAME algorithm: 3-correspondence RANSAC with the same hypothesis schedule (the splitmix-style
`mix(seed + 3h + j) % n` sample indices), the same degeneracy rules (coincident indices; a sample
edge shorter than 1e-9; collinearity |v_perp| <= 1e-3 |u|), the same inlier rule
(|R x + t - y|^2 <= d^2, fp64 arithmetic on fp32 points), the same winner (most inliers, then
lowest hypothesis index) and the same accept / reject rule as the reference's call site: a null
transform (`t.isNull()`, rgbd_handler.cpp:699-713) = not enough inliers (rtabmap Vis/MinInliers)
or degenerate geometry -> `success = false`; otherwise success with the refined pose.
"""
import numpy as np

M64 = (1 << 64) - 1


def mix(x):
    """geom.hip `mix`: 64-bit finaliser, low 32 bits returned."""
    x &= M64
    x ^= x >> 33
    x = (x * 0xff51afd7ed558ccd) & M64
    x ^= x >> 33
    x = (x * 0xc4ceb9fe1a85ec53) & M64
    x ^= x >> 33
    return x & 0xFFFFFFFF


def sample_indices(seed, h, n):
    return [mix(seed + 3 * h + j) % n for j in range(3)]


def frame_of(p0, p1, p2):
    """Orthonormal frame (columns u, v, w) of a 3-point sample, or None when degenerate."""
    u = p1 - p0
    v = p2 - p0
    nu = np.sqrt(u @ u)
    if not nu > 1e-9:
        return None
    u = u / nu
    v = v - (v @ u) * u
    nv = np.sqrt(v @ v)
    if not nv > 1e-3 * nu:
        return None
    v = v / nv
    w = np.cross(u, v)
    return np.stack([u, v, w], axis=1)


def hypothesis(src, dst, seed, h):
    """(R, t) of hypothesis h, or None."""
    n = src.shape[0]
    i0, i1, i2 = sample_indices(seed, h, n)
    if i0 == i1 or i0 == i2 or i1 == i2:
        return None
    a = src[[i0, i1, i2]].astype(np.float64)
    b = dst[[i0, i1, i2]].astype(np.float64)
    fa, fb = frame_of(*a), frame_of(*b)
    if fa is None or fb is None:
        return None
    r = fb @ fa.T
    return r, b[0] - r @ a[0]


def residual2(r, t, src, dst):
    d = src.astype(np.float64) @ r.T + t - dst.astype(np.float64)
    return np.einsum("ij,ij->i", d, d)


def kabsch(src, dst):
    """Least-squares rigid transform dst ~ R src + t, and the singular values of the covariance."""
    ca, cb = src.mean(axis=0), dst.mean(axis=0)
    hmat = (src - ca).T @ (dst - cb)
    u, sv, vt = np.linalg.svd(hmat)
    d = np.sign(np.linalg.det(vt.T @ u.T))
    r = vt.T @ np.diag([1.0, 1.0, d]) @ u.T
    return r, cb - r @ ca, sv


def ransac_rigid3d(src, dst, iterations, inlier_distance, seed):
    """What `cslam_ransac_rigid3d` returns: (best hypothesis or -1, sample (R, t), inlier mask)."""
    src = np.ascontiguousarray(src, np.float32).reshape(-1, 3)
    dst = np.ascontiguousarray(dst, np.float32).reshape(-1, 3)
    n = src.shape[0]
    d2 = float(inlier_distance) ** 2
    best = (-1, -1)   # (count, -h): most inliers, then lowest h
    best_h = -1
    if n >= 3:
        for h in range(int(iterations)):
            hyp = hypothesis(src, dst, seed, h)
            if hyp is None:
                continue
            cnt = int((residual2(hyp[0], hyp[1], src, dst) <= d2).sum())
            if (cnt, -h) > best:
                best, best_h = (cnt, -h), h
    if best_h < 0:
        return -1, (np.eye(3), np.zeros(3)), np.zeros(n, bool)
    r, t = hypothesis(src, dst, seed, best_h)
    return best_h, (r, t), residual2(r, t, src, dst) <= d2


def compute_transformation(points_from, points_to, iterations=300, inlier_distance=0.1, min_inliers=20, seed=0):
    """The accept / reject ladder of cslam_amd/geometric_verification.py::GeometricVerifier, i.e.
    `!t.isNull()` of rgbd_handler.cpp:699: (success, 4x4 or None, inliers, message)."""
    src = np.ascontiguousarray(points_from, np.float32).reshape(-1, 3)
    dst = np.ascontiguousarray(points_to, np.float32).reshape(-1, 3)
    n = src.shape[0]
    if n < min_inliers or n < 3:
        return False, None, 0, f"Not enough matches ({n} < {min_inliers})"
    best_h, _, mask = ransac_rigid3d(src, dst, iterations, inlier_distance, seed)
    if best_h < 0:
        return False, None, 0, "No valid hypothesis (degenerate geometry)"
    inl = int(mask.sum())
    if inl < max(3, min_inliers):
        return False, None, inl, f"Not enough inliers {inl}/{min_inliers}"
    r, t, sv = kabsch(src[mask].astype(np.float64), dst[mask].astype(np.float64))
    if not sv[1] > 1e-6 * sv[0]:
        return False, None, inl, "Degenerate geometry (collinear inliers)"
    inl = int((np.sqrt(residual2(r, t, src, dst)) <= inlier_distance).sum())
    if inl < min_inliers:
        return False, None, inl, f"Not enough inliers {inl}/{min_inliers}"
    tf = np.eye(4)
    tf[:3, :3], tf[:3, 3] = r, t
    return True, tf, inl, ""
