"""BASELINE.json configs[4] — the full front-end stream on one MI355X, synthetic data:

    images ─CosPlace (PyTorch-ROCm)→ descriptors ─(device)→ kNN pools of two robots
           → candidates → MAC selection → RANSAC geometric verification → fixed edges

Two robots drive through `places` distinct places (one keyframe per visit); robot 1 revisits a
random half of robot 0's places, in another order, so those keyframes are true inter-robot loop
closures.  Images are smooth random textures per place + per-visit noise, the CosPlace network
is randomly initialised (no checkpoints in this image), the 3D keypoints handed to the verifier
are consistent for true revisits and unrelated otherwise.  Nothing is read from the host on the
descriptor path; per-stage wall times are reported.

    python tools/cfg5_stream.py [--keyframes 4541] [--batch 64] [--dim 512] [--budget 100]

(4541 keyframes per robot = KITTI-00 scale.)"""
import argparse
import io
import json
import os
import sys
import time
from contextlib import redirect_stdout

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class _Log(object):
    def info(self, *a, **k):
        pass
    error = warn = debug = info


class _Node(object):
    def get_logger(self):
        return _Log()


def place_images(torch, place_ids, gen, device, size=224, noise=0.03):
    """[B, 3, size, size] float images (already normalised): a smooth texture that depends only
    on the place id, plus visit noise."""
    out = []
    for p in place_ids:
        g = torch.Generator(device="cpu").manual_seed(1000003 * int(p) + 17)
        coarse = torch.rand((1, 3, 7, 7), generator=g)
        out.append(coarse)
    x = torch.cat(out).to(device)
    x = torch.nn.functional.interpolate(x, size=(size, size), mode="bicubic", align_corners=False)
    x = x + noise * torch.randn(x.shape, generator=gen, device=device)
    return (x - 0.45) / 0.225


def run(keyframes=600, batch=50, dim=256, budget=60, select_every=4, backbone="ResNet18", seed=0,
        threshold=0.0, verbose=False):
    import torch
    from cslam_amd.algebraic_connectivity_maximization import EdgeInterRobot
    from cslam_amd.geometric_verification import GeometricVerifier
    from cslam_amd.loop_closure_sparse_matching import LoopClosureSparseMatching
    from cslam_amd.vpr import GeoLocalizationNet

    dev = torch.device("cuda", 0)
    torch.manual_seed(seed)
    rs = np.random.RandomState(seed)
    net = GeoLocalizationNet(backbone, dim).to(dev).eval().half()
    gen = torch.Generator(device=dev).manual_seed(seed)

    # itineraries: robot 0 visits places 0..K-1; robot 1 revisits a random half of them
    K = keyframes
    place = [np.arange(K), None]
    revisit = rs.rand(K) < 0.5
    p1 = np.where(revisit, rs.permutation(K), K + np.arange(K))   # >= K: places only robot 1 sees
    place[1] = p1
    truth = {}                                                     # kf of robot 1 → kf of robot 0
    for kf1, p in enumerate(p1):
        if p < K:
            truth[kf1] = int(p)

    params = {"max_nb_robots": 2, "frontend.sensor_type": "stereo", "frontend.similarity_threshold": threshold,
              "frontend.nb_best_matches": 10, "frontend.intra_loop_min_inbetween_keyframes": 10}
    robots = [LoopClosureSparseMatching(dict(params, robot_id=r), _Node()) for r in range(2)]
    verifier = GeometricVerifier(iterations=200, inlier_distance=0.1, min_inliers=20, seed=seed)
    t = {"vpr": 0.0, "knn": 0.0, "mac": 0.0, "verify": 0.0}
    stats = {"candidates": 0, "selected": 0, "verified_ok": 0, "verified_ok_true": 0, "rejected": 0,
             "rejected_true": 0, "best_match_correct": 0, "revisits_seen": 0}
    fixed_pairs = set()

    def sync():
        torch.cuda.synchronize()

    n_steps = (K + batch - 1) // batch
    t_start = time.perf_counter()
    for step in range(n_steps):
        lo, hi = step * batch, min(K, (step + 1) * batch)
        kfs = list(range(lo, hi))
        embs = []
        sync()
        t0 = time.perf_counter()
        with torch.no_grad():
            for r in range(2):
                img = place_images(torch, place[r][lo:hi], gen, dev).half()
                embs.append(net(img).float().contiguous())
        sync()
        t["vpr"] += time.perf_counter() - t0
        t0 = time.perf_counter()
        for r in range(2):
            # local add + match against the mirror of the other robot, then "publish" to the other
            robots[r].add_local_global_descriptors_device(embs[r], kfs)
            edges = robots[1 - r].add_other_robot_global_descriptors_device(r, embs[r], kfs)
            if r == 1:
                for kf1, e in zip(kfs, edges):
                    if kf1 in truth and truth[kf1] < hi:      # its counterpart is already in the pool
                        stats["revisits_seen"] += 1
                        if e is not None and e.robot0_keyframe_id == truth[kf1]:
                            stats["best_match_correct"] += 1
        sync()
        t["knn"] += time.perf_counter() - t0

        if (step + 1) % select_every == 0 or step == n_steps - 1:
            broker = robots[0]                                   # lowest id in range selects
            stats["candidates"] = max(stats["candidates"], len(broker.candidate_selector.candidate_edges))
            t0 = time.perf_counter()
            with redirect_stdout(io.StringIO()):
                selection = broker.select_candidates(budget, {0: True, 1: True})
            t["mac"] += time.perf_counter() - t0
            stats["selected"] += len(selection)
            t0 = time.perf_counter()
            for e in selection:
                is_true = truth.get(e.robot1_keyframe_id) == e.robot0_keyframe_id
                pts = rs.uniform(-10, 10, (150, 3))
                if is_true:
                    ang = rs.uniform(-0.3, 0.3)
                    rot = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1.0]])
                    other = pts @ rot.T + rs.uniform(-1, 1, 3) + rs.normal(0, 0.01, pts.shape)
                    other[rs.rand(150) < 0.3] = rs.uniform(-10, 10, 3)
                else:
                    other = rs.uniform(-10, 10, (150, 3))
                res = verifier.verify_candidate(e, pts, other)
                edge = EdgeInterRobot(res.robot0_id, res.robot0_keyframe_id, res.robot1_id,
                                      res.robot1_keyframe_id, 1.0)
                for rb in robots:                                # both robots hear the result
                    if res.success:
                        rb.candidate_selector.candidate_edges_to_fixed([edge])
                    else:
                        rb.candidate_selector.remove_candidate_edges([edge], failed=True)
                if res.success:
                    stats["verified_ok"] += 1
                    stats["verified_ok_true"] += int(is_true)
                    fixed_pairs.add((e.robot0_keyframe_id, e.robot1_keyframe_id))
                else:
                    stats["rejected"] += 1
                    stats["rejected_true"] += int(is_true)
            t["verify"] += time.perf_counter() - t0
        if verbose and step % 10 == 0:
            print(f"step {step}/{n_steps}", {k: round(v, 3) for k, v in t.items()}, flush=True)
    total = time.perf_counter() - t_start
    out = {"config": {"keyframes_per_robot": K, "batch": batch, "descriptor_dim": dim, "backbone": backbone,
                      "budget": budget, "select_every_steps": select_every, "data": "synthetic"},
           "seconds": {k: round(v, 4) for k, v in t.items()}, "total_seconds": round(total, 3),
           "keyframes_per_second": round(2 * K / total, 1), **stats,
           "fixed_edges": len(robots[0].candidate_selector.fixed_edges),
           "pool_rows": [robots[0].local_nnsm.n, robots[1].local_nnsm.n]}
    out["_fixed_pairs"] = fixed_pairs
    out["_truth"] = truth
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--keyframes", type=int, default=4541)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--dim", type=int, default=512)
    ap.add_argument("--budget", type=int, default=100)
    ap.add_argument("--select-every", type=int, default=8)
    a = ap.parse_args()
    r = run(a.keyframes, a.batch, a.dim, a.budget, a.select_every, verbose=True)
    r = {k: v for k, v in r.items() if not k.startswith("_")}
    print(json.dumps(r), flush=True)
