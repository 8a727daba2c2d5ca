"""BASELINE.json configs[4]: the full front-end stream (CosPlace on PyTorch-ROCm → device kNN
pools → MAC selection → RANSAC verification → fixed edges) at reduced scale, synthetic data.
Functional properties; the reference has no test for this composition."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_full_front_end_stream_small(gpu):
    import cfg5_stream
    r = cfg5_stream.run(keyframes=240, batch=40, dim=128, budget=40, select_every=2, seed=1)
    truth, fixed = r["_truth"], r["_fixed_pairs"]
    assert r["pool_rows"] == [240, 240]
    # place recognition: a revisited place is the best match of its keyframe
    assert r["revisits_seen"] > 20
    assert r["best_match_correct"] >= 0.9 * r["revisits_seen"]
    # selection respects the budget every round; verification separates true from false matches
    assert 0 < r["selected"] <= 40 * (240 // 40 // 2 + 1)
    assert r["verified_ok"] > 0 and r["verified_ok"] == r["verified_ok_true"]
    assert r["rejected_true"] <= 0.1 * max(1, r["verified_ok_true"])
    # every fixed edge is a true loop closure, and fixed edges left the candidate set
    assert all(truth.get(kf1) == kf0 for kf0, kf1 in fixed)
    assert r["fixed_edges"] == len(fixed) == r["verified_ok"]


def test_full_front_end_stream_kitti00_scale(gpu):
    """The same composition at the scale BASELINE.json names: 2 robots x 4541 keyframes
    (KITTI-00), 512-d CosPlace/ResNet18 descriptors, MAC selection every 8 steps with budget 100
    (~6 s on one MI355X, profiles/r01_cfg5_stream.json)."""
    import cfg5_stream
    K = 4541
    r = cfg5_stream.run(keyframes=K, batch=64, dim=512, budget=100, select_every=8, seed=0)
    truth, fixed = r["_truth"], r["_fixed_pairs"]
    assert r["pool_rows"] == [K, K]
    assert r["revisits_seen"] > 500
    assert r["best_match_correct"] >= 0.97 * r["revisits_seen"]
    n_rounds = len(range(7, (K + 63) // 64, 8)) + 1
    assert 0 < r["selected"] <= 100 * n_rounds
    assert r["candidates"] > 2000                       # thousands of candidates before a selection
    assert r["verified_ok"] > 100 and r["verified_ok"] == r["verified_ok_true"]
    assert r["rejected_true"] <= 0.1 * r["verified_ok_true"]
    assert all(truth.get(kf1) == kf0 for kf0, kf1 in fixed)
    assert r["fixed_edges"] == len(fixed) == r["verified_ok"]
