"""Host bookkeeping of AlgebraicConnectivityMaximization against traces of the reference
(tests/golden/acm_ref.json) and the properties the reference's own
tests/test_algebraic_connectivity.py checks.  Only the branches that never reach the MAC
solver run here (no GPU); the solver branches are in tests/test_mac_gpu.py."""
import json
import os

import numpy as np
import pytest

from cslam_amd.algebraic_connectivity_maximization import (AlgebraicConnectivityMaximization,
                                                           EdgeInterRobot)
from cslam_amd.mac.utils import Edge

T = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "acm_ref.json")))


def E(lst):
    return EdgeInterRobot(int(lst[0]), int(lst[1]), int(lst[2]), int(lst[3]), float(lst[4]))


def as_set(edges):
    return sorted((e.robot0_id, e.robot0_keyframe_id, e.robot1_id, e.robot1_keyframe_id,
                   round(e.weight, 12)) for e in edges)


def test_greedy_branch_matches_reference():
    t = T["greedy_no_fixed"]
    acm = AlgebraicConnectivityMaximization(t["robot_id"], t["max_nb_robots"])
    for op, arg in t["ops"]:
        assert op == "add_match"
        acm.add_match(E(arg))
    sel = acm.select_candidates(t["budget"], {int(k): v for k, v in t["in_range"].items()})
    assert as_set(sel) == as_set([E(x) for x in t["selected"]])
    assert [tuple(x[:4]) for x in map(list, map(tuple, sel))] == [tuple(x[:4]) for x in t["selected"]]
    assert len(acm.candidate_edges) == t["remaining"]


def test_add_match_orientation_quirk():
    t = T["add_match_quirk"]
    acm = AlgebraicConnectivityMaximization(1, 2)
    acm.add_match(EdgeInterRobot(1, 5, 0, 7, 0.9))
    acm.add_match(EdgeInterRobot(1, 5, 0, 7, 0.2))
    assert {str(k): v.weight for k, v in acm.candidate_edges.items()} == t["robot1_first"]
    acm = AlgebraicConnectivityMaximization(0, 2)
    acm.add_match(EdgeInterRobot(0, 7, 1, 5, 0.9))
    acm.add_match(EdgeInterRobot(0, 7, 1, 5, 0.2))
    assert {str(k): v.weight for k, v in acm.candidate_edges.items()} == t["robot0_first"]


def test_connection_biased_greedy_matches_reference():
    t = T["connection_biased"]
    for budget, key in ((t["budget"], "selected"), (500, "selected_big_budget")):
        acm = AlgebraicConnectivityMaximization(t["robot_id"], t["max_nb_robots"])
        acm.set_graph([E(x) for x in t["fixed"]], [E(x) for x in t["cand"]])
        sel = acm.select_candidates(budget, {0: True, 1: True, 2: True})
        assert as_set(sel) == as_set([E(x) for x in t[key]])


def test_edge_equality_and_hash():
    a = EdgeInterRobot(0, 1, 2, 3, 0.5)
    assert a == EdgeInterRobot(2, 3, 0, 1, 0.9) and a == EdgeInterRobot(0, 1, 2, 3, 0.1)
    assert not (a == EdgeInterRobot(0, 1, 2, 4, 0.5))
    assert hash(a) == hash((0, 1, 2, 3, 0.5))
    assert a in [EdgeInterRobot(2, 3, 0, 1, 0.0)]


def test_offsets_and_rekey_round_trip():
    """tests/test_algebraic_connectivity.py:392-524 in spirit: offsets are cumulative over the
    included robots, excluded robots sit at 0, rekey ∘ recover is the identity."""
    rs = np.random.RandomState(3)
    acm = AlgebraicConnectivityMaximization(0, 4)
    cand = []
    for _ in range(200):
        r0, r1 = rs.choice(4, 2, replace=False)
        cand.append(EdgeInterRobot(int(r0), int(rs.randint(30)), int(r1), int(rs.randint(30)),
                                   float(rs.rand())))
    acm.set_graph([], cand)
    inc = {0: True, 1: True, 2: False, 3: True}
    acm.compute_offsets(inc)
    assert acm.offsets[0] == 0 and acm.offsets[1] == acm.nb_poses[0]
    assert acm.offsets[2] == 0 and acm.offsets[3] == acm.nb_poses[0] + acm.nb_poses[1]
    kept = [e for e in cand if inc[e.robot0_id] and inc[e.robot1_id]]
    rek = acm.rekey_edges(cand, inc)
    assert len(rek) == len(kept) and all(isinstance(e, Edge) for e in rek)
    back = acm.recover_inter_robot_edges(rek, inc)
    assert [tuple(e) for e in back] == [tuple(e) for e in kept]
    # odometry covers every known robot, excluded ones at offset 0 (the reference's quirk)
    odom = acm.fill_odometry()
    assert len(odom) == sum(max(acm.nb_poses[r] - 1, 0) for r in range(4))


def test_bookkeeping_add_remove_fix():
    """tests/test_algebraic_connectivity.py:81-319 in spirit."""
    acm = AlgebraicConnectivityMaximization(0, 2)
    c = [EdgeInterRobot(0, i, 1, 2 * i, 0.1 * (i + 1)) for i in range(6)]
    acm.set_graph([EdgeInterRobot(0, 9, 1, 9, 1.0)], list(c))
    assert acm.nb_poses == {0: 10, 1: 11} and acm.initial_fixed_edge_exists == {0: True, 1: True}
    acm.remove_candidate_edges([EdgeInterRobot(1, 4, 0, 2, 0.0)], failed=True)  # reversed orientation
    assert (0, 2, 1, 4) not in acm.candidate_edges and len(acm.candidate_edges) == 5
    acm.add_match(c[2])  # blacklisted: never comes back (acm.py:169-170)
    assert len(acm.candidate_edges) == 5
    moved = [c[0], c[1]]
    acm.candidate_edges_to_fixed(moved)
    assert len(acm.fixed_edges) == 3 and all(e.weight == 1.0 for e in acm.fixed_edges)
    assert moved[0].weight == 1.0  # the caller's list is rewritten in place
    assert len(acm.candidate_edges) == 3
    # keep-max rule for robot0 < robot1
    acm.add_match(EdgeInterRobot(0, 5, 1, 10, 0.01))
    assert acm.candidate_edges[(0, 5, 1, 10)].weight == pytest.approx(0.6)
    acm.add_match(EdgeInterRobot(0, 5, 1, 10, 0.99))
    assert acm.candidate_edges[(0, 5, 1, 10)].weight == 0.99


def test_budget_clamp_and_empty():
    acm = AlgebraicConnectivityMaximization(0, 2)
    assert acm.select_candidates(5, {0: True, 1: True}) == []
    acm.add_match(EdgeInterRobot(0, 1, 1, 1, 0.5))
    acm.add_match(EdgeInterRobot(0, 2, 1, 3, 0.7))
    sel = acm.select_candidates(50, {0: True, 1: True})  # no fixed edge → greedy, clamped to 2
    assert len(sel) == 2 and len(acm.candidate_edges) == 0
    # robot 1 out of range: nothing to select
    acm.add_match(EdgeInterRobot(0, 3, 1, 4, 0.9))
    assert acm.select_candidates(1, {0: True, 1: False}) == []


def test_candidate_store_is_a_dict_that_keeps_its_rows_in_dict_order():
    """`candidate_edges` keeps a float64 row per candidate in dict order (what select_candidates
    reads at 200k-candidate scale): random inserts / overwrites / removals / re-inserts against
    a plain dict."""
    import random
    from cslam_amd.algebraic_connectivity_maximization import _CandidateStore
    rs = random.Random(3)
    plain, store = {}, _CandidateStore()
    for _ in range(30000):
        op = rs.random()
        key = (0, rs.randrange(200), 1, rs.randrange(20))
        if op < 0.55:
            e = EdgeInterRobot(key[0], key[1], key[2], key[3], rs.random())
            plain[key] = e
            store[key] = e
        elif op < 0.9:
            assert plain.pop(key, None) == store.pop(key, None)
        elif plain:
            k2 = next(iter(plain))
            del plain[k2]
            del store[k2]
    assert list(plain) == list(store) and plain == dict(store)
    np.testing.assert_array_equal(store.live_rows(),
                                  np.array([list(v) for v in plain.values()]).reshape(-1, 5))
    acm = AlgebraicConnectivityMaximization(0, 2)
    acm.candidate_edges = dict(plain)                 # plain-dict assignment is adopted
    np.testing.assert_array_equal(acm.candidate_edges.live_rows(), store.live_rows())
    store.clear()
    assert len(store) == 0 and store.live_rows().shape == (0, 5)
