"""Broker (next-row N2): vertex covers identical to the reference's (networkx-based)
`cslam/broker.py` on the selections recorded in tests/golden/broker_ref.json, plus the
properties the reference's tests/test_broker.py:62-116 checks (every edge covered, no more
vertices than edges, no duplicates)."""
import json
import os

import numpy as np
import pytest

from cslam_amd.algebraic_connectivity_maximization import EdgeInterRobot
from cslam_amd.broker import Broker

T = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "broker_ref.json")))


def _sel(t):
    return [EdgeInterRobot(int(e[0]), int(e[1]), int(e[2]), int(e[3]), float(e[4]))
            for e in t["selection"]]


@pytest.mark.parametrize("case", [c for c in T if c != "single_robot"])
def test_vertex_cover_matches_reference(case):
    t = T[case]
    b = Broker(_sel(t), t["in_range"])
    assert b.is_bipartite == t["is_bipartite"]
    covers = b.brokerage(True)
    got = [sorted([list(v) for v in c]) for c in covers]
    assert got == t["covers"]
    # properties of the reference's own test: all in-range edges covered, |cover| <= |edges|
    chosen = set().union(*covers) if covers else set()
    in_range = set(t["in_range"])
    nb_edges = 0
    for e in _sel(t):
        if e.robot0_id in in_range and e.robot1_id in in_range:
            nb_edges += 1
            assert (e.robot0_id, e.robot0_keyframe_id) in chosen or \
                   (e.robot1_id, e.robot1_keyframe_id) in chosen
    assert len(chosen) <= nb_edges
    assert sum(len(c) for c in covers) == len(chosen)   # components are disjoint


@pytest.mark.parametrize("case", [c for c in T if c != "single_robot"])
def test_simple_dialog_matches_reference(case):
    t = T[case]
    np.random.seed(99)
    got = [sorted([list(v) for v in c]) for c in Broker(_sel(t), t["in_range"]).brokerage(False)]
    assert got == t["dialog"]


def test_single_robot_has_nothing_to_broker():
    t = T["single_robot"]
    assert Broker(_sel(t), t["in_range"]).brokerage(True) == []
    assert Broker(_sel(t), t["in_range"]).brokerage(False) == []


def test_manual_two_components():
    """tests/test_broker.py:213-265 (manual case): two components, one vertex each suffices."""
    edges = [EdgeInterRobot(0, 1, 1, 1, 1.0), EdgeInterRobot(0, 1, 1, 2, 1.0),
             EdgeInterRobot(0, 1, 1, 3, 1.0), EdgeInterRobot(0, 5, 1, 7, 1.0),
             EdgeInterRobot(0, 6, 1, 7, 1.0)]
    covers = Broker(edges, [0, 1]).brokerage(True)
    assert len(covers) == 2 and sorted(len(c) for c in covers) == [1, 1]
    assert set().union(*covers) == {(0, 1), (1, 7)}
