"""Topic surface of GlobalDescriptorLoopClosureDetection (SURVEY.md §8b ring 1, next-row N1)
against a node double: topic names, chunked publication, resend window, dedup of received
descriptors, buffer pruning, broker election, LocalDescriptorsRequest emission, loop-closure
feedback, log keys.  The matcher is a stub here (no GPU); tests/test_mac_gpu.py and the -m gpu
scenario below run the same surface on the real HIP path."""
import numpy as np
import pytest

from cslam_amd import ros_compat as rc
from cslam_amd.algebraic_connectivity_maximization import (AlgebraicConnectivityMaximization,
                                                           EdgeInterRobot)
from cslam_amd.global_descriptor_loop_closure_detection import GlobalDescriptorLoopClosureDetection
from cslam_amd.loop_closure_detection_node import PARAMETERS, default_params, flatten_parameters
from fake_ros import FakeBus, FakeNode


class StubLcm(object):
    """Records calls; every other-robot descriptor with an even keyframe id 'matches'."""

    def __init__(self, robot_id, nb_robots):
        self.robot_id = robot_id
        self.candidate_selector = AlgebraicConnectivityMaximization(robot_id, nb_robots)
        self.local, self.other, self.batches = [], [], []
        self.selection = []

    def add_local_global_descriptor(self, emb, kf):
        self.local.append(kf)
        return [EdgeInterRobot(self.robot_id, kf, 1 - self.robot_id, 100 + kf, 0.95)] if kf % 3 == 0 else []

    def match_local_loop_closures(self, emb, kf):
        return (kf - 20, np.array([0.99])) if kf >= 20 else (None, np.array([]))

    def add_other_robot_global_descriptors(self, msgs):
        self.batches.append([m.keyframe_id for m in msgs])
        out = []
        for m in msgs:
            self.other.append((m.robot_id, m.keyframe_id))
            out.append(EdgeInterRobot(self.robot_id, 7, m.robot_id, m.keyframe_id, 0.9)
                       if m.keyframe_id % 2 == 0 else None)
        return out

    def select_candidates(self, budget, in_range):
        self.asked = (budget, dict(in_range))
        return list(self.selection)


def make_robot(bus, rid, nb=2, **over):
    ros_params = dict(PARAMETERS)
    ros_params.update({"robot_id": rid, "max_nb_robots": nb,
                       "neighbor_management.enable_neighbor_monitoring": True,
                       "neighbor_management.init_delay_sec": -1.0,
                       "neighbor_management.max_heartbeat_delay_sec": 1e9,
                       "frontend.detection_publication_max_elems_per_msg": 4})
    ros_params.update(over)
    node = FakeNode(bus, f"/r{rid}", ros_params)
    params = flatten_parameters(node)
    lcm = StubLcm(rid, nb)
    return node, GlobalDescriptorLoopClosureDetection(params, node, lcm=lcm), lcm


def heartbeat(bus, rid):
    bus.publish(f"/r{rid}/cslam/heartbeat", rc.UInt32(data=rid))


def test_parameter_flattening_matches_reference_keys():
    p = default_params()
    assert p["frontend.similarity_threshold"] == 0.9 and p["max_nb_robots"] == 10
    assert "frontend.global_descriptors_topic" not in p       # added by the surface itself
    assert p["frontend.sensor_type"] == "stereo"


def test_topics_and_chunked_publication_with_resend_window():
    bus = FakeBus()
    n0, g0, l0 = make_robot(bus, 0)
    n1, g1, l1 = make_robot(bus, 1)
    assert g0.params["frontend.global_descriptors_topic"] == "/cslam/global_descriptors"
    for kf in range(10):   # descriptors arrive from the C++ front end on the relative topic
        bus.publish("/r0/cslam/processed_global_descriptor",
                    rc.GlobalDescriptor(keyframe_id=kf, robot_id=0, descriptor=[0.1 * kf] * 8))
    assert l0.local == list(range(10)) and l1.local == []
    assert len(g0.global_descriptors_buffer) == 10
    # nobody is alive yet: the timer publishes nothing new (from = latest + 1) and prunes nothing
    n0.timers[0].fire()
    assert bus.published("/cslam/global_descriptors") == []
    heartbeat(bus, 1)      # robot 1 comes in range of robot 0
    n0.timers[0].fire()
    msgs = bus.published("/cslam/global_descriptors")
    assert [len(m.descriptors) for m in msgs] == [4, 4, 2]     # chunks of 4
    assert l1.batches == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9]]  # one device call per message
    assert l0.other == []                                     # own messages are ignored
    # pruning deletes while iterating like the reference (gdlcd.py:176-178): every other stale key
    assert list(g0.global_descriptors_buffer.keys()) == [1, 3, 5, 7, 9]
    # a resend of the same keyframes is deduplicated on the receiving side
    g1.global_descriptor_callback(msgs[0])
    assert l1.batches[-1] == []
    # matches found by robot 1 were buffered: kf 0,2,4,6,8
    assert len(g1.inter_robot_matches_buffer) == 5


def test_matches_publication_filters_pairs_known_to_both_robots():
    bus = FakeBus()
    n0, g0, l0 = make_robot(bus, 0, nb=3)
    n1, g1, l1 = make_robot(bus, 1, nb=3)
    n2, g2, l2 = make_robot(bus, 2, nb=3)
    heartbeat(bus, 1)
    for kf in (0, 3, 6):
        g0.add_global_descriptor_to_map(np.ones(4) * kf, kf)   # kf % 3 == 0 → a match each
    assert len(g0.inter_robot_matches_buffer) == 3
    n0.timers[1].fire()    # only robots 0 and 1 in range, all matches are 0-1 pairs
    out = bus.published("/cslam/inter_robot_matches")
    # the reference removes from the list it iterates: every second element survives
    assert [len(m.matches) for m in out] == [1]
    assert out[0].robot_id == 0 and out[0].matches[0].robot0_keyframe_id == 3
    # robot 1 and 2 take the received match into their candidate graph, robot 0 ignores its own
    assert len(l1.candidate_selector.candidate_edges) == 1
    assert len(l2.candidate_selector.candidate_edges) == 1
    assert len(l0.candidate_selector.candidate_edges) == 0


def test_broker_election_requests_and_feedback():
    bus = FakeBus()
    n0, g0, l0 = make_robot(bus, 0, **{"evaluation.enable_logs": True})
    n1, g1, l1 = make_robot(bus, 1)
    sel = [EdgeInterRobot(0, 4, 1, 9, 0.9), EdgeInterRobot(0, 4, 1, 11, 0.8),
           EdgeInterRobot(0, 6, 1, 13, 0.7)]
    l0.selection = sel
    l1.selection = sel
    requests = []
    for r in (0, 1):
        bus.subscribe(f"/r{r}/cslam/local_descriptors_request", lambda m, r=r: requests.append((r, m)))
    # NB: with no neighbour alive the reference still brokers (its own id is always in the
    # in-range list, gdlcd.py:315): a single robot has nothing to exchange → no requests
    g0.detect_inter()
    assert requests == []
    heartbeat(bus, 0)
    heartbeat(bus, 1)
    g1.detect_inter()          # robot 1 sees robot 0 alive with a lower id: not the broker
    assert requests == [] and not hasattr(l1, "asked")
    g0.detect_inter()
    assert l0.asked == (5, {0: True, 1: True})
    # minimum vertex cover of {(0,4)-(1,9), (0,4)-(1,11), (0,6)-(1,13)}: (0,4) + one of the last edge
    got = sorted((r, m.keyframe_id, tuple(m.matches_robot_id), tuple(m.matches_keyframe_id))
                 for r, m in requests)
    assert (0, 4, (1, 1), (9, 11)) in got and len(got) == 2
    keys = [m.key for m in bus.published("/r0/cslam/log_info")]
    assert {"sparsification_cumulative_computation_time", "nb_vertices_transmitted",
            "nb_matches_selected"} <= set(keys)
    # geometric verification feedback
    l0.candidate_selector.set_graph([], list(sel))
    bus.publish("/cslam/inter_robot_loop_closure",
                rc.InterRobotLoopClosure(robot0_id=0, robot0_keyframe_id=4, robot1_id=1,
                                         robot1_keyframe_id=9, success=True))
    bus.publish("/cslam/inter_robot_loop_closure",
                rc.InterRobotLoopClosure(robot0_id=1, robot0_keyframe_id=13, robot1_id=0,
                                         robot1_keyframe_id=6, success=False))
    cs = l0.candidate_selector
    assert len(cs.fixed_edges) == 1 and cs.fixed_edges[0].weight == cs.fixed_weight
    assert list(cs.candidate_edges) == [(0, 4, 1, 11)]
    assert (0, 6, 1, 13) in cs.already_considered_matches
    assert [m.key for m in bus.published("/r0/cslam/log_info")][-2:] == ["nb_matches", "nb_failed_matches"]


def test_intra_robot_match_publication_and_neighbors_topic():
    bus = FakeBus()
    n0, g0, l0 = make_robot(bus, 0, **{"frontend.enable_intra_robot_loop_closures": True})
    make_robot(bus, 1)
    for kf in (5, 25):
        g0.add_global_descriptor_to_map(np.ones(4), kf)
    out = bus.published("/r0/cslam/local_keyframe_match")
    assert len(out) == 1 and (out[0].keyframe0_id, out[0].keyframe1_id) == (25, 5)
    heartbeat(bus, 1)
    bus.publish("/r0/cslam/get_current_neighbors", rc.String(data=""))
    nb = bus.published("/r0/cslam/current_neighbors")
    assert nb[-1].robots.ids == [1] and nb[-1].origins.ids == [1]


def test_monitoring_disabled_quirk():
    """neighbor_monitor.py:52-53: with monitoring disabled is_alive() returns None (falsy)."""
    bus = FakeBus()
    n0, g0, l0 = make_robot(bus, 0, **{"neighbor_management.enable_neighbor_monitoring": False})
    assert g0.neighbor_manager.neighbors_monitors[1].is_alive() is None
    assert g0.neighbor_manager.check_neighbors_in_range() == ({0: True, 1: False}, [0])
    assert g0.neighbor_manager.local_robot_is_broker()


@pytest.mark.gpu
def test_two_robot_scenario_on_the_hip_path(gpu):
    """Two real GlobalDescriptorLoopClosureDetection instances on one GPU talking over the bus:
    descriptors flow, both directions produce candidates, the broker selects within budget and
    requests local descriptors from the right namespaces."""
    import io
    import os
    from contextlib import redirect_stdout
    desc = np.load(os.path.join(os.path.dirname(__file__), "golden", "lcsm_desc.npz"))["desc"]
    bus = FakeBus()
    robots = []
    for rid in (0, 1):
        ros_params = dict(PARAMETERS)
        ros_params.update({"robot_id": rid, "max_nb_robots": 2,
                           "frontend.similarity_threshold": 0.6,
                           "frontend.inter_robot_loop_closure_budget": 6,
                           "neighbor_management.enable_neighbor_monitoring": True,
                           "neighbor_management.init_delay_sec": -1.0,
                           "neighbor_management.max_heartbeat_delay_sec": 1e9})
        node = FakeNode(bus, f"/r{rid}", ros_params)
        robots.append((node, GlobalDescriptorLoopClosureDetection(flatten_parameters(node), node)))
    heartbeat(bus, 0)
    heartbeat(bus, 1)
    for step in range(4):
        for rid, (node, g) in enumerate(robots):
            for kf in range(step * 10, step * 10 + 10):
                bus.publish(f"/r{rid}/cslam/processed_global_descriptor",
                            rc.GlobalDescriptor(keyframe_id=kf, robot_id=rid,
                                                descriptor=desc[rid, kf].tolist()))
        for node, g in robots:
            node.timers[0].fire()
            node.timers[1].fire()
    c0 = robots[0][1].lcm.candidate_selector
    c1 = robots[1][1].lcm.candidate_selector
    # each robot detects matches in both directions itself; with exactly two robots in range the
    # reference does not forward them (gdlcd.py:251-260), so the two candidate graphs overlap
    # without having to be equal (arrival order decides which keyframe was the best at the time)
    k0, k1 = set(c0.candidate_edges), set(c1.candidate_edges)
    assert len(k0) > 10 and len(k1) > 10 and len(k0 & k1) > 0.5 * min(len(k0), len(k1))
    assert robots[0][1].lcm.local_nnsm.n == 40 and robots[0][1].lcm.other_robots_nnsm[1].n == 40
    # a verified loop closure makes the MAC branch reachable (every robot has a fixed edge)
    first = next(iter(c0.candidate_edges.values()))
    bus.publish("/cslam/inter_robot_loop_closure",
                rc.InterRobotLoopClosure(robot0_id=first.robot0_id,
                                         robot0_keyframe_id=first.robot0_keyframe_id,
                                         robot1_id=first.robot1_id,
                                         robot1_keyframe_id=first.robot1_keyframe_id, success=True))
    requests = []
    for r in (0, 1):
        bus.subscribe(f"/r{r}/cslam/local_descriptors_request", lambda m, r=r: requests.append((r, m)))
    before = len(c0.candidate_edges)
    with redirect_stdout(io.StringIO()):
        robots[1][1].detect_inter()     # not the broker
        assert requests == []
        robots[0][1].detect_inter()
    assert len(c0.candidate_edges) == before - 6 and 1 <= len(requests) <= 6
    covered = set()
    for r, m in requests:
        for rr, kk in zip(m.matches_robot_id, m.matches_keyframe_id):
            covered.add(frozenset([(r, m.keyframe_id), (rr, kk)]))
    assert len(covered) >= 6           # every selected edge has an endpoint that ships features
