"""Scripted scenarios of the GlobalDescriptorLoopClosureDetection topic surface, written against
NOTHING but the node double of tests/fake_ros.py and duck-typed message classes, so the very same
script drives
  * the reference's own classes (tests/golden/gen_golden.py, build container only) — the traces
    it returns are committed as tests/golden/gdlcd_ref.json, and
  * cslam_amd's classes on the HIP path (tests/test_ros_surface.py, -m gpu), whose traces must
    equal that fixture.
A trace is the ordered list of everything published on the bus (topic + the message's fields)
plus snapshots of the buffers and of the candidate selector.

`make_robot(bus, rid, nb_robots, overrides) -> (node, glcd)` builds one robot of the
implementation under test; `msgs` is the module/namespace with the message classes
(GlobalDescriptor, InterRobotLoopClosure, UInt32, String)."""
import contextlib
import io

BASE = {"neighbor_management.enable_neighbor_monitoring": True,
        "neighbor_management.init_delay_sec": -1.0,
        "neighbor_management.max_heartbeat_delay_sec": 1e9,
        "frontend.similarity_threshold": 0.6}

# log values that depend on the wall clock are recorded by key only
_TIMED_KEYS = {"sparsification_cumulative_computation_time"}


def _ser(msg):
    name = type(msg).__name__
    if name == "GlobalDescriptors":
        return {"descriptors": [[d.robot_id, d.keyframe_id, len(d.descriptor),
                                 float(d.descriptor[0])] for d in msg.descriptors]}
    if name == "GlobalDescriptor":
        return {"robot_id": msg.robot_id, "keyframe_id": msg.keyframe_id, "dim": len(msg.descriptor)}
    if name == "InterRobotMatches":
        return {"robot_id": msg.robot_id,
                "matches": [[m.robot0_id, m.robot0_keyframe_id, m.robot1_id, m.robot1_keyframe_id,
                             float(m.weight)] for m in msg.matches]}
    if name == "LocalDescriptorsRequest":
        return {"keyframe_id": msg.keyframe_id, "matches_robot_id": list(msg.matches_robot_id),
                "matches_keyframe_id": list(msg.matches_keyframe_id)}
    if name == "LocalKeyframeMatch":
        return {"keyframe0_id": msg.keyframe0_id, "keyframe1_id": msg.keyframe1_id}
    if name == "KeyValue":
        return {"key": msg.key, "value": None if msg.key in _TIMED_KEYS else msg.value}
    if name == "InterRobotLoopClosure":
        return {"robot0_id": msg.robot0_id, "robot0_keyframe_id": msg.robot0_keyframe_id,
                "robot1_id": msg.robot1_id, "robot1_keyframe_id": msg.robot1_keyframe_id,
                "success": bool(msg.success)}
    if name == "RobotIdsAndOrigin":
        return {"robots": list(msg.robots.ids), "origins": list(msg.origins.ids)}
    if name in ("UInt32", "String"):
        return {"data": msg.data}
    raise TypeError(f"unexpected message type on the bus: {name}")


def _bus_trace(bus, start=0):
    return [[topic, type(msg).__name__, _ser(msg)] for topic, msg in bus.log[start:]]


def _selector_state(g):
    cs = g.lcm.candidate_selector
    return {"candidates": sorted([list(k) + [float(v.weight)] for k, v in cs.candidate_edges.items()]),
            "fixed": [[e.robot0_id, e.robot0_keyframe_id, e.robot1_id, e.robot1_keyframe_id,
                       float(e.weight)] for e in cs.fixed_edges],
            "considered": sorted(list(k) for k in cs.already_considered_matches)}


def _buffers(g):
    return {"descriptors": list(g.global_descriptors_buffer.keys()),
            "matches": [[k, [e.robot0_id, e.robot0_keyframe_id, e.robot1_id, e.robot1_keyframe_id,
                             float(e.weight)]] for k, e in g.inter_robot_matches_buffer.items()]}


def _heartbeat(bus, msgs, rid):
    bus.publish(f"/r{rid}/cslam/heartbeat", msgs.UInt32(data=rid))


def _feed(bus, msgs, desc, rid, kfs):
    for kf in kfs:
        bus.publish(f"/r{rid}/cslam/processed_global_descriptor",
                    msgs.GlobalDescriptor(keyframe_id=int(kf), robot_id=rid,
                                          descriptor=[float(x) for x in desc[rid, kf]]))


# ------------------------------------------------------------------------------------------------
def chunked_publication(make_robot, msgs, bus_cls, desc):
    """Two robots, 4 descriptors per message: nothing is published while nobody is alive; the
    whole backlog goes out in chunks once robot 1 is heard; buffers are pruned to what somebody
    may still need; a second round sends only the new keyframes; a replayed message is
    deduplicated by the receiver; the receiver's matches land in its own buffer."""
    bus = bus_cls()
    over = dict(BASE, **{"frontend.detection_publication_max_elems_per_msg": 4})
    n0, g0 = make_robot(bus, 0, 2, over)
    n1, g1 = make_robot(bus, 1, 2, over)
    out = {}
    _feed(bus, msgs, desc, 1, range(30))          # robot 1 already has a map of its own
    _feed(bus, msgs, desc, 0, range(10))
    mark = len(bus.log)
    n0.timers[0].fire()
    out["before_heartbeat"] = _bus_trace(bus, mark)
    _heartbeat(bus, msgs, 1)
    mark = len(bus.log)
    n0.timers[0].fire()
    out["first_round"] = _bus_trace(bus, mark)
    out["buffers_after_first_round"] = _buffers(g0)
    out["robot1_after_first_round"] = {"buffers": _buffers(g1), "selector": _selector_state(g1),
                                       "pool_sizes": [int(g1.lcm.local_nnsm.n),
                                                      int(g1.lcm.other_robots_nnsm[0].n)]}
    _feed(bus, msgs, desc, 0, range(10, 13))
    mark = len(bus.log)
    n0.timers[0].fire()
    out["second_round"] = _bus_trace(bus, mark)
    first_msg = [m for t, m in bus.log if t == "/cslam/global_descriptors"][0]
    before = int(g1.lcm.other_robots_nnsm[0].n)
    g1.global_descriptor_callback(first_msg)      # replay
    out["replay_added"] = int(g1.lcm.other_robots_nnsm[0].n) - before
    _heartbeat(bus, msgs, 0)
    mark = len(bus.log)
    n1.timers[1].fire()                           # robot 1 publishes (or withholds) its matches
    out["robot1_matches_round"] = _bus_trace(bus, mark)
    out["robot0_selector_after"] = _selector_state(g0)
    return out


def match_forwarding_three_robots(make_robot, msgs, bus_cls, desc):
    """Three robots.  While only robots 0 and 1 hear each other, 0-1 pairs are withheld (with the
    reference's remove-while-iterating behaviour); once robot 2 is heard too everything pending is
    forwarded and every robot's candidate graph takes the received pairs."""
    bus = bus_cls()
    robots = [make_robot(bus, r, 3, dict(BASE)) for r in range(3)]
    (n0, g0), (n1, g1), (n2, g2) = robots
    out = {}
    _heartbeat(bus, msgs, 0)
    _heartbeat(bus, msgs, 1)
    _feed(bus, msgs, desc, 1, range(40))
    n1.timers[0].fire()                            # robot 1's descriptors reach robots 0 and 2
    _feed(bus, msgs, desc, 0, range(25))           # robot 0 matches against its mirror of robot 1
    out["robot0_buffers"] = _buffers(g0)
    mark = len(bus.log)
    n0.timers[1].fire()
    out["two_in_range"] = _bus_trace(bus, mark)
    _heartbeat(bus, msgs, 2)
    _feed(bus, msgs, desc, 0, range(25, 40))
    mark = len(bus.log)
    n0.timers[1].fire()
    out["three_in_range"] = _bus_trace(bus, mark)
    out["selectors"] = [_selector_state(g) for _, g in robots]
    out["robot0_buffers_after"] = _buffers(g0)
    return out


def broker_requests_and_feedback(make_robot, msgs, bus_cls, desc):
    """Two robots exchange 40 keyframes each; a verified loop closure puts a fixed edge in the
    graph; only the lowest id brokers; its selection (budget 6, MAC) turns into
    LocalDescriptorsRequest messages on the right namespaces (vertex cover); success / failure
    feedback moves candidates to fixed edges / to the blacklist; log keys in order."""
    bus = bus_cls()
    over = dict(BASE, **{"frontend.inter_robot_loop_closure_budget": 6, "evaluation.enable_logs": True})
    robots = [make_robot(bus, r, 2, over) for r in range(2)]
    (n0, g0), (n1, g1) = robots
    out = {}
    _heartbeat(bus, msgs, 0)
    _heartbeat(bus, msgs, 1)
    for step in range(4):
        for rid in (0, 1):
            _feed(bus, msgs, desc, rid, range(step * 10, step * 10 + 10))
        for n, g in robots:
            n.timers[0].fire()
            n.timers[1].fire()
    out["selectors_before"] = [_selector_state(g) for _, g in robots]
    first = sorted(g0.lcm.candidate_selector.candidate_edges.items())[0][1]
    bus.publish("/cslam/inter_robot_loop_closure",
                msgs.InterRobotLoopClosure(robot0_id=first.robot0_id,
                                           robot0_keyframe_id=first.robot0_keyframe_id,
                                           robot1_id=first.robot1_id,
                                           robot1_keyframe_id=first.robot1_keyframe_id, success=True))
    mark = len(bus.log)
    with contextlib.redirect_stdout(io.StringIO()):
        g1.detect_inter()            # not the broker: nothing
        g0.detect_inter()
    out["detect_inter"] = _bus_trace(bus, mark)
    out["selector0_after_selection"] = _selector_state(g0)
    # feedback on the first two selected pairs: one verified, one failed
    reqs = [m for t, m in bus.log[mark:] if type(m).__name__ == "LocalDescriptorsRequest"]
    pairs = []
    for t, m in bus.log[mark:]:
        if type(m).__name__ != "LocalDescriptorsRequest":
            continue
        r = int(t.split("/")[1][1:])
        for rr, kk in zip(m.matches_robot_id, m.matches_keyframe_id):
            pairs.append((r, m.keyframe_id, rr, kk))
    pairs = sorted(set(pairs))[:2]
    mark = len(bus.log)
    for i, (a, ka, b, kb) in enumerate(pairs):
        bus.publish("/cslam/inter_robot_loop_closure",
                    msgs.InterRobotLoopClosure(robot0_id=a, robot0_keyframe_id=ka, robot1_id=b,
                                               robot1_keyframe_id=kb, success=(i == 0)))
    out["feedback"] = _bus_trace(bus, mark)
    out["selectors_after_feedback"] = [_selector_state(g) for _, g in robots]
    out["n_requests"] = len(reqs)
    return out


def intra_robot_and_neighbors(make_robot, msgs, bus_cls, desc):
    """Intra-robot loop closures enabled: LocalKeyframeMatch for every keyframe whose top matches
    hold one far enough and similar enough; the neighbours topic answers with ids and origins."""
    bus = bus_cls()
    over = dict(BASE, **{"frontend.enable_intra_robot_loop_closures": True,
                         "frontend.intra_loop_min_inbetween_keyframes": 10})
    n0, g0 = make_robot(bus, 0, 3, over)
    make_robot(bus, 1, 3, over)
    mark = len(bus.log)
    _feed(bus, msgs, desc, 0, range(60))
    out = {"local_matches": [e for e in _bus_trace(bus, mark) if e[1] == "LocalKeyframeMatch"]}
    _heartbeat(bus, msgs, 1)
    mark = len(bus.log)
    bus.publish("/r0/cslam/get_current_neighbors", msgs.String(data=""))
    out["neighbors"] = _bus_trace(bus, mark)
    out["in_range"] = [[int(k), bool(v)] for k, v in
                       g0.neighbor_manager.check_neighbors_in_range()[0].items()]
    out["is_broker"] = bool(g0.neighbor_manager.local_robot_is_broker())
    return out


SCENARIOS = {"chunked_publication": chunked_publication,
             "match_forwarding_three_robots": match_forwarding_three_robots,
             "broker_requests_and_feedback": broker_requests_and_feedback,
             "intra_robot_and_neighbors": intra_robot_and_neighbors}


def run_all(make_robot, msgs, bus_cls, desc):
    return {name: fn(make_robot, msgs, bus_cls, desc) for name, fn in SCENARIOS.items()}
