#!/usr/bin/env python
"""Generates the golden vectors under tests/golden/ by RUNNING THE REFERENCE (nhahn/cslam at
/root/reference) on seeded inputs.  Build-container only: the reference never travels to the
GPU box, only the .npz/.json written here do.

Shims (all outside /root/reference):
  * `numba` stub (cslam/mac/mac.py:12-13 and utils.py:9-10 import it, never use it)
  * NearestNeighborsMatching.device forced to CPU right after construction
    (cslam/nns_matching.py:16 hard-wires 'cuda')
  * fake `node` with get_logger() for LoopClosureSparseMatching (lcsm.py:72)
  * gen_gdlcd only: the ROS 2 packages the node surface imports (rclpy, cslam_common_interfaces,
    std_msgs, diagnostic_msgs, cv2, cv_bridge — none exists in this image) are replaced by
    field-container message classes and empty modules; the surface runs on the node double of
    tests/fake_ros.py.  No arithmetic lives in those packages.

Usage:  python tests/golden/gen_golden.py
"""
import contextlib
import io
import json
import os
import random
import sys
import types
from collections import namedtuple

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
# `--out DIR` (or CSLAM_GOLDEN_OUT) writes the fixtures somewhere else, e.g. to diff a fresh run
# against the committed files; helper modules (fake_ros, ros_scenarios, test_multi_robot) and the
# repo root are looked up next to this script, else under /root/repo
OUT = os.environ.get("CSLAM_GOLDEN_OUT", HERE)
if "--out" in sys.argv:
    OUT = sys.argv[sys.argv.index("--out") + 1]
    os.makedirs(OUT, exist_ok=True)
TESTS = os.path.dirname(HERE) if os.path.exists(os.path.join(os.path.dirname(HERE), "fake_ros.py")) \
    else "/root/repo/tests"
ROOT = os.path.dirname(TESTS)


def _install_shims():
    nb = types.ModuleType("numba")
    nb.jit = lambda *a, **k: (lambda f: f)
    nb.njit = nb.jit
    ex = types.ModuleType("numba.experimental")
    ex.jitclass = lambda *a, **k: (lambda c: c)
    nb.experimental = ex
    sys.modules["numba"] = nb
    sys.modules["numba.experimental"] = ex
    sys.path.insert(0, REF)


_install_shims()
import torch  # noqa: E402
from cslam.algebraic_connectivity_maximization import (  # noqa: E402
    AlgebraicConnectivityMaximization, EdgeInterRobot)
from cslam.loop_closure_sparse_matching import LoopClosureSparseMatching  # noqa: E402
from cslam.mac.mac import MAC  # noqa: E402
from cslam.nns_matching import NearestNeighborsMatching  # noqa: E402

GlobalDescriptor = namedtuple("GlobalDescriptor", ["keyframe_id", "robot_id", "descriptor"])


class _Logger:
    def info(self, *a, **k):
        pass

    debug = error = warn = info


class FakeNode:
    def get_logger(self):
        return _Logger()


def cpu_matcher():
    m = NearestNeighborsMatching()
    m.device = torch.device("cpu")
    return m


# ----------------------------------------------------------------------------------------------
# kNN
# ----------------------------------------------------------------------------------------------
def gen_knn():
    out = {}
    cases = []

    def add_case(name, X, Q, k):
        m = cpu_matcher()
        for i in range(X.shape[0]):
            m.add_item(X[i], i)
        idx = np.full((Q.shape[0], k), -1, np.int32)
        sim = np.full((Q.shape[0], k), -np.inf, np.float32)
        for qi in range(Q.shape[0]):
            items, s = m.search(Q[qi], k)
            idx[qi, :len(items)] = items
            sim[qi, :len(items)] = s
        out[f"{name}_X"] = X
        out[f"{name}_Q"] = Q
        out[f"{name}_idx"] = idx
        out[f"{name}_sim"] = sim
        out[f"{name}_k"] = np.int32(k)
        cases.append(name)

    rs = np.random.RandomState(0)
    X = rs.rand(1000, 128).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    add_case("cfg1_unit128", X, rs.rand(16, 128).astype(np.float32), 10)

    X = rs.rand(100, 100).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    Q = rs.rand(10, 100).astype(np.float32)
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    add_case("test_similarity_100", X, Q, 100)  # tests/test_sparse_matching.py:51-81 shape

    X = rs.rand(96, 4096).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    add_case("cfg2_slice4096", X, rs.rand(4, 4096).astype(np.float32), 10)

    X = (rs.randn(400, 37) * 2.5).astype(np.float32)
    add_case("raw_randn37", X, rs.randn(6, 37).astype(np.float32), 7)

    X = rs.rand(64, 16).astype(np.float32)
    X[5] = 0.0          # zero row → similarity 0 (eps clamp)
    X[9] = X[3]         # duplicate rows → exact tie
    X[40] = 2.0 * X[3]  # scaled duplicate → tie up to rounding
    add_case("ties_zero16", X, np.vstack([X[3], rs.rand(16)]).astype(np.float32), 5)

    out["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(OUT, "knn_ref.npz"), **out)
    print("knn_ref.npz:", cases)


# ----------------------------------------------------------------------------------------------
# MAC
# ----------------------------------------------------------------------------------------------
def build_multi_robot_graph(rs, nb_poses, nb_cand, nb_robots, random_weights=True):
    """tests/test_algebraic_connectivity.py:38-74 with a seeded RNG and generic weights."""
    fixed = [EdgeInterRobot(i, nb_poses - 1, i + 1, nb_poses - 1, 1.0)
             for i in range(nb_robots - 1)]
    cand = {}
    while len(cand) < nb_cand:
        r0 = int(rs.randint(nb_robots))
        r1 = int(rs.choice([r for r in range(nb_robots) if r != r0]))
        w = float(rs.rand()) if random_weights else 1.0
        e = EdgeInterRobot(r0, int(rs.randint(nb_poses)), r1, int(rs.randint(nb_poses)), w)
        if e.robot0_id < e.robot1_id:
            key = (e.robot0_id, e.robot0_keyframe_id, e.robot1_id, e.robot1_keyframe_id)
        else:
            key = (e.robot1_id, e.robot1_keyframe_id, e.robot0_id, e.robot0_keyframe_id)
        cand[key] = e
    return fixed, list(cand.values())


def edges_to_array(edges):
    return np.array([[e.robot0_id, e.robot0_keyframe_id, e.robot1_id, e.robot1_keyframe_id,
                      e.weight] for e in edges], dtype=np.float64).reshape(-1, 5)


def gen_mac():
    out = {}
    cases = []
    specs = [("g3x10", 10, 40, 3, 5), ("g3x100", 100, 300, 3, 20),
             ("g2x1000", 1000, 2000, 2, 100), ("g5x200", 200, 1500, 5, 50)]
    for name, nb_poses, nb_cand, nb_robots, k in specs:
        rs = np.random.RandomState(abs(hash(name)) % (2 ** 31) if False else sum(map(ord, name)))
        fixed, cand = build_multi_robot_graph(rs, nb_poses, nb_cand, nb_robots)
        in_range = {i: True for i in range(nb_robots)}

        # (1) end-to-end select_candidates on the reference
        acm = AlgebraicConnectivityMaximization(0, nb_robots)
        acm.set_graph(list(fixed), list(cand))
        with contextlib.redirect_stdout(io.StringIO()):
            selected = acm.select_candidates(k, dict(in_range))

        # (2) the same problem opened up: rekeyed edge arrays + instrumented fw_subset
        acm2 = AlgebraicConnectivityMaximization(0, nb_robots)
        acm2.set_graph(list(fixed), list(cand))
        inc = acm2.check_graph_disconnections(dict(in_range))
        acm2.compute_offsets(inc)
        rf = acm2.rekey_edges(acm2.fixed_edges, inc)
        rf.extend(acm2.fill_odometry())
        rc = acm2.rekey_edges(acm2.candidate_edges.values(), inc)
        total = sum(acm2.nb_poses.values())
        w_init = acm2.greedy_initialization(k, rc)
        mac = MAC(rf, rc, total)
        trace = []
        orig = mac.evaluate_fiedler_pair

        def spy(w, method="tracemin_lu", tol=1e-8, _orig=orig, _trace=trace):
            f, v = _orig(w, method, tol)
            _trace.append((np.array(w, dtype=np.float64), float(f), np.array(v)))
            return f, v

        mac.evaluate_fiedler_pair = spy
        with contextlib.redirect_stdout(io.StringIO()):
            rounded, relaxed, upper = mac.fw_subset(w_init, k, max_iters=20)
        L0 = mac.combined_laplacian(trace[0][0]).tocsr()
        L0.sum_duplicates()
        L0.sort_indices()
        Llast = mac.combined_laplacian(trace[-1][0]).tocsr()
        Llast.sort_indices()
        v0 = trace[0][2]
        v0 = v0 if v0[np.argmax(np.abs(v0))] >= 0 else -v0
        vl = trace[-1][2]
        vl = vl if vl[np.argmax(np.abs(vl))] >= 0 else -vl
        g0 = mac.grad_from_fiedler(trace[0][2])

        out[f"{name}_nb"] = np.array([nb_poses, nb_cand, nb_robots, k, total], np.int64)
        out[f"{name}_fixed_edges"] = edges_to_array(fixed)
        out[f"{name}_cand_edges"] = edges_to_array(cand)
        out[f"{name}_selected"] = edges_to_array(selected)
        out[f"{name}_fi"] = np.array([e.i for e in rf], np.int32)
        out[f"{name}_fj"] = np.array([e.j for e in rf], np.int32)
        out[f"{name}_fw"] = np.array([e.weight for e in rf], np.float64)
        out[f"{name}_ci"] = np.array([e.i for e in rc], np.int32)
        out[f"{name}_cj"] = np.array([e.j for e in rc], np.int32)
        out[f"{name}_cw"] = np.array([e.weight for e in rc], np.float64)
        out[f"{name}_w_init"] = w_init
        out[f"{name}_lambda_trace"] = np.array([t[1] for t in trace])
        out[f"{name}_w_last"] = trace[-1][0]
        out[f"{name}_L0_indptr"] = L0.indptr.astype(np.int32)
        out[f"{name}_L0_indices"] = L0.indices.astype(np.int32)
        out[f"{name}_L0_data"] = L0.data
        out[f"{name}_v0"] = v0
        out[f"{name}_vlast"] = vl
        out[f"{name}_g0"] = g0
        out[f"{name}_rounded"] = rounded
        out[f"{name}_relaxed"] = relaxed
        out[f"{name}_upper"] = np.float64(upper)
        # cross-check named by BASELINE.json cfg3: scipy eigsh shift-invert on the last L(w)
        import scipy.sparse.linalg as spla
        ev = spla.eigsh(Llast.tocsc(), 2, sigma=-1e-6, which="LM", return_eigenvectors=False)
        out[f"{name}_eigsh_lambda_last"] = np.float64(np.sort(ev)[1])
        cases.append(name)
        print(name, "iters", len(trace), "lambda0", trace[0][1], "last", trace[-1][1], "eigsh",
              float(np.sort(ev)[1]), "selected", len(selected))
    out["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(OUT, "mac_ref.npz"), **out)


# ----------------------------------------------------------------------------------------------
# MAC at BASELINE configs[2] size: the reference's Frank-Wolfe trajectory on the cfg3 graph
# ----------------------------------------------------------------------------------------------
CFG3_TEACHER_ITERATES = (0, 1, 2, 5, 8, 13, 16, 19)
CFG3_TOP = 1200          # gradient entries kept around the top-k boundary (k = 500)


def _sparse(w):
    idx = np.nonzero(w)[0].astype(np.int32)
    return idx, np.asarray(w)[idx].astype(np.float64)


def gen_mac_cfg3():
    """`MAC.fw_subset` (mac.py:191-233) of the REFERENCE on the cfg3 graph (2 x 25 000 poses,
    200 000 candidates, budget 500, 20 iterations, greedy w_init), twice:
      * "ref":   as shipped — networkx TraceMIN-LU stopped at its default tol = 1e-8 (mac.py:35);
      * "tight": the same code with `evaluate_fiedler_pair(w, tol=1e-12)` (the reference's own
                 keyword, mac.py:79): its converged-solver trajectory.
    Kept per run: f_i, u_i, the direction s_i of every iteration, the rounded selection and its
    lambda_2; for "ref" also the iterates w_i (sparse) and the top of the gradient at
    CFG3_TEACHER_ITERATES — what a teacher-forced parity test needs.  ~3 min on one core."""
    sys.path.insert(0, ROOT)
    from tools.mac_bench_util import cfg3_graph, greedy_w
    n, fixed, cand = cfg3_graph()
    k = 500
    w0 = greedy_w(cand, k)
    out = {"nb": np.array([n, len(fixed), len(cand), k, 20], np.int64),
           "graph_checksum": np.array([sum(e.i for e in cand), sum(e.j for e in cand)], np.int64),
           "graph_weight_sum": np.float64(sum(e.weight for e in cand)),
           "teacher_iterates": np.array(CFG3_TEACHER_ITERATES, np.int32)}
    wi, wv = _sparse(w0)
    out["w_init_idx"] = wi
    for tag, tol in (("ref", None), ("tight", 1e-12)):
        mac = MAC(fixed, cand, n)
        trace = []
        orig = mac.evaluate_fiedler_pair

        def spy(w, method="tracemin_lu", tol_=1e-8, _orig=orig, _trace=trace, _tol=tol):
            f, v = _orig(w, method, tol_ if _tol is None else _tol)
            _trace.append((np.array(w, dtype=np.float64), float(f), np.array(v)))
            return f, v

        mac.evaluate_fiedler_pair = spy
        with contextlib.redirect_stdout(io.StringIO()):
            rounded, relaxed, upper = mac.fw_subset(w0, k, max_iters=20)
        assert len(trace) == 20
        u = float("inf")
        s_all, u_all = [], []
        for it, (w, f, v) in enumerate(trace):
            g = mac.grad_from_fiedler(v)
            s = mac.round_solution(g, k)
            u = min(u, f + g @ (s - w))
            s_all.append(np.nonzero(s)[0].astype(np.int32))
            u_all.append(u)
            if tag == "ref" and it in CFG3_TEACHER_ITERATES:
                out[f"ref_w{it}_idx"], out[f"ref_w{it}_val"] = _sparse(w)
                top = np.argsort(-g, kind="stable")[:CFG3_TOP].astype(np.int32)
                out[f"ref_g{it}_idx"] = top
                out[f"ref_g{it}_val"] = g[top]
        assert u == upper
        mac.evaluate_fiedler_pair = orig
        lam_sel = mac.evaluate_objective(rounded)   # lambda_2 of the rounded selection (default tol)
        out[f"{tag}_f"] = np.array([t[1] for t in trace])
        out[f"{tag}_u"] = np.array(u_all)
        out[f"{tag}_s"] = np.stack(s_all)
        out[f"{tag}_rounded_idx"] = np.nonzero(rounded)[0].astype(np.int32)
        out[f"{tag}_relaxed_idx"], out[f"{tag}_relaxed_val"] = _sparse(relaxed)
        out[f"{tag}_upper"] = np.float64(upper)
        out[f"{tag}_lambda2_of_selection"] = np.float64(lam_sel)
        print("mac_cfg3", tag, "f first/last", trace[0][1], trace[-1][1], "lambda2(selection)", lam_sel)
    ov = len(set(out["ref_rounded_idx"]) & set(out["tight_rounded_idx"]))
    print("mac_cfg3: |sel(ref) & sel(tight)| =", ov, "of", k)
    np.savez_compressed(os.path.join(OUT, "mac_cfg3_ref.npz"), **out)


# ----------------------------------------------------------------------------------------------
# AlgebraicConnectivityMaximization bookkeeping traces (quirks of SURVEY.md Appendix A)
# ----------------------------------------------------------------------------------------------
def _edge_list(edges):
    return [[int(e.robot0_id), int(e.robot0_keyframe_id), int(e.robot1_id),
             int(e.robot1_keyframe_id), float(e.weight)] for e in edges]


def gen_acm():
    traces = {}

    # (a) greedy branch: no fixed inter-robot edge (tests/test_sparse_matching.py:110-207 path)
    rs = np.random.RandomState(21)
    acm = AlgebraicConnectivityMaximization(0, 3)
    ops = []
    for _ in range(60):
        r1 = int(rs.randint(1, 3))
        e = EdgeInterRobot(0, int(rs.randint(50)), r1, int(rs.randint(50)), float(rs.rand()))
        acm.add_match(e)
        ops.append(["add_match", _edge_list([e])[0]])
    sel = acm.select_candidates(10, {0: True, 1: True, 2: True})
    traces["greedy_no_fixed"] = dict(robot_id=0, max_nb_robots=3, ops=ops, budget=10,
                                     in_range={"0": True, "1": True, "2": True},
                                     selected=_edge_list(sel),
                                     remaining=len(acm.candidate_edges))

    # (b) add_match orientation quirk (acm.py:564-571 vs :84-89)
    acm = AlgebraicConnectivityMaximization(1, 2)
    acm.add_match(EdgeInterRobot(1, 5, 0, 7, 0.9))
    acm.add_match(EdgeInterRobot(1, 5, 0, 7, 0.2))
    q1 = {str(k): v.weight for k, v in acm.candidate_edges.items()}
    acm = AlgebraicConnectivityMaximization(0, 2)
    acm.add_match(EdgeInterRobot(0, 7, 1, 5, 0.9))
    acm.add_match(EdgeInterRobot(0, 7, 1, 5, 0.2))
    q2 = {str(k): v.weight for k, v in acm.candidate_edges.items()}
    traces["add_match_quirk"] = dict(robot1_first=q1, robot0_first=q2)

    # (c) excluded robot → disconnected L → exception ladder → greedy w_init
    rs = np.random.RandomState(33)
    np.random.seed(1234)  # pseudo_greedy_initialization uses the global RNG (acm.py:236)
    fixed = [EdgeInterRobot(0, 19, 1, 19, 1.0)]
    cand = []
    seen = set()
    while len(cand) < 30:
        e = EdgeInterRobot(0, int(rs.randint(20)), 1, int(rs.randint(20)), float(rs.rand()))
        if (e.robot0_keyframe_id, e.robot1_keyframe_id) in seen:
            continue
        seen.add((e.robot0_keyframe_id, e.robot1_keyframe_id))
        cand.append(e)
    cand.append(EdgeInterRobot(0, 3, 2, 4, 0.5))  # robot 2 known (5 poses) but out of range
    acm = AlgebraicConnectivityMaximization(0, 3)
    acm.set_graph(list(fixed), list(cand))
    with contextlib.redirect_stdout(io.StringIO()):
        sel = acm.select_candidates(5, {0: True, 1: True, 2: False})
    traces["excluded_robot"] = dict(robot_id=0, max_nb_robots=3, fixed=_edge_list(fixed),
                                    cand=_edge_list(cand), budget=5,
                                    in_range={"0": True, "1": True, "2": False},
                                    selected=_edge_list(sel),
                                    remaining=len(acm.candidate_edges))

    # (d) full life-cycle: select → candidate_edges_to_fixed / remove(failed) → select again
    rs = np.random.RandomState(44)
    fixed, cand = build_multi_robot_graph(rs, 60, 200, 3)
    acm = AlgebraicConnectivityMaximization(0, 3)
    acm.set_graph(list(fixed), list(cand))
    rounds = []
    for rd in range(3):
        with contextlib.redirect_stdout(io.StringIO()):
            sel = acm.select_candidates(8, {0: True, 1: True, 2: True})
        ok = [e for i, e in enumerate(sel) if i % 2 == 0]
        bad = [e for i, e in enumerate(sel) if i % 2 == 1]
        rounds.append(dict(selected=_edge_list(sel)))
        acm.candidate_edges_to_fixed(list(ok))
        acm.remove_candidate_edges(list(bad), failed=True)
        rounds[-1]["nb_fixed"] = len(acm.fixed_edges)
        rounds[-1]["nb_cand"] = len(acm.candidate_edges)
    traces["lifecycle"] = dict(robot_id=0, max_nb_robots=3, fixed=_edge_list(fixed),
                               cand=_edge_list(cand), budget=8, rounds=rounds)

    # (e) budget larger than #candidates (acm.py:495-496) and connection-biased greedy with one
    #     robot lacking a fixed edge (acm.py:256-288)
    rs = np.random.RandomState(55)
    fixed = [EdgeInterRobot(0, 9, 1, 9, 1.0)]
    cand = []
    for _ in range(12):
        cand.append(EdgeInterRobot(0, int(rs.randint(10)), 1, int(rs.randint(10)), float(rs.rand())))
    for _ in range(6):
        cand.append(EdgeInterRobot(1, int(rs.randint(10)), 2, int(rs.randint(10)), float(rs.rand())))
    acm = AlgebraicConnectivityMaximization(0, 3)
    acm.set_graph(list(fixed), list(cand))
    sel = acm.select_candidates(4, {0: True, 1: True, 2: True})
    acm_b = AlgebraicConnectivityMaximization(0, 3)
    acm_b.set_graph(list(fixed), list(cand))
    sel_b = acm_b.select_candidates(500, {0: True, 1: True, 2: True})
    traces["connection_biased"] = dict(robot_id=0, max_nb_robots=3, fixed=_edge_list(fixed),
                                       cand=_edge_list(cand), budget=4,
                                       selected=_edge_list(sel),
                                       selected_big_budget=_edge_list(sel_b))

    with open(os.path.join(OUT, "acm_ref.json"), "w") as f:
        json.dump(traces, f)
    print("acm_ref.json:", list(traces))


# ----------------------------------------------------------------------------------------------
# LoopClosureSparseMatching end to end (cfg1-shaped, reduced)
# ----------------------------------------------------------------------------------------------
def gen_lcsm():
    def params(robot_id, nb, thr=0.0):
        return {"robot_id": robot_id, "max_nb_robots": nb, "frontend.sensor_type": "stereo",
                "frontend.similarity_threshold": thr, "frontend.nb_best_matches": 10,
                "frontend.intra_loop_min_inbetween_keyframes": 10,
                "frontend.enable_sparsification": True,
                "evaluation.enable_sparsification_comparison": False}

    def make(p):
        l = LoopClosureSparseMatching(p, FakeNode())
        l.local_nnsm.device = torch.device("cpu")
        for m in l.other_robots_nnsm.values():
            m.device = torch.device("cpu")
        return l

    rs = np.random.RandomState(7)
    nb_robots, nkf, dim = 3, 120, 64
    desc = rs.rand(nb_robots, nkf, dim).astype(np.float32)
    desc /= np.linalg.norm(desc, axis=2, keepdims=True)
    l = make(params(0, nb_robots, thr=0.6))
    events = []
    for kf in range(nkf):
        ms = l.add_local_global_descriptor(desc[0, kf], kf)
        events.append(["local", kf, _edge_list(ms)])
        kf_match, sims = l.match_local_loop_closures(desc[0, kf], kf)
        events.append(["intra", kf, None if kf_match is None else int(kf_match),
                       [float(s) for s in sims]])
        for r in (1, 2):
            m = l.add_other_robot_global_descriptor(
                GlobalDescriptor(kf, r, desc[r, kf].tolist()))
            events.append(["other", r, kf, None if m is None else _edge_list([m])[0]])
    cand = {str(k): float(v.weight) for k, v in l.candidate_selector.candidate_edges.items()}
    sel = l.select_candidates(10, {0: True, 1: True, 2: True})
    np.savez_compressed(os.path.join(OUT, "lcsm_desc.npz"), desc=desc)
    with open(os.path.join(OUT, "lcsm_ref.json"), "w") as f:
        json.dump(dict(params=params(0, nb_robots, thr=0.6), events=events, candidates=cand,
                       selected=_edge_list(sel)), f)
    print("lcsm_ref.json: events", len(events), "candidates", len(cand), "selected", len(sel))


# ----------------------------------------------------------------------------------------------
# BASELINE.json configs[0] ("cfg1"): 2 robots x 1000 keyframes x 128-d, interleaved
# add_local / add_other, threshold 0.0, one fixed edge, select_candidates(100)
# ----------------------------------------------------------------------------------------------
def cfg1_descriptors():
    d = np.random.RandomState(0).rand(2, 1000, 128).astype(np.float32)
    return d / np.linalg.norm(d, axis=2, keepdims=True)


def gen_cfg1():
    def params():
        return {"robot_id": 0, "max_nb_robots": 2, "frontend.sensor_type": "stereo",
                "frontend.similarity_threshold": 0.0, "frontend.nb_best_matches": 10,
                "frontend.intra_loop_min_inbetween_keyframes": 10,
                "frontend.enable_sparsification": True,
                "evaluation.enable_sparsification_comparison": False}
    desc = cfg1_descriptors()
    l = LoopClosureSparseMatching(params(), FakeNode())
    l.local_nnsm.device = torch.device("cpu")
    for m in l.other_robots_nnsm.values():
        m.device = torch.device("cpu")
    matches = []
    for kf in range(1000):
        ms = l.add_local_global_descriptor(desc[0, kf], kf)
        matches.append([[int(x[3]), float(x[4])] for x in ms])
        m = l.add_other_robot_global_descriptor(GlobalDescriptor(kf, 1, desc[1, kf].tolist()))
        matches.append(None if m is None else [int(m[1]), float(m[4])])
    cand = {str(k): float(v.weight) for k, v in l.candidate_selector.candidate_edges.items()}
    l.candidate_selector.add_fixed_edge(EdgeInterRobot(0, 999, 1, 999, 1.0))
    with contextlib.redirect_stdout(io.StringIO()):
        sel = l.select_candidates(100, {0: True, 1: True})
    with open(os.path.join(OUT, "cfg1_ref.json"), "w") as f:
        json.dump(dict(params=params(), matches=matches, candidates=cand,
                       selected=_edge_list(sel)), f)
    print("cfg1_ref.json: candidates", len(cand), "selected", len(sel))


# ----------------------------------------------------------------------------------------------
# Broker (cslam/broker.py): vertex covers of selections, bipartite (Hopcroft-Karp + Koenig via
# networkx) and multi-robot (networkx local-ratio 2-approximation), and the random dialog
# ----------------------------------------------------------------------------------------------
def gen_broker():
    from cslam.broker import Broker
    cases = {}
    specs = [("bip_2robots", 2, 40, 60, 25), ("bip_dense", 2, 12, 80, 40),
             ("multi_3robots", 3, 30, 90, 30), ("multi_5robots", 5, 20, 120, 45),
             ("subset_in_range", 4, 25, 100, 35)]
    for name, nb_robots, nb_poses, nb_cand, nb_sel in specs:
        rs = np.random.RandomState(sum(map(ord, name)))
        _, cand = build_multi_robot_graph(rs, nb_poses, nb_cand, nb_robots)
        sel = [cand[i] for i in rs.permutation(len(cand))[:nb_sel]]
        in_range = list(range(nb_robots)) if name != "subset_in_range" else [0, 2, 3]
        b = Broker(sel, in_range)
        covers = b.brokerage(True)
        np.random.seed(99)
        dialog = Broker(sel, in_range).brokerage(False)
        cases[name] = dict(selection=_edge_list(sel), in_range=in_range,
                           is_bipartite=bool(getattr(b, "is_bipartite", False)),
                           covers=[sorted([list(v) for v in c]) for c in covers],
                           dialog=[sorted([list(v) for v in c]) for c in dialog])
    # single robot in range → no brokerage
    rs = np.random.RandomState(5)
    _, cand = build_multi_robot_graph(rs, 10, 20, 2)
    cases["single_robot"] = dict(selection=_edge_list(cand[:5]), in_range=[0],
                                 covers=Broker(cand[:5], [0]).brokerage(True), dialog=[],
                                 is_bipartite=False)
    with open(os.path.join(OUT, "broker_ref.json"), "w") as f:
        json.dump(cases, f)
    print("broker_ref.json:", {k: [len(c) for c in v["covers"]] for k, v in cases.items()})


def gen_vpr():
    """Aggregation heads of the two VPR front ends, run through the reference's
    own layer classes on CPU: cslam/vpr/cosplace_utils/layers.py (imported by path: it only needs
    torch) and NetVLADLayer of cslam/vpr/netvlad.py (module imports torchvision / PIL / sklearn at
    the top, none of which the layer uses: stubbed for the import when absent)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_cosplace_layers",
                                                  os.path.join(REF, "cslam/vpr/cosplace_utils/layers.py"))
    layers = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(layers)
    for name in ("torchvision", "torchvision.transforms", "torchvision.datasets", "torchvision.models", "PIL",
                 "PIL.Image"):
        try:
            __import__(name)
        except ImportError:
            sys.modules[name] = types.ModuleType(name)
    if not hasattr(sys.modules["torchvision"], "transforms"):
        sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
        sys.modules["torchvision"].datasets = sys.modules["torchvision.datasets"]
        sys.modules["torchvision"].models = sys.modules["torchvision.models"]
    if not hasattr(sys.modules["PIL"], "Image"):
        sys.modules["PIL"].Image = sys.modules["PIL.Image"]
    from cslam.vpr.netvlad import NetVLADLayer

    g = torch.Generator().manual_seed(5)
    out = {}
    # CosPlace head: L2Norm -> GeM -> Flatten -> Linear -> L2Norm (network.py:33-39)
    feats = torch.randn(3, 96, 7, 5, generator=g) * torch.rand(3, 96, 1, 1, generator=g) * 3.0
    fc = torch.nn.Linear(96, 40)
    with torch.no_grad():
        fc.weight.copy_(torch.randn(40, 96, generator=g) * 0.2)
        fc.bias.copy_(torch.randn(40, generator=g) * 0.1)
    gem = layers.GeM()
    with torch.no_grad():
        gem.p.fill_(2.7)
    head = torch.nn.Sequential(layers.L2Norm(), gem, layers.Flatten(), fc, layers.L2Norm()).eval()
    with torch.no_grad():
        out["cos_feats"] = feats.numpy()
        out["cos_fc_w"] = fc.weight.numpy().copy()
        out["cos_fc_b"] = fc.bias.numpy().copy()
        out["cos_p"] = np.float32(2.7)
        out["cos_gem"] = torch.nn.Sequential(layers.L2Norm(), gem, layers.Flatten())(feats).numpy()
        out["cos_out"] = head(feats).numpy()
    # NetVLAD layer, both variants
    for tag, v2 in (("v1", False), ("v2", True)):
        layer = NetVLADLayer(num_clusters=6, dim=24, vladv2=v2).eval()
        with torch.no_grad():
            layer.conv.weight.copy_(torch.randn(6, 24, 1, 1, generator=g))
            if v2:
                layer.conv.bias.copy_(torch.randn(6, generator=g) * 0.3)
            layer.centroids.copy_(torch.rand(6, 24, generator=g))
            x = torch.randn(2, 24, 5, 4, generator=g)
            out[f"vlad_{tag}_x"] = x.numpy()
            out[f"vlad_{tag}_conv_w"] = layer.conv.weight.numpy().copy()
            if v2:
                out[f"vlad_{tag}_conv_b"] = layer.conv.bias.numpy().copy()
            out[f"vlad_{tag}_centroids"] = layer.centroids.numpy().copy()
            out[f"vlad_{tag}_out"] = layer(x).numpy()
    np.savez_compressed(os.path.join(OUT, "vpr_ref.npz"), **out)
    print("vpr_ref.npz", {k: v.shape for k, v in out.items() if hasattr(v, "shape")})


# ----------------------------------------------------------------------------------------------
# ROS surface: the reference's GlobalDescriptorLoopClosureDetection / NeighborManager /
# NeighborMonitor driven through the scripted scenarios of tests/ros_scenarios.py
# ----------------------------------------------------------------------------------------------
# loop_closure_detection_node.py:20-44 — the node's parameter table (names and defaults)
REF_NODE_PARAMETERS = {
    'frontend.similarity_threshold': 0.9, 'frontend.global_descriptor_technique': 'cosplace',
    'frontend.netvlad.pca_checkpoint': '', 'frontend.nn_checkpoint': 'models/resnet18_64.pth',
    'robot_id': 0, 'max_nb_robots': 10, 'frontend.inter_robot_loop_closure_budget': 5,
    'frontend.inter_robot_detection_period_sec': 5, 'frontend.nb_best_matches': 10,
    'frontend.image_crop_size': 376, 'frontend.intra_loop_min_inbetween_keyframes': 10,
    'neighbor_management.max_heartbeat_delay_sec': 5.0, 'neighbor_management.init_delay_sec': 5.0,
    'neighbor_management.heartbeat_period_sec': 0.5,
    'frontend.detection_publication_period_sec': 1.0,
    'frontend.detection_publication_max_elems_per_msg': 10,
    'neighbor_management.enable_neighbor_monitoring': False,
    'frontend.enable_intra_robot_loop_closures': False,
    'frontend.global_descriptors_topic': "global_descriptors",
    'frontend.inter_robot_matches_topic': "inter_robot_matches",
    'frontend.enable_sparsification': True, 'frontend.use_vertex_cover_selection': True,
    'frontend.sensor_type': "stereo", 'evaluation.enable_logs': False,
    'evaluation.enable_sparsification_comparison': False,
}


def _msg_class(name, **fields):
    def __init__(self, **kw):
        for k, d in fields.items():
            setattr(self, k, d() if callable(d) else d)
        for k, v in kw.items():
            if k not in fields:
                raise AttributeError(f"{name} has no field {k}")
            setattr(self, k, v)
    return type(name, (object,), {"__init__": __init__})


def _install_ros_stand_ins():
    """Field containers with the message definitions' field names (cslam_common_interfaces/msg,
    std_msgs, diagnostic_msgs) and empty rclpy / cv2 / cv_bridge modules."""
    def module(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    ids = _msg_class("UInt32MultiArrayLike", ids=list)
    msgs = dict(
        GlobalDescriptor=_msg_class("GlobalDescriptor", keyframe_id=0, robot_id=0, descriptor=list),
        GlobalDescriptors=_msg_class("GlobalDescriptors", descriptors=list),
        InterRobotMatch=_msg_class("InterRobotMatch", robot0_id=0, robot0_keyframe_id=0, robot1_id=0,
                                   robot1_keyframe_id=0, weight=0.0),
        InterRobotMatches=_msg_class("InterRobotMatches", robot_id=0, matches=list),
        InterRobotLoopClosure=_msg_class("InterRobotLoopClosure", robot0_id=0, robot0_keyframe_id=0,
                                         robot1_id=0, robot1_keyframe_id=0, success=False),
        LocalDescriptorsRequest=_msg_class("LocalDescriptorsRequest", keyframe_id=0,
                                           matches_robot_id=list, matches_keyframe_id=list),
        LocalKeyframeMatch=_msg_class("LocalKeyframeMatch", keyframe0_id=0, keyframe1_id=0),
        RobotIdsAndOrigin=_msg_class("RobotIdsAndOrigin", robots=ids, origins=ids),
        KeyframeRGB=_msg_class("KeyframeRGB"), KeyframePointCloud=_msg_class("KeyframePointCloud"))
    pkg = module("cslam_common_interfaces")
    pkg.msg = module("cslam_common_interfaces.msg", **msgs)
    std = module("std_msgs")
    std.msg = module("std_msgs.msg", String=_msg_class("String", data=""),
                     UInt32=_msg_class("UInt32", data=0))
    dia = module("diagnostic_msgs")
    dia.msg = module("diagnostic_msgs.msg", KeyValue=_msg_class("KeyValue", key="", value=""))
    rclpy = module("rclpy")
    rclpy.node = module("rclpy.node", Node=type("Node", (object,), {}))
    rclpy.clock = module("rclpy.clock", Clock=type("Clock", (object,), {"now": lambda self: 0.0}))
    module("cv2")
    module("cv_bridge", CvBridge=type("CvBridge", (object,), {}))
    ns = types.SimpleNamespace(**msgs)
    ns.UInt32 = std.msg.UInt32
    ns.String = std.msg.String
    return ns


def gen_gdlcd():
    sys.path.insert(0, TESTS)                     # tests/: fake_ros, ros_scenarios
    msgs = _install_ros_stand_ins()
    from fake_ros import FakeBus, FakeNode as BusNode
    import ros_scenarios
    from cslam.global_descriptor_loop_closure_detection import GlobalDescriptorLoopClosureDetection

    def make_robot(bus, rid, nb, overrides):
        ros_params = dict(REF_NODE_PARAMETERS)
        ros_params.update({"robot_id": rid, "max_nb_robots": nb})
        ros_params.update(overrides)
        node = BusNode(bus, f"/r{rid}", ros_params)
        # loop_closure_detection_node.py:45-89: the flat dict handed to the surface
        params = {k: v for k, v in ros_params.items()
                  if k not in ('frontend.netvlad.pca_checkpoint', 'frontend.global_descriptors_topic',
                               'frontend.inter_robot_matches_topic')}
        params["frontend.sensor_type"] = params["frontend.sensor_type"].lower()
        g = GlobalDescriptorLoopClosureDetection(params, node)
        g.lcm.local_nnsm.device = torch.device("cpu")
        for m in g.lcm.other_robots_nnsm.values():
            m.device = torch.device("cpu")
        return node, g

    src = OUT if os.path.exists(os.path.join(OUT, "lcsm_desc.npz")) else HERE
    desc = np.load(os.path.join(src, "lcsm_desc.npz"))["desc"]
    traces = ros_scenarios.run_all(make_robot, msgs, FakeBus, desc)
    with open(os.path.join(OUT, "gdlcd_ref.json"), "w") as f:
        json.dump(traces, f)
    print("gdlcd_ref.json:", {k: sorted(v) for k, v in traces.items()})


# ----------------------------------------------------------------------------------------------
# cfg4: R reference LoopClosureSparseMatching instances fed the block stream of
# tests/test_multi_robot.py in the arrival order the sharded protocol documents
# ----------------------------------------------------------------------------------------------
def gen_cfg4():
    sys.path.insert(0, TESTS)
    sys.path.append(ROOT)   # repo root: oracle/ (the block generator's module imports it)
    import test_multi_robot as tm

    def make(robot_id, nb, k, gap):
        p = {"robot_id": robot_id, "max_nb_robots": nb, "frontend.sensor_type": "stereo",
             "frontend.similarity_threshold": tm.THR, "frontend.nb_best_matches": k,
             "frontend.intra_loop_min_inbetween_keyframes": gap,
             "frontend.enable_sparsification": True,
             "evaluation.enable_sparsification_comparison": False}
        l = LoopClosureSparseMatching(p, FakeNode())
        l.local_nnsm.device = torch.device("cpu")
        for m in l.other_robots_nnsm.values():
            m.device = torch.device("cpu")
        return l

    out = {}
    for world, steps, seed, k, gap in ((3, 12, 0, 4, 3), (4, 10, 5, 4, 3), (2, 10, 5, 4, 3)):
        blocks = tm.make_blocks(world, steps, seed=seed)
        robots = [make(r, world, k, gap) for r in range(world)]
        intra = {r: [] for r in range(world)}
        for t in range(steps):
            for r in range(world):                       # (1) deliveries, rank order
                x, kfs = blocks[(t, r)]
                for s in range(world):
                    if s == r:
                        continue
                    for i, kf in enumerate(kfs):
                        robots[s].add_other_robot_global_descriptor(
                            GlobalDescriptor(int(kf), r, x[i].tolist()))
            for r in range(world):                       # (2) own insertions (+ intra-robot match)
                x, kfs = blocks[(t, r)]
                for i, kf in enumerate(kfs):
                    robots[r].add_local_global_descriptor(x[i], int(kf))
                    m, _ = robots[r].match_local_loop_closures(x[i], int(kf))
                    if m is not None:
                        intra[r].append([int(kf), int(m)])
        union = {}
        per_robot = []
        for r in range(world):
            ce = robots[r].candidate_selector.candidate_edges
            per_robot.append(sorted(list(k_) + [float(v.weight)] for k_, v in ce.items()))
            for k_, v in ce.items():
                key = str(tuple(int(a) for a in k_))
                union[key] = max(union.get(key, -1.0), float(v.weight))
        out[f"world{world}_steps{steps}_seed{seed}"] = dict(
            world=world, steps=steps, seed=seed, nb_best_matches=k, min_inbetween=gap,
            threshold=tm.THR, union=union, per_robot=per_robot,
            intra={str(r): v for r, v in intra.items()},
            pool_sizes=[int(robots[r].local_nnsm.n) for r in range(world)])
    with open(os.path.join(OUT, "cfg4_ref.json"), "w") as f:
        json.dump(out, f)
    print("cfg4_ref.json:", {k: (len(v["union"]), sum(len(x) for x in v["intra"].values()))
                             for k, v in out.items()})


def gen_misc():
    """List helpers of the publication path (cslam/utils/misc.py:6-32) on seeded random inputs,
    incl. empty lists, starts outside the list and crossed clamp bounds."""
    from sortedcontainers import SortedDict
    from cslam.utils import misc as ref
    rs = random.Random(5)
    cases = []
    for _ in range(400):
        n = rs.randint(0, 12)
        l = [rs.randint(0, 99) for _ in range(n)]
        start, chunk = rs.randint(-3, 15), rs.randint(1, 5)
        keys = sorted({rs.randint(0, 30) for _ in range(n)})
        d = SortedDict({k: rs.randint(0, 99) for k in keys})
        a, b, c = rs.randint(-5, 5), rs.randint(-5, 5), rs.randint(-5, 5)
        cases.append({"list": l, "start": start, "chunk": chunk,
                      "dict": [[k, d[k]] for k in d.keys()],
                      "clamp_args": [a, b, c], "clamp": ref.clamp(a, b, c),
                      "list_clamp": ref.list_clamp(l, start) if n else None,
                      "list_range": ref.list_range(l, start),
                      "list_chunks": ref.list_chunks(l, start, chunk),
                      "dict_to_list_chunks": ref.dict_to_list_chunks(d, start, chunk)})
    with open(os.path.join(OUT, "misc_ref.json"), "w") as f:
        json.dump(cases, f)
    print("misc_ref.json:", len(cases), "cases")


if __name__ == "__main__":
    random.seed(0)
    np.random.seed(0)
    gen_misc()
    gen_knn()
    gen_mac()
    if "--skip-cfg3" not in sys.argv:
        gen_mac_cfg3()
    gen_acm()
    gen_lcsm()
    gen_broker()
    gen_cfg1()
    gen_vpr()
    gen_cfg4()
    gen_gdlcd()      # last: installs stand-ins for the ROS packages
