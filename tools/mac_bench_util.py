"""cfg3-scale MAC workload (BASELINE.json configs[2]): 2 robots x 25 000-pose odometry chains,
one fixed inter-robot edge, 200 000 distinct inter-robot candidates with U(0,1) weights,
budget 500.  Used by bench.py (extras) and tests/test_mac_gpu.py; lives outside the product package."""
import io
import time
from contextlib import redirect_stdout

import numpy as np

from cslam_amd.mac.mac import MAC
from cslam_amd.mac.utils import Edge


def cfg3_graph(nb_robots=2, poses_per_robot=25_000, nb_candidates=200_000, seed=0):
    rs = np.random.RandomState(seed)
    n = nb_robots * poses_per_robot
    fixed = []
    for r in range(nb_robots):
        base = r * poses_per_robot
        fixed.extend(Edge(base + k, base + k + 1, 1.0) for k in range(poses_per_robot - 1))
    for r in range(nb_robots - 1):
        fixed.append(Edge(r * poses_per_robot + poses_per_robot - 1,
                          (r + 1) * poses_per_robot + poses_per_robot - 1, 1.0))
    # distinct inter-robot pairs
    r0 = rs.randint(nb_robots, size=nb_candidates * 2)
    r1 = (r0 + 1 + rs.randint(max(nb_robots - 1, 1), size=nb_candidates * 2)) % nb_robots
    a = r0 * poses_per_robot + rs.randint(poses_per_robot, size=nb_candidates * 2)
    b = r1 * poses_per_robot + rs.randint(poses_per_robot, size=nb_candidates * 2)
    lo, hi = np.minimum(a, b), np.maximum(a, b)
    _, first = np.unique(lo.astype(np.int64) * n + hi, return_index=True)
    keep = np.sort(first)[:nb_candidates]
    w = rs.rand(len(keep))
    cand = [Edge(int(i), int(j), float(x)) for i, j, x in zip(a[keep], b[keep], w)]
    return n, fixed, cand


def greedy_w(cand, k):
    w = np.zeros(len(cand))
    w[np.argsort([c.weight for c in cand])[-k:]] = 1.0
    return w


def fiedler_bench(budget=500, reps=5):
    n, fixed, cand = cfg3_graph()
    w0 = greedy_w(cand, budget)
    # untimed first use in this process (code-object load, first launches); "cold" below means
    # "from the pseudo-random start vector", not "first launch"
    mac = MAC(fixed, cand, n)
    mac.evaluate_fiedler_pair(w0)
    mac.close()
    t0 = time.perf_counter()
    mac = MAC(fixed, cand, n)
    t_create = time.perf_counter() - t0
    out = {"mac_graph": f"cfg3: n={n}, fixed={len(fixed)}, candidates={len(cand)}, budget={budget}",
           "mac_create_ms": t_create * 1e3}
    lam, _ = mac.evaluate_fiedler_pair(w0)                 # cold start
    out["fiedler_lambda2"] = lam
    out["fiedler_cold_iters"] = mac.last_iters
    out["fiedler_cold_device_ms"] = mac.stats()[1]
    ts = []
    for _ in range(reps):                                   # warm start from the last vector
        t0 = time.perf_counter()
        mac.evaluate_fiedler_pair(w0)
        ts.append(time.perf_counter() - t0)
    out["fiedler_warm_ms"] = float(np.median(ts)) * 1e3
    out["fiedler_warm_iters"] = mac.last_iters
    mac.close()
    mac = MAC(fixed, cand, n)
    with redirect_stdout(io.StringIO()):
        t0 = time.perf_counter()
        rounded, relaxed, upper = mac.fw_subset(w0, budget, max_iters=20)
        out["fw_subset_20iters_ms"] = (time.perf_counter() - t0) * 1e3
    out["fw_subset_device_ms"] = mac.stats()[1]
    out["fw_subset_solver_iters"] = mac.stats()[0]
    out["fw_lambda2_first_last"] = [float(mac.lambda_trace[0]), float(mac.lambda_trace[-1])]
    mac.close()
    return out


def acm_select_bench(poses_per_robot=25_000, nb_candidates=200_000, budget=500, seed=0):
    """End-to-end `AlgebraicConnectivityMaximization.select_candidates` at cfg3 scale: graph
    assembly on the host + MAC create + 20 Frank-Wolfe iterations + rounding + bookkeeping —
    what the node pays per selection round (cslam/algebraic_connectivity_maximization.py:467-542)."""
    from cslam_amd.algebraic_connectivity_maximization import (AlgebraicConnectivityMaximization,
                                                       EdgeInterRobot)
    rs = np.random.RandomState(seed)
    acm = AlgebraicConnectivityMaximization(robot_id=0, max_nb_robots=2)
    acm.add_fixed_edge(EdgeInterRobot(0, poses_per_robot - 1, 1, poses_per_robot - 1, acm.fixed_weight))
    a = rs.randint(poses_per_robot, size=nb_candidates)
    b = rs.randint(poses_per_robot, size=nb_candidates)
    w = rs.rand(nb_candidates)
    for i in range(nb_candidates):
        acm.add_match(EdgeInterRobot(0, int(a[i]), 1, int(b[i]), float(w[i])))
    n_cand = len(acm.candidate_edges)
    with redirect_stdout(io.StringIO()):
        t0 = time.perf_counter()
        selected = acm.select_candidates(budget, {0: True, 1: True})
        dt = time.perf_counter() - t0
    return {"acm_select_candidates_ms": dt * 1e3, "acm_candidates": n_cand, "acm_selected": len(selected),
            "acm_solver_exceptions": acm.last_solver_exceptions}
