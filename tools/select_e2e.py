"""End-to-end latency of AlgebraicConnectivityMaximization.select_candidates at cfg3 scale
(2 robots x 25 000 poses, 200 000 candidates, budget 500), with a cProfile breakdown."""
import cProfile, io, os, pstats, sys, time
from contextlib import redirect_stdout
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cslam_amd.algebraic_connectivity_maximization import AlgebraicConnectivityMaximization, EdgeInterRobot

rs = np.random.RandomState(0)
P, M = 25_000, 200_000
acm = AlgebraicConnectivityMaximization(robot_id=0, max_nb_robots=2)
t0 = time.perf_counter()
acm.add_fixed_edge(EdgeInterRobot(0, P - 1, 1, P - 1, acm.fixed_weight))
a, b, w = rs.randint(P, size=M), rs.randint(P, size=M), rs.rand(M)
for i in range(M):
    acm.add_match(EdgeInterRobot(0, int(a[i]), 1, int(b[i]), float(w[i])))
print("add_match x", M, ":", round(time.perf_counter() - t0, 2), "s; candidates", len(acm.candidate_edges), flush=True)
# first call in the process pays HIP runtime / context start-up: do a tiny one first
warm = AlgebraicConnectivityMaximization(robot_id=0, max_nb_robots=2)
warm.add_fixed_edge(EdgeInterRobot(0, 5, 1, 5, 1.0))
for i in range(8):
    warm.add_match(EdgeInterRobot(0, i, 1, 7 - i, 0.5 + 0.01 * i))
with redirect_stdout(io.StringIO()):
    warm.select_candidates(2, {0: True, 1: True})
pr = cProfile.Profile()
t0 = time.perf_counter()
with redirect_stdout(io.StringIO()):
    pr.enable()
    sel = acm.select_candidates(500, {0: True, 1: True})
    pr.disable()
print("select_candidates:", round(time.perf_counter() - t0, 3), "s; selected", len(sel), flush=True)
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18)
print(s.getvalue()[:3500])
