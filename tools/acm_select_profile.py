"""cProfile of AlgebraicConnectivityMaximization.select_candidates at cfg3 scale (199 972 candidates, budget 500):
where the host time around the ~23 ms of Frank-Wolfe goes."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from contextlib import redirect_stdout
import numpy as np
from cslam_amd.algebraic_connectivity_maximization import AlgebraicConnectivityMaximization, EdgeInterRobot
rs = np.random.RandomState(0)
P, N = 25000, 200000
acm = AlgebraicConnectivityMaximization(robot_id=0, max_nb_robots=2)
acm.add_fixed_edge(EdgeInterRobot(0, P - 1, 1, P - 1, acm.fixed_weight))
a = rs.randint(P, size=N); b = rs.randint(P, size=N); w = rs.rand(N)
for i in range(N):
    acm.add_match(EdgeInterRobot(0, int(a[i]), 1, int(b[i]), float(w[i])))
for rep in range(3):
    with redirect_stdout(io.StringIO()):
        t0 = time.perf_counter(); sel = acm.select_candidates(500, {0: True, 1: True}); dt = time.perf_counter() - t0
    print(f"select_candidates: {dt * 1e3:.2f} ms, {len(sel)} selected", flush=True)
    # put the selected edges back so that every repetition sees the same problem
    for e in sel:
        acm.candidate_edges[acm.edge_key(e)] = e
pr = cProfile.Profile()
with redirect_stdout(io.StringIO()):
    pr.enable(); sel = acm.select_candidates(500, {0: True, 1: True}); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18); print(s.getvalue()[:5000])
