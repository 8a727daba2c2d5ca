"""Determinism soak of the persistent Fiedler solver / Frank-Wolfe loop: the cfg3 selection repeated
on fresh handles, every lambda trace, relaxed solution and rounded set compared bit for bit with the
first run; then many cold Fiedler solves on one handle (the stamped exchange slots are reused
across thousands of rounds)."""
import io, os, sys, time
from contextlib import redirect_stdout
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mac_bench_util import cfg3_graph, greedy_w
from cslam_amd.mac.mac import MAC
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n, fixed, cand = cfg3_graph()
w0 = greedy_w(cand, 500)
first = None; bad = 0; t0 = time.perf_counter()
for r in range(reps):
    mac = MAC(fixed, cand, n)
    with redirect_stdout(io.StringIO()):
        rounded, relaxed, upper = mac.fw_subset(w0, 500, max_iters=20)
    cur = (np.asarray(mac.lambda_trace).tobytes(), np.asarray(relaxed).tobytes(), np.asarray(rounded).tobytes(), float(upper))
    if first is None: first = cur
    bad += int(cur != first)
    fb = mac.stats()
    mac.close()
print(f"fw_subset x {reps}: {bad} runs differ from the first; {(time.perf_counter() - t0) / reps * 1e3:.1f} ms per run incl. create", flush=True)
ws = [greedy_w(cand, 500), greedy_w(cand, 300)]
seqs = []
for run in range(2):                                         # the same sequence on two fresh handles
    mac = MAC(fixed, cand, n)
    seq = []
    for r in range(reps * 10):
        lam, v = mac.evaluate_fiedler_pair(ws[r % 2])        # alternating graphs: warm starts that re-converge
        seq.append((lam, np.asarray(v).tobytes()))
    seqs.append(seq)
    mac.close()
bad2 = sum(int(a != b) for a, b in zip(*seqs))
print(f"{reps * 10} alternating warm-started Fiedler solves, run twice on fresh handles: {bad2} solves differ between the runs")
sys.exit(1 if bad or bad2 else 0)
