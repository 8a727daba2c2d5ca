"""Randomised comparison of the device MAC path with the numpy oracle (which follows the
reference's mac.py + networkx TraceMIN): lambda traces, relaxed solutions and selected sets of
`fw_subset` on random multi-robot pose graphs.  Not part of the suite; run by hand:
    python tools/fuzz_mac.py [cases] [seed]"""
import io
import os
import sys
from contextlib import redirect_stdout

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cslam_amd.mac.mac import MAC  # noqa: E402
from cslam_amd.mac.utils import Edge  # noqa: E402
from oracle import mac_oracle as mo  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad_sel = bad_num = 0
for c in range(cases):
    robots = int(rs.randint(2, 5))
    per = int(rs.choice([10, 40, 150, 600, 1500]))
    n = robots * per
    fixed = []
    for r in range(robots):
        fixed += [Edge(r * per + i, r * per + i + 1, float(rs.choice([1.0, 0.5 + rs.rand()])))
                  for i in range(per - 1)]
    for r in range(robots - 1):   # one fixed inter-robot edge per neighbouring pair → connected
        fixed.append(Edge(r * per + int(rs.randint(per)), (r + 1) * per + int(rs.randint(per)), 1.0))
    m = int(rs.choice([30, 200, 1500]))
    m = min(m, (robots * (robots - 1) // 2) * per * per // 3)   # enough distinct inter-robot pairs exist
    pairs = set()
    while len(pairs) < m:
        a, b = int(rs.randint(n)), int(rs.randint(n))
        if a // per != b // per:
            pairs.add((min(a, b), max(a, b)))
    cand = [Edge(i, j, float(rs.rand())) for i, j in sorted(pairs)]
    k = int(rs.randint(1, max(2, m // 3)))
    w0 = np.zeros(m)
    w0[np.argsort([e.weight for e in cand])[-k:]] = 1.0
    iters = int(rs.choice([3, 10, 20]))
    mac = MAC(fixed, cand, n)
    with redirect_stdout(io.StringIO()):
        r_dev, relaxed_dev, u_dev = mac.fw_subset(w0, k, max_iters=iters)
    trace_dev = np.array(mac.lambda_trace)
    mac.close()
    orc = mo.MacOracle([e.i for e in fixed], [e.j for e in fixed], [e.weight for e in fixed],
                       [e.i for e in cand], [e.j for e in cand], [e.weight for e in cand], n)
    tr = []
    r_orc, relaxed_orc, u_orc = orc.fw_subset(w0, k, max_iters=iters, trace=tr)
    trace_orc = np.array([t["f"] for t in tr])
    num_ok = len(trace_dev) == len(trace_orc) and np.allclose(trace_dev, trace_orc, atol=1e-8) and \
        np.allclose(relaxed_dev, relaxed_orc, atol=1e-8)
    sel_ok = np.array_equal(r_dev, r_orc)
    bad_num += not num_ok
    bad_sel += not sel_ok
    if not (num_ok and sel_ok):
        diff = int((r_dev != r_orc).sum())
        print("DIFF", dict(case=c, n=n, m=m, k=k, iters=iters, num_ok=bool(num_ok), sel_diff=diff,
                           dl=float(np.abs(trace_dev[:len(trace_orc)] - trace_orc[:len(trace_dev)]).max())), flush=True)
    print(f"case {c}/{cases} n={n} m={m} k={k} iters={iters} ok={bool(num_ok and sel_ok)}", flush=True)
print("cases", cases, "numeric mismatches", bad_num, "selection mismatches", bad_sel)
