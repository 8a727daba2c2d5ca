"""cfg3 Frank-Wolfe selection only (for rocprofv3 kernel traces)."""
import io
import os
import sys
import time
from contextlib import redirect_stdout

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mac_bench_util import cfg3_graph, greedy_w  # noqa: E402
from cslam_amd.mac.mac import MAC  # noqa: E402

n, fixed, cand = cfg3_graph()
w0 = greedy_w(cand, 500)
mac = MAC(fixed, cand, n)
for rep in range(2):
    with redirect_stdout(io.StringIO()):
        t0 = time.perf_counter()
        mac.fw_subset(w0, 500, max_iters=20)
        dt = time.perf_counter() - t0
    print("fw_subset ms", dt * 1e3, "device ms", mac.stats()[1], "iters", mac.stats()[0], flush=True)
mac.close()
