import os, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
os.environ["CSLAM_MAC_CREATE_PROF"] = "1"
import numpy as np
from mac_bench_util import cfg3_graph, greedy_w, acm_select_bench
from cslam_amd.mac.mac import MAC
n, fixed, cand = cfg3_graph()
for rep in range(3):
    t0 = time.perf_counter(); mac = MAC(fixed, cand, n); t1 = time.perf_counter()
    print("MAC(...) from Edge lists: %.2f ms" % ((t1 - t0) * 1e3), flush=True)
    mac.close()
fi = np.array([e.i for e in fixed]); fj = np.array([e.j for e in fixed]); fw = np.array([e.weight for e in fixed])
ci = np.array([e.i for e in cand]); cj = np.array([e.j for e in cand]); cw = np.array([e.weight for e in cand])
for rep in range(3):
    t0 = time.perf_counter(); mac = MAC.from_arrays(fi, fj, fw, ci, cj, cw, n); t1 = time.perf_counter()
    print("MAC.from_arrays: %.2f ms" % ((t1 - t0) * 1e3), flush=True)
    w0 = greedy_w(cand, 500)
    t0 = time.perf_counter(); r = mac.fw_subset(w0, 500, max_iters=20); t1 = time.perf_counter()
    print("fw_subset: %.2f ms" % ((t1 - t0) * 1e3), flush=True)
    mac.close()
print(acm_select_bench())
