import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CSLAM_MAC_PROF"] = "1"
from mac_bench_util import cfg3_graph, greedy_w
from cslam_amd.mac.mac import MAC
n, fixed, cand = cfg3_graph()
w0 = greedy_w(cand, 500)
for e in sys.argv[1:] or ["2"]:
    os.environ["CSLAM_MAC_PERSIST_ELEMS"] = e
    mac = MAC(fixed, cand, n)
    lam, _ = mac.evaluate_fiedler_pair(w0)
    print("E", e, "lambda", lam, "iters", mac.last_iters, "device ms", mac.stats()[1], flush=True)
    mac.close()
