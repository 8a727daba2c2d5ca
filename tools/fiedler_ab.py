"""A/B of the Fiedler solver variants on the cfg3 graph (env knobs read at cslam_mac_create)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mac_bench_util import fiedler_bench  # noqa: E402

variants = [("graph", {"CSLAM_MAC_PERSISTENT": "0"}),
            ("persist_e4", {"CSLAM_MAC_PERSIST_ELEMS": "4"}),
            ("persist_e2", {"CSLAM_MAC_PERSIST_ELEMS": "2"}),
            ("persist_e1", {"CSLAM_MAC_PERSIST_ELEMS": "1"}),
            ("persist_e8", {"CSLAM_MAC_PERSIST_ELEMS": "8"})]
for name, env in variants:
    for k in ("CSLAM_MAC_PERSISTENT", "CSLAM_MAC_PERSIST_ELEMS", "CSLAM_MAC_PERSIST_BLOCKS"):
        os.environ.pop(k, None)
    os.environ.update(env)
    r = fiedler_bench()
    print(name, json.dumps({k: r[k] for k in ("fiedler_lambda2", "fiedler_cold_iters",
                                              "fiedler_cold_device_ms", "fiedler_warm_ms",
                                              "fw_subset_20iters_ms", "fw_subset_device_ms",
                                              "fw_subset_solver_iters", "fw_lambda2_first_last")}),
          flush=True)
