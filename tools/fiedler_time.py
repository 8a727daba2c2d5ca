"""Cold Fiedler solve and 20 Frank-Wolfe iterations on the cfg3 graph, three times (for before/after
comparisons of solver changes: iteration counts, device ms)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mac_bench_util import fiedler_bench  # noqa: E402
for rep in range(3):
    r = fiedler_bench()
    print(json.dumps({k: r[k] for k in ("fiedler_lambda2", "fiedler_cold_iters", "fiedler_cold_device_ms",
                                        "fw_subset_device_ms", "fw_subset_solver_iters", "fw_lambda2_first_last")}), flush=True)
