"""Selection parity of the Frank-Wolfe loop at BASELINE configs[2] size, measured.

Compares the HIP `MAC.fw_subset` on the cfg3 graph (tools/mac_bench_util.py::cfg3_graph) with the
REFERENCE's own runs recorded in tests/golden/mac_cfg3_ref.npz (tests/golden/gen_golden.py::
gen_mac_cfg3: cslam/mac/mac.py:191-233 at its default solver tol 1e-8 = "ref", and with
`evaluate_fiedler_pair(tol=1e-12)` = "tight"):
  * free-running: full lambda trace of both, first iteration whose direction s_i differs,
    |s_i(hip) & s_i(reference)| per iteration, overlap of the rounded selections, lambda_2 of each
    rounded selection;
  * teacher-forced: on the reference's iterates w_i the HIP lambda_2 and top-k mask, mismatches
    classified by their relative distance to the k-th gradient value.
Usage:  python tools/mac_cfg3_parity.py [--json out.json]      (needs a GPU; the fixture travels)
        CSLAM_MAC_TOL_FACTOR=1e-3 python tools/mac_cfg3_parity.py   (solver tolerance A/B)
"""
import io
import json
import os
import sys
from contextlib import redirect_stdout

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FIXTURE = os.path.join(ROOT, "tests", "golden", "mac_cfg3_ref.npz")


def dense(idx, val, m):
    w = np.zeros(m)
    w[idx] = val
    return w


def load_problem():
    from tools.mac_bench_util import cfg3_graph
    ref = np.load(FIXTURE)
    n, fixed, cand = cfg3_graph()
    nb = ref["nb"]
    assert (n, len(fixed), len(cand)) == (int(nb[0]), int(nb[1]), int(nb[2])), "cfg3 graph differs"
    assert [sum(e.i for e in cand), sum(e.j for e in cand)] == ref["graph_checksum"].tolist()
    k = int(nb[3])
    w0 = np.zeros(len(cand))
    w0[ref["w_init_idx"]] = 1.0
    return ref, n, fixed, cand, k, w0


def free_running(mac, ref, k, w0, iters=20):
    mac.trace_directions = True
    with redirect_stdout(io.StringIO()):
        rounded, relaxed, upper = mac.fw_subset(w0, k, max_iters=iters)
    out = {"hip_lambda_trace": [float(x) for x in mac.lambda_trace],
           "hip_upper": float(upper),
           "hip_solver_iters": [int(x) for x in mac.solver_iters_trace]}
    sel = set(np.nonzero(rounded)[0].tolist())
    for tag in ("ref", "tight"):
        f = ref[f"{tag}_f"]
        s = ref[f"{tag}_s"]
        ov = [len(set(mac.direction_trace[i].tolist()) & set(s[i].tolist())) for i in range(len(f))]
        first = next((i for i, o in enumerate(ov) if o != k), None)
        out[tag] = {
            "lambda_trace": [float(x) for x in f],
            "lambda_abs_diff": [float(abs(a - b)) for a, b in zip(mac.lambda_trace, f)],
            "direction_overlap": ov,
            "first_differing_direction": first,
            "selection_overlap": len(sel & set(ref[f"{tag}_rounded_idx"].tolist())),
            "upper": float(ref[f"{tag}_upper"]),
            "lambda2_of_selection": float(ref[f"{tag}_lambda2_of_selection"]),
        }
    out["hip_lambda2_of_selection"] = float(mac.evaluate_objective(rounded))
    # the reference's selections evaluated by the HIP solver (same yardstick for all three)
    m = len(w0)
    for tag in ("ref", "tight"):
        w = np.zeros(m)
        w[ref[f"{tag}_rounded_idx"]] = 1.0
        out[tag]["lambda2_of_selection_by_hip"] = float(mac.evaluate_objective(w))
    return out, rounded


def teacher_forced(mac, ref, k, m):
    rows = []
    for it in ref["teacher_iterates"].tolist():
        w = dense(ref[f"ref_w{it}_idx"], ref[f"ref_w{it}_val"], m)
        lam, v = mac.evaluate_fiedler_pair(w)
        g = mac.grad_from_fiedler(v)
        mask = mac.round_solution(g, k) > 0.5
        gi, gv = ref[f"ref_g{it}_idx"], ref[f"ref_g{it}_val"]
        kth = gv[k - 1]
        ref_set = set(gi[:k].tolist())
        hip_set = set(np.nonzero(mask)[0].tolist())
        ref_g = dict(zip(gi.tolist(), gv.tolist()))
        mism = sorted(ref_set ^ hip_set)
        bands = [abs(ref_g[j] - kth) / kth if j in ref_g else float("inf") for j in mism]
        rows.append({"iteration": it, "lambda_hip": float(lam), "lambda_ref": float(ref["ref_f"][it]),
                     "abs_diff": float(abs(lam - ref["ref_f"][it])), "solver_iters": int(mac.last_iters),
                     "mask_mismatches": len(mism) // 2,
                     "max_rel_distance_to_kth": float(max(bands, default=0.0)),
                     "rel_gap_at_k": float((gv[k - 1] - gv[k]) / gv[k - 1]),
                     "max_rel_grad_diff_top": float(np.max(np.abs(g[gi] - gv) / kth))})
    return rows


def main():
    from cslam_amd.mac.mac import MAC
    ref, n, fixed, cand, k, w0 = load_problem()
    mac = MAC(fixed, cand, n)
    out = {"tol_factor": os.environ.get("CSLAM_MAC_TOL_FACTOR", "default")}
    fr, _ = free_running(mac, ref, k, w0)
    out["free_running"] = fr
    mac.close()
    mac = MAC(fixed, cand, n)
    out["teacher_forced"] = teacher_forced(mac, ref, k, len(cand))
    mac.close()
    out["reference_self"] = {
        "selection_overlap_ref_vs_tight": len(set(ref["ref_rounded_idx"].tolist()) &
                                              set(ref["tight_rounded_idx"].tolist())),
        "direction_overlap_ref_vs_tight": [len(set(a.tolist()) & set(b.tolist()))
                                           for a, b in zip(ref["ref_s"], ref["tight_s"])]}
    text = json.dumps(out, indent=1)
    if "--json" in sys.argv:
        with open(sys.argv[sys.argv.index("--json") + 1], "w") as f:
            f.write(text)
    fr = out["free_running"]
    print("tol factor:", out["tol_factor"])
    print("it   hip lambda2        ref (tol 1e-8)     tight (tol 1e-12)  |s&ref| |s&tight| solver its")
    for i, lam in enumerate(fr["hip_lambda_trace"]):
        print(f"{i:2d}  {lam:.10e}  {fr['ref']['lambda_trace'][i]:.10e}  {fr['tight']['lambda_trace'][i]:.10e}"
              f"   {fr['ref']['direction_overlap'][i]:4d}   {fr['tight']['direction_overlap'][i]:4d}"
              f"   {fr['hip_solver_iters'][i]:5d}")
    for tag in ("ref", "tight"):
        print(f"vs {tag}: first differing direction = {fr[tag]['first_differing_direction']}, "
              f"rounded selection overlap = {fr[tag]['selection_overlap']} of {k}, "
              f"lambda2(selection) = {fr[tag]['lambda2_of_selection_by_hip']:.6e}")
    print(f"hip: lambda2(selection) = {fr['hip_lambda2_of_selection']:.6e}")
    print("reference vs itself (tol 1e-8 vs 1e-12): selection overlap",
          out["reference_self"]["selection_overlap_ref_vs_tight"])
    print("teacher-forced on the reference's iterates:")
    for r in out["teacher_forced"]:
        print(f"  w_{r['iteration']:<2d} |dlambda| = {r['abs_diff']:.1e}  mask mismatches = {r['mask_mismatches']}"
              f"  max rel distance to k-th = {r['max_rel_distance_to_kth']:.1e}  gap at k = {r['rel_gap_at_k']:.1e}"
              f"  max |dg|/g_k = {r['max_rel_grad_diff_top']:.1e}  ({r['solver_iters']} its)")


if __name__ == "__main__":
    main()
