"""PMC summary of the multi-tile pass of `bench.py --workload cfg4` (80 gathered descriptors x ~25 700 rows x 4096-d)
from rocprofv3 --pmc passes (tools/pmc_cfg4.sh):
  pass 1: SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES   (matrix-pipe busy, clock)
  pass 5: FETCH_SIZE    pass 6: WRITE_SIZE                                          (HBM traffic; FETCH x2 on gfx950)
The pass is the K-split launch (round 4: k_scan_mq<5, 64, 3, 6, 0, 2, 8>, round 5: <5, 64, 2, 6, 2, 2, 4>, two blocks per CU) + k_mq_combine<true>; both are summed.
usage: pmc_mq.py mfma_dir fetch_dir write_dir "<roofline.kernel of the bench line>" out.json"""
import csv, glob, json, os, statistics, sys


def rows(directory):
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            yield from csv.DictReader(f)


def per_dispatch(directory, want):
    per = {}
    for r in rows(directory):
        if want(r["Kernel_Name"]):
            d = per.setdefault(int(r["Dispatch_Id"]), {"dur": float(r["End_Timestamp"]) - float(r["Start_Timestamp"])})
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return [v for _, v in sorted(per.items())][3:]


def main():
    mfma_dir, fetch_dir, write_dir, label, out = sys.argv[1:6]
    N, D, M = 25_000, 4096, 80
    rows_mid = N + 10 * (57 + 3 + 10)                     # pool rows at the middle of the timed steps
    scan = lambda n: "k_scan_mq<5" in n
    comb = lambda n: "k_mq_combine" in n
    disp = [d for d in per_dispatch(mfma_dir, scan) if "GRBM_GUI_ACTIVE" in d and "SQ_VALU_MFMA_BUSY_CYCLES" in d]
    util = [d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["GRBM_GUI_ACTIVE"] / 8 * 1024) for d in disp]
    clk = [d["GRBM_GUI_ACTIVE"] / 8 / d["dur"] for d in disp]
    cdur = [d["dur"] for d in per_dispatch(mfma_dir, comb)]
    kb = lambda directory, counter, want: statistics.mean(d[counter] for d in per_dispatch(directory, want) if counter in d)
    fetch = kb(fetch_dir, "FETCH_SIZE", scan) + kb(fetch_dir, "FETCH_SIZE", comb)
    write = kb(write_dir, "WRITE_SIZE", scan) + kb(write_dir, "WRITE_SIZE", comb)
    hbm = 2.0 * fetch * 1024 + write * 1024
    alg = rows_mid * D * 4 + M * D * 4 + M * 8 * 2
    dur = statistics.median(d["dur"] for d in disp)
    rec = {"kernel": label, "keyframes": N, "dim": D,
           "command": "rocprofv3 --pmc <counters> --kernel-trace --output-format csv -- python3 bench.py --workload cfg4 --steps 20 --warmup 3 (separate passes, tools/pmc_cfg4.sh)",
           "dispatches": len(disp), "scan_dur_ms_median": dur * 1e-6,
           "combine_dur_ms_median": statistics.median(cdur) * 1e-6 if cdur else None,
           "clock_GHz_median": statistics.median(clk), "MfmaUtil_median": statistics.median(util),
           "note": "MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) of the K-split launch; durations under a counter pass are longer than un-profiled ones",
           "FETCH_SIZE_KB_avg": fetch, "WRITE_SIZE_KB_avg": write,
           "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": hbm / alg,
           "partial_bytes_written": ((rows_mid + 15) // 16) * (D // 512) * 5 * 1024,
           "flops_per_launch": 2.0 * rows_mid * D * M,
           "tflops_at_median_scan_duration": 2.0 * rows_mid * D * M / (dur * 1e-9) / 1e12}
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
