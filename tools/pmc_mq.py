"""PMC summary of the multi-tile scan (k_scan_mq) from rocprofv3 --pmc passes of
`bench.py --workload cfg4 --steps 20 --warmup 3` (80 gathered descriptors x 25 000 rows x 4096-d, one launch per step):
  pass 1: SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES   (matrix-pipe busy, clock)
  pass 2: FETCH_SIZE    pass 3: WRITE_SIZE                            (HBM traffic; FETCH x2 on gfx950)
usage: pmc_mq.py mfma_dir fetch_dir write_dir out.json"""
import csv, glob, json, os, statistics, sys


def rows(directory):
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            yield from csv.DictReader(f)


def main():
    mfma_dir, fetch_dir, write_dir, out = sys.argv[1:5]
    N, D, M = 25_000 + 10 * (57 + 3 + 10), 4096, 80      # pool rows at the middle of the timed steps
    per = {}
    for r in rows(mfma_dir):
        if "k_scan_mq<5" in r["Kernel_Name"] and ", 1, 4>" in r["Kernel_Name"]:     # main launch, best mode
            d = per.setdefault(r["Dispatch_Id"], {"dur": float(r["End_Timestamp"]) - float(r["Start_Timestamp"])})
            d[r["Counter_Name"]] = float(r["Counter_Value"])
    disp = [d for d in per.values() if "GRBM_GUI_ACTIVE" in d and "SQ_VALU_MFMA_BUSY_CYCLES" in d][3:]
    util = [d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["GRBM_GUI_ACTIVE"] / 8 * 1024) for d in disp]
    clk = [d["GRBM_GUI_ACTIVE"] / 8 / d["dur"] for d in disp]
    fetch = [float(r["Counter_Value"]) for r in rows(fetch_dir) if r["Counter_Name"] == "FETCH_SIZE" and "k_scan_mq<5" in r["Kernel_Name"] and ", 1, 4>" in r["Kernel_Name"]][3:]
    write = [float(r["Counter_Value"]) for r in rows(write_dir) if r["Counter_Name"] == "WRITE_SIZE" and "k_scan_mq<5" in r["Kernel_Name"] and ", 1, 4>" in r["Kernel_Name"]][3:]
    hbm = 2.0 * statistics.mean(fetch) * 1024 + statistics.mean(write) * 1024
    n_main = N
    alg = n_main * D * 4 + M * D * 4 + M * 8 * 2
    rec = {"kernel": "k_scan_mq<T=5,KW=128,NST=2,NW=4,best> (80 gathered descriptors x ~25 700 rows x 4096-d, split window: the cfg4 step)",
           "command": "rocprofv3 --pmc <counters> --kernel-trace --output-format csv -- python3 bench.py --workload cfg4 --steps 20 --warmup 3 (separate passes)",
           "dispatches": len(disp), "dur_ms_median": statistics.median(d["dur"] for d in disp) * 1e-6,
           "clock_GHz_median": statistics.median(clk), "MfmaUtil_median": statistics.median(util),
           "note": "MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); durations under a counter pass are longer than un-profiled ones",
           "FETCH_SIZE_KB_avg": statistics.mean(fetch), "WRITE_SIZE_KB_avg": statistics.mean(write),
           "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": hbm / alg,
           "flops_per_launch": 2.0 * n_main * D * M,
           "tflops_at_median_duration": 2.0 * n_main * D * M / (statistics.median(d["dur"] for d in disp) * 1e-9) / 1e12}
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
