"""Per-kernel means of whatever counters a set of rocprofv3 --pmc passes collected (csv output).
usage: pmc_summary.py KERNEL_SUBSTR dir [dir ...]   -> JSON: {counter: mean over dispatches (first 3 skipped)}"""
import csv, glob, json, os, statistics, sys

want = sys.argv[1]
out = {}
durs = []
for d in sys.argv[2:]:
    per = {}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                if want in r["Kernel_Name"]:
                    per.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
                    per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
                    if "Start_Timestamp" in r:
                        durs.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    for c, byd in per.items():
        vals = [v for _, v in sorted(byd.items(), key=lambda kv: int(kv[0]))][3:]
        if vals:
            out[c] = statistics.mean(vals)
            out[c + "_n"] = len(vals)
if durs:
    out["dur_us_median_under_counters"] = statistics.median(durs) / 1e3
print(json.dumps(out, indent=1))
