"""HBM traffic of the scan kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE are in
separate passes: TCC has 4 slots, FETCH_SIZE takes 3 and WRITE_SIZE 2 — MI355X_MICROARCH.md
§rocprofv3 PMC slots), corrected as that guide's §HBM prescribes for gfx950 (FETCH_SIZE counts
the 128-B requests of a 16-B-per-lane streaming read at 64 B → x2; WRITE_SIZE is exact).

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- \
        python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- (same)
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write <kernel substring> \
        "<roofline.kernel of the bench line>" profiles/r02_scan_pmc.json
"""
import csv
import glob
import json
import os
import sys


def collect(directory, counter, kernel_sub):
    vals = []
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] == counter and kernel_sub in row["Kernel_Name"]:
                    vals.append(float(row["Counter_Value"]))
    return vals


def main():
    fetch_dir, write_dir, kernel_sub, kernel_label, out = sys.argv[1:6]
    keyframes = int(sys.argv[6]) if len(sys.argv) > 6 else 100_000
    dim = int(sys.argv[7]) if len(sys.argv) > 7 else 4096
    queries = 16
    f = collect(fetch_dir, "FETCH_SIZE", kernel_sub)
    w = collect(write_dir, "WRITE_SIZE", kernel_sub)
    if not f or not w:
        raise SystemExit(f"no {kernel_sub} dispatches found (fetch {len(f)}, write {len(w)})")
    f, w = f[3:] or f, w[3:] or w       # drop the first launches (cold caches, page tables)
    fetch_kb = sum(f) / len(f)
    write_kb = sum(w) / len(w)
    alg = keyframes * dim * 4 + queries * dim * 4 + queries * 8
    hbm = 2.0 * fetch_kb * 1024 + write_kb * 1024
    rec = {"kernel": kernel_label, "kernel_symbol_contains": kernel_sub, "keyframes": keyframes, "dim": dim,
           "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- "
                      "python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras (separate passes)",
           "FETCH_SIZE_KB_avg": fetch_kb, "WRITE_SIZE_KB_avg": write_kb,
           "dispatches": [len(f), len(w)],
           "correction": "gfx950: FETCH_SIZE counts 128-B requests at 64 B for 16-B/lane streaming reads "
                         "(global_load and LDS-DMA alike) -> x2 (MI355X_MICROARCH.md §HBM); WRITE_SIZE exact",
           "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg,
           "traffic_over_algorithmic": hbm / alg}
    with open(out, "w") as fh:
        json.dump(rec, fh, indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
