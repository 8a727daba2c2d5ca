# Round-end measurements on one box: default bench line (plain, under rocprofv3 --kernel-trace --stats, PMC traffic),
# cfg4 line (plain, kernel trace, PMC), two-rank gloo rehearsal.  Summaries land in gpurun_out/r05_*.
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp
python3 $R/bench.py > $O/r05_bench_default.json 2> $O/r05_bench_default.err
rocprofv3 --kernel-trace --stats -d /tmp/prof_default -- python3 $R/bench.py --no-extras --no-cpu-baseline > $O/r05_bench_under_rocprof.json 2>/dev/null
python3 $R/tools/rocpd_stats.py $(find /tmp/prof_default -name "*.db" | head -1) | cut -c1-220 > $O/r05_bench_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_fetch -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_write -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > /dev/null 2>&1
python3 $R/tools/pmc_traffic.py /tmp/pmc_fetch /tmp/pmc_write "k_scan_dma<8, 4, 2, 2, true, 256, false>" "k_scan_dma<NS=8,NW=4,NST=2,nt,best>" $O/r05_scan_pmc.json > /dev/null
python3 $R/bench.py --workload cfg4 > $O/r05_bench_cfg4_n1.json 2> $O/r05_bench_cfg4_n1.err
rocprofv3 --kernel-trace --stats -d /tmp/prof_cfg4 -- python3 $R/bench.py --workload cfg4 > $O/r05_bench_cfg4_under_rocprof.json 2>/dev/null
DB=$(find /tmp/prof_cfg4 -name "*.db" | head -1)
python3 $R/tools/rocpd_stats.py $DB | cut -c1-220 > $O/r05_bench_cfg4_kernel_stats.csv
python3 $R/tools/rocpd_timeline.py $DB 36 | cut -c1-150 > $O/r05_bench_cfg4_timeline.txt
i=0
for C in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_c4_$i -- python3 $R/bench.py --workload cfg4 --steps 20 --warmup 3 > /dev/null 2>&1
done
KN=$(python3 -c "import json; print(json.loads(open('$O/r05_bench_cfg4_n1.json').read().strip().splitlines()[-1])['roofline']['kernel'])")
python3 $R/tools/pmc_mq.py /tmp/pmc_c4_1 /tmp/pmc_c4_2 /tmp/pmc_c4_3 "$KN" $O/r05_mq_pmc.json > /dev/null
CSLAM_BENCH_SAME_DEVICE=1 CSLAM_BENCH_BACKEND=gloo python3 $R/bench.py --gpus 2 --steps 20 --warmup 5 > $O/r05_bench_gloo2_rehearsal.json 2> $O/r05_bench_gloo2_rehearsal.err
cd $R
python3 - <<'PY'
import json
for f in ("r05_bench_default.json", "r05_bench_under_rocprof.json", "r05_bench_cfg4_n1.json", "r05_bench_cfg4_under_rocprof.json", "r05_bench_gloo2_rehearsal.json"):
    try:
        d = json.loads(open("gpurun_out/" + f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["roofline"]["kernel"], round(d["roofline"]["frac"], 4), d["roofline"]["avg_kernel_ms"], (d.get("dist") or {}).get("selfcheck"))
    except Exception as e:
        print(f, "FAILED", e)
for f in ("r05_scan_pmc.json", "r05_mq_pmc.json"):
    try:
        d = json.load(open("gpurun_out/" + f)); print(f, {k: d[k] for k in d if k in ("traffic_over_algorithmic", "MfmaUtil_median", "scan_dur_ms_median", "combine_dur_ms_median", "clock_GHz_median", "hbm_bytes_per_launch")})
    except Exception as e:
        print(f, "FAILED", e)
PY
head -6 gpurun_out/r05_bench_kernel_stats.csv | cut -c1-160
