# A/B of the fused fold + final level (k_mq_combine_final) in ONE call: tree (variant 2: block array + last-block
# reduce), tools/ab_libs/final0.so (one atomic slot per query), final1.so (8 slots), and the two-launch form
# (CSLAM_MQ_FINAL=0), alternating; then the kernel time line of the tree's fused form.
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2 3; do
  for lib in tree twolaunch; do
    unset CSLAM_PROBE_LIB CSLAM_MQ_FINAL
    if [ $lib = twolaunch ]; then export CSLAM_MQ_FINAL=0; elif [ $lib != tree ]; then export CSLAM_PROBE_LIB=$R/tools/ab_libs/$lib.so; fi
    MQ_M=80,48 python3 tools/mq_time.py 25000 12000 2>/dev/null | grep "N=" | sed "s/^/$lib /"
  done
done
unset CSLAM_PROBE_LIB CSLAM_MQ_FINAL
cd /tmp
MQ_M=80 rocprofv3 --kernel-trace -d /tmp/prof_final -- python3 $R/tools/mq_time.py 25000 > /dev/null 2>&1
python3 $R/tools/rocpd_timeline.py $(find /tmp/prof_final -name "*.db" | head -1) 9 | cut -c1-120
cd $R
for i in 1 2; do for f in 1 0; do CSLAM_MQ_FINAL=$f python3 bench.py --workload cfg4 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg4 final=$f', round(d['value']), round(d['ms_per_step'],4), d['roofline']['kernel'], round(d['roofline']['avg_kernel_ms'],4), round(d['roofline']['frac'],4))"; done; done
# the K-split launch as two four-wave blocks per CU (CSLAM_MQ_PART_BLOCKS2=1) against the shipped eight-wave block
MQ_AB="one eight-wave block per CU=;two four-wave blocks per CU=CSLAM_MQ_PART_BLOCKS2:1" MQ_M=80,48 python3 tools/mq_plan_ab.py 25000 12000 40000 2>&1 | grep "N="
