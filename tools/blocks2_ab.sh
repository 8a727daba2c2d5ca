# A/B in ONE call: the all-K-split launch as one eight-wave block per CU (shipped in round 4) against two four-wave
# blocks per CU (CSLAM_MQ_PART_BLOCKS2), two-launch combine (CSLAM_MQ_FINAL=0): pass times, then the cfg4 bench line.
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R
export CSLAM_MQ_FINAL=0
MQ_AB="one eight-wave block per CU=CSLAM_MQ_PART_BLOCKS2:0;two four-wave blocks per CU=CSLAM_MQ_PART_BLOCKS2:1" MQ_M=80,64,48 python3 tools/mq_plan_ab.py 25000 12000 40000 8000 2>&1 | grep "N="
for i in 1 2 3; do for f in 0 1; do CSLAM_MQ_PART_BLOCKS2=$f python3 bench.py --workload cfg4 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg4 blocks2=$f', round(d['value']), round(d['ms_per_step'],4), d['roofline']['kernel'], round(d['roofline']['avg_kernel_ms'],4), round(d['roofline']['frac'],4))"; done; done
