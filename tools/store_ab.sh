# A/B of the chunk-partial store of k_scan_mq's K-split mode in ONE call: tree (plain) vs tools/ab_libs/store1.so (nt)
# vs store2.so (sc0 sc1), alternating; then the kernel time line of each (gap between the K-split launch and k_mq_combine).
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2 3; do
  for lib in tree store1 store2; do
    if [ $lib = tree ]; then unset CSLAM_PROBE_LIB; else export CSLAM_PROBE_LIB=$R/tools/ab_libs/$lib.so; fi
    MQ_M=80 python3 tools/mq_time.py 25000 100000 2>/dev/null | grep "N=" | sed "s/^/$lib /"
  done
done
cd /tmp
for lib in tree store1 store2; do
  if [ $lib = tree ]; then unset CSLAM_PROBE_LIB; else export CSLAM_PROBE_LIB=$R/tools/ab_libs/$lib.so; fi
  MQ_M=80 rocprofv3 --kernel-trace -d /tmp/prof_$lib -- python3 $R/tools/mq_time.py 25000 > /dev/null 2>&1
  echo "== $lib"; python3 $R/tools/rocpd_timeline.py $(find /tmp/prof_$lib -name "*.db" | head -1) 12 | cut -c1-120
done
