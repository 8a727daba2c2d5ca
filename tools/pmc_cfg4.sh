export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
for W in 8 4; do
  i=0
  for C in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    CSLAM_MQ_PART_WAVES=$W rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_w${W}_$i -- python3 $R/bench.py --workload cfg4 --steps 20 --warmup 3 > /tmp/pmc_w${W}_$i.json 2> /tmp/pmc_w${W}_$i.err || echo "pass $W $i failed"
  done
  python3 $R/tools/pmc_summary.py "k_scan_mq<5" /tmp/pmc_w${W}_1 /tmp/pmc_w${W}_2 /tmp/pmc_w${W}_3 /tmp/pmc_w${W}_4 /tmp/pmc_w${W}_5 /tmp/pmc_w${W}_6 > $R/gpurun_out/r04_mq_pmc_ksplit_w$W.json
  python3 $R/tools/pmc_summary.py "k_mq_combine" /tmp/pmc_w${W}_1 /tmp/pmc_w${W}_5 /tmp/pmc_w${W}_6 > $R/gpurun_out/r04_mq_pmc_combine_w$W.json
  cat $R/gpurun_out/r04_mq_pmc_ksplit_w$W.json
done
