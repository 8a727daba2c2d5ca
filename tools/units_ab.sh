# time of the all-K-split pass vs units per wave on ONE pool (25 000 x 4096-d x 80): groups 32 / 16 / 8 x 8 K parts
for w in 8 4; do for g in 32 16 8; do
  CSLAM_MQ_PART_WAVES=$w CSLAM_MQ_ALLPART_GROUPS=$g CSLAM_MQ_ALLPART_KS=8 MQ_M=80 python tools/mq_time.py 25000 2>/dev/null | grep "N=" | sed "s/^/waves=$w groups=$g /"
done; done
