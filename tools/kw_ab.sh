for rep in 1 2; do
  for kw in 128 64; do
    CSLAM_MQ_KW=$kw MQ_M=80,48 python tools/mq_time.py 100000 2>/dev/null | grep "N=" | sed "s/^/KW=$kw main /"
    CSLAM_MQ_KW=$kw CSLAM_MQ_PART_WAVES=4 MQ_M=80 python tools/mq_time.py 25000 2>/dev/null | grep "N=" | sed "s/^/KW=$kw part4 /"
  done
done
