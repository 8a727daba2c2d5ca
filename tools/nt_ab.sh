for rep in 1 2 3; do for nt in 1 0; do CSLAM_MQ_PART_NT=$nt MQ_M=80,48 python tools/mq_time.py 25000 40000 2>/dev/null | grep "N=" | sed "s/^/nt=$nt /"; done; done
