for ns in 6 3 2 4 6 3; do CSLAM_MQ_PART_NS=$ns MQ_M=80 python tools/mq_time.py 25000 2>/dev/null | sed "s/^/NS=$ns /"; done
