# A/B of two builds of the library in ONE call (boxes differ by up to 10 %): tools/ab_libs/base.so vs the tree's
# usage: bash tools/lib_ab.sh "<rows ...>" "<M list>"      (environment knobs apply to both)
ROWS=${1:-"25000 100000"}; MS=${2:-"80"}
for rep in 1 2 3; do
  for lib in base new; do
    if [ $lib = base ]; then export CSLAM_PROBE_LIB=tools/ab_libs/base.so; else unset CSLAM_PROBE_LIB; fi
    MQ_M=$MS python tools/mq_time.py $ROWS 2>/dev/null | grep "N=" | sed "s/^/$lib /"
  done
done
