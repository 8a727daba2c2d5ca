#!/bin/bash
# Timing probes of k_scan_mq: the library with one ingredient of the stage loop compiled out
# (results are wrong; time only) - and PROBES=AUDIT: the address audit (tools/probes/mq_probe_audit.hpp: results
# are the product's, every global address is checked before use; tools/mq_audit.py).  Output: tools/probe_libs/libcslam_hip_<name>.so, loaded by
# tools/mq_time.py through CSLAM_PROBE_LIB.
set -e
cd "$(dirname "$0")/../cslam_amd/csrc"
make -j4 >/dev/null
mkdir -p ../../tools/probe_libs build/probe
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math"
# the probe code lives in tools/probes/*.hpp and enters through the hook points of csrc/probe_hooks.hpp
for v in ${PROBES:-NO_DMA NO_MFMA CLOCK}; do
  f=$(echo $v | tr A-Z a-z)
  /opt/rocm/bin/hipcc $FLAGS -DCSLAM_PROBE_HOOKS_FILE="\"../../tools/probes/mq_probe_$f.hpp\"" -c scan_mq.hip -o build/probe/scan_mq_$v.o &
done
wait
if [ -n "$DMA_CLOCK" ]; then   # k_scan_dma with time stamps (knn.hip)
  /opt/rocm/bin/hipcc $FLAGS -DCSLAM_PROBE_HOOKS_FILE="\"../../tools/probes/dma_probe_clock.hpp\"" -c knn.hip -o build/probe/knn_DMA_CLOCK.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/probe_libs/libcslam_hip_DMA_CLOCK.so \
     build/common.o build/probe/knn_DMA_CLOCK.o build/scan_mq.o build/mac.o build/geom.o build/comm.o build/shard.o -ldl
fi
for v in ${PROBES:-NO_DMA NO_MFMA CLOCK}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/probe_libs/libcslam_hip_$v.so \
     build/common.o build/knn.o build/probe/scan_mq_$v.o build/mac.o build/geom.o build/comm.o build/shard.o -ldl
done
ls -la ../../tools/probe_libs
