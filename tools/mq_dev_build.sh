#!/bin/bash
# Compile scan_mq.hip with ONE variant (T = 5, KW = 128) into /tmp/mqdev and report its registers,
# scratch and the scratch / vmcnt(0) sites inside the stage loop.  Seconds instead of a minute.
set -e
cd "$(dirname "$0")/../cslam_amd/csrc"
mkdir -p /tmp/mqdev
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -save-temps=obj \
    -ffp-contract=off -fno-fast-math -DCSLAM_MQ_DEV_T5 "$@" -c scan_mq.hip -o /tmp/mqdev/scan_mq.o
python3 - <<'PY'
import re
s = open('/tmp/mqdev/scan_mq-hip-amdgcn-amd-amdhsa-gfx950.s').read()
for blk in re.split(r"\n  - \.agpr_count:", s)[1:]:
    blk = ".agpr_count:" + blk
    g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, blk).group(1)
    if 'k_scan_mq' in g('name'):
        print(g('name')[25:60], 'vgpr', g('vgpr_count'), 'acc', g('agpr_count'), 'scratch', g('private_segment_fixed_size'))
lines = s.split('\n')
for mode in (1, 0, 2):
    name = '_ZN12_GLOBAL__N_19k_scan_mqILi5ELi128ELi2ELi6ELi2ELi%dELi4EEEvN5cslam6MqArgsE' % (mode if mode != 2 else 2)
    if mode == 2: name = name.replace('ELi2ELi2ELi4', 'ELi0ELi2ELi4')
    st = [i for i, l in enumerate(lines) if l.startswith(name + ':')]
    if not st: continue
    fe = [i for i in range(st[0], len(lines)) if lines[i].startswith('.Lfunc_end')][0]
    body = lines[st[0]:fe]
    print('mode', mode, 'lines', len(body), 'scratch ops', sum('scratch_' in l for l in body),
          'vmcnt(0)', sum('vmcnt(0)' in l for l in body), 'mfma', sum('v_mfma' in l for l in body),
          'accvgpr_read', sum('v_accvgpr_read' in l for l in body), 'accvgpr_write', sum('v_accvgpr_write' in l for l in body),
          'accvgpr_mov', sum('v_accvgpr_mov' in l for l in body))
PY
