"""Registers, AccVGPRs, scratch and LDS of every kernel in the kept device assembly
(cslam_amd/csrc/build/*.s).  usage: kernel_resources.py [name-substring ...]"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        return dict(zip(names, out))
    except OSError:
        return {n: n for n in names}


def main():
    want = sys.argv[1:]
    rows = []
    for f in sorted(glob.glob(os.path.join(ROOT, "cslam_amd", "csrc", "build", "*-gfx950.s"))):
        s = open(f).read()
        for blk in re.split(r"\n  - \.agpr_count:", s)[1:]:
            blk = ".agpr_count:" + blk
            g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, blk)
            name = g("name").group(1)
            rows.append((name, int(g("vgpr_count").group(1)), int(g("agpr_count").group(1)),
                         int(g("private_segment_fixed_size").group(1)), int(g("group_segment_fixed_size").group(1)),
                         int(g("sgpr_count").group(1))))
    dm = demangle([r[0] for r in rows])
    for name, vg, ag, scr, lds, sg in rows:
        d = dm.get(name, name)
        if want and not any(w in d for w in want):
            continue
        print(f"{d[:110]:110s} vgpr {vg:3d} (acc {ag:3d}) sgpr {sg:3d} scratch {scr:4d} B lds {lds}")


if __name__ == "__main__":
    main()
