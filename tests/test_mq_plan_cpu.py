"""The multi-tile scan's planning (csrc/mq_plan.hpp: one launch / main + K-split tail / the whole window through
the K-split launch; groups x K parts; blocks, rounds, candidates per query) is plain host arithmetic: compiled for
the host and checked without a GPU - invariants over 43 000 (pool size, descriptor length, queries, CUs, knobs)
combinations and the plans DESIGN.md quotes for the benchmark shapes (tests/native/mq_plan_check.cpp).  What the
GPU does with a plan is pinned by tests/test_mq_plan_gpu.py."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not found")
def test_plans_hold_their_invariants(tmp_path):
    exe = str(tmp_path / "mq_plan_check")
    build = subprocess.run([HIPCC, "-O1", "-std=c++17", "-x", "c++", "-I", os.path.join(ROOT, "cslam_amd", "csrc"), "-o", exe,
                            os.path.join(ROOT, "tests", "native", "mq_plan_check.cpp")], capture_output=True, text=True, timeout=600)
    assert build.returncode == 0, build.stderr[-3000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and run.stdout.strip().splitlines()[-1].startswith("ok"), run.stdout[-3000:] + run.stderr[-1000:]
