"""The partitions the one-launch rank select works on (csrc/topk_parts.hpp: WaveParts, PanelParts, StripParts) are
plain index arithmetic shared by host and device: compiled for the HOST and checked without a GPU - the parts tile
the window exactly and a panel's rows are the ones the scan's own arithmetic gives it (tests/native/parts_check.cpp).
On the GPU the same structs are pinned end to end by tests/test_knn_gpu.py::test_multi_tile_top_k_from_parts."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not found")
def test_parts_tile_the_window(tmp_path):
    exe = str(tmp_path / "parts_check")
    build = subprocess.run([HIPCC, "-O1", "-std=c++17", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "cslam_amd", "csrc"),
                            "-I", os.path.join(ROOT, "include"), "-o", exe, os.path.join(ROOT, "tests", "native", "parts_check.cpp")],
                           capture_output=True, text=True, timeout=600)
    assert build.returncode == 0, build.stderr[-3000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and run.stdout.strip().startswith("ok"), run.stdout[-3000:] + run.stderr[-1000:]
