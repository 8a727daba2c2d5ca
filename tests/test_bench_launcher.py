"""bench.py's self-launcher without a GPU: ranks that cannot get a device make the parent exit
non-zero promptly (no rank left waiting in a rendezvous), and a world/--gpus mismatch is refused."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gpu_present():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_gpu_present(), reason="exercises the failure path of a box without a GPU")
def test_self_launch_fails_fast_without_a_gpu():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
                          "--warmup", "1", "--keyframes", "1000", "--dim", "256"], env=env, timeout=120,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert out.returncode != 0
    assert "rank(s) failed" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]     # no JSON line on failure


def test_world_size_mismatch_is_refused():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, timeout=60,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert out.returncode != 0 and "WORLD_SIZE=3" in out.stderr
