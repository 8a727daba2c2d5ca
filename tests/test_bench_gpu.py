"""bench.py's own control flow on the GPU box, at toy size: the N = 1 line carries the contract's
keys (roofline, traffic rule, extras off), and a bare `python bench.py --gpus 2` launches its own
two ranks and runs the sharded protocol step (publish, exchange, split best-match scan, records,
fetch) — here with both ranks on cuda:0 and the exchange through gloo/host memory, because RCCL
cannot place two ranks on one device; the driver's multi-GPU run uses RCCL."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env=None, timeout=300):
    env = dict(os.environ, **(extra_env or {}))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, timeout=timeout,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]           # ONE JSON line
    return json.loads(lines[0])


def test_bench_single_gpu_line(gpu):
    d = _run(["--keyframes", "20000", "--dim", "512", "--steps", "8", "--warmup", "2", "--no-extras",
              "--no-cpu-baseline"])
    assert d["n_gpus"] == 1 and d["steps"] == 8 and d["warmup"] == 2 and d["scaling"] == "weak"
    assert d["unit"] == "candidates/s" and d["value"] > 0 and d["higher_is_better"] is True
    assert abs(d["value"] - 16 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["kernel"].startswith("k_scan_dma") and r["launches_timed"] >= 1
    assert r["traffic"] is None and r["traffic_source"] is None   # the PMC summary is about another shape
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["algorithmic_bytes_per_launch"] == 20000 * 512 * 4 + 16 * 512 * 4 + 16 * 8


def test_bench_self_launches_two_ranks(gpu):
    d = _run(["--gpus", "2", "--keyframes", "6000", "--dim", "256", "--steps", "6", "--warmup", "2"],
             {"CSLAM_BENCH_SAME_DEVICE": "1", "CSLAM_BENCH_BACKEND": "gloo"})
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "robot-per-gpu x2"
    dist = d["dist"]
    assert dist["world_size"] == 2 and dist["control_plane_backend"] == "gloo" and dist["communicators"] == 1
    # the first N > 1 run validates itself: communicator size, rank-stamped all-gather, the last step's
    # records recomputed from the regenerated peer blocks and compared bit for bit
    assert dist["selfcheck"] == "ok" and set(dist["selfcheck_detail"].values()) == {"ok"}, dist
    # ONE all-gather of blocks + ids at publish time, ONE of the records at match time
    assert dist["new_descriptors_per_rank_per_step"] == 8 and abs(dist["collectives_per_step"] - 2) < 0.5
    # threshold 0 on non-negative data: every foreign descriptor matches its best old row, every own
    # one its intra-robot match (the cfg2 protocol's default: all 16 gathered descriptors are evaluated)
    assert dist["match_records_per_step"] >= 2 * 16
    assert dist["allgather_payload_bytes_per_step"] > 2 * 8 * 256 * 4
    assert d["config"]["evaluations_per_step_per_gpu"] == 16
    assert abs(d["value"] - 2 * 16 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    assert "best+dots" in d["roofline"]["kernel"]        # one fused scan: composites per wave + dots, K = 8 + 10
    # inter-robot matching only: the FOREIGN descriptors are the evaluations, 8 of the 16 gathered per rank
    d = _run(["--gpus", "2", "--keyframes", "6000", "--dim", "256", "--steps", "6", "--warmup", "2"],
             {"CSLAM_BENCH_SAME_DEVICE": "1", "CSLAM_BENCH_BACKEND": "gloo", "CSLAM_BENCH_INTRA": "0"})
    assert d["config"]["evaluations_per_step_per_gpu"] == 8 and d["dist"]["match_records_per_step"] >= 16
    assert abs(d["value"] - 2 * 8 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    assert d["roofline"]["kernel"].endswith("best>")     # the split best-match scan ran


def test_bench_cfg4_workload_on_one_gpu(gpu):
    """--workload cfg4 at N = 1: robot 0 of 8 over the loopback communicator, 10 descriptors per robot
    and step, all 80 gathered ones matched in ONE k_scan_mq pass (toy pool size here)."""
    d = _run(["--workload", "cfg4", "--keyframes", "6000", "--dim", "256", "--steps", "6", "--warmup", "2"])
    assert d["n_gpus"] == 1 and d["config"]["queries_per_step_per_gpu"] == 80
    assert d["config"]["evaluations_per_step_per_gpu"] == 70
    assert abs(d["value"] - 70 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    dist = d["dist"]
    assert dist["robots"] == 8 and dist["rccl_ranks"] is None and abs(dist["collectives_per_step"] - 2) < 0.5
    assert dist["match_records_per_step"] >= 70
    r = d["roofline"]
    assert r["kernel"].startswith("k_scan_mq<T=5") and r["kernel"].endswith("best>")
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9


def test_bench_self_check_bites(gpu):
    """A record that differs from its recomputation ends the run non-zero on every rank, no JSON line."""
    env = dict(os.environ, CSLAM_BENCH_SAME_DEVICE="1", CSLAM_BENCH_BACKEND="gloo", CSLAM_BENCH_SELFCHECK_CORRUPT="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--keyframes", "6000", "--dim",
                          "256", "--steps", "6", "--warmup", "2"], env=env, timeout=300, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True)
    assert out.returncode != 0
    assert "self-check FAILED" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
