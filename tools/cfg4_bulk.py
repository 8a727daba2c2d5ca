"""BASELINE.json configs[3], the BULK exchange, on ONE GPU: robot 0 of 8 with a 25 000 x 4096 pool
receives 25 000 descriptors from each of the 7 other robots (loopback communicators: the peers'
blocks are copies of the own block) and matches all 175 000 of them against its pool — the
GEMM-tile path of cslam_pool_search_range_dev behind cslam_shard_match (csrc/shard.hip, bulk mode).
What a rank of the 8-GPU node computes in the first step after the robots meet; the all-gather of
the 8 x 410 MB blocks itself is RCCL's (≈2.7 ms per peer link at 153 GB/s).
Usage: python tools/cfg4_bulk.py [rows_per_robot] [robots]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cslam_amd.multi_robot import DeviceShard, LoopbackComm  # noqa: E402
from cslam_amd.nns_matching import NearestNeighborsMatching  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 25_000
R = int(sys.argv[2]) if len(sys.argv) > 2 else 8
D = 4096
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
st = torch.cuda.current_stream().cuda_stream
pool = NearestNeighborsMatching(dim=D, capacity_hint=3 * B)
cd, cr = LoopbackComm(0, R, 0), LoopbackComm(0, R, 0)
sh = DeviceShard(pool, 0, R, B, 0.0, cd, cr, nb_best_matches=0)
blocks = [torch.rand((B, D), device=dev, generator=g) for _ in range(2)]
kf = 0
out = []
for t in range(2):        # step 0: empty pool (append only); step 1: 25k rows resident, 175k queries
    ids = list(range(kf, kf + B))
    kf += B
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sh.publish(blocks[t], ids, st)
    sh.match(st)
    rec = sh.fetch_records()
    dt = time.perf_counter() - t0
    out.append((t, pool.n, len(rec), dt))
    print(f"step {t}: pool {pool.n} rows, {len(rec)} records, {dt * 1e3:.1f} ms", flush=True)
n_q = (R - 1) * B
flops = 2.0 * B * D * n_q + 2.0 * B * D * n_q      # old rows [0, B) and new rows [B, 2B)
print(f"bulk step: {n_q} foreign descriptors x (25k old + 25k new rows): {flops / out[1][3] / 1e12:.1f} TFLOP/s end to end, "
      f"{n_q * 2 / out[1][3]:.0f} best-match evaluations/s")
sh.close(); cd.close(); cr.close(); pool.close()
