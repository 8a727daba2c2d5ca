"""Determinism soak of the search paths: the same queries against the same pool, thousands of
launches per shape, every result compared bit for bit with the first one (an intermittent race in
the hand-counted vmcnt / LDS-DMA pipelines would show as a flipped bit sooner or later) and the
first one with the oracle.  Shapes cover k_scan_dma (big pool), k_scan_dma_ks (small pool, long
descriptors), the split best-match scan, the narrow-slice variants (128 / 64 floats per slice), a ragged length
(staged queries), the register-ring scan (dim <= 16), k = 10, - round 3 - the multi-tile scan k_scan_mq and
- round 4 - its all-K-split form, k_topk_from_parts at 17+ queries and the one-launch k_small_best."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cslam_amd import _lib
from cslam_amd.nns_matching import NearestNeighborsMatching
from oracle import knn as oracle_knn
L = _lib.load()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
bad = 0
for (n, d, m, k, split) in [(100_000, 4096, 16, 1, -1), (100_000, 4096, 16, 1, 99_990), (3000, 4096, 16, 1, -1),
                            (20_000, 512, 16, 10, -1), (5000, 100, 7, 1, -1), (900, 2048, 3, 5, -1),
                            (400_000, 128, 16, 1, -1), (400_000, 128, 16, 1, 399_990), (300_000, 64, 16, 1, -1),
                            (50_000, 384, 9, 10, -1), (3000, 10, 5, 3, -1),
                            # round 3: k_scan_mq (T = 2..5), plain / split / k > 1, with and without the K-split tail
                            (100_000, 4096, 80, 1, -1), (100_000, 4096, 80, 1, 99_990), (100_000, 4096, 32, 1, -1),
                            (25_000, 4096, 80, 1, 24_990), (60_000, 1024, 48, 1, -1), (200_000, 128, 64, 1, -1),
                            (30_000, 512, 40, 10, -1), (7001, 192, 80, 1, -1),
                            # round 4: the all-K-split pass, top-k from parts at 17+ queries, k_small_best (atomics + ticket)
                            (25_000, 4096, 80, 1, -1), (25_000, 4096, 80, 26, -1), (100_000, 4096, 80, 26, -1),
                            (12_000, 1024, 48, 10, -1), (1000, 128, 1, 1, -1), (4000, 128, 16, 1, -1),
                            (1000, 4096, 2, 1, -1), (300, 1000, 4, 1, -1)]:
    rs = np.random.RandomState(n + d)
    X = rs.rand(n, d).astype(np.float32); Q = rs.rand(m, d).astype(np.float32)
    pool = NearestNeighborsMatching(dim=d, capacity_hint=n); pool.add_items(X, list(range(n)))
    qd = torch.from_numpy(Q).cuda()
    if split < 0:
        idx = torch.empty((m, k), dtype=torch.int32, device="cuda"); sim = torch.empty((m, k), device="cuda")
        def run():
            _lib.check(L.cslam_pool_search_dev(pool.handle, ctypes.c_void_p(qd.data_ptr()), d, m, k,
                                               ctypes.c_void_p(idx.data_ptr()), ctypes.c_void_p(sim.data_ptr()), None))
        outs = (idx, sim)
    else:
        o = [torch.empty(m, dtype=torch.int32, device="cuda"), torch.empty(m, device="cuda"),
             torch.empty(m, dtype=torch.int32, device="cuda"), torch.empty(m, device="cuda")]
        def run():
            _lib.check(L.cslam_pool_search_best_split_dev(pool.handle, ctypes.c_void_p(qd.data_ptr()), d, m, split, -1,
                                                          *[ctypes.c_void_p(t.data_ptr()) for t in o], None))
        outs = tuple(o)
    run(); torch.cuda.synchronize()
    first = [t.clone() for t in outs]
    if split < 0 and n * d * m <= 40_000_000_000:
        oi, os_, _ = oracle_knn.search(X, Q, k)
        assert np.array_equal(first[0].cpu().numpy(), oi) and first[1].cpu().numpy().tobytes() == os_.tobytes()
    t0 = time.perf_counter(); flips = 0
    for i in range(reps):
        run()
        if i % 8 == 7:
            flips += sum(int(not torch.equal(a, b)) for a, b in zip(outs, first))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    bad += flips
    print(f"n={n} d={d} m={m} k={k} split={split}: {reps} launches, {flips} differing results, "
          f"{dt / reps * 1e6:.1f} us per call, kernel {L.cslam_pool_scan_kernel(pool.handle).decode()[:40]}", flush=True)
    pool.close()
print("soak: total differing results", bad)
sys.exit(1 if bad else 0)
