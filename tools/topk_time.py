"""search with k = 10 at 1 / 4 / 16 queries over 100k x 4096 (and 100k x 128): the fused select
(k_scan_dma<best+dots> + k_topk_from_waves) against the two-level select over every row
(CSLAM_TOPK_WAVES=0), alternating in one process, next to the k = 1 best-match step."""
import os, sys, time
os.environ["CSLAM_KNOBS_LIVE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes
import numpy as np
import torch
from cslam_amd import _lib
from cslam_amd.nns_matching import NearestNeighborsMatching

L = _lib.load()


def run(n, d):
    g = torch.Generator(device="cuda").manual_seed(1)
    X = torch.rand((n, d), generator=g, device="cuda", dtype=torch.float32) - 0.3
    mt = NearestNeighborsMatching(dim=d, capacity_hint=n)
    mt.add_items_tensor(X, list(range(n)))
    st = torch.cuda.current_stream().cuda_stream
    for m, k in ((16, 1), (16, 10), (4, 10), (1, 10), (16, 2), (16, 16)):
        q = torch.rand((m, d), generator=g, device="cuda", dtype=torch.float32)
        idx = torch.empty((m, k), dtype=torch.int32, device="cuda")
        sim = torch.empty((m, k), dtype=torch.float32, device="cuda")

        def call():
            _lib.check(L.cslam_pool_search_range_dev(mt.handle, ctypes.c_void_p(q.data_ptr()), q.stride(0), m, k, 0, n,
                                                     ctypes.c_void_p(idx.data_ptr()), ctypes.c_void_p(sim.data_ptr()),
                                                     ctypes.c_void_p(st)))
        res = {}
        for rep in range(3):
            for mode in ("1", "0"):
                os.environ["CSLAM_TOPK_WAVES"] = mode
                for _ in range(20):
                    call()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(200):
                    call()
                torch.cuda.synchronize()
                res.setdefault(mode, []).append((time.perf_counter() - t0) / 200 * 1e3)
                if rep == 0:
                    res.setdefault("out" + mode, (idx.cpu().numpy().copy(), sim.cpu().numpy().copy()))
        same = np.array_equal(res["out1"][0], res["out0"][0]) and res["out1"][1].tobytes() == res["out0"][1].tobytes()
        print(f"{n}x{d} m={m:2d} k={k:2d}: fused {min(res['1']):.4f} ms, two-level {min(res['0']):.4f} ms, "
              f"same bytes {same}, kernel {L.cslam_pool_scan_kernel(mt.handle).decode()}", flush=True)
    mt.close()


run(100000, 4096)
run(100000, 128)
run(25000, 4096)
