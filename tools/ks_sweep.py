"""16 best-match queries over pools of 3k..24k rows x 4096-d: end-to-end step with the K-split kernel
allowed up to 1 / 2 / 3 strips per block (CSLAM_SCAN_KSPLIT_STRIPS, re-read per launch; beyond the cap: k_scan_dma)."""
import ctypes, os, sys, time
os.environ["CSLAM_KNOBS_LIVE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cslam_amd import _lib
from cslam_amd.nns_matching import NearestNeighborsMatching
L = _lib.load(); D = 4096
g = torch.Generator(device="cuda").manual_seed(0)
for N in (3000, 4096, 5000, 6000, 8000, 8192, 10000, 12000, 12288, 14000, 16000):
    X = torch.rand((N, D), device="cuda", generator=g)
    m = NearestNeighborsMatching(dim=D, capacity_hint=N)
    m.add_items_device(X.data_ptr(), N, D, items=None, stream=None)
    Q = torch.rand((16, D), device="cuda", generator=g)
    idx = torch.empty((16, 1), dtype=torch.int32, device="cuda"); sim = torch.empty((16, 1), dtype=torch.float32, device="cuda")
    def run():
        _lib.check(L.cslam_pool_search_dev(m.handle, ctypes.c_void_p(Q.data_ptr()), D, 16, 1, ctypes.c_void_p(idx.data_ptr()), ctypes.c_void_p(sim.data_ptr()), None))
    out = []
    for rounds in ("1", "2", "3", "1", "2", "3"):
        os.environ["CSLAM_SCAN_KSPLIT_STRIPS"] = rounds
        for _ in range(20): run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): run()
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 50
        out.append(f"strips<={rounds}: {t*1e6:.1f} us ({L.cslam_pool_scan_kernel(m.handle).decode()[:13]})")
    print(f"N={N}: " + " | ".join(out), flush=True)
    m.close(); del X
