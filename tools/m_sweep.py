"""Time of one best-match search (device pointers, k = 1) vs number of queries on 100k x 4096-d: k_scan_dma (<= 16),
k_scan_mq (17..80 per pass), the GEMM tiles beyond.  Steady state, as bench.py measures it: the first ~40 launches of
a process run 3-10 % slow (profiles/r02_launch_ramp.txt), so 60 untimed launches first, then 10 + 40 per batch size
(M_SWEEP_COLD=1: the old 3 + 10 right after start-up)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cslam_amd import _lib
from cslam_amd.nns_matching import NearestNeighborsMatching
import torch
L = _lib.load(); N, D = 100000, 4096
g = torch.Generator(device="cuda").manual_seed(0)
X = torch.rand((N, D), device="cuda", generator=g)
m = NearestNeighborsMatching(dim=D, capacity_hint=N)
_lib.check(L.cslam_pool_add_dev(m.handle, ctypes.c_void_p(X.data_ptr()), N, D, None)); torch.cuda.synchronize(); m.n = N
cold = os.environ.get("M_SWEEP_COLD") == "1"
if not cold:
    Q0 = torch.rand((16, D), device="cuda", generator=g)
    i0 = torch.empty((16, 1), dtype=torch.int32, device="cuda"); s0 = torch.empty((16, 1), dtype=torch.float32, device="cuda")
    for _ in range(60):
        _lib.check(L.cslam_pool_search_dev(m.handle, ctypes.c_void_p(Q0.data_ptr()), D, 16, 1, ctypes.c_void_p(i0.data_ptr()), ctypes.c_void_p(s0.data_ptr()), None))
    torch.cuda.synchronize()
warm, reps = (3, 10) if cold else (10, 40)
for M in (8, 16, 17, 32, 33, 48, 64, 65, 80, 95, 96, 128, 192, 256):
    Q = torch.rand((M, D), device="cuda", generator=g)
    idx = torch.empty((M, 1), dtype=torch.int32, device="cuda"); sim = torch.empty((M, 1), dtype=torch.float32, device="cuda")
    def run():
        _lib.check(L.cslam_pool_search_dev(m.handle, ctypes.c_void_p(Q.data_ptr()), D, M, 1, ctypes.c_void_p(idx.data_ptr()), ctypes.c_void_p(sim.data_ptr()), None))
    for _ in range(warm): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): run()
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / reps
    print(f"M={M:4d}: {t*1e3:.3f} ms  {t*1e6/M:.1f} us/query  {2.0*N*D*M/t/1e12:.1f} TFLOP/s  {L.cslam_pool_scan_kernel(m.handle).decode()}", flush=True)
