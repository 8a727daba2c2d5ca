"""k_scan_mq (17..80 queries per pass): kernel time by HIP events on the launch stream and end-to-end
time of search (k = 1) per batch size, on pools of 100k and 25k rows x 4096-d (cfg2 / cfg4 shapes).
usage: mq_time.py [rows ...]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cslam_amd import _lib
from cslam_amd.nns_matching import NearestNeighborsMatching
if os.environ.get('CSLAM_PROBE_LIB'): _lib.LIB_PATH = os.environ['CSLAM_PROBE_LIB']
L = _lib.load(); D = 4096
MS = [int(x) for x in os.environ.get('MQ_M', '16,32,48,64,80,96,128,160').split(',')]
sizes = [int(a) for a in sys.argv[1:]] or [100000, 25000]
g = torch.Generator(device="cuda").manual_seed(0)
for N in sizes:
    X = torch.rand((N, D), device="cuda", generator=g)
    m = NearestNeighborsMatching(dim=D, capacity_hint=N)
    m.add_items_device(X.data_ptr(), N, D, items=None, stream=None)
    for M in MS:
        Q = torch.rand((M, D), device="cuda", generator=g)
        idx = torch.empty((M, 1), dtype=torch.int32, device="cuda"); sim = torch.empty((M, 1), dtype=torch.float32, device="cuda")
        def run():
            _lib.check(L.cslam_pool_search_dev(m.handle, ctypes.c_void_p(Q.data_ptr()), D, M, 1, ctypes.c_void_p(idx.data_ptr()), ctypes.c_void_p(sim.data_ptr()), None))
        for _ in range(20): run()
        torch.cuda.synchronize()
        reps = 30
        _lib.check(L.cslam_pool_profile_begin(m.handle, reps * 4))
        t0 = time.perf_counter()
        for _ in range(reps): run()
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / reps
        ms = ctypes.c_double(); nl = ctypes.c_int()
        _lib.check(L.cslam_pool_profile_end(m.handle, ctypes.byref(ms), ctypes.byref(nl)))
        per_call = ms.value * nl.value / reps
        print(f"N={N} M={M:4d}: step {t*1e3:.3f} ms  kernels {per_call:.3f} ms ({nl.value // reps} launch(es))  "
              f"{2.0*N*D*M/t/1e12:.1f} TFLOP/s  {N*D*4/t/1e12:.2f} TB/s  {L.cslam_pool_scan_kernel(m.handle).decode()}", flush=True)
    m.close(); del X
