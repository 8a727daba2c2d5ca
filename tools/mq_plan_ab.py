"""k_scan_mq planning A/B (round 4): how the strips of a pool are dealt over the waves.
One process, settings alternating (CSLAM_KNOBS_LIVE), results of every setting compared bit for bit
with the first.  For each pool size: 80 (and 48) best-match queries, step time and summed kernel time
(HIP events on the launch stream) per setting.
usage: mq_plan_ab.py [rows ...]      env MQ_M=80,48  MQ_AB=name=KEY:VALUE,KEY:VALUE;name2=...  CSLAM_PROBE_LIB=other.so"""
import ctypes, os, sys, time
os.environ["CSLAM_KNOBS_LIVE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cslam_amd import _lib
from cslam_amd.nns_matching import NearestNeighborsMatching
if os.environ.get('CSLAM_PROBE_LIB'): _lib.LIB_PATH = os.environ['CSLAM_PROBE_LIB']   # another build of the library
L = _lib.load(); D = 4096
MS = [int(x) for x in os.environ.get("MQ_M", "80").split(",")]
sizes = [int(a) for a in sys.argv[1:]] or [25000, 100000, 60000, 12000, 5000]
SETTINGS = [
    ("round-3 rule (rounds of one-strip-per-wave blocks)", {"CSLAM_MQ_TAIL_SHAPE": "2", "CSLAM_MQ_ALLPART": "0"}),
    ("main + one wave of tail blocks, no all-K-split", {"CSLAM_MQ_ALLPART": "0"}),
    ("round-4 default", {}),
    ("all-K-split with one wave per SIMD", {"CSLAM_MQ_PART_WAVES": "4"}),
    ("all-K-split forced, two waves per SIMD", {"CSLAM_MQ_ALLPART": "2"}),
    ("all-K-split forced, one wave per SIMD", {"CSLAM_MQ_ALLPART": "2", "CSLAM_MQ_PART_WAVES": "4"}),
]
if os.environ.get("MQ_AB"):   # other settings: MQ_AB="name=KEY:VALUE,KEY:VALUE;name2=..." (an empty list = the default)
    SETTINGS = [(part.split("=")[0], dict(kv.split(":") for kv in part.split("=")[1].split(",") if kv))
                for part in os.environ["MQ_AB"].split(";")]
KEYS = sorted({k for _, e in SETTINGS for k in e})
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
        ref = None
        res = {name: [] for name, _ in SETTINGS}
        for rep in range(3):
            for name, env in SETTINGS:
                for k in KEYS: os.environ.pop(k, None)
                os.environ.update(env)
                for _ in range(12): run()
                torch.cuda.synchronize()
                reps = 30
                _lib.check(L.cslam_pool_profile_begin(m.handle, reps * 4))
                t0 = time.perf_counter()
                for _ in range(reps): run()
                torch.cuda.synchronize(); t = (time.perf_counter() - t0) / reps
                ms = ctypes.c_double(); nl = ctypes.c_int()
                _lib.check(L.cslam_pool_profile_end(m.handle, ctypes.byref(ms), ctypes.byref(nl)))
                out = (idx.cpu().numpy().tobytes(), sim.cpu().numpy().tobytes())
                if ref is None: ref = out
                res[name].append((t * 1e3, ms.value * nl.value / reps, out == ref))
        for k in KEYS: os.environ.pop(k, None)
        for name, _ in SETTINGS:
            r = res[name]
            print(f"N={N} M={M:3d} {name:52s} step " + " ".join(f"{a:.4f}" for a, _, _ in r) + "  kernels " +
                  " ".join(f"{b:.4f}" for _, b, _ in r) + f" ms  {2.0*N*D*M/min(a for a,_,_ in r)/1e9:.1f} TFLOP/s  "
                  f"bits {'same' if all(c for _, _, c in r) else 'DIFFER'}", flush=True)
    m.close(); del X
