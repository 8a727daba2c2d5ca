"""How many CUs does the HBM-bound scan need, and should panels be whole strips?
CSLAM_SCAN_EVEN_PANELS (1 = a wave's rows in equal panels, 0 = panels of 128 rows and a remainder) and CSLAM_SCAN_RESERVE_CUS A/B in one
process (CSLAM_KNOBS_LIVE: the library re-reads the knobs at every launch): 16 best-match queries over 100k x 4096, alternating settings."""
import ctypes, os, sys, time
os.environ["CSLAM_KNOBS_LIVE"] = "1"   # the library re-reads its measurement knobs at every launch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cslam_amd import _lib
from cslam_amd.nns_matching import NearestNeighborsMatching
L = _lib.load()
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
st = torch.cuda.current_stream().cuda_stream
D = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = int(os.environ.get("AB_ROWS", 1_638_400_000 // (D * 4)))
settings = sys.argv[2:] or ["0:0", "1:0", "0:16", "1:16"]   # panel unit : reserved CUs
X = torch.rand((n, D), device=dev, generator=g)
pool = NearestNeighborsMatching(dim=D, capacity_hint=n)
pool.add_items_device(X.data_ptr(), n, D, items=None, stream=st)
q = torch.rand((16, D), device=dev, generator=g)
oi = torch.empty((16, 1), dtype=torch.int32, device=dev)
os_ = torch.empty((16, 1), dtype=torch.float32, device=dev)
def run():
    _lib.check(L.cslam_pool_search_dev(pool.handle, ctypes.c_void_p(q.data_ptr()), D, 16, 1,
                                       ctypes.c_void_p(oi.data_ptr()), ctypes.c_void_p(os_.data_ptr()),
                                       ctypes.c_void_p(st)))
ref = None
avg = {s: 0.0 for s in settings}
best = {s: 1e9 for s in settings}
R = 6
for rnd in range(R):
    for s in settings:
        os.environ[os.environ.get("AB_KNOB", "CSLAM_SCAN_EVEN_PANELS")], os.environ["CSLAM_SCAN_RESERVE_CUS"] = s.split(":")
        for _ in range(30): run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100): run()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
        res = (oi.cpu().numpy().tobytes(), os_.cpu().numpy().tobytes())
        if ref is None: ref = res
        assert res == ref
        avg[s] += dt / R; best[s] = min(best[s], dt)
for s in settings:
    print(f"D={D} rows={n} {os.environ.get('AB_KNOB', 'CSLAM_SCAN_EVEN_PANELS')}:reserved CUs {s:>6s}: best {best[s]*1e3:.4f} ms avg {avg[s]*1e3:.4f} ms  {n*D*4/avg[s]/1e12:.2f} TB/s", flush=True)
