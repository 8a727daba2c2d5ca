"""Row-stride padding A/B in ONE process (boxes and processes differ in clock mode): pools of the
same ~1.6 GB at several CSLAM_ROW_PAD values, measured alternately.  Usage: pad_ab.py D pad pad ..."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cslam_amd import _lib
from cslam_amd.nns_matching import NearestNeighborsMatching
L = _lib.load()
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
st = torch.cuda.current_stream().cuda_stream
D = int(sys.argv[1])
pads = [int(a) for a in sys.argv[2:]]
n = 1_638_400_000 // (D * 4)
X = torch.rand((n, D), device=dev, generator=g)
q = torch.rand((16, D), device=dev, generator=g)
oi = torch.empty((16, 1), dtype=torch.int32, device=dev)
os_ = torch.empty((16, 1), dtype=torch.float32, device=dev)
pools = []
for pad in pads:
    os.environ["CSLAM_ROW_PAD"] = str(pad)
    pool = NearestNeighborsMatching(dim=D, capacity_hint=n)
    pool.add_items_device(X.data_ptr(), n, D, items=None, stream=st)
    pools.append(pool)
torch.cuda.synchronize()
ref = None
def run(pool):
    _lib.check(L.cslam_pool_search_dev(pool.handle, ctypes.c_void_p(q.data_ptr()), D, 16, 1,
                                       ctypes.c_void_p(oi.data_ptr()), ctypes.c_void_p(os_.data_ptr()),
                                       ctypes.c_void_p(st)))
best = {p: 1e9 for p in pads}
avg = {p: 0.0 for p in pads}
R = 6
for rnd in range(R):
    for pad, pool in zip(pads, pools):
        for _ in range(30): run(pool)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100): run(pool)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
        res = (oi.cpu().numpy().tobytes(), os_.cpu().numpy().tobytes())
        if ref is None: ref = res
        assert res == ref, "results differ between paddings"
        best[pad] = min(best[pad], dt); avg[pad] += dt / R
for pad in pads:
    print(f"D={D} n={n} pad={pad:4d} floats: best {best[pad]*1e3:.4f} ms  avg {avg[pad]*1e3:.4f} ms  "
          f"{n*D*4/avg[pad]/1e12:.2f} TB/s (avg)", flush=True)
