"""Where do the 20 us between `search` and `append 16 rows, then search` go?  Same pool, same
stream: search only / append only / append + search / append on a second stream + search."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cslam_amd import _lib
from cslam_amd.nns_matching import NearestNeighborsMatching
L = _lib.load()
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
N, D = 100_000, 4096
st = torch.cuda.current_stream().cuda_stream
side = torch.cuda.Stream()
X = torch.rand((N, D), device=dev, generator=g)
pool = NearestNeighborsMatching(dim=D, capacity_hint=N + 16 * 4000)
pool.add_items_device(X.data_ptr(), N, D, items=None, stream=st)
blocks = torch.rand((4, 16, D), device=dev, generator=g)
oi = torch.empty((16, 1), dtype=torch.int32, device=dev)
os_ = torch.empty((16, 1), dtype=torch.float32, device=dev)
def search(i):
    b = blocks[i % 4]
    _lib.check(L.cslam_pool_search_dev(pool.handle, ctypes.c_void_p(b.data_ptr()), D, 16, 1,
                                       ctypes.c_void_p(oi.data_ptr()), ctypes.c_void_p(os_.data_ptr()), ctypes.c_void_p(st)))
def append(i, stream=st):
    b = blocks[i % 4]
    _lib.check(L.cslam_pool_add_dev(pool.handle, ctypes.c_void_p(b.data_ptr()), 16, D, ctypes.c_void_p(stream)))
def timed(fn, reps=200):
    for i in range(60): fn(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(reps): fn(i)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for rnd in range(3):
    a = timed(search)
    b = timed(lambda i: append(i))
    c = timed(lambda i: (append(i), search(i)))
    d = timed(lambda i: (append(i, side.cuda_stream), search(i)))
    print(f"search {a:.4f} ms   append {b:.4f} ms   append+search {c:.4f} ms   append (other stream)+search {d:.4f} ms   rows {L.cslam_pool_size(pool.handle)}", flush=True)
