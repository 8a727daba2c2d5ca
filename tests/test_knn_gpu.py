"""Parity of the HIP kNN path (through the C ABI) with the CPU oracle — bit exact.

Mirrors the reference's tests/test_sparse_matching.py:51-81 (test_similarity) and adds the
edge cases of cslam/nns_matching.py (empty pool, growth, k > n, ragged dims, ties, zero rows).
"""
import numpy as np
import pytest

from oracle import knn as oracle_knn

pytestmark = pytest.mark.gpu


def _pool(gpu, X, **kw):
    from cslam_amd.nns_matching import NearestNeighborsMatching
    m = NearestNeighborsMatching(**kw)
    m.add_items(X, list(range(X.shape[0])))
    return m


def _check(gpu, X, Q, k):
    m = _pool(gpu, X)
    rows, sims = m.search_batch(Q, k)
    o_idx, o_sim, o_cnt = oracle_knn.search(X, Q, k)
    kk = int(o_cnt[0])
    assert rows.shape == (Q.shape[0], kk)
    np.testing.assert_array_equal(rows, o_idx[:, :kk])
    assert sims.tobytes() == o_sim[:, :kk].astype(np.float32).tobytes()
    m.close()


@pytest.mark.parametrize("n,d,m,k", [
    (100, 100, 100, 100),     # the reference's test_similarity shape (k == n → bitonic path)
    (1000, 128, 7, 10),       # cfg1
    (1, 10, 1, 1),
    (17, 10, 3, 5),
    (4097, 64, 16, 10),       # two select parts
    (5000, 4096, 16, 10),     # cfg2 row shape
    (3000, 4096, 33, 1),      # T=4 scan
    (3000, 256, 20, 3),       # T=2 scan
    (9000, 48, 300, 64),      # three GEMM tiles in one chunk, k = 64
    (2000, 64, 1100, 3),      # query chunking (> 1024 queries per call)
    (600, 33, 2, 200),        # bitonic path, ragged dim
    (2000, 4096, 130, 10),    # LDS-tiled 32x32x2 GEMM path (>= 96 queries), ragged row/query tiles
    (777, 100, 256, 5),       # GEMM path, ragged dim, full 256-query chunk
    (129, 16, 97, 129),       # GEMM path, one k-block, k == n (bitonic select)
])
def test_search_matches_oracle(gpu, n, d, m, k):
    rs = np.random.RandomState(n * 31 + d)
    X = rs.rand(n, d).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    Q = rs.rand(m, d).astype(np.float32)
    _check(gpu, X, Q, k)


def test_negative_and_unnormalised(gpu):
    rs = np.random.RandomState(5)
    X = (rs.randn(2500, 200) * 3).astype(np.float32)
    Q = rs.randn(9, 200).astype(np.float32)
    _check(gpu, X, Q, 10)


def test_ties_duplicates_and_zero_rows(gpu):
    rs = np.random.RandomState(11)
    X = rs.rand(500, 32).astype(np.float32)
    X[100] = X[7]          # exact duplicates → tie broken by lowest row
    X[300] = X[7]
    X[50] = 0.0            # zero row → similarity 0 (clamped norm, nns_matching.py:61 eps)
    Q = np.vstack([X[7], rs.rand(32).astype(np.float32), np.zeros(32, np.float32)])
    m = _pool(gpu, X)
    rows, sims = m.search_batch(Q, 5)
    o_idx, o_sim, _ = oracle_knn.search(X, Q, 5)
    np.testing.assert_array_equal(rows, o_idx)
    assert sims.tobytes() == o_sim.tobytes()
    assert list(rows[0][:3]) == [7, 100, 300]
    assert np.all(sims[2] == 0.0) and list(rows[2]) == [0, 1, 2, 3, 4]
    m.close()


def test_inverse_norms_match_oracle(gpu):
    rs = np.random.RandomState(3)
    X = rs.rand(300, 1000).astype(np.float32) * 7
    m = _pool(gpu, X)
    got = np.empty(300, np.float32)
    from cslam_amd import _lib
    _lib.check(gpu.cslam_pool_read_inv_norms(m.handle, 0, 300, _lib.ptr(got)))
    assert got.tobytes() == oracle_knn.inv_norms(X).tobytes()
    m.close()


def test_similarity_ordering_like_reference(gpu):
    """tests/test_sparse_matching.py:51-81: cosine order == euclidean order on unit vectors."""
    from cslam_amd.nns_matching import NearestNeighborsMatching
    rs = np.random.RandomState(0)
    nnsm = NearestNeighborsMatching()
    for i in range(100):
        d = rs.rand(100)
        nnsm.add_item(d / np.linalg.norm(d), i)
    k = 100
    for _ in range(100):
        query = rs.rand(100)
        query = query / np.linalg.norm(query)
        ds = np.linalg.norm(query[np.newaxis, :] - nnsm.data.cpu().numpy()[:nnsm.n], axis=1)
        ns_dist = np.argsort(ds)[:k]
        ns_sim, sims = nnsm.search(query, k)
        assert np.all(sims[:-1] >= sims[1:])
        for j in range(k):
            id_d, id_s = ns_dist[j], ns_sim[j]
            if id_d != id_s:
                if abs(sims[id_d] - sims[id_s]) < 1e-6 or abs(ds[id_d] - ds[id_s]) < 1e-6:
                    continue
            assert ns_dist[j] == ns_sim[j]
        best, _ = nnsm.search_best(query)
        assert ns_dist[0] == best


def test_growth_and_data_view(gpu):
    """Pool growth 1000 → 2000 → 4000 (nns_matching.py:33-38) keeps rows and results."""
    from cslam_amd.nns_matching import NearestNeighborsMatching
    rs = np.random.RandomState(2)
    X = rs.rand(2500, 24).astype(np.float32)
    m = NearestNeighborsMatching()
    assert m.search(X[0], 3) == ([], []) and m.search_best(X[0]) == (None, None)
    for i in range(1200):
        m.add_item(X[i].astype(np.float64), 10 * i)   # float64 input is cast (nns_matching.py:31)
    assert len(m.data) == 2000 and m.n == 1200
    m.add_items(X[1200:], [10 * i for i in range(1200, 2500)])
    assert len(m.data) == 4000 and m.n == 2500
    np.testing.assert_array_equal(m.data.cpu().numpy()[:2500], X)
    assert m.data[1234, 5] == X[1234, 5]
    np.testing.assert_array_equal(m.data[17], X[17])
    items, sims = m.search(X[777], 4)
    o_idx, o_sim, _ = oracle_knn.search(X, X[777:778], 4)
    assert items == [10 * int(i) for i in o_idx[0]] and items[0] == 7770
    assert sims.tobytes() == o_sim[0].tobytes()
    m.close()


def test_device_resident_queries(gpu):
    """cslam_pool_search_dev with queries/outputs in HBM (the bench / multi-GPU entry)."""
    torch = pytest.importorskip("torch")
    import ctypes
    from cslam_amd import _lib
    rs = np.random.RandomState(9)
    X = rs.rand(3000, 512).astype(np.float32)
    Q = rs.rand(10, 512).astype(np.float32)
    m = _pool(gpu, X)
    q = torch.from_numpy(Q).cuda()
    idx = torch.empty((10, 10), dtype=torch.int32, device="cuda")
    sim = torch.empty((10, 10), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(gpu.cslam_pool_search_dev(m.handle, ctypes.c_void_p(q.data_ptr()), 512, 10, 10,
                                         ctypes.c_void_p(idx.data_ptr()),
                                         ctypes.c_void_p(sim.data_ptr()), ctypes.c_void_p(st)))
    torch.cuda.synchronize()
    o_idx, o_sim, _ = oracle_knn.search(X, Q, 10)
    np.testing.assert_array_equal(idx.cpu().numpy(), o_idx)
    assert sim.cpu().numpy().tobytes() == o_sim.tobytes()
    m.close()


def test_range_search_and_sharded_step_on_gpu(gpu):
    """cslam_pool_search_range_dev (rows appended since the step began) and the single-rank
    step of cslam_amd/multi_robot.py on the real pool."""
    torch = pytest.importorskip("torch")
    from cslam_amd.multi_robot import HipPoolMatcher, ShardedLoopClosureMatching
    rs = np.random.RandomState(4)
    X = rs.rand(700, 96).astype(np.float32)
    Q = rs.rand(9, 96).astype(np.float32)
    hm = HipPoolMatcher(96)
    hm.add_rows(torch.from_numpy(X[:500]).cuda(), list(range(500)))
    hm.add_rows(torch.from_numpy(X[500:]).cuda(), list(range(500, 700)))
    q = torch.from_numpy(Q).cuda()
    for lo, hi in ((0, 700), (0, 500), (500, 700), (123, 124), (650, 700)):
        rows, sims = hm.search_best(q, lo, hi)
        o_idx, o_sim, _ = oracle_knn.search(X[lo:hi], Q, 1)
        np.testing.assert_array_equal(rows, o_idx[:, 0] + lo)
        assert sims.tobytes() == o_sim[:, 0].tobytes()
    rows, sims = hm.search_best(q, 5, 5)
    assert np.all(rows == -1)

    class Solo:
        rank, world = 0, 1

        def __init__(self):
            self.torch = torch

        def all_gather(self, t):
            return t.unsqueeze(0)

    sh = ShardedLoopClosureMatching(Solo(), HipPoolMatcher(96), 0.5, max_block=8)
    assert sh.step(torch.from_numpy(X[:4]).cuda(), [0, 1, 2, 3]) == []
    assert sh.matcher.size == 4


def test_gemm_tail_tiles_and_range_search_on_a_large_pool(gpu):
    """(a) 512 queries against 70 000 rows (the bulk regime: up to ~500 queries the k_scan_mq passes of
    round 3 win): 2188 GEMM tiles = four whole waves of 128 x 128 tiles plus a remainder that runs
    as 64 x 128 tiles (k_gemm_dma<1,2>) — bit-exact against the oracle.
    (b) the row-range search (what the sharded step uses) on a window large enough for the
    LDS-DMA scan, with a non-zero first row: indices are absolute, similarities identical."""
    torch = pytest.importorskip("torch")
    import ctypes
    from cslam_amd import _lib
    rs = np.random.RandomState(12)
    n, d = 70_000, 64
    X = (rs.rand(n, d).astype(np.float32) - 0.4)
    Q = rs.rand(512, d).astype(np.float32)
    m = _pool(gpu, X, dim=d, capacity_hint=n)
    rows, sims = m.search_batch(Q, 3)
    assert gpu.cslam_pool_scan_kernel(m.handle).decode().startswith("k_gemm_dma")
    o_idx, o_sim, _ = oracle_knn.search(X, Q, 3)
    np.testing.assert_array_equal(rows, o_idx)
    assert sims.tobytes() == o_sim.tobytes()
    m.close()
    # (b)
    n, d, lo, hi = 40_000, 256, 7_001, 39_000
    X = rs.rand(n, d).astype(np.float32)
    Q = rs.rand(9, d).astype(np.float32)
    m = _pool(gpu, X, dim=d, capacity_hint=n)
    qd = torch.from_numpy(Q).cuda()
    for k in (1, 4):
        idx = torch.empty((9, k), dtype=torch.int32, device="cuda")
        sim = torch.empty((9, k), dtype=torch.float32, device="cuda")
        _lib.check(gpu.cslam_pool_search_range_dev(m.handle, ctypes.c_void_p(qd.data_ptr()), d, 9, k, lo, hi,
                                                   ctypes.c_void_p(idx.data_ptr()),
                                                   ctypes.c_void_p(sim.data_ptr()), None))
        torch.cuda.synchronize()
        assert gpu.cslam_pool_scan_kernel(m.handle).decode().startswith("k_scan_dma")
        o_idx, o_sim, _ = oracle_knn.search(X[lo:hi], Q, k)
        np.testing.assert_array_equal(idx.cpu().numpy(), o_idx + lo)
        assert sim.cpu().numpy().tobytes() == o_sim.tobytes()
    m.close()


def test_search_best_device_queries_strided_and_unaligned(gpu):
    """search_best (k = 1) through the device-pointer ABI on the DMA path: a strided query
    view is read in place, a view that starts off a 16-byte boundary goes through the staging
    kernel; both equal the oracle bit for bit, ties resolve to the lowest row."""
    torch = pytest.importorskip("torch")
    import ctypes
    from cslam_amd import _lib
    rs = np.random.RandomState(77)
    n, d = 18000, 512
    X = rs.rand(n, d).astype(np.float32)
    X[9000] = X[17]                                     # tie → row 17
    m = _pool(gpu, X, dim=d, capacity_hint=n)
    Qh = rs.rand(6, d).astype(np.float32)
    Qh[2] = X[17]
    big = torch.zeros((6, d + 64 + 1), device="cuda")
    o_idx, o_sim, _ = oracle_knn.search(X, Qh, 1)
    for off in (0, 1):                                  # 0: aligned, in place; 1: staged
        view = big[:, off:off + d]
        view.copy_(torch.from_numpy(Qh))
        idx = torch.empty((6, 1), dtype=torch.int32, device="cuda")
        sim = torch.empty((6, 1), dtype=torch.float32, device="cuda")
        _lib.check(gpu.cslam_pool_search_dev(m.handle, ctypes.c_void_p(view.data_ptr()),
                                             view.stride(0), 6, 1, ctypes.c_void_p(idx.data_ptr()),
                                             ctypes.c_void_p(sim.data_ptr()), None))
        torch.cuda.synchronize()
        assert gpu.cslam_pool_scan_kernel(m.handle).decode().endswith("best>")
        np.testing.assert_array_equal(idx.cpu().numpy(), o_idx)
        assert sim.cpu().numpy().tobytes() == o_sim.tobytes()
        assert idx[2, 0].item() == 17
    m.close()


def test_device_append_is_ordered_before_host_api(gpu):
    """A device-pointer append on a busy caller stream followed, with NO synchronisation, by the
    host-pointer API (own non-blocking stream): the search must see the appended rows, and a
    host add that grows the pool must not lose them (ordering by event, csrc/knn.hip mark_dev /
    order_after_dev)."""
    torch = pytest.importorskip("torch")
    from cslam_amd.nns_matching import NearestNeighborsMatching
    rs = np.random.RandomState(21)
    base = rs.rand(900, 512).astype(np.float32)
    blk = rs.rand(90, 512).astype(np.float32)
    m = NearestNeighborsMatching(dim=512)              # capacity 1000 like the reference
    m.add_items(base, list(range(900)))
    side = torch.cuda.Stream()
    big = torch.rand((6144, 6144), device="cuda")
    blk_d = torch.from_numpy(blk).cuda()
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for _ in range(12):                            # keeps the side stream busy for a while
            big = big @ big
            big /= big.norm()
        m.add_items_tensor(blk_d, list(range(900, 990)))
    rows, sims = m.search_batch(blk[37:38], 1)         # host API, nothing synchronised
    assert rows[0, 0] == 937 and abs(sims[0, 0] - 1.0) < 1e-6
    with torch.cuda.stream(side):
        for _ in range(12):
            big = big @ big
            big /= big.norm()
        m.add_items_tensor(blk_d[:5].contiguous(), list(range(990, 995)))
    m.add_items(base[:20], list(range(995, 1015)))     # grows 1000 -> 2000 right behind it
    got = m.data.cpu().numpy()
    np.testing.assert_array_equal(got[990:995], blk[:5])
    np.testing.assert_array_equal(got[995:1015], base[:20])
    np.testing.assert_array_equal(got[900:990], blk)
    m.close()


def test_flush_dev_pins_the_ordering_event_behind_the_device_call(gpu):
    """cslam_pool_flush_dev records the (lazily kept) ordering event NOW: the caller's stream may then be
    destroyed, and unrelated work queued on it afterwards does not stand between the device-pointer append
    and the next host-pointer search (include/cslam_hip.h, INTEGRATION.md "Streams of the device-pointer calls")."""
    import time
    torch = pytest.importorskip("torch")
    from cslam_amd import _lib
    from cslam_amd.nns_matching import NearestNeighborsMatching
    rs = np.random.RandomState(5)
    base = rs.rand(600, 256).astype(np.float32)
    blk = rs.rand(40, 256).astype(np.float32)
    m = NearestNeighborsMatching(dim=256)
    m.add_items(base, list(range(600)))
    assert gpu.cslam_pool_flush_dev(m.handle) == 0          # nothing pending: no-op
    side = torch.cuda.Stream()
    big = torch.rand((8192, 8192), device="cuda")
    blk_d = torch.from_numpy(blk).cuda()
    torch.cuda.synchronize()
    m.search_batch(blk[:1], 1)                              # warm the host path
    with torch.cuda.stream(side):
        m.add_items_tensor(blk_d, list(range(600, 640)))
        _lib.check(gpu.cslam_pool_flush_dev(m.handle))
        for _ in range(40):                                 # unrelated work BEHIND the flushed event
            big = big @ big
            big /= big.norm()
    t0 = time.perf_counter()
    rows, sims = m.search_batch(blk[11:12], 1)              # waits for the append only
    t_host = time.perf_counter() - t0
    t1 = time.perf_counter()
    side.synchronize()
    t_rest = time.perf_counter() - t1
    assert rows[0, 0] == 611 and abs(sims[0, 0] - 1.0) < 1e-6
    assert t_rest > 5 * t_host, (t_host, t_rest)            # the matmuls were still running when the search returned
    # the stream goes away behind a flush; the pool's next calls never touch it again
    with torch.cuda.stream(side):
        m.add_items_tensor(blk_d[:3].contiguous(), list(range(640, 643)))
        _lib.check(gpu.cslam_pool_flush_dev(m.handle))
    del side
    rows, _ = m.search_batch(blk[2:3], 1)
    assert rows[0, 0] in (602, 642)
    m.close()


def test_ring_and_dma_scan_variants_are_bit_identical(gpu, monkeypatch):
    """CSLAM_SCAN_KERNEL=ring (round 1's register-ring kernel) and the default LDS-DMA scan (here on
    209 one-strip panels with partial strips) must give the same bits: both implement the one
    canonical order.  (The LDS-staged and producer/consumer A/B variants of round 1 left the
    product library in round 3; profiles/r01_scan_lds_vs_ring.txt, r01_scan_pc_variants.txt keep
    their measurements.)"""
    import subprocess, sys, os, json
    code = (
        "import sys, json, numpy as np; sys.path.insert(0, '.');"
        "from cslam_amd.nns_matching import NearestNeighborsMatching;"
        "rs = np.random.RandomState(8); X = rs.randn(3333, 512).astype(np.float32);"
        "Q = rs.randn(7, 512).astype(np.float32); m = NearestNeighborsMatching();"
        "m.add_items(X, list(range(3333))); r, s = m.search_batch(Q, 10);"
        "print(json.dumps([r.tolist(), s.view(np.int32).tolist()]))")
    outs = []
    for kern in ("ring", "dma"):
        env = dict(os.environ, CSLAM_SCAN_KERNEL=kern)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True,
                             cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), timeout=300)
        assert out.returncode == 0, out.stderr
        outs.append(json.loads(out.stdout.strip().splitlines()[-1]))
    assert outs[0] == outs[1]


@pytest.mark.parametrize("n,d,m,k", [
    (100000, 128, 16, 10),    # lcsm.py:69's k = 10 at the reference's own descriptor length
    (20011, 512, 5, 16),      # ragged panels
    (20011, 512, 16, 18),     # the sharded step's fused search at two robots: block of 8 + nb_best_matches 10
    (100000, 128, 9, 32),     # the largest fused k
    (20011, 64, 3, 2),        # the smallest
    (140000, 256, 9, 10),     # two panels per wave: 137 rows per wave, candidates past the register cache
    (600000, 64, 3, 16),      # 586 rows per wave; the zero query overflows the list: arg-max rounds
    (17, 128, 16, 16),        # two waves, k = 16 of 17 rows
    (5, 128, 4, 10),          # k > n: the tail of every result row is (-1, -inf)
    (1, 64, 1, 2),
    (40000, 768, 16, 3),
    (17000, 4096, 7, 10),     # cfg2's row shape
])
def test_fused_top_k_select_matches_oracle(gpu, knobs, n, d, m, k):
    """search with 2 <= k <= 32 and <= 16 queries: k_scan_dma<best+dots> + k_topk_from_waves (the rows
    of the k best waves only) against the C oracle bit for bit, and against the two-level select
    over every row (CSLAM_TOPK_WAVES=0).  Ties (duplicated rows, a zero query) keep the lowest row
    first; a row window (cslam_pool_search_range_dev) selects inside the window."""
    import ctypes
    torch = pytest.importorskip("torch")
    knobs.set("CSLAM_SMALL_BEST", "0")   # (round 4: windows of <= 4096 rows are k_small_search's; this kernel's small shapes stay covered)
    rs = np.random.RandomState(n + d + k)
    X = (rs.rand(n, d).astype(np.float32) - 0.3)
    if n > 300:
        X[n // 2] = X[7]          # duplicates in different waves
        X[n - 1] = X[7]
        X[50] = 0.0
    Q = rs.rand(m, d).astype(np.float32)
    Q[0] = X[min(7, n - 1)]
    if m > 2:
        Q[2] = 0.0                # every similarity 0: rows 0..k-1 in order
    mt = _pool(gpu, X, dim=d, capacity_hint=n)
    rows, sims = mt.search_batch(Q, k)
    assert "best+dots" in gpu.cslam_pool_scan_kernel(mt.handle).decode()
    o_idx, o_sim, o_cnt = oracle_knn.search(X, Q, k)
    kk = int(o_cnt[0])
    np.testing.assert_array_equal(rows[:, :kk], o_idx[:, :kk])
    assert sims[:, :kk].tobytes() == o_sim[:, :kk].astype(np.float32).tobytes()
    if n > 300:
        assert list(rows[0][:3]) == [7, n // 2, n - 1][:min(k, 3)]
    # a row window through the device-pointer API
    lo, hi = n // 3, n - n // 5
    if hi - lo >= 1:
        q = torch.from_numpy(Q).cuda()
        idx = torch.empty((m, k), dtype=torch.int32, device="cuda")
        sim = torch.empty((m, k), dtype=torch.float32, device="cuda")
        rc = gpu.cslam_pool_search_range_dev(mt.handle, ctypes.c_void_p(q.data_ptr()), q.stride(0), m, k, lo, hi,
                                             ctypes.c_void_p(idx.data_ptr()), ctypes.c_void_p(sim.data_ptr()),
                                             ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
        torch.cuda.synchronize()
        w_idx, w_sim, w_cnt = oracle_knn.search(X[lo:hi], Q, k)
        wk = int(w_cnt[0])
        np.testing.assert_array_equal(idx.cpu().numpy()[:, :wk], w_idx[:, :wk] + lo)
        assert sim.cpu().numpy()[:, :wk].tobytes() == w_sim[:, :wk].astype(np.float32).tobytes()
        if wk < k:
            assert np.all(idx.cpu().numpy()[:, wk:] == -1)
    mt.close()
    # the two-level select over every row gives the same bytes
    knobs.set("CSLAM_TOPK_WAVES", "0")
    mt = _pool(gpu, X, dim=d, capacity_hint=n)
    rows2, sims2 = mt.search_batch(Q, k)
    assert "best+dots" not in gpu.cslam_pool_scan_kernel(mt.handle).decode()
    np.testing.assert_array_equal(rows2, rows)
    assert sims2.tobytes() == sims.tobytes()
    mt.close()


@pytest.mark.parametrize("n,d,m,k", [
    (16384, 256, 16, 10),     # exactly the switch-over size, one 256-float slice
    (20011, 512, 5, 10),      # ragged rows: 1024 panels of 19-20 rows (partial strips everywhere)
    (17000, 4096, 3, 1),      # cfg2 row shape
    (40000, 768, 16, 3),      # odd number of slices: the last 512-float chunk is a half chunk
    (140000, 256, 9, 10),     # more than 128 rows per wave: two panels per wave
    (16384, 256, 16, 1),      # k = 1: fused arg-max epilogue, queries read in place
    (20011, 512, 5, 1),       # k = 1, partial strips, 5 of 16 query columns valid
    (20000, 250, 7, 1),       # k = 1, dim padded to 256: staged queries
    (140000, 256, 1, 1),      # k = 1, two panels per wave (query norm only from the first)
    # slice width 128 (dim % 128 == 0, % 256 != 0): two rows per DMA instruction
    (20011, 128, 16, 1),      # the reference's own descriptor length; ragged panels, odd tails
    (20011, 128, 5, 10),
    (300000, 128, 16, 1),     # three panels per wave
    (33, 128, 3, 1),          # one full stage + a one-row stage
    (1, 128, 1, 1),
    (9000, 384, 7, 1),        # three slices: the 512-float fold falls after the row's end
    (9000, 640, 16, 4),       # five slices: one whole chunk + a quarter
    (5000, 120, 9, 1),        # dim padded to 128: staged queries
    # slice width 64: four rows per DMA instruction
    (20011, 64, 16, 1),
    (20011, 64, 4, 10),
    (600000, 64, 16, 1),      # five panels per wave
    (67, 64, 2, 1),           # 64 rows + a three-row piece
    (2, 64, 1, 1),
    (7001, 192, 16, 1),       # three slices
    (7001, 320, 3, 7),        # five slices
    (4000, 576, 11, 1),       # nine slices: 512 + 64
    # mid-size pools at the benchmark's row length (configs[3]: 25 000 keyframes per robot): one or
    # two strips per wave, ragged splits
    (20000, 4096, 16, 1),     # 1250 strips on 1024 waves: 19.5 rows per wave
    (25000, 4096, 16, 1),     # cfg4's pool
    (25000, 4096, 16, 10),    # ... with the fused top-k select
    (40000, 4096, 5, 1),      # 39 rows per wave: three stages per slice
    (60000, 4096, 16, 1),     # 58.6 rows per wave
    (5000, 4096, 16, 1),      # one strip per wave, waves dealt round-robin over the CUs
    (9000, 4096, 3, 10),      # ... 563 strips: two or three waves per CU
])
def test_dma_scan_matches_oracle(gpu, knobs, n, d, m, k):
    """k_scan_dma (the default scan for <= 16 queries; slice width 256 / 128 / 64 floats) against the C
    oracle, bit for bit, on shapes that cover its panel / strip / slice edge cases.  (The best match on
    a pool of <= 4096 rows is k_small_best's since round 4 - tests/test_small_best_gpu.py; switched
    off here so that this kernel's small shapes stay covered.)"""
    knobs.set("CSLAM_SMALL_BEST", "0")
    rs = np.random.RandomState(n + d)
    X = (rs.rand(n, d).astype(np.float32) - 0.3)
    Q = rs.rand(m, d).astype(np.float32)
    mt = _pool(gpu, X, dim=d, capacity_hint=n)
    rows, sims = mt.search_batch(Q, k)
    assert gpu.cslam_pool_scan_kernel(mt.handle).decode().startswith("k_scan_dma")
    o_idx, o_sim, _ = oracle_knn.search(X, Q, k)
    np.testing.assert_array_equal(rows, o_idx)
    assert sims.tobytes() == o_sim.tobytes()
    mt.close()


@pytest.mark.parametrize("n,d,m,k", [
    (25000, 1024, 80, 18),     # all K-split (1563 strips = 1563 parts > the list: the zero query takes every row)
    (25000, 1024, 80, 26),     # the sharded step's fused search at eight robots: block of 10 + ... <= 26
    (12000, 1024, 48, 10),     # fewer strips than waves, three tiles
    (70000, 512, 80, 10),      # main launch + K-split tail: panels of four strips and single strips mixed
    (100000, 128, 33, 10),     # lcsm.py:69's k at the reference's descriptor length: six strips per panel
    (300000, 128, 40, 32),     # several rounds per block (p_max > 1), the largest k
    (20011, 192, 17, 2),       # 64-float slices, two tiles (never all K-split), the smallest k
    (4100, 640, 64, 5),        # a chunk and a quarter
    (40, 128, 20, 16),         # three strips in the pool
    (9, 128, 17, 12),          # k > n: (-1, -inf) tails
    (9000, 4096, 96, 10),      # two balanced passes of 48
])
def test_multi_tile_top_k_from_parts(gpu, knobs, n, d, m, k):
    """search with 2 <= k <= 32 and 17+ queries (the intra + inter search of the sharded step,
    loop_closure_sparse_matching.py:78-97 at K = block + nb_best_matches): k_scan_mq leaves the best
    composite of every panel / K-split strip beside the raw dots and k_topk_from_parts ranks the rows
    of the k best parts only.  Against the C oracle bit for bit, against the two-level select over
    every row (CSLAM_TOPK_WAVES=0), through a row window, with ties across parts and a zero query."""
    import ctypes
    torch = pytest.importorskip("torch")
    rs = np.random.RandomState(n + d + k + m)
    X = (rs.rand(n, d).astype(np.float32) - 0.3)
    if n > 300:
        X[n // 2] = X[7]          # duplicates in different parts
        X[n - 1] = X[7]
        X[50] = 0.0
    Q = rs.rand(m, d).astype(np.float32)
    Q[0] = X[min(7, n - 1)]
    Q[2] = 0.0                    # every similarity 0: rows 0..k-1 in order
    Q[m - 1] = X[min(7, n - 1)]
    mt = _pool(gpu, X, dim=d, capacity_hint=n)
    rows, sims = mt.search_batch(Q, k)
    name = gpu.cslam_pool_scan_kernel(mt.handle).decode()
    assert name.startswith("k_scan_mq") and "best" in name, name
    o_idx, o_sim, o_cnt = oracle_knn.search(X, Q, k)
    kk = int(o_cnt[0])
    np.testing.assert_array_equal(rows[:, :kk], o_idx[:, :kk])
    assert sims[:, :kk].tobytes() == o_sim[:, :kk].astype(np.float32).tobytes()
    if n > 300:
        assert list(rows[0][:3]) == [7, n // 2, n - 1][:min(k, 3)]
        assert list(rows[m - 1][:3]) == [7, n // 2, n - 1][:min(k, 3)]
        assert list(rows[2]) == list(range(k))
    # a row window through the device-pointer API
    lo, hi = n // 3, n - n // 5
    if hi - lo >= 1:
        q = torch.from_numpy(Q).cuda()
        idx = torch.empty((m, k), dtype=torch.int32, device="cuda")
        sim = torch.empty((m, k), dtype=torch.float32, device="cuda")
        rc = gpu.cslam_pool_search_range_dev(mt.handle, ctypes.c_void_p(q.data_ptr()), q.stride(0), m, k, lo, hi,
                                             ctypes.c_void_p(idx.data_ptr()), ctypes.c_void_p(sim.data_ptr()),
                                             ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
        torch.cuda.synchronize()
        w_idx, w_sim, w_cnt = oracle_knn.search(X[lo:hi], Q, k)
        wk = int(w_cnt[0])
        np.testing.assert_array_equal(idx.cpu().numpy()[:, :wk], w_idx[:, :wk] + lo)
        assert sim.cpu().numpy()[:, :wk].tobytes() == w_sim[:, :wk].astype(np.float32).tobytes()
        if wk < k:
            assert np.all(idx.cpu().numpy()[:, wk:] == -1)
    mt.close()
    # the dots pass + two-level select over every row gives the same bytes
    knobs.set("CSLAM_TOPK_WAVES", "0")
    mt = _pool(gpu, X, dim=d, capacity_hint=n)
    rows2, sims2 = mt.search_batch(Q, k)
    assert "best" not in gpu.cslam_pool_scan_kernel(mt.handle).decode()
    np.testing.assert_array_equal(rows2, rows)
    assert sims2.tobytes() == sims.tobytes()
    mt.close()


@pytest.mark.parametrize("n,d,m,k", [
    # 4096-d (cfg2 / cfg4 row shape): every tile count, ragged query counts, k = 1 (fused arg-max
    # epilogue) and k > 1 (raw dots + select)
    (9000, 4096, 17, 1), (9000, 4096, 33, 1), (9000, 4096, 48, 1), (9000, 4096, 64, 1),
    (9000, 4096, 80, 1), (9000, 4096, 96, 1),          # 96 = two balanced passes of 48
    (9000, 4096, 17, 10), (5000, 4096, 48, 3), (5000, 4096, 80, 10),
    # 128-d: the reference's own descriptor length (one slice per row: a step per panel round)
    (20011, 128, 17, 1), (20011, 128, 33, 1), (20011, 128, 48, 1), (20011, 128, 64, 1),
    (20011, 128, 80, 1), (20011, 128, 96, 1), (20011, 128, 33, 10), (20011, 128, 80, 10),
    # panel / strip / slice edge cases
    (300000, 128, 40, 1),     # several rounds per block (p_max > 1)
    (70000, 256, 80, 1),      # two slices, rounds of six strips
    (33, 128, 20, 1),         # three strips in the whole pool: blocks of one strip, idle waves
    (1, 128, 17, 1),          # one row
    (4100, 384, 50, 1),       # three slices: the 512-float fold falls behind the row's end
    (4100, 640, 33, 4),       # five slices: one whole chunk + a quarter
    (7001, 192, 25, 1),       # slice width 64 (dim % 128 != 0): four rows per DMA instruction
    (7001, 64, 80, 1),
    (7001, 320, 64, 3),
    (2500, 1000, 72, 1),      # dim padded to 1024: staged queries
    (640, 4096, 80, 1),       # fewer strips than waves on the chip
])
def test_multi_tile_scan_matches_oracle(gpu, n, d, m, k):
    """k_scan_mq (17..80 queries per pass, the pool streamed once through LDS-DMA whatever the
    number of query tiles; SURVEY.md 8e's gathered block of 8 robots x 10 descriptors) against the
    C oracle, bit for bit: indices and similarity bits."""
    rs = np.random.RandomState(n + 7 * d + m)
    X = (rs.rand(n, d).astype(np.float32) - 0.3)
    Q = rs.rand(m, d).astype(np.float32)
    if n > 100:
        X[n - 2] = X[5]        # a tie across panels resolves to the lower row
        Q[m - 1] = X[5]
    mt = _pool(gpu, X, dim=d, capacity_hint=n)
    rows, sims = mt.search_batch(Q, k)
    assert gpu.cslam_pool_scan_kernel(mt.handle).decode().startswith("k_scan_mq")
    o_idx, o_sim, o_cnt = oracle_knn.search(X, Q, k)
    kk = int(o_cnt[0])
    np.testing.assert_array_equal(rows, o_idx[:, :kk])
    assert sims.tobytes() == o_sim[:, :kk].astype(np.float32).tobytes()
    if n > 100:
        assert rows[m - 1, 0] == 5
    mt.close()


def test_full_size_best_match_equals_oracle(gpu):
    """BASELINE cfg2 at FULL size, the variant bench.py times: 100 000 x 4096, 16 queries, k = 1
    (k_scan_dma<...,best> with the 893-wave partition that is only taken near 100k rows) - and the
    split-window call with the split near the end - bit for bit against the C oracle on the whole
    pool (about 2 s of CPU); then the cfg4 steady-state block of 80 gathered descriptors on the
    same pool (k_scan_mq<T=5,...,best>)."""
    torch = pytest.importorskip("torch")
    import ctypes
    from cslam_amd import _lib
    from cslam_amd.nns_matching import NearestNeighborsMatching
    N, D = 100_000, 4096
    g = torch.Generator(device="cuda").manual_seed(9)
    X = torch.rand((N, D), device="cuda", generator=g)
    X[99_990] = X[4242]                      # a tie across the split and across waves
    m = NearestNeighborsMatching(dim=D, capacity_hint=N)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(gpu.cslam_pool_add_dev(m.handle, ctypes.c_void_p(X.data_ptr()), N, D, ctypes.c_void_p(st)))
    torch.cuda.synchronize()
    m.n = N
    Q = torch.rand((80, D), device="cuda", generator=g)
    Q[3] = X[4242]
    Xh, Qh = X.cpu().numpy(), Q.cpu().numpy()
    o_idx, o_sim, _ = oracle_knn.search(Xh, Qh, 1)

    def best(mq):
        idx = torch.empty((mq, 1), dtype=torch.int32, device="cuda")
        sim = torch.empty((mq, 1), dtype=torch.float32, device="cuda")
        _lib.check(gpu.cslam_pool_search_dev(m.handle, ctypes.c_void_p(Q.data_ptr()), D, mq, 1,
                                             ctypes.c_void_p(idx.data_ptr()),
                                             ctypes.c_void_p(sim.data_ptr()), ctypes.c_void_p(st)))
        torch.cuda.synchronize()
        return idx.cpu().numpy()[:, 0], sim.cpu().numpy()[:, 0]

    i16, s16 = best(16)
    name = gpu.cslam_pool_scan_kernel(m.handle).decode()
    assert name.startswith("k_scan_dma") and name.endswith("best>"), name
    np.testing.assert_array_equal(i16, o_idx[:16, 0])
    assert s16.tobytes() == o_sim[:16, 0].tobytes()
    assert i16[3] == 4242

    i80, s80 = best(80)
    name = gpu.cslam_pool_scan_kernel(m.handle).decode()
    assert name.startswith("k_scan_mq<T=5") and name.endswith("best>"), name
    np.testing.assert_array_equal(i80, o_idx[:, 0])
    assert s80.tobytes() == o_sim[:, 0].tobytes()

    # split window (the sharded step: best OLD row, best row of the block just appended)
    split = N - 16
    oa_idx, oa_sim, _ = oracle_knn.search(Xh[:split], Qh, 1)
    oc_idx, oc_sim, _ = oracle_knn.search(Xh[split:], Qh, 1)
    for mq in (16, 80):
        out = [torch.full((mq,), -7, dtype=torch.int32, device="cuda"), torch.zeros(mq, device="cuda"),
               torch.full((mq,), -7, dtype=torch.int32, device="cuda"), torch.zeros(mq, device="cuda")]
        _lib.check(gpu.cslam_pool_search_best_split_dev(
            m.handle, ctypes.c_void_p(Q.data_ptr()), D, mq, split, -1,
            *[ctypes.c_void_p(t.data_ptr()) for t in out], ctypes.c_void_p(st)))
        torch.cuda.synchronize()
        ia, sa, ic, sc = [t.cpu().numpy() for t in out]
        np.testing.assert_array_equal(ia, oa_idx[:mq, 0])
        np.testing.assert_array_equal(ic, oc_idx[:mq, 0] + split)
        assert sa.tobytes() == oa_sim[:mq, 0].tobytes()
        assert sc.tobytes() == oc_sim[:mq, 0].tobytes()
        assert ia[3] == 4242 and ic[3] == 99_990
    m.close()


def test_full_size_properties(gpu):
    """BASELINE cfg2 size (100k x 4096) through size-independent properties: every sampled pool
    row retrieves itself first with similarity 1 (to 1e-6), results are sorted, a duplicated row
    ties and resolves to the lower index, and search is idempotent."""
    torch = pytest.importorskip("torch")
    import ctypes
    from cslam_amd import _lib
    from cslam_amd.nns_matching import NearestNeighborsMatching
    N, D = 100_000, 4096
    g = torch.Generator(device="cuda").manual_seed(5)
    X = torch.rand((N, D), device="cuda", generator=g)
    X[77_777] = X[123]                       # exact duplicate far away
    m = NearestNeighborsMatching(dim=D, capacity_hint=N)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(gpu.cslam_pool_add_dev(m.handle, ctypes.c_void_p(X.data_ptr()), N, D, ctypes.c_void_p(st)))
    torch.cuda.synchronize()
    m.n = N
    probe = torch.tensor([0, 123, 4095, 50_000, 77_777, 99_999, 31_337, 65_536, 16, 15,
                          99_998, 1, 2, 3, 4, 5], device="cuda")
    Q = X[probe].contiguous()
    idx = torch.empty((16, 10), dtype=torch.int32, device="cuda")
    sim = torch.empty((16, 10), dtype=torch.float32, device="cuda")

    def run():
        _lib.check(gpu.cslam_pool_search_dev(m.handle, ctypes.c_void_p(Q.data_ptr()), D, 16, 10,
                                             ctypes.c_void_p(idx.data_ptr()),
                                             ctypes.c_void_p(sim.data_ptr()), ctypes.c_void_p(st)))
        torch.cuda.synchronize()
        return idx.cpu().numpy().copy(), sim.cpu().numpy().copy()

    i1, s1 = run()
    i2, s2 = run()
    np.testing.assert_array_equal(i1, i2)
    assert s1.tobytes() == s2.tobytes()                       # idempotent, deterministic
    assert np.all(s1[:, :-1] >= s1[:, 1:])                    # sorted
    expect_first = probe.cpu().numpy().copy()
    expect_first[4] = 123                                     # duplicate → lower row wins
    np.testing.assert_array_equal(i1[:, 0], expect_first)
    assert np.all(np.abs(s1[:, 0] - 1.0) < 1e-6)              # fp32, 4096 terms
    assert i1[1, 1] == 77_777 and s1[1, 0] == s1[1, 1]        # the tie partner comes second
    assert len(set(map(int, i1[0]))) == 10                    # no duplicates in a result list
    # the oracle on the rows that won (cheap: 10 rows) reproduces the similarities bit for bit
    rows = X[torch.from_numpy(i1[0].astype(np.int64)).cuda()].cpu().numpy()
    o = oracle_knn.similarities(rows, Q[:1].cpu().numpy())
    assert o[0].tobytes() == s1[0].tobytes()
    m.close()


@pytest.mark.parametrize("n,d,m,split", [
    (18000, 512, 16, 17990),      # the sharded step's shape: a handful of new rows behind the split
    (18000, 512, 5, 17999),       # one new row
    (40000, 256, 16, 123),        # split inside the first panel
    (3000, 256, 9, 1500),         # small pool: fewer panels than waves
    (70, 256, 16, 64),            # one strip per panel, partial last strip
    (2500, 1024, 40, 2400),       # > 16 queries: one k_scan_mq pass (T = 3)
    (18000, 128, 80, 17990),      # cfg4's gathered block: 80 descriptors, one pass (T = 5)
    (9000, 4096, 80, 8990),
    (30000, 256, 100, 29000),     # two balanced passes of 64 (T = 4)
    (1200, 48, 7, 1000),          # dim % 64 != 0: two range searches behind the same entry point
    (18000, 128, 16, 17990),      # slice width 128
    (75, 128, 16, 70),
    (18000, 64, 16, 17999),       # slice width 64
    (5000, 192, 3, 2),
])
def test_search_best_split_equals_two_range_searches(gpu, n, d, m, split):
    """cslam_pool_search_best_split_dev: best row before `split` and best row from `split` on in
    one scan launch — bit-identical to the oracle on the two row ranges, ties to the lowest row on
    either side."""
    torch = pytest.importorskip("torch")
    import ctypes
    from cslam_amd import _lib
    rs = np.random.RandomState(n + d + m)
    X = rs.rand(n, d).astype(np.float32)
    Q = rs.rand(m, d).astype(np.float32)
    if split + 2 < n:
        X[n - 1] = X[split]                         # tie behind the split → row `split`
        Q[0] = X[split]
    if split > 40:
        X[split - 1] = X[3]                         # tie in front of it → row 3
        Q[m - 1] = X[3]
    pool = _pool(gpu, X, dim=d, capacity_hint=n)
    qd = torch.from_numpy(Q).cuda()
    out = [torch.full((m,), -7, dtype=torch.int32, device="cuda"), torch.zeros(m, device="cuda"),
           torch.full((m,), -7, dtype=torch.int32, device="cuda"), torch.zeros(m, device="cuda")]
    oa_idx, oa_sim, _ = oracle_knn.search(X[:split], Q, 1)
    oc_idx, oc_sim, _ = oracle_knn.search(X[split:], Q, 1)
    for rep in range(3):
        _lib.check(gpu.cslam_pool_search_best_split_dev(
            pool.handle, ctypes.c_void_p(qd.data_ptr()), d, m, split, -1,
            *[ctypes.c_void_p(t.data_ptr()) for t in out], None))
        torch.cuda.synchronize()
        ia, sa, ic, sc = [t.cpu().numpy() for t in out]
        np.testing.assert_array_equal(ia, oa_idx[:, 0])
        np.testing.assert_array_equal(ic, oc_idx[:, 0] + split)
        assert sa.tobytes() == oa_sim[:, 0].tobytes()
        assert sc.tobytes() == oc_sim[:, 0].tobytes()
    if split + 2 < n:
        assert ic[0] == split
    if split > 40:
        assert ia[m - 1] == 3
    # an explicit upper row bound (the sharded step: later blocks are already in the pool)
    hi = n - 3
    if split < hi:
        _lib.check(gpu.cslam_pool_search_best_split_dev(
            pool.handle, ctypes.c_void_p(qd.data_ptr()), d, m, split, hi,
            *[ctypes.c_void_p(t.data_ptr()) for t in out], None))
        torch.cuda.synchronize()
        ia, sa, ic, sc = [t.cpu().numpy() for t in out]
        oc_idx, oc_sim, _ = oracle_knn.search(X[split:hi], Q, 1)
        np.testing.assert_array_equal(ia, oa_idx[:, 0])
        np.testing.assert_array_equal(ic, oc_idx[:, 0] + split)
        assert sa.tobytes() == oa_sim[:, 0].tobytes() and sc.tobytes() == oc_sim[:, 0].tobytes()
    # the plain search_best on the same handle still agrees (one window)
    rows, sims = pool.search_batch(Q[:min(m, 16)], 1)
    o_idx, o_sim, _ = oracle_knn.search(X, Q[:min(m, 16)], 1)
    np.testing.assert_array_equal(rows, o_idx)
    assert sims.tobytes() == o_sim.tobytes()
    pool.close()


@pytest.mark.parametrize("n,d,m,k", [
    (1000, 4096, 16, 1),       # the benchmark's descriptor size on a young robot's pool
    (1000, 4096, 1, 10),
    (17, 2048, 5, 3),          # two strips, the second with one row; 4 chunks = one per wave
    (4096, 2048, 16, 1),       # exactly as many strips as CUs
    (333, 8192, 7, 400),       # 16 chunks, k > n
    (50, 3072, 16, 1),         # 6 chunks: two waves with two, one with... none for the last
    (640, 2304, 3, 5),         # odd number of slices (9): the last chunk is half a chunk
    # CSLAM_SCAN_KSPLIT_STRIPS=3: up to three strips per block, the waves' runs of chunks cross strip boundaries
    (5000, 4096, 16, 1),       # 313 strips on 256 blocks: 57 blocks with two strips
    (8192, 4096, 7, 10),       # two strips for every block: each wave a half strip
    (10007, 4096, 16, 1),      # 626 strips, the last with 7 rows: blocks of two and three strips
    (12288, 4096, 3, 1),       # three strips for every block: 24 chunks, six per wave
    (9000, 2304, 16, 4),       # three strips x 5 chunks (the last one slice only): runs of 3/4/4/4 chunks
    (6000, 1792, 5, 1),        # 7 slices = 4 chunks, the last a single slice; two strips per block
    (7000, 8192, 2, 1),        # 16 chunks: 438 strips would need two strips = 32 partials per block: k_scan_dma
])
def test_small_pool_k_split_scan_is_bit_identical(gpu, knobs, n, d, m, k):
    """k_scan_dma_ks (pools of up to three strips per CU, descriptors of >= 4 chunks): the waves of
    a block split the chunks of the block's strips evenly and wave s folds the chunk partials of
    strip s in order — bit-identical to the oracle's sequential fold."""
    knobs.set("CSLAM_SCAN_KSPLIT_STRIPS", "3")   # default 1: more strips per block measured slower
    knobs.set("CSLAM_SMALL_BEST", "0")           # (round 4: the best match on <= 4096 rows is k_small_best's)
    rs = np.random.RandomState(n + d + m + k)
    X = (rs.rand(n, d).astype(np.float32) - 0.3)
    Q = rs.rand(m, d).astype(np.float32) - 0.3
    X[n // 2] = X[0]                                    # a tie → lowest row
    Q[0] = X[0]
    pool = _pool(gpu, X, dim=d, capacity_hint=n)
    rows, sims = pool.search_batch(Q, k)
    kernel = gpu.cslam_pool_scan_kernel(pool.handle).decode()
    assert kernel.startswith("k_scan_dma<" if (n, d) == (7000, 8192) else "k_scan_dma_ks"), kernel
    o_idx, o_sim, _ = oracle_knn.search(X, Q, k)
    kk = min(k, n)
    np.testing.assert_array_equal(rows, o_idx[:, :kk])
    assert sims.tobytes() == o_sim[:, :kk].tobytes()
    assert rows[0, 0] == 0
    pool.close()


def test_narrow_gemm_tiles_for_129_to_192_queries(gpu):
    """k_gemm_dma<2,2,TJ=1> (128 x 64 tiles: three 64-query column blocks instead of two padded
    128-query ones) — same (k, k+4) operand pairing and chunk folds, bit-identical to the oracle;
    a pool with a partial last row tile and more rows than one wave of tiles.  Since round 3 only
    descriptors of <= 16 floats get here (everything longer takes k_scan_mq passes up to ~500
    queries), hence the 16-d pool."""
    rs = np.random.RandomState(192)
    n, d, m = 70_001, 16, 180
    X = rs.rand(n, d).astype(np.float32) - 0.2
    Q = rs.rand(m, d).astype(np.float32) - 0.2
    pool = _pool(gpu, X, dim=d, capacity_hint=n)
    rows, sims = pool.search_batch(Q, 3)
    assert gpu.cslam_pool_scan_kernel(pool.handle).decode().endswith("TJ=1>")
    o_idx, o_sim, _ = oracle_knn.search(X, Q, 3)
    np.testing.assert_array_equal(rows, o_idx)
    assert sims.tobytes() == o_sim.tobytes()
    pool.close()


@pytest.mark.parametrize("dim,stride", [(4096, 4096 + 128), (1024, 1024 + 128), (512, 512), (128, 128),
                                        (100, 128), (33, 64), (10, 16), (3000, 3008)])
def test_row_layout_padding_is_invisible_to_callers(gpu, dim, stride):
    """Rows are stored zero-padded to a multiple of 64 floats (16 for tiny descriptors) and 512 B
    further apart when the stride would be a multiple of 4 KiB: the device view
    reports that stride, `.data` still reads back the dense rows, the zero padding is really zero,
    and a pool grown across a reallocation keeps every row."""
    import ctypes
    torch = pytest.importorskip("torch")
    from cslam_amd import _lib
    rs = np.random.RandomState(dim)
    X = rs.rand(1500, dim).astype(np.float32)
    m = _pool(gpu, X[:900], dim=dim)              # capacity 1000 ...
    m.add_items(X[900:], list(range(900, 1500)))  # ... grown to 2000
    rows_dev, row_stride = ctypes.c_void_p(), ctypes.c_int64()
    _lib.check(gpu.cslam_pool_device_view(m.handle, ctypes.byref(rows_dev), ctypes.byref(row_stride)))
    assert row_stride.value == stride
    np.testing.assert_array_equal(m.data.cpu().numpy()[:1500], X)
    dp = (dim + 63) // 64 * 64 if dim > 16 else 16
    if dp > dim:                                  # the padding behind row 1234 is zero
        pad = torch.empty(dp - dim, dtype=torch.float32, device="cuda")
        _lib.check(gpu.cslam_device_copy(ctypes.c_void_p(pad.data_ptr()),
                                         ctypes.c_void_p(rows_dev.value + (1234 * stride + dim) * 4),
                                         (dp - dim) * 4, 2, None, 1))
        assert float(pad.abs().max()) == 0.0
    q = X[[7, 1234]]
    rows, sims = m.search_batch(q, 1)
    np.testing.assert_array_equal(rows[:, 0], [7, 1234])
    m.close()


@pytest.mark.parametrize("n,d,m,k", [
    (4097, 100, 16, 10),      # 112-float rows: register-ring scan, two select parts
    (3000, 1000, 7, 1),       # 1008-float rows, best match through dots + select
    (2500, 4096, 16, 1),      # dense 16-KiB rows (no stride pad): the LDS-DMA scan on the round-1 layout
    (900, 48, 5, 3),
])
def test_round1_row_layout_knobs_still_bit_exact(gpu, monkeypatch, knobs, n, d, m, k):
    """CSLAM_DIM_ALIGN=16 / CSLAM_ROW_PAD=0 (read at pool creation) keep the tight round-1 layout:
    lengths that are no multiple of 64 floats then take the register-ring scan.  Same bits."""
    import ctypes
    from cslam_amd import _lib
    monkeypatch.setenv("CSLAM_DIM_ALIGN", "16")
    monkeypatch.setenv("CSLAM_ROW_PAD", "0")
    knobs.set("CSLAM_SMALL_BEST", "0")           # (the general path's kernels on this layout)
    rs = np.random.RandomState(n + d + 1)
    X = (rs.rand(n, d).astype(np.float32) - 0.4)
    Q = rs.rand(m, d).astype(np.float32)
    mt = _pool(gpu, X, dim=d, capacity_hint=n)
    rows_dev, row_stride = ctypes.c_void_p(), ctypes.c_int64()
    _lib.check(gpu.cslam_pool_device_view(mt.handle, ctypes.byref(rows_dev), ctypes.byref(row_stride)))
    assert row_stride.value == (d + 15) // 16 * 16
    rows, sims = mt.search_batch(Q, k)
    kernel = gpu.cslam_pool_scan_kernel(mt.handle).decode()
    assert kernel.startswith("k_scan<" if row_stride.value % 64 else "k_scan_dma"), kernel
    o_idx, o_sim, _ = oracle_knn.search(X, Q, k)
    np.testing.assert_array_equal(rows, o_idx)
    assert sims.tobytes() == o_sim.tobytes()
    mt.close()
