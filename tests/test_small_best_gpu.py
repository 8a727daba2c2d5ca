"""search_best - and search with 2 <= k <= 32 - on a SMALL pool in one launch (k_small_search, round 4): the reference's per-keyframe call
(nns_matching.py:66-78: search(query, 1) on a pool of a few hundred 128-d descriptors, BASELINE
configs[0]) without the query copy, the stage kernel and the final-level kernel of the general path.
Bit for bit against the C oracle and against the general path (CSLAM_SMALL_BEST=0); queries from host
memory, from HBM (aligned, strided, unaligned), through a row window; ties, zero rows, zero queries;
call after call (the kernel leaves its scratch zeroed for the next one)."""
import ctypes

import numpy as np
import pytest

from oracle import knn as oracle_knn

pytestmark = pytest.mark.gpu


def _pool(X):
    from cslam_amd.nns_matching import NearestNeighborsMatching
    m = NearestNeighborsMatching(dim=X.shape[1], capacity_hint=max(1, X.shape[0]))
    m.add_items(X, list(range(X.shape[0])))
    return m


def _data(n, d, m, seed):
    rs = np.random.RandomState(seed)
    X = rs.rand(n, d).astype(np.float32) - 0.3
    Q = rs.rand(m, d).astype(np.float32) - 0.2
    if n > 40:
        X[n - 2] = X[5]            # a tie across strips and blocks resolves to the lower row
        X[20] = 0.0                # a zero row: similarity 0
        Q[m - 1] = X[5]
    if m > 2:
        Q[1] = 0.0                 # a zero query: every similarity 0, row 0 wins
    return X, Q


SHAPES = [
    (1000, 128, 1),     # cfg1: one query, 63 strips, four strips per block
    (1, 128, 1), (15, 128, 2), (16, 128, 1), (17, 128, 3),
    (4096, 128, 16),    # the largest window, a full query tile
    (1500, 4096, 1),    # eight chunks per strip: the waves of a block split K (24 MB of rows: the limit)
    (700, 4096, 2),     # two queries of 4096 floats: the most query floats taken
    (300, 1000, 4),     # dim padded to 1024: two chunks, two strips per block
    (333, 1536, 5),     # three chunks: one strip per block, a wave idle
    (100, 33, 5),       # dim padded to 64
    (50, 10, 2),        # dim padded to 16 (the LDS-DMA scans do not take such rows)
    (700, 8192, 1),     # sixteen chunks
    (999, 200, 7),      # dim % 4 == 0 but rows of 200 floats: 13 k-blocks, a ragged batch
]


@pytest.mark.parametrize("n,d,m", SHAPES)
def test_small_pool_best_match_in_one_launch(gpu, knobs, n, d, m):
    torch = pytest.importorskip("torch")
    X, Q = _data(n, d, m, 7 * n + d + m)
    o_idx, o_sim, _ = oracle_knn.search(X, Q, 1)
    p = _pool(X)
    # host-pointer API, twice (scratch reset), then per query like the reference does
    for _ in range(2):
        rows, sims = p.search_batch(Q, 1)
        assert gpu.cslam_pool_scan_kernel(p.handle).decode() == "k_small_best"
        np.testing.assert_array_equal(rows, o_idx[:, :1])
        assert sims.tobytes() == o_sim[:, :1].astype(np.float32).tobytes()
    for j in range(m):
        item, sim = p.search_best(Q[j])
        assert item == int(o_idx[j, 0]) and np.float32(sim).tobytes() == np.float32(o_sim[j, 0]).tobytes()
    if n > 40:
        assert rows[m - 1, 0] == 5
        if m > 2:
            assert rows[1, 0] == 0 and sims[1, 0] == 0.0
    # device-pointer API: aligned, strided (row stride d + 4) and unaligned (offset by one float) queries
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    idx = torch.empty((m, 1), dtype=torch.int32, device="cuda")
    sim = torch.empty((m, 1), dtype=torch.float32, device="cuda")
    wide = torch.zeros((m, d + 4), dtype=torch.float32, device="cuda")
    wide[:, :d] = torch.from_numpy(Q)
    flat = torch.zeros(m * d + 1, dtype=torch.float32, device="cuda")
    flat[1:] = torch.from_numpy(Q).reshape(-1)
    for ptr, stride in ((torch.from_numpy(Q).cuda(), d), (wide, d + 4), (flat[1:], d)):
        idx.fill_(-7)
        assert gpu.cslam_pool_search_dev(p.handle, ctypes.c_void_p(ptr.data_ptr()), stride, m, 1,
                                         ctypes.c_void_p(idx.data_ptr()), ctypes.c_void_p(sim.data_ptr()), st) == 0
        torch.cuda.synchronize()
        assert gpu.cslam_pool_scan_kernel(p.handle).decode() == "k_small_best"
        np.testing.assert_array_equal(idx.cpu().numpy(), o_idx[:, :1])
        assert sim.cpu().numpy().tobytes() == o_sim[:, :1].astype(np.float32).tobytes()
    # a row window
    lo, hi = n // 3, n - n // 5
    if hi > lo:
        q = torch.from_numpy(Q).cuda()
        assert gpu.cslam_pool_search_range_dev(p.handle, ctypes.c_void_p(q.data_ptr()), d, m, 1, lo, hi,
                                               ctypes.c_void_p(idx.data_ptr()), ctypes.c_void_p(sim.data_ptr()), st) == 0
        torch.cuda.synchronize()
        w_idx, w_sim, _ = oracle_knn.search(X[lo:hi], Q, 1)
        np.testing.assert_array_equal(idx.cpu().numpy(), w_idx[:, :1] + lo)
        assert sim.cpu().numpy().tobytes() == w_sim[:, :1].astype(np.float32).tobytes()
    # the host-pointer call's variants: query read from the pinned buffer or copied first, completion by
    # polling or by stream synchronisation
    for q_src, poll in (("1", "1"), ("0", "1"), ("1", "0"), ("0", "0")):
        knobs.set("CSLAM_SMALL_BEST_Q", q_src)
        knobs.set("CSLAM_SMALL_BEST_POLL", poll)
        for _ in range(3):
            rows3, sims3 = p.search_batch(Q, 1)
            assert gpu.cslam_pool_scan_kernel(p.handle).decode() == "k_small_best"
            np.testing.assert_array_equal(rows3, rows)
            assert sims3.tobytes() == sims.tobytes()
    # the general path gives the same bytes
    knobs.set("CSLAM_SMALL_BEST", "0")
    rows2, sims2 = p.search_batch(Q, 1)
    assert gpu.cslam_pool_scan_kernel(p.handle).decode() != "k_small_best"
    np.testing.assert_array_equal(rows2, rows)
    assert sims2.tobytes() == sims.tobytes()
    p.close()


def test_small_pool_launch_has_its_limits(gpu):
    """Beyond 4096 rows, 24 MB of rows or 8192 query floats per call, or for k > 32, the general path runs."""
    rs = np.random.RandomState(0)
    X = rs.rand(4097, 128).astype(np.float32)
    p = _pool(X)
    Q = rs.rand(3, 128).astype(np.float32)
    rows, sims = p.search_batch(Q, 1)
    assert gpu.cslam_pool_scan_kernel(p.handle).decode() != "k_small_best"
    o_idx, o_sim, _ = oracle_knn.search(X, Q, 1)
    np.testing.assert_array_equal(rows, o_idx[:, :1])
    p.close()
    X = rs.rand(500, 4096).astype(np.float32)
    p = _pool(X)
    for m, fused in ((2, True), (3, False)):
        Q = rs.rand(m, 4096).astype(np.float32)
        rows, sims = p.search_batch(Q, 1)
        assert (gpu.cslam_pool_scan_kernel(p.handle).decode() == "k_small_best") == fused
        o_idx, o_sim, _ = oracle_knn.search(X, Q, 1)
        np.testing.assert_array_equal(rows, o_idx[:, :1])
        assert sims.tobytes() == o_sim[:, :1].astype(np.float32).tobytes()
    rows, sims = p.search_batch(Q[:1], 4)
    assert gpu.cslam_pool_scan_kernel(p.handle).decode() == "k_small_topk"
    rows, sims = p.search_batch(Q[:1], 33)
    assert not gpu.cslam_pool_scan_kernel(p.handle).decode().startswith("k_small")
    p.close()
    X = rs.rand(1600, 4096).astype(np.float32)
    p = _pool(X)
    rows, sims = p.search_batch(Q[:1], 1)
    assert gpu.cslam_pool_scan_kernel(p.handle).decode() != "k_small_best"
    p.close()


@pytest.mark.parametrize("n,d,m,k", [
    (1000, 128, 1, 10),     # the intra-robot match of lcsm.py:78-97 at cfg1's scale: nb_best_matches = 10
    (1000, 128, 1, 2), (1000, 128, 1, 32),
    (4096, 128, 16, 10),    # 256 strips, a full query tile: sixteen rank selects by the last block
    (17, 128, 3, 16),       # two strips, k of 17 rows
    (5, 128, 4, 10),        # k > n: (-1, -inf) tails
    (1, 64, 1, 2),
    (1500, 4096, 1, 10),    # eight chunks per strip
    (300, 1000, 4, 5),
    (100, 33, 5, 3), (50, 10, 2, 7),
    (4000, 64, 3, 16),      # the zero query: 250 strips tie at the bound, rows 0..15 in order
])
def test_small_pool_top_k_in_one_launch(gpu, knobs, n, d, m, k):
    """search with 2 <= k <= 32 on a small pool (k_small_search<true>, reported as k_small_topk): the blocks
    leave the strips' composites and the raw dots, the last block to finish ranks the rows of the k best
    strips (topk_from_parts_body, the code of k_topk_from_parts).  Against the oracle bit for bit, against
    the general path, host- and device-pointer API, a row window, call after call."""
    torch = pytest.importorskip("torch")
    X, Q = _data(n, d, m, 11 * n + d + m + k)
    if n > 300:
        X[n // 2] = X[5]           # three equal rows in different strips: lowest row first
    o_idx, o_sim, o_cnt = oracle_knn.search(X, Q, k)
    kk = int(o_cnt[0])
    p = _pool(X)
    for _ in range(3):
        rows, sims = p.search_batch(Q, k)
        assert gpu.cslam_pool_scan_kernel(p.handle).decode() == "k_small_topk"
        assert rows.shape == (m, kk)
        np.testing.assert_array_equal(rows, o_idx[:, :kk])
        assert sims.tobytes() == o_sim[:, :kk].astype(np.float32).tobytes()
    if n > 300:
        assert list(rows[m - 1][:3]) == [5, n // 2, n - 2][:min(k, 3)]
        if m > 2:
            assert list(rows[1]) == list(range(kk))
    # reference-shaped call: search(query, k) -> items, similarities
    items, s1 = p.search(Q[0], k)
    assert items == [int(r) for r in o_idx[0, :kk]] and np.asarray(s1).tobytes() == o_sim[0, :kk].astype(np.float32).tobytes()
    # device-pointer API on a row window
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    lo, hi = n // 3, n - n // 5
    if hi > lo:
        q = torch.from_numpy(Q).cuda()
        idx = torch.empty((m, k), dtype=torch.int32, device="cuda")
        sim = torch.empty((m, k), dtype=torch.float32, device="cuda")
        assert gpu.cslam_pool_search_range_dev(p.handle, ctypes.c_void_p(q.data_ptr()), d, m, k, lo, hi,
                                               ctypes.c_void_p(idx.data_ptr()), ctypes.c_void_p(sim.data_ptr()), st) == 0
        torch.cuda.synchronize()
        assert gpu.cslam_pool_scan_kernel(p.handle).decode() == "k_small_topk"
        w_idx, w_sim, w_cnt = oracle_knn.search(X[lo:hi], Q, k)
        wk = int(w_cnt[0])
        np.testing.assert_array_equal(idx.cpu().numpy()[:, :wk], w_idx[:, :wk] + lo)
        assert sim.cpu().numpy()[:, :wk].tobytes() == w_sim[:, :wk].astype(np.float32).tobytes()
        assert np.all(idx.cpu().numpy()[:, wk:] == -1)
    # k = 1 only (the round's first form of the launch), then the general path: same bytes
    for setting in ("1", "0"):
        knobs.set("CSLAM_SMALL_BEST", setting)
        rows2, sims2 = p.search_batch(Q, k)
        assert not gpu.cslam_pool_scan_kernel(p.handle).decode().startswith("k_small")
        np.testing.assert_array_equal(rows2, rows)
        assert sims2.tobytes() == sims.tobytes()
    p.close()


def test_growing_pool_keyframe_by_keyframe(gpu):
    """The reference's pattern (loop_closure_sparse_matching.py:26-32, 78-97): add a descriptor, match the
    next one, from the empty pool on — every answer equal to the oracle's on the pool so far."""
    rs = np.random.RandomState(3)
    from cslam_amd.nns_matching import NearestNeighborsMatching
    d = 128
    D = rs.rand(80, d).astype(np.float32) - 0.4
    D[40] = D[3]
    p = NearestNeighborsMatching()
    assert p.search_best(D[0]) == (None, None)
    for i in range(80):
        if i > 0:
            item, sim = p.search_best(D[i])
            o_idx, o_sim, _ = oracle_knn.search(D[:i], D[i:i + 1], 1)
            assert item == int(o_idx[0, 0]), i
            assert np.float32(sim).tobytes() == np.float32(o_sim[0, 0]).tobytes()
            assert gpu.cslam_pool_scan_kernel(p.handle).decode() == "k_small_best"
            items, sims = p.search(D[i], 10)            # match_local_loop_closures' call (lcsm.py:85)
            o_idx, o_sim, o_cnt = oracle_knn.search(D[:i], D[i:i + 1], 10)
            kk = int(o_cnt[0])
            assert items == [int(r) for r in o_idx[0, :kk]], i
            assert np.asarray(sims).tobytes() == o_sim[0, :kk].astype(np.float32).tobytes()
            assert gpu.cslam_pool_scan_kernel(p.handle).decode() == "k_small_topk"
        p.add_item(D[i], i)
    item, sim = p.search_best(D[3])
    assert item == 3
    p.close()


@pytest.mark.parametrize("n,d,m,k", [
    (20000, 128, 1, 1),       # k_scan_dma<best> + k_best_pair, one block: signals without a ticket
    (20000, 128, 16, 1),      # sixteen blocks: the last one to take a ticket signals
    (20000, 128, 5, 10),      # k_topk_from_parts<WaveParts>
    (9000, 512, 40, 1),       # k_scan_mq + k_best_pair
    (9000, 512, 40, 7),       # k_scan_mq + k_topk_from_parts<PanelParts>
    (9000, 512, 96, 1),       # two passes: only the last pass's final kernel signals
    (5000, 64, 3, 200),       # two-level select: no signalling launch site, the call synchronises the stream
    (5000, 33, 9, 1),         # 1188 B of queries: the copy kernel's float4 body + a one-float tail
    (5000, 30, 7, 3),         # 840 B: a two-float tail
    (20000, 4096, 16, 1),     # 256 KiB of queries: polled since the queries travel by the copy kernel
])
def test_polled_completion_of_the_host_call(gpu, knobs, n, d, m, k):
    """cslam_pool_search's small calls return when the call's final kernel has written the call's number into a
    pinned word (signal_done), not after a stream synchronisation.  Call after call with DIFFERENT queries -
    a stale result or an early return would show - against the oracle, polling forced for every size (=2),
    default (<= 512 KiB of queries) and off; queries moved by the copy kernel and by hipMemcpyAsync."""
    rs = np.random.RandomState(n + d + m + k)
    X = rs.rand(n, d).astype(np.float32) - 0.3
    p = _pool(X)
    for setting, qcopy in (("2", "0"), ("1", "0"), ("0", "0"), ("2", "1"), ("1", "1")):
        knobs.set("CSLAM_SMALL_BEST_POLL", setting)
        knobs.set("CSLAM_HOST_Q_COPY", qcopy)      # 1: hipMemcpyAsync instead of the copy kernel
        for call in range(4):
            Q = rs.rand(m, d).astype(np.float32) - 0.2
            Q[0] = X[rs.randint(n)]
            rows, sims = p.search_batch(Q, k)
            o_idx, o_sim, o_cnt = oracle_knn.search(X, Q, k)
            kk = int(o_cnt[0])
            np.testing.assert_array_equal(rows, o_idx[:, :kk])
            assert sims.tobytes() == o_sim[:, :kk].astype(np.float32).tobytes()
    p.close()


def test_device_calls_on_several_streams_stay_ordered(gpu):
    """The event that orders device-pointer calls against calls on OTHER streams is recorded lazily (round 4: by
    the first call that needs it, on the stream the work went to).  Searches on two caller streams and the
    host-pointer API alternate on one pool - they share its candidate / dots buffers - and every result is the
    oracle's."""
    torch = pytest.importorskip("torch")
    rs = np.random.RandomState(5)
    n, d = 30000, 512
    X = rs.rand(n, d).astype(np.float32) - 0.3
    p = _pool(X)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    for rnd in range(4):
        want, got = [], []
        for m, k, stream in ((40, 1, sa), (16, 10, sb), (80, 5, sa), (3, 1, sb)):
            Q = rs.rand(m, d).astype(np.float32) - 0.2
            with torch.cuda.stream(stream):
                q = torch.from_numpy(Q).cuda()
                idx = torch.empty((m, k), dtype=torch.int32, device="cuda")
                sim = torch.empty((m, k), dtype=torch.float32, device="cuda")
                assert gpu.cslam_pool_search_dev(p.handle, ctypes.c_void_p(q.data_ptr()), d, m, k,
                                                 ctypes.c_void_p(idx.data_ptr()), ctypes.c_void_p(sim.data_ptr()),
                                                 ctypes.c_void_p(stream.cuda_stream)) == 0
            want.append(oracle_knn.search(X, Q, k))
            got.append((idx, sim, q))
            if rnd % 2 == 0:       # a host-pointer call right behind it (the pool's own stream)
                Qh = rs.rand(2, d).astype(np.float32)
                rows, sims = p.search_batch(Qh, 3)
                o_idx, o_sim, _ = oracle_knn.search(X, Qh, 3)
                np.testing.assert_array_equal(rows, o_idx)
                assert sims.tobytes() == o_sim.tobytes()
        torch.cuda.synchronize()
        for (o_idx, o_sim, _), (idx, sim, _) in zip(want, got):
            np.testing.assert_array_equal(idx.cpu().numpy(), o_idx)
            assert sim.cpu().numpy().tobytes() == o_sim.tobytes()
    p.close()


def test_two_threads_two_pools(gpu):
    """Two host threads, each with a pool of its own (the node matches against one pool per robot), call the
    host-pointer API at the same time - every call polls its own pool's completion word and uses its own pool's
    scratch: every answer is the oracle's."""
    import threading
    rs = np.random.RandomState(9)
    pools, data = [], []
    for n, d in ((1000, 128), (3000, 128), (9000, 256)):
        X = rs.rand(n, d).astype(np.float32) - 0.3
        Q = rs.rand(40, d).astype(np.float32) - 0.2
        o1 = oracle_knn.search(X, Q, 1)
        o10 = oracle_knn.search(X, Q, 10)
        pools.append(_pool(X))
        data.append((Q, o1, o10))
    errors = []

    def work(i):
        try:
            p = pools[i]
            Q, (i1, s1, _), (i10, s10, _) = data[i]
            for rep in range(150):
                j = rep % len(Q)
                item, sim = p.search_best(Q[j])
                assert item == int(i1[j, 0]) and np.float32(sim).tobytes() == s1[j, :1].astype(np.float32).tobytes(), (i, rep)
                items, sims = p.search(Q[j], 10)
                assert items == [int(r) for r in i10[j]] and np.asarray(sims).tobytes() == s10[j].astype(np.float32).tobytes(), (i, rep)
        except Exception as e:     # surfaced in the main thread
            errors.append(e)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(pools))]
    [t.start() for t in threads]
    [t.join(timeout=300) for t in threads]
    assert not errors, errors
    for p in pools:
        p.close()
