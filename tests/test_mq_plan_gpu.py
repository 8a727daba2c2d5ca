"""How the multi-tile scan deals a pool's strips over the waves (round 4, DESIGN.md §5 "k_scan_mq, round 4"): which form a
pool size takes - main launch + K-split tail, or the whole window through the K-split launch + k_mq_combine - and that
every form gives the same bits as the C oracle: the (strip, 512-float chunk) partials are folded in chunk order whoever
computed them (nns_matching.py:44-63 has one order-free answer per query; ours is pinned by oracle/knn_oracle.c)."""
import ctypes

import numpy as np
import pytest

from oracle import knn as oracle_knn

pytestmark = pytest.mark.gpu


def _pool(X):
    from cslam_amd.nns_matching import NearestNeighborsMatching
    m = NearestNeighborsMatching(dim=X.shape[1], capacity_hint=X.shape[0])
    m.add_items(X, list(range(X.shape[0])))
    return m


def _data(n, d, m, seed):
    rs = np.random.RandomState(seed)
    X = rs.rand(n, d).astype(np.float32) - 0.25
    Q = rs.rand(m, d).astype(np.float32)
    X[n - 3] = X[11]           # a tie across strips / chunks / blocks resolves to the lower row
    Q[m - 1] = X[11]
    return X, Q


@pytest.mark.parametrize("n,d,m,expect", [
    (25000, 1024, 80, "part"),      # 1563 strips, five tiles: 1.5 strips per wave -> all K-split
    (12000, 1024, 48, "part"),      # fewer strips than waves, three tiles
    (70000, 1024, 80, "best"),      # 4.3 strips per wave: main launch (+ tail)
    (25000, 1024, 32, "best"),      # two tiles lean on the HBM: never all K-split
])
def test_plan_by_pool_size_and_bits(gpu, n, d, m, expect):
    X, Q = _data(n, d, m, n + m)
    p = _pool(X)
    rows, sims = p.search_batch(Q, 1)
    name = gpu.cslam_pool_scan_kernel(p.handle).decode()
    assert name.startswith("k_scan_mq")
    assert ("part>" in name and "k_mq_combine" in name) == (expect == "part"), name
    o_idx, o_sim, _ = oracle_knn.search(X, Q, 1)
    np.testing.assert_array_equal(rows, o_idx[:, :1])
    assert sims.tobytes() == o_sim[:, :1].astype(np.float32).tobytes()
    assert rows[m - 1, 0] == 11
    p.close()


@pytest.mark.parametrize("n,d,m,k", [
    (9000, 4096, 80, 1),       # eight chunks, 563 strips
    (20011, 192, 80, 1),       # 64-float slices only, a ragged last strip
    (20011, 64, 48, 1),        # one chunk per row: the K split degenerates to one part
    (7001, 640, 40, 3),        # five slices = one whole chunk + a quarter; k > 1: dots through k_mq_combine
    (4100, 2048, 33, 10),
])
def test_every_form_of_the_pass_gives_the_same_bits(gpu, knobs, n, d, m, k):
    """Default plan, main + tail only, all K-split forced with one and two waves per SIMD: identical results."""
    X, Q = _data(n, d, m, 3 * n + d)
    o_idx, o_sim, o_cnt = oracle_knn.search(X, Q, k)
    kk = int(o_cnt[0])
    seen = set()
    for env in ({}, {"CSLAM_MQ_ALLPART": "0"}, {"CSLAM_MQ_ALLPART": "2"}, {"CSLAM_MQ_ALLPART": "2", "CSLAM_MQ_PART_WAVES": "4"},
                {"CSLAM_MQ_ALLPART": "0", "CSLAM_MQ_TAIL_SHAPE": "2"}, {"CSLAM_MQ_ALLPART": "2", "CSLAM_MQ_PART_BLOCKS2": "0"}):
        knobs.only(env)
        p = _pool(X)
        rows, sims = p.search_batch(Q, k)
        seen.add(gpu.cslam_pool_scan_kernel(p.handle).decode())
        np.testing.assert_array_equal(rows, o_idx[:, :kk])
        assert sims.tobytes() == o_sim[:, :kk].astype(np.float32).tobytes()
        p.close()
    assert len(seen) >= 2, seen     # the knobs did select different forms


def test_split_window_through_the_k_split_pass(gpu):
    """cslam_pool_search_best_split_dev on a pool that takes the all-K-split form: best row before and from a split
    row that falls INSIDE a strip, against the oracle on the two row ranges (the sharded step's two questions,
    loop_closure_sparse_matching.py:37-76)."""
    import torch
    from cslam_amd import _lib
    n, d, m = 25000, 512, 80
    X, Q = _data(n, d, m, 99)
    p = _pool(X)
    qd = torch.from_numpy(Q).cuda()
    for split in (n - 10, n - 37, 16 * 700 + 5, 1):
        ia = torch.empty(m, dtype=torch.int32, device="cuda"); sa = torch.empty(m, device="cuda")
        ic = torch.empty(m, dtype=torch.int32, device="cuda"); sc = torch.empty(m, device="cuda")
        _lib.check(gpu.cslam_pool_search_best_split_dev(p.handle, ctypes.c_void_p(qd.data_ptr()), d, m, split, n,
                                                        ctypes.c_void_p(ia.data_ptr()), ctypes.c_void_p(sa.data_ptr()),
                                                        ctypes.c_void_p(ic.data_ptr()), ctypes.c_void_p(sc.data_ptr()), None))
        torch.cuda.synchronize()
        name = gpu.cslam_pool_scan_kernel(p.handle).decode()
        assert "part>" in name, name
        oa, osa, _ = oracle_knn.search(X[:split], Q, 1)
        oc, osc, _ = oracle_knn.search(X[split:], Q, 1)
        np.testing.assert_array_equal(ia.cpu().numpy(), oa[:, 0])
        assert sa.cpu().numpy().tobytes() == osa[:, 0].astype(np.float32).tobytes()
        np.testing.assert_array_equal(ic.cpu().numpy(), oc[:, 0] + split)
        assert sc.cpu().numpy().tobytes() == osc[:, 0].astype(np.float32).tobytes()
    p.close()


def test_pools_of_60000_then_12000_rows_in_one_process_under_alternating_plans(gpu, knobs):
    """The sequence behind the unexplained memory fault of profiles/r04_mq_plan_ab_2.txt: one process, a
    60 000-row x 4096-d pool scanned with 80 queries under the seven plan settings of that day, alternating, then -
    its buffers freed - a 12 000-row pool under the same alternation (the record stops where the second pool starts).
    Every setting must give the oracle's bits on both pools; mq_pass checks every launch against the pool's
    allocations on the host, tools/mq_audit.py runs the same sequence with every device address checked
    (profiles/README.md, r05_mq_audit.txt)."""
    import torch
    from cslam_amd import _lib
    from cslam_amd.nns_matching import NearestNeighborsMatching
    settings = [
        {"CSLAM_MQ_TAIL_SHAPE": "2", "CSLAM_MQ_ALLPART": "0"}, {}, {"CSLAM_MQ_TAIL_SHAPE": "1", "CSLAM_MQ_ALLPART": "0"},
        {"CSLAM_MQ_TAIL": "0"}, {"CSLAM_MQ_ALLPART_GROUPS": "32", "CSLAM_MQ_ALLPART_KS": "8"},
        {"CSLAM_MQ_ALLPART_GROUPS": "64", "CSLAM_MQ_ALLPART_KS": "4"}, {"CSLAM_MQ_ALLPART_GROUPS": "128", "CSLAM_MQ_ALLPART_KS": "2"},
    ]
    d, m = 4096, 80
    gen = torch.Generator(device="cuda").manual_seed(618)
    for n in (60000, 12000):
        Xd = torch.rand((n, d), device="cuda", generator=gen)
        Qd = torch.rand((m, d), device="cuda", generator=gen)
        p = NearestNeighborsMatching(dim=d, capacity_hint=n)
        p.add_items_device(Xd.data_ptr(), n, d, items=None, stream=None)
        idx = torch.empty((m, 1), dtype=torch.int32, device="cuda"); sim = torch.empty((m, 1), device="cuda")
        X = Xd.cpu().numpy(); Q = Qd.cpu().numpy()
        # the oracle on 16 of the 80 queries (strict-chain C: the full block would take the CPU half a minute)
        sel = np.arange(0, m, 5)
        o_idx, o_sim, _ = oracle_knn.search(X, Q[sel], 1)
        ref = None
        seen = set()
        for rep in range(2):
            for env in settings:
                knobs.only(env)
                for _ in range(4):
                    _lib.check(gpu.cslam_pool_search_dev(p.handle, ctypes.c_void_p(Qd.data_ptr()), d, m, 1,
                                                         ctypes.c_void_p(idx.data_ptr()), ctypes.c_void_p(sim.data_ptr()), None))
                torch.cuda.synchronize()
                seen.add(gpu.cslam_pool_scan_kernel(p.handle).decode())
                out = (idx.cpu().numpy().copy(), sim.cpu().numpy().copy())
                if ref is None:
                    ref = out
                    np.testing.assert_array_equal(out[0][sel, 0], o_idx[:, 0])
                    assert out[1][sel, 0].tobytes() == o_sim[:, 0].astype(np.float32).tobytes()
                assert out[0].tobytes() == ref[0].tobytes() and out[1].tobytes() == ref[1].tobytes(), (n, env)
        assert len(seen) >= 2, seen
        p.close()
        del Xd, Qd, idx, sim


def test_both_forms_of_the_k_split_launch_on_one_pool(gpu, knobs):
    """The whole window through the K-split launch as two four-wave blocks per CU (round 5, default) and as one
    eight-wave block (round 4, CSLAM_MQ_PART_BLOCKS2=0): same bits as the oracle, query counts going down and up
    on one pool, the split search in between, ties -> lowest row."""
    import torch
    from cslam_amd import _lib
    n, d = 25000, 512
    X, Q = _data(n, d, 80, 4242)
    X[n - 1] = X[5]; X[16 * 700 + 3] = X[5]; Q[7] = X[5]          # three equal rows in different strips
    p = _pool(X)
    qd = torch.from_numpy(Q).cuda()
    o_idx, o_sim, _ = oracle_knn.search(X, Q, 1)
    names = set()
    for blocks2 in ("1", "0"):
        knobs.set("CSLAM_MQ_PART_BLOCKS2", blocks2)
        for m in (80, 33, 48, 80, 17):
            rows, sims = p.search_batch(Q[:m], 1)
            name = gpu.cslam_pool_scan_kernel(p.handle).decode()
            assert ("part>" in name) == (m > 32), (m, name)   # two tiles never take the all-K-split form
            if m > 32:
                names.add(name)
            np.testing.assert_array_equal(rows, o_idx[:m, :1])
            assert sims.tobytes() == o_sim[:m, :1].astype(np.float32).tobytes()
            assert m <= 7 or rows[7, 0] == 5
            if m == 48:   # the split search through the same launch, then back
                split = 16 * 700 + 5
                ia = torch.empty(m, dtype=torch.int32, device="cuda"); sa = torch.empty(m, device="cuda")
                ic = torch.empty(m, dtype=torch.int32, device="cuda"); sc = torch.empty(m, device="cuda")
                _lib.check(gpu.cslam_pool_search_best_split_dev(p.handle, ctypes.c_void_p(qd.data_ptr()), d, m, split, n,
                                                                ctypes.c_void_p(ia.data_ptr()), ctypes.c_void_p(sa.data_ptr()),
                                                                ctypes.c_void_p(ic.data_ptr()), ctypes.c_void_p(sc.data_ptr()), None))
                torch.cuda.synchronize()
                oa, osa, _ = oracle_knn.search(X[:split], Q[:m], 1)
                oc, osc, _ = oracle_knn.search(X[split:], Q[:m], 1)
                np.testing.assert_array_equal(ia.cpu().numpy(), oa[:, 0])
                assert sa.cpu().numpy().tobytes() == osa[:, 0].astype(np.float32).tobytes()
                np.testing.assert_array_equal(ic.cpu().numpy(), oc[:, 0] + split)
                assert sc.cpu().numpy().tobytes() == osc[:, 0].astype(np.float32).tobytes()
    assert any("2 per CU" in x for x in names) and any("NW=8" in x for x in names), names
    p.close()
