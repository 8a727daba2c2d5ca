"""The device-side sharded matching step (csrc/shard.hip, csrc/comm.hip) through the C ABI.

* RCCL itself: a one-rank communicator on the one GPU of the test box (two ranks cannot share a
  device under RCCL) — unique id, init, the descriptor all-gather, in-place byte all-gather.
* The step protocol with R > 1: robot shards as THREADS of this process sharing the GPU, each
  with its own pool / shard / communicators over the `LocalGroup` transport, compared with the
  reference semantics replayed sequentially on the oracle
  (cslam/loop_closure_sparse_matching.py:37-97; replay order documented in
  cslam_amd/multi_robot.py and SURVEY.md §8e).
"""
import ctypes
import threading

import numpy as np
import pytest

from oracle import knn as oracle_knn
from test_multi_robot import DIM, MAXB, THR, _sequential_reference_with, make_blocks

pytestmark = pytest.mark.gpu


def test_rccl_one_rank_communicator(gpu):
    torch = pytest.importorskip("torch")
    from cslam_amd import _lib
    from cslam_amd.multi_robot import RcclComm
    uid = RcclComm.unique_id()
    assert len(uid) == 128 and any(uid)
    comm = RcclComm(0, 1, uid, 0)
    assert gpu.cslam_comm_rank(comm.handle) == 0 and gpu.cslam_comm_size(comm.handle) == 1
    rs = np.random.RandomState(0)
    x = torch.from_numpy(rs.rand(3, 80).astype(np.float32)).cuda()      # row stride 80 > dim 64
    all_rows = torch.full((1, 5, 64), -7.0, device="cuda")
    counts = torch.full((1,), -1, dtype=torch.int32, device="cuda")
    _lib.check(gpu.cslam_allgather_descriptors(comm.handle, ctypes.c_void_p(x.data_ptr()), 80, 3, 5, 64,
                                               ctypes.c_void_p(all_rows.data_ptr()),
                                               ctypes.c_void_p(counts.data_ptr()), None))
    torch.cuda.synchronize()
    assert counts.cpu().tolist() == [3]
    got = all_rows.cpu().numpy()[0]
    np.testing.assert_array_equal(got[:3], x.cpu().numpy()[:, :64])
    assert np.all(got[3:] == 0.0)
    # byte all-gather on the communicator's own stream, in place
    buf = torch.arange(100, dtype=torch.uint8, device="cuda")
    _lib.check(gpu.cslam_comm_allgather(comm.handle, ctypes.c_void_p(buf.data_ptr()),
                                        ctypes.c_void_p(buf.data_ptr()), 100, _lib.CSLAM_COMM_OWN_STREAM))
    torch.cuda.synchronize()
    assert buf.cpu().tolist() == list(range(100))
    st = comm.stats()
    assert st["collectives"] == 3 and st["bytes_gathered"] == 5 * 64 * 4 + 4 + 100
    assert st["rccl_version"] > 0
    comm.close()


def _intra_reference(blocks, world, steps, k, min_inbetween, thr):
    """match_local_loop_closures (lcsm.py:78-97) replayed per robot: every new keyframe is
    inserted, then searched in the pool as it is (itself included), top-k, self dropped when
    first, first far-enough and similar-enough match kept."""
    out = {r: [] for r in range(world)}
    for r in range(world):
        rows = np.zeros((0, DIM), np.float32)
        kfs = []
        for t in range(steps):
            x, ids = blocks[(t, r)]
            for i, kf in enumerate(ids):
                rows = np.vstack([rows, x[i:i + 1]])
                kfs.append(kf)
                idx, sim, cnt = oracle_knn.search(rows, x[i:i + 1], k)
                lst = [(kfs[int(j)], float(s)) for j, s in zip(idx[0][:cnt[0]], sim[0][:cnt[0]])]
                if lst and lst[0][0] == kf:
                    lst = lst[1:]
                for kfm, s in lst:
                    if abs(kfm - kf) < min_inbetween or s < thr:
                        continue
                    out[r].append((kf, kfm, np.float32(s)))
                    break
    return out


def _cfg4_fixture():
    import json
    import os
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden",
                                       "cfg4_ref.json")))


def _run_shards(world, steps, blocks, max_block, nb_best, min_inbetween, dim=DIM, thr=THR, one_comm=False):
    torch = pytest.importorskip("torch")
    from cslam_amd.multi_robot import CallbackComm, DeviceShard, LocalGroup
    from cslam_amd.nns_matching import NearestNeighborsMatching
    grp_d, grp_r = LocalGroup(world), LocalGroup(world)
    results, local, errors = [None] * world, [None] * world, []
    records = [None] * world

    def run(rank):
        try:
            torch.cuda.set_device(0)
            pool = NearestNeighborsMatching(dim=dim)
            cd = CallbackComm(rank, world, 0, grp_d.transport(rank))
            cr = cd if one_comm else CallbackComm(rank, world, 0, grp_r.transport(rank))
            sh = DeviceShard(pool, rank, world, max_block, thr, cd, cr, nb_best_matches=nb_best,
                             intra_min_inbetween=min_inbetween)
            mine, recs = [], []
            for t in range(steps):
                x, kfs = blocks[(t, rank)]
                xd = torch.from_numpy(x).cuda() if len(kfs) else torch.zeros((0, dim), device="cuda")
                sh.publish(xd, kfs)
                sh.match()
                r = sh.fetch_records()
                recs.append(r)
                _, loc = sh.apply(r)
                mine.extend(loc)
            results[rank] = {k: float(v.weight) for k, v in sh.candidate_selector.candidate_edges.items()}
            local[rank] = mine
            records[rank] = np.concatenate(recs) if recs else None
            assert pool.n == sum(len(blocks[(t, rank)][1]) for t in range(steps))
            sh.close(); cd.close()
            if not one_comm:
                cr.close()
            pool.close()
        except Exception as e:  # surface worker failures in the main thread
            errors.append(e)
            grp_d.barrier.abort()
            grp_r.barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in threads]
    [t.join(timeout=600) for t in threads]
    assert not errors, errors
    return results, local, records


@pytest.mark.parametrize("world", [2, 4])
def test_device_shard_matches_reference_semantics(gpu, world):
    """Fused mode (one scan per step, top-(b+k) split on the device): inter-robot candidate set
    identical to the sequential reference, replicas identical on every rank, every rank fetched
    the same record stream, intra-robot matches equal to match_local_loop_closures replayed."""
    steps, k, gap = 10, 4, 3
    blocks = make_blocks(world, steps, seed=5)
    results, local, records = _run_shards(world, steps, blocks, MAXB, k, gap)
    union, _ = _sequential_reference_with(blocks, world, steps)
    assert len(union) > 5
    for r in range(world):
        assert set(results[r]) == set(union)
        for key in union:
            assert abs(results[r][key] - union[key]) <= 1e-7
        assert results[r] == results[0]
        assert records[r].tobytes() == records[0].tobytes()
    # ... and with the REFERENCE itself: R reference LoopClosureSparseMatching instances fed the
    # same stream in the documented arrival order (tests/golden/gen_golden.py::gen_cfg4)
    ref = _cfg4_fixture()[f"world{world}_steps{steps}_seed5"]
    ref_union = {tuple(int(x) for x in key.strip("()").split(",")): w for key, w in ref["union"].items()}
    assert set(results[0]) == set(ref_union)
    for key, w in ref_union.items():
        assert abs(results[0][key] - w) <= 1e-6
    for r in range(world):
        assert [[a, b] for a, b, _ in local[r]] == ref["intra"][str(r)]
    intra = _intra_reference(blocks, world, steps, k, gap, THR)
    assert sum(len(v) for v in intra.values()) > 3
    for r in range(world):
        assert [(a, b) for a, b, _ in local[r]] == [(a, b) for a, b, _ in intra[r]]
        for (_, _, s), (_, _, so) in zip(local[r], intra[r]):
            assert np.float32(s) == so


def test_device_shard_bulk_blocks(gpu):
    """Blocks too large for the fused top-K (max_block + k > 64): old-row and new-row matches come
    from two searches around the append; same candidate set as the sequential reference."""
    world, steps, maxb = 3, 3, 70
    rs = np.random.RandomState(9)
    blocks, nxt = {}, [0] * world
    for t in range(steps):
        for r in range(world):
            b = int(rs.randint(40, maxb + 1))
            blocks[(t, r)] = (rs.rand(b, DIM).astype(np.float32), list(range(nxt[r], nxt[r] + b)))
            nxt[r] += b
    results, local, _ = _run_shards(world, steps, blocks, maxb, 0, 0)
    union, _ = _sequential_reference_with(blocks, world, steps)
    assert len(union) > 20
    for r in range(world):
        assert set(results[r]) == set(union)
        for key in union:
            assert abs(results[r][key] - union[key]) <= 1e-7
        assert local[r] == []                      # intra-robot matching is a fused-mode feature (refused below)


def test_device_shard_bulk_thousands_of_descriptors_per_step(gpu):
    """cfg4's bulk case in miniature: blocks of ~2000 descriptors per rank and step (chunked
    searches of 256 queries, GEMM-tile dots path), two ranks, against the sequential reference."""
    world, steps, maxb = 2, 2, 2048
    rs = np.random.RandomState(4)
    blocks, nxt = {}, [0] * world
    for t in range(steps):
        for r in range(world):
            b = int(rs.randint(1500, maxb + 1))
            blocks[(t, r)] = (rs.rand(b, DIM).astype(np.float32), list(range(nxt[r], nxt[r] + b)))
            nxt[r] += b
    results, _, _ = _run_shards(world, steps, blocks, maxb, 0, 0, thr=0.8)
    union, _ = _sequential_reference_with(blocks, world, steps, thr=0.8)
    assert len(union) > 200
    for r in range(world):
        assert set(results[r]) == set(union)
        for key in union:
            assert abs(results[r][key] - union[key]) <= 1e-7


def test_intra_matching_with_bulk_blocks_is_refused(gpu):
    """nb_best_matches > 0 needs max_block + nb_best_matches <= 64 (one fused top-K list per
    descriptor); a bulk block with intra-robot matching requested is an error, not a silent
    inter-robot-only shard."""
    from cslam_amd import _lib
    from cslam_amd.multi_robot import DeviceShard
    from cslam_amd.nns_matching import NearestNeighborsMatching
    pool = NearestNeighborsMatching(dim=64)
    with pytest.raises(_lib.CslamHipError):
        DeviceShard(pool, 0, 1, 70, 0.5, nb_best_matches=10, intra_min_inbetween=1)
    sh = DeviceShard(pool, 0, 1, 70, 0.5, nb_best_matches=0)
    sh.close(); pool.close()


@pytest.mark.parametrize("world,maxb", [(8, 10), (5, 16)])
def test_device_shard_split_best_mode_many_descriptors(gpu, world, maxb):
    """cfg4's steady state: 8 robots x 10 descriptors per message (loop_closure_detection_node.py:34)
    = 80 gathered descriptors, inter-robot matching only: ONE k_scan_mq pass with the split window
    per step, records built from its candidates - against the sequential reference semantics; the
    descriptor and the record exchange share ONE communicator (the form bench.py uses)."""
    steps, dim = 6, 128
    rs = np.random.RandomState(71 + world)
    places = rs.rand(40, dim).astype(np.float32)
    blocks, nxt = {}, [0] * world
    for t in range(steps):
        for r in range(world):
            b = int(rs.randint(maxb - 3, maxb + 1))
            x = places[rs.randint(0, len(places), size=b)] + 0.05 * rs.rand(b, dim).astype(np.float32)
            blocks[(t, r)] = (x.astype(np.float32), list(range(nxt[r], nxt[r] + b)))
            nxt[r] += b
    results, local, records = _run_shards(world, steps, blocks, maxb, 0, 0, dim=dim, thr=0.9, one_comm=True)
    union, _ = _sequential_reference_with(blocks, world, steps, thr=0.9)
    assert len(union) > 50
    for r in range(world):
        assert set(results[r]) == set(union)
        for key in union:
            assert abs(results[r][key] - union[key]) <= 1e-7
        assert results[r] == results[0]
        assert records[r].tobytes() == records[0].tobytes()


def test_pipelined_keyframe_table_growth(gpu):
    """Round-2 advisor finding: the keyframe table grew (1024 -> 2048 rows) on the exchange stream
    while the record kernel of an earlier step - queued on the MATCHING stream behind a scan - was
    still to write its rows' ids into the old table; those ids were lost (records carried keyframe
    -1).  Here: separate producer / matching streams, publish two steps ahead, the pool crosses
    1000 (pool growth) and 1024 rows (table growth), and every step's first descriptor revisits a
    row appended ONE step earlier - exactly the rows whose ids were at risk."""
    torch = pytest.importorskip("torch")
    from cslam_amd.multi_robot import DeviceShard, LoopbackComm
    from cslam_amd.nns_matching import NearestNeighborsMatching
    R, me, b, dim, steps = 2, 0, 16, 4096, 12
    rs = np.random.RandomState(12)
    base = rs.rand(960, dim).astype(np.float32)
    pool = NearestNeighborsMatching(dim=dim, capacity_hint=1000)
    pool.add_items(base, [500000 + i for i in range(960)])
    cd = LoopbackComm(me, R, 0)
    sh = DeviceShard(pool, me, R, b, 0.5, cd, cd, nb_best_matches=0)
    blocks = []
    for t in range(steps):
        x = rs.rand(b, dim).astype(np.float32)
        if t > 0:
            x[0] = blocks[t - 1][0][3]               # revisit row 3 of the previous block
        blocks.append((x, [1000 * (t + 1) + i for i in range(b)]))
    front, work = torch.cuda.Stream(), torch.cuda.Stream()
    dev = [torch.from_numpy(x).cuda() for x, _ in blocks]
    torch.cuda.synchronize()
    got, pub = [], 0
    for t in range(steps):
        while pub < steps and pub <= t + 1:          # the exchange runs two steps ahead of the fetch
            with torch.cuda.stream(front):
                sh.publish(dev[pub], blocks[pub][1], front.cuda_stream)
            pub += 1
        sh.match(work.cuda_stream)
        if t >= 1:
            got.append(sh.fetch_records())
    got.append(sh.fetch_records())
    assert pool.n == 960 + steps * b
    for t in range(1, steps):
        recs = got[t]
        # slot of the foreign copy (owner 1) of descriptor 0: best OLD row = row 3 of block t-1
        mine = [r for r in recs if int(r[0]) == me and int(r[2]) == 1 and int(r[3]) == blocks[t][1][0]]
        assert mine, f"step {t}: no record for the revisiting descriptor"
        assert int(mine[0][1]) == blocks[t - 1][1][3], (t, mine[0])
        assert all(int(r[1]) >= 0 and int(r[3]) >= 0 for r in recs)
    sh.close(); cd.close(); pool.close()


@pytest.mark.parametrize("R,b,n,dim,nb", [
    (8, 10, 20000, 128, 0),    # five query tiles, 1250 strips: 157 blocks, staged at publish time
    (3, 10, 25000, 256, 0),    # two tiles, packed because the shard calls
    (8, 10, 9000, 128, 0),     # fewer strips than waves: no CUs to spare, staged by the scan
    # with intra-robot matching (round 4: k_topk_records behind pool_topk_candidates, K = b + 10 = 20 / 18)
    (8, 10, 20000, 128, 10),   # 80 descriptors: one k_scan_mq pass, panels + K-split strips as parts
    (8, 10, 70000, 128, 10),   # ... main launch + tail
    (2, 8, 25000, 256, 10),    # 16 descriptors: k_scan_dma<best+dots>, the waves as parts
])
def test_sharded_step_on_a_mid_size_pool_matches_the_oracle(gpu, R, b, n, dim, nb):
    """The cfg4 step on a pool of tens of thousands of rows: k_scan_mq runs on just enough blocks for the
    same strips per wave and the gathered block is staged for it at publish time, on the exchange stream
    (csrc/shard.hip).  Robot 0 of R over the loopback communicator (every slot of the
    gathered block holds a copy of the own block); every foreign slot's record must carry the oracle's best
    OLD row of its descriptor - keyframe id and similarity bits - for three pipelined steps."""
    torch = pytest.importorskip("torch")
    from cslam_amd.multi_robot import DeviceShard, LoopbackComm
    from cslam_amd.nns_matching import NearestNeighborsMatching
    from oracle import knn as oracle_knn
    me, steps = 0, 3
    rs = np.random.RandomState(R + n)
    base = rs.rand(n, dim).astype(np.float32)
    pool = NearestNeighborsMatching(dim=dim, capacity_hint=n + steps * b)
    pool.add_items(base, [700000 + i for i in range(n)])
    cd = LoopbackComm(me, R, 0)
    sh = DeviceShard(pool, me, R, b, 0.0, cd, cd, nb_best_matches=nb, intra_min_inbetween=10)
    blocks = []
    for t in range(steps):
        x = rs.rand(b, dim).astype(np.float32)
        x[1] = base[rs.randint(n)]                    # an exact revisit of an old row
        blocks.append((x, [1000 * (t + 1) + i for i in range(b)]))
    dev = [torch.from_numpy(x).cuda() for x, _ in blocks]
    got = []
    sh.publish(dev[0], blocks[0][1])
    for t in range(steps):
        if t + 1 < steps:
            sh.publish(dev[t + 1], blocks[t + 1][1])
        sh.match()
        got.append(sh.fetch_records())
    rows_so_far = base
    kf_of_row = [700000 + i for i in range(n)]
    for t in range(steps):
        x, kfs = blocks[t]
        o_idx, o_sim, _ = oracle_knn.search(rows_so_far, x, 1)      # best OLD row: the rows before this step's block
        recs = got[t]
        for s_owner in range(1, R):                                     # the foreign slots (copies of the own block)
            # (a descriptor may carry a second record: its best row of THIS step's block - not the one checked here)
            mine = {int(r["robot1_keyframe_id"]): r for r in recs
                    if int(r["robot0_id"]) == me and int(r["robot1_id"]) == s_owner and int(r["robot0_keyframe_id"]) not in kfs}
            assert len(mine) == b, (t, s_owner, len(mine))
            for i in range(b):
                r = mine[kfs[i]]
                assert int(r["robot0_keyframe_id"]) == kf_of_row[int(o_idx[i, 0])], (t, s_owner, i)
                assert np.float32(r["weight"]).tobytes() == np.float32(o_sim[i, 0]).tobytes(), (t, s_owner, i)
        rows_so_far = np.vstack([rows_so_far, x])
        kf_of_row += kfs
    assert pool.n == n + steps * b
    sh.close(); cd.close(); pool.close()


def test_device_shard_single_robot_and_pipelining(gpu):
    """One robot, no communicator: nothing to match against, blocks are appended; up to four steps
    may be in flight, a fifth publish without a fetch is refused loudly."""
    torch = pytest.importorskip("torch")
    from cslam_amd import _lib
    from cslam_amd.multi_robot import DeviceShard
    from cslam_amd.nns_matching import NearestNeighborsMatching
    pool = NearestNeighborsMatching(dim=256)
    sh = DeviceShard(pool, 0, 1, 4, 0.5, nb_best_matches=3, intra_min_inbetween=1)
    rs = np.random.RandomState(1)
    x = rs.rand(16, 256).astype(np.float32)
    x[9] = x[2]                                           # keyframe 9 revisits keyframe 2
    xd = torch.from_numpy(x).cuda()
    for t in range(4):
        sh.publish(xd[4 * t:4 * t + 4], list(range(4 * t, 4 * t + 4)))
    with pytest.raises(_lib.CslamHipError):
        sh.publish(xd[:4], [100, 101, 102, 103])
    local = []
    for t in range(4):
        sh.match()
    for t in range(4):
        _, loc = sh.apply(sh.fetch_records())
        local.extend(loc)
    assert pool.n == 16 and [pool.items[i] for i in range(16)] == list(range(16))
    assert (9, 2) in [(a, b) for a, b, _ in local]
    assert sh.candidate_selector.candidate_edges == {}
    sh.close(); pool.close()


@pytest.mark.parametrize("world,maxb", [(2, 5), (3, 5), (4, 4)])
def test_device_shard_split_best_mode(gpu, world, maxb):
    """Inter-robot matching only (nb_best_matches = 0: frontend.enable_intra_robot_loop_closures
    off, the node default) with <= 16 gathered descriptors per step: ONE scan launch per step
    answers "best old row" and "best row of the block just appended" (split best-match scan, its
    candidates reduced by the record kernel).  256-d so the LDS-DMA scan runs; same candidate set as the sequential
    reference semantics, identical replicas and record streams."""
    steps, dim = 12, 256
    rs = np.random.RandomState(31 + world)
    places = rs.rand(25, dim).astype(np.float32)
    blocks, nxt = {}, [0] * world
    for t in range(steps):
        for r in range(world):
            b = int(rs.randint(0, maxb + 1))
            x = places[rs.randint(0, len(places), size=b)] + 0.05 * rs.rand(b, dim).astype(np.float32)
            blocks[(t, r)] = (x.astype(np.float32), list(range(nxt[r], nxt[r] + b)))
            nxt[r] += b
    results, local, records = _run_shards(world, steps, blocks, maxb, 0, 0, dim=dim, thr=0.9)
    union, _ = _sequential_reference_with(blocks, world, steps, thr=0.9)
    assert len(union) > 10
    for r in range(world):
        assert set(results[r]) == set(union)
        for key in union:
            assert abs(results[r][key] - union[key]) <= 1e-7
        assert results[r] == results[0]
        assert records[r].tobytes() == records[0].tobytes()
        assert local[r] == []


def test_loopback_communicator_step_matches_oracle(gpu):
    """cslam_comm_init_loopback (every all-gather copies the caller's block into all slots, one
    kernel) under the split best-match step: robot 1 of 4 sees its own descriptors as everybody's.
    Each fetched record stream is rebuilt on the host from the oracle: per foreign slot the best
    OLD row (lcsm.py:69-75) and, when it beats every older row, the best row of the block just
    appended (lcsm.py:47-55).  Pipelined publishes (two steps ahead) exercise the explicit row
    windows: rows of later blocks are already in the pool when a step is matched."""
    torch = pytest.importorskip("torch")
    from cslam_amd.multi_robot import DeviceShard, LoopbackComm
    from cslam_amd.nns_matching import NearestNeighborsMatching
    R, me, bmax, dim, thr, steps = 4, 1, 4, 256, 0.8, 9
    rs = np.random.RandomState(3)
    places = rs.rand(12, dim).astype(np.float32)
    pool = NearestNeighborsMatching(dim=dim)
    cd, cr = LoopbackComm(me, R, 0), LoopbackComm(me, R, 0)
    sh = DeviceShard(pool, me, R, bmax, thr, cd, cr, nb_best_matches=0)
    blocks, kf = [], 0
    for t in range(steps):
        b = int(rs.randint(1, bmax + 1))
        x = (places[rs.randint(0, len(places), size=b)] + 0.04 * rs.rand(b, dim)).astype(np.float32)
        blocks.append((x, list(range(kf, kf + b))))
        kf += b
    dev = [torch.from_numpy(x).cuda() for x, _ in blocks]
    got = []
    sh.publish(dev[0], blocks[0][1])
    for t in range(steps):
        if t + 1 < steps:
            sh.publish(dev[t + 1], blocks[t + 1][1])        # one step ahead of the match
        sh.match()
        got.append(sh.fetch_records())
    # host replay
    rows = np.zeros((0, dim), np.float32)
    kfs = []
    for t, (x, ids) in enumerate(blocks):
        n0 = rows.shape[0]
        rows = np.vstack([rows, x])
        kfs = kfs + ids
        exp = []
        for s in range(R):                                   # every rank's records are the same
            for slot_rank in range(R):                       # gathered descriptors, rank order
                if slot_rank == me:
                    continue
                for j, kf_s in enumerate(ids):
                    ia, sa = -1, 0.0
                    if n0:
                        i, v, _ = oracle_knn.search(rows[:n0], x[j:j + 1], 1)
                        ia, sa = int(i[0, 0]), np.float32(v[0, 0])
                    i, v, _ = oracle_knn.search(rows[n0:], x[j:j + 1], 1)
                    ic, sc = int(i[0, 0]) + n0, np.float32(v[0, 0])
                    if ia >= 0 and float(sa) >= thr:
                        exp.append((me, kfs[ia], slot_rank, kf_s, sa))
                    if float(sc) >= thr and not (ia >= 0 and sa >= sc):
                        exp.append((slot_rank, kf_s, me, kfs[ic], sc))
        rec = got[t]
        assert len(rec) == len(exp), (t, len(rec), len(exp))
        for r, e in zip(rec, exp):
            assert (int(r[0]), int(r[1]), int(r[2]), int(r[3])) == e[:4]
            assert np.float32(r[4]) == e[4]
    assert sum(len(g) for g in got) > 20
    assert pool.n == kf
    sh.close(); cd.close(); cr.close(); pool.close()


def test_fused_step_without_the_one_launch_rank_select(gpu):
    """Round 4 runs the fused step's rank select and records in ONE launch behind the scan (k_topk_records over
    pool_topk_candidates).  The path it replaced - the pool's own search + k_make_records<true>, still taken for
    shapes that scan does not cover - stays covered: the fused tests again with CSLAM_SHARD_FUSE_TOPK=0 (read once
    per process, hence a child process)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CSLAM_SHARD_FUSE_TOPK="0")
    out = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                          "tests/test_shard_gpu.py::test_device_shard_matches_reference_semantics"],
                         cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "2 passed" in out.stdout, out.stdout[-500:]
