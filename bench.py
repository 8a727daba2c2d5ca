#!/usr/bin/env python
"""bench.py — headline benchmark of the loop-closure hot path on MI355X.

Metric (BASELINE.json): loop-closure candidates/sec (kNN+MAC), 100k keyframes x 4096-d;
Fiedler-solve ms.

Workload (configs[1], fits one GPU): every rank is one robot holding a resident pool of
100 000 keyframe descriptors x 4096 fp32 (rows L2-normalised, seed = rank).

  N = 1   one step = the best match (k = 1, the `search_best` of
          cslam/loop_closure_sparse_matching.py:49,69) of 16 new descriptors, already in HBM,
          in the pool: one HBM-bound scan + exact top-1.  value = 16 / step time.
  N > 1   one step = the sharded protocol of SURVEY.md §8e through the C ABI
          (cslam_amd.multi_robot.DeviceShard → csrc/shard.hip, csrc/comm.hip): every rank
          publishes 16/N new descriptors, ONE RCCL all-gather of the blocks + keyframe ids, the own
          block is appended to the pool, ONE scan launch of the pool for all 16 gathered descriptors
          (foreign descriptors: best old row and best row of the new block, the inter-robot match of
          lcsm.py:58-76 / 37-56; the rank's own ones: match_local_loop_closures, lcsm.py:78-97, with
          nb_best_matches = 10 - frontend.enable_intra_robot_loop_closures on, so that every gathered
          descriptor is a candidate evaluation exactly as the 16 own ones are at N = 1;
          CSLAM_BENCH_INTRA=0 leaves the own ones out and counts the 16 (N-1)/N foreign ones only),
          20-byte match records built on the device, ONE RCCL all-gather of the records, records
          fetched by the host.  Per-GPU work fixed as N grows (weak scaling): 16 descriptors scanned
          and matched per rank and step: value = N * 16 / step time.  The exchange of step t+1 is issued
          before the scan of step t, records are fetched one step late.  Control plane (barriers,
          max over ranks, communicator id): a gloo process group; the only GPU collectives are the
          library's own RCCL communicator (ONE, shared by both exchanges).

  --workload cfg4   BASELINE configs[3]'s own shape: 25 000 keyframes x 4096-d per robot, every robot
          publishes 10 descriptors per step (frontend.detection_publication_max_elems_per_msg), all
          R x 10 gathered descriptors are matched against the pool (k_scan_mq: one pass, split
          window).  N = 1: robot 0 of 8 over the loopback communicator (80 gathered, 70 foreign);
          N > 1: R = N over RCCL.  The default N = 1 line above is unchanged by this option.

`python bench.py --gpus N` launches its own N ranks (children started before anything touches
the GPU); under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` it is one
of the ranks.

Extra keys: `roofline` (the scan kernel: algorithmic bytes / HIP-event time on its launch
stream), `cpu_baseline` (oracle/knn_oracle.c timed on this host; rank 0, N = 1 only), `extras`
(host-pointer rate incl. PCIe, append+search step, single-query latency, k = 10, 32/64/1024-query
batches, MAC Fiedler solve / Frank-Wolfe selection), `dist` (N > 1: backend, world size,
transport, all-gather payload).
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(1, os.path.join(ROOT, "tools"))

N_KEYFRAMES = 100_000
DIM = 4096
BATCH = 16
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md:36 (spec; 6290 measured copy)
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md:42 (fp32-input MFMA, dense)
METRIC = "loop-closure candidates/sec (kNN+MAC), 100k keyframes x 4096-d; Fiedler-solve ms"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--keyframes", type=int, default=N_KEYFRAMES)
    ap.add_argument("--dim", type=int, default=DIM)
    ap.add_argument("--workload", choices=["cfg2", "cfg4"], default="cfg2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------
# self-launch: `python bench.py --gpus N` with no launcher around it
# ------------------------------------------------------------------------------------------------
def launch_ranks(args):
    """Start N fresh rank processes of this script and relay rank 0's JSON line.  Runs before
    torch or the HIP runtime are imported: the parent never touches a GPU, nothing is exec'd."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                      env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
                                      text=True))
    # rank 0's stdout is drained by a thread; the parent watches all ranks so that one that
    # dies (a rank that cannot get its GPU, say) takes the others down instead of leaving them
    # waiting in the rendezvous for ever — only the exact processes started here are signalled
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = None
    while True:
        rcs = [p.poll() for p in procs]
        if any(rc not in (None, 0) for rc in rcs):
            failed = [(r, rc) for r, rc in enumerate(rcs) if rc not in (None, 0)]
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            break
        if all(rc == 0 for rc in rcs):
            break
        time.sleep(0.2)
    for p in procs:
        try:
            p.wait(timeout=30)
        except subprocess.TimeoutExpired:
            p.kill()
    reader.join(timeout=10)
    out0 = "".join(c for c in chunks if c)
    if out0:
        sys.stdout.write(out0)
        sys.stdout.flush()
    if failed:
        sys.stderr.write(f"bench.py: rank(s) failed: {failed}\n")
        return 1
    return 0


class Deadline(object):
    """`with Deadline(seconds, what):` - a watchdog thread ends THIS process with status 6 when the
    body has not finished in time (a collective that never completes, a hung device).  Fresh-process
    semantics only: nothing is re-executed, the parent (launch_ranks) sees the failed rank and stops the rest."""

    def __init__(self, seconds, what):
        self.seconds, self.what, self.timer = seconds, what, None

    def _expire(self):
        sys.stderr.write(f"bench.py: rank {os.environ.get('RANK', '0')}: '{self.what}' did not finish within "
                         f"{self.seconds:.0f} s - exiting\n")
        sys.stderr.flush()
        os._exit(6)

    def __enter__(self):
        import threading
        self.timer = threading.Timer(self.seconds, self._expire)
        self.timer.daemon = True
        self.timer.start()
        return self

    def __exit__(self, *exc):
        self.timer.cancel()
        return False


def robot_blocks(torch, dev, robot, per_rank, dim):
    """The 8 blocks of new descriptors robot `robot` cycles through (seeded by the robot alone)."""
    gb = torch.Generator(device=dev).manual_seed(10_000 + robot)
    return torch.rand((8, per_rank, dim), device=dev, generator=gb)


def selfcheck_exchange(torch, L, _lib, comm, rank, world, nb_robots, dev, out):
    """Before timing: the communicator counts `world` ranks, and a rank-stamped block gathered through
    the library's own collective (cslam_comm_allgather) arrives in the right slot on every rank."""
    loopback = world == 1
    out["communicator_size"] = "ok" if loopback or comm.size() == world else f"FAILED: {comm.size()} != {world}"
    n = 1024
    i = torch.arange(n, dtype=torch.int64, device=dev)
    send = (rank * 1_000_003 + i * 7919) % 2_147_483_647
    recv = torch.full((nb_robots * n,), -1, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(L.cslam_comm_allgather(comm.handle, ctypes.c_void_p(send.data_ptr()), ctypes.c_void_p(recv.data_ptr()),
                                      n * 8, ctypes.c_void_p(st)))
    torch.cuda.synchronize()
    got = recv.view(nb_robots, n).cpu()
    bad = []
    for r in range(nb_robots):
        src = 0 if loopback else r              # the loopback transport replicates the caller's block
        want = ((src * 1_000_003 + torch.arange(n, dtype=torch.int64) * 7919) % 2_147_483_647)
        if not torch.equal(got[r], want):
            bad.append(r)
    out["stamped_allgather"] = "ok" if not bad else f"FAILED: slots {bad} do not hold their rank's stamp"


def selfcheck_records(torch, L, _lib, pool, state, rank, world, nb_robots, per_rank, N, D, dev, stream, out):
    """After the timed loop: the records THIS rank contributed to the last fetched step are recomputed
    from scratch - the peers' blocks regenerated from their seeds, one row-range search per window
    (cslam_pool_search_range_dev) - and compared bit for bit with what came back through the record
    exchange: content, not count.  (Keyframe id = pool row in this bench.)"""
    import numpy as np
    if world == 1:
        out["last_step_records"] = "skipped (loopback: no peers)"   # every slot is a copy of the own block
        return
    recs = state.get("last_records")
    m = state["fetched"] - 1                      # the step those records belong to
    if recs is None or m < 0:
        out["last_step_records"] = "FAILED: no step was fetched"
        return
    if os.environ.get("CSLAM_BENCH_SELFCHECK_CORRUPT") == "1":   # tests: prove that the check bites
        recs = recs.copy()
        recs["weight"] = recs["weight"] + np.float32(1e-3)
    n0 = N + m * per_rank                         # pool rows before step m's own block
    problems = []
    seen = 0
    for s_ in range(nb_robots):
        if s_ == rank:
            continue
        q = robot_blocks(torch, dev, s_, per_rank, D)[m % 8].contiguous()
        res = {}
        for name, lo, hi in (("old", 0, n0), ("new", n0, n0 + per_rank)):
            idx = torch.empty((per_rank, 1), dtype=torch.int32, device=dev)
            sim = torch.empty((per_rank, 1), dtype=torch.float32, device=dev)
            _lib.check(L.cslam_pool_search_range_dev(pool.handle, ctypes.c_void_p(q.data_ptr()), D, per_rank, 1, lo, hi,
                                                     ctypes.c_void_p(idx.data_ptr()), ctypes.c_void_p(sim.data_ptr()),
                                                     ctypes.c_void_p(stream)))
            torch.cuda.synchronize()
            res[name] = (idx[:, 0].cpu().numpy(), sim[:, 0].cpu().numpy())
        for j in range(per_rank):
            kf_s = N + m * per_rank + j
            ia, sa = int(res["old"][0][j]), res["old"][1][j]
            ic, sc = int(res["new"][0][j]), res["new"][1][j]
            want = [(rank, ia, s_, kf_s, sa)]                             # lcsm.py:58-76, r0 slot
            if ic >= 0 and not (ia >= 0 and sa >= sc):
                want.append((s_, kf_s, rank, ic, sc))                     # lcsm.py:37-56 on the peer's behalf
            for w in want:
                hit = recs[(recs["robot0_id"] == w[0]) & (recs["robot0_keyframe_id"] == w[1]) &
                           (recs["robot1_id"] == w[2]) & (recs["robot1_keyframe_id"] == w[3])]
                seen += 1
                if len(hit) != 1 or np.float32(hit["weight"][0]).tobytes() != np.float32(w[4]).tobytes():
                    problems.append((w, [tuple(h) for h in hit][:2]))
    out["last_step_records"] = "ok" if not problems and seen >= (nb_robots - 1) * per_rank else \
        f"FAILED: {len(problems)} of {seen} recomputed records differ, e.g. {problems[:2]}"


def main():
    args = parse()
    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and world_env is None:
        sys.exit(launch_ranks(args))
    world = int(world_env or "1")
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} "
                         "(launch N ranks for --gpus N, or none at all)")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import torch
    import torch.distributed as dist

    from cslam_amd import _lib
    from cslam_amd.nns_matching import NearestNeighborsMatching

    # rehearsal knobs (one-GPU boxes): CSLAM_BENCH_BACKEND=gloo exchanges through host memory,
    # CSLAM_BENCH_SAME_DEVICE=1 puts every rank on cuda:0.  The driver's runs use neither.
    backend = os.environ.get("CSLAM_BENCH_BACKEND", "nccl")
    if os.environ.get("CSLAM_BENCH_SAME_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        # Control plane only (barrier, max over ranks, the communicator id): gloo over TCP.  The
        # GPU collectives are the library's own RCCL communicator - no torch NCCL group shares the
        # device with the scan, one sequence of collectives per rank.
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)

    L = _lib.load()
    cfg4 = args.workload == "cfg4"
    N, D = args.keyframes, args.dim
    if cfg4 and args.keyframes == N_KEYFRAMES:
        N = 25_000
    nb_robots = world if world > 1 else (8 if cfg4 else 1)      # cfg4 at N = 1: robot 0 of 8, loopback
    per_rank = 10 if cfg4 else max(1, BATCH // world)
    gathered_rows = per_rank * nb_robots
    total_steps = args.warmup + args.steps

    # ---- resident pool (robot = rank) -------------------------------------------------------
    g = torch.Generator(device=dev).manual_seed(rank)
    X = torch.rand((N, D), device=dev, generator=g)
    X /= X.norm(dim=1, keepdim=True)
    grow = (total_steps + 90) * per_rank if nb_robots > 1 else 0
    pool = NearestNeighborsMatching(dim=D, device_id=local_rank, capacity_hint=N + grow)
    stream = torch.cuda.current_stream().cuda_stream
    pool.add_items_device(X.data_ptr(), N, D, items=None, stream=stream)
    torch.cuda.synchronize()

    new_local = torch.rand((per_rank, D), device=dev, generator=g)
    out_idx = torch.empty((gathered_rows, 1), dtype=torch.int32, device=dev)
    out_sim = torch.empty((gathered_rows, 1), dtype=torch.float32, device=dev)

    def search(q, k, oi, os_):
        _lib.check(L.cslam_pool_search_dev(pool.handle, ctypes.c_void_p(q.data_ptr()), D,
                                           q.shape[0], k, ctypes.c_void_p(oi.data_ptr()),
                                           ctypes.c_void_p(os_.data_ptr()),
                                           ctypes.c_void_p(stream)))

    dist_info = None
    state = {"published": 0, "matched": 0, "fetched": 0, "records": 0}
    # cfg2 protocol at N > 1: the rank's OWN new descriptors are matched too (intra-robot matching,
    # match_local_loop_closures, nb_best_matches = 10) - every one of the 16 gathered descriptors is a
    # candidate evaluation, as at N = 1, so the per-GPU work AND the per-GPU count are fixed as N grows.
    # cfg4: inter-robot matching only, the node's default.  CSLAM_BENCH_INTRA=0/1 overrides either.
    intra_env = os.environ.get("CSLAM_BENCH_INTRA")
    intra = (not cfg4) if intra_env is None else intra_env == "1"
    if nb_robots == 1:
        def step(t):
            search(new_local, 1, out_idx, out_sim)

        def drain():
            pass
    else:
        from cslam_amd.multi_robot import (CallbackComm, DeviceShard, LoopbackComm, RcclComm,
                                           torch_dist_transport)
        if world == 1:
            comm = LoopbackComm(0, nb_robots, local_rank)       # cfg4 on one GPU: compute side of the step
        elif backend == "nccl":
            # ONE RCCL communicator for both exchanges; its id travels over the gloo group
            ids = [RcclComm.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(ids, src=0)
            comm = RcclComm(rank, world, ids[0], local_rank)
        else:
            comm = CallbackComm(rank, world, local_rank, torch_dist_transport(),
                                name=f"{backend} through host memory (one-GPU rehearsal)")
        # threshold 0: every evaluation on non-negative data yields a candidate;
        # inter-robot matching only, as SURVEY §8e defines the step and as the node runs by default
        # (frontend.enable_intra_robot_loop_closures = False, loop_closure_detection_node.py:36).
        # CSLAM_BENCH_INTRA=1 adds match_local_loop_closures (nb_best_matches = 10, lcsm.py:78-97).
        shard = DeviceShard(pool, rank, nb_robots, per_rank, 0.0, comm, comm,
                            nb_best_matches=10 if intra else 0, intra_min_inbetween=10)
        # new keyframes, cycled; from a generator of their own so that ANY rank can regenerate what rank r
        # publishes (the self-check below recomputes the last step's records from it)
        blocks = robot_blocks(torch, dev, rank, per_rank, D)
        next_kf = [N]
        torch.cuda.synchronize()
        # new descriptors come from the front end's stream (the VPR network), not from the
        # matching stream: publish orders the exchange behind THAT stream, the matching stream
        # carries scans and record kernels only
        front = torch.cuda.Stream(device=dev)

        def publish():
            kfs = list(range(next_kf[0], next_kf[0] + per_rank))
            next_kf[0] += per_rank
            shard.publish(blocks[state["published"] % 8], kfs, front.cuda_stream)
            state["published"] += 1

        # exchanges in flight ahead of the scan (1 = the exchange of step t+1 overlaps the scan of step t)
        # 2 (default): the exchange of step t+2 is queued while step t is scanned, so that by the time the
        # host queues match(t+1) the event of publish(t+1) has completed and no cross-stream barrier packet
        # goes in front of the scan (~5 us per step on the cfg4 workload, profiles/r04_bench_cfg4_ab.txt)
        ahead = max(1, min(2, int(os.environ.get("CSLAM_BENCH_PUBLISH_AHEAD", "2"))))

        def fetch():
            recs = shard.fetch_records()
            state["records"] += len(recs)
            state["last_records"] = recs           # of step state["fetched"]
            state["fetched"] += 1

        def step(t):
            # exchange of step t+1 first (own stream), then step t's scan + records;
            # the records of step t-1 are fetched while step t runs
            while state["published"] - state["matched"] < ahead:
                publish()
            publish()
            shard.match(stream)
            state["matched"] += 1
            if state["matched"] - state["fetched"] > 1:
                fetch()

        def drain():
            while state["fetched"] < state["matched"]:
                fetch()

        # first N > 1 run on hardware validates itself: communicator size, a rank-stamped all-gather
        # compared on every rank - BEFORE anything is timed, under a deadline (a collective that does
        # not complete makes this rank exit non-zero; the parent then stops the others)
        selfcheck = {"communicator_size": None, "stamped_allgather": None, "last_step_records": None}
        with Deadline(float(os.environ.get("CSLAM_BENCH_SELFCHECK_DEADLINE_S", "180")), "self-check all-gather"):
            selfcheck_exchange(torch, L, _lib, comm, rank, world, nb_robots, dev, selfcheck)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Device settle: the first ~40 scan launches of a process run 3-10 % slow (power management
    # transient: 264 -> 286 -> 274 -> 258 us per launch over chunks of 10,
    # profiles/r02_launch_ramp.txt).  Untimed steps of the same work in front of the W warm-up
    # steps bring the chip to the state a long-running node is in; reported as `settle_steps`.
    settle = max(0, 60 - args.warmup)
    for t in range(settle + args.warmup):
        step(t)
    drain()
    fence()
    # HIP events around every 4th scan launch of the timed region (a pair costs ~8 us of stream
    # time, tools/event_cost.py: bracketing every launch would tax the very step being timed)
    prof_every = 4 if args.steps >= 8 else 1
    _lib.check(L.cslam_pool_profile_begin_sampled(pool.handle, args.steps, prof_every))
    bytes0 = 0
    if nb_robots > 1:
        state["records"] = 0
        bytes0 = comm.stats()["bytes_gathered"]
        coll0 = comm.stats()["collectives"]
    fence()
    t0 = time.perf_counter()
    for t in range(args.steps):
        step(t)
    drain()                       # every timed step's records are on the host
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    fence()
    scan_ms = ctypes.c_double()
    n_l = ctypes.c_int()
    _lib.check(L.cslam_pool_profile_end(pool.handle, ctypes.byref(scan_ms), ctypes.byref(n_l)))
    scan_kernel = L.cslam_pool_scan_kernel(pool.handle).decode()   # of the timed steps (the self-check below searches too)

    if nb_robots > 1:
        with Deadline(float(os.environ.get("CSLAM_BENCH_SELFCHECK_DEADLINE_S", "180")), "self-check of the last step's records"):
            selfcheck_records(torch, L, _lib, pool, state, rank, world, nb_robots, per_rank, N, D, dev, stream,
                              selfcheck)
        if world > 1:       # every rank must agree before a number is printed
            oks = [None] * world
            dist.all_gather_object(oks, selfcheck)
            bad = [(r, o) for r, o in enumerate(oks) if any(v not in ("ok", "skipped (loopback: no peers)") for v in o.values())]
            if bad:
                sys.stderr.write(f"bench.py: self-check FAILED on rank(s) {bad}\n")
                sys.exit(5)
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)          # gloo, host tensor
        dt = float(tt.item())
    if nb_robots > 1:
        sd = comm.stats()
        dist_info = {"control_plane_backend": dist.get_backend() if world > 1 else None,
                     "world_size": world, "robots": nb_robots,
                     "transport": comm.transport if world > 1 else "loopback (one GPU: compute side of the step)",
                     "rccl_ranks": comm.size() if world > 1 and backend == "nccl" else None,
                     "rccl_version": sd["rccl_version"], "communicators": 1,
                     "allgather_payload_bytes_per_step": (sd["bytes_gathered"] - bytes0) / args.steps,
                     "collectives_per_step": (sd["collectives"] - coll0) / args.steps,
                     "match_records_per_step": state["records"] / args.steps,
                     "new_descriptors_per_rank_per_step": per_rank,
                     "gathered_descriptors_per_step": gathered_rows,
                     "selfcheck": "ok" if all(v in ("ok", "skipped (loopback: no peers)") for v in selfcheck.values()) else "FAILED",
                     "selfcheck_detail": selfcheck}

    ms_per_step = dt / args.steps * 1e3
    # candidates evaluated per step: N = 1 (cfg2) all 16 best-match evaluations; sharded step: the
    # FOREIGN descriptors of every rank (its own are skipped by the record kernel unless the
    # intra-robot match is on)
    evals_per_rank = gathered_rows if (nb_robots == 1 or intra) else gathered_rows - per_rank
    value = world * evals_per_rank / (dt / args.steps)

    # scan kernel roofline: algorithmic bytes / flops of one launch:
    #   pool rows N*D*4 + queries M*D*4 + results M*k*8;  2*N*D*M flop
    n_rows_scanned = N if nb_robots == 1 else N + (settle + args.warmup + args.steps // 2) * per_rank
    k_sel = 1 if nb_robots == 1 else (per_rank + 10 if intra else 2)
    alg_bytes = n_rows_scanned * D * 4 + gathered_rows * D * 4 + gathered_rows * k_sel * 8
    alg_flops = 2.0 * n_rows_scanned * D * gathered_rows
    achieved = alg_bytes / (scan_ms.value * 1e-3) / 1e9 if scan_ms.value > 0 else 0.0
    # HBM traffic from the PMC counters is collected in separate rocprofv3 --pmc passes and
    # summarised in profiles/r0X_scan_pmc.json; it is only quoted when that summary is about
    # the kernel variant and the shape that ran here - `traffic_source` says where it came from
    traffic, traffic_source = None, None
    mfma_busy = None
    for name in ("r05_scan_pmc.json", "r05_mq_pmc.json", "r04_scan_pmc.json", "r04_mq_pmc.json", "r03_scan_pmc.json", "r02_scan_pmc.json"):
        pmc = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(pmc):
            continue
        try:
            rec = json.load(open(pmc))
            if rec.get("kernel") == scan_kernel and rec.get("keyframes") == N and rec.get("dim") == D:
                traffic = rec.get("hbm_bytes_per_launch")
                mfma_busy = rec.get("MfmaUtil_median")
                traffic_source = ("profiles/" + name + " (separate rocprofv3 --pmc passes of this command, "
                                  "not measured in this run)")
                break
        except Exception:
            pass

    workload = ("cfg2: 100k keyframes x 4096-d fp32 pool per GPU, 16 new descriptors per step, "
                "best match (k=1) in the pool")
    if nb_robots > 1:
        where = ("RCCL all-gather" if world > 1 else "loopback all-gather (one GPU, robot 0 of %d)" % nb_robots)
        workload = (("cfg4: %d keyframes x %d-d per robot, " % (N, D) if cfg4 else "cfg2 pool per GPU under the cfg4 protocol: ") +
                    f"every robot publishes {per_rank} new descriptors per step, ONE {where} of blocks + ids, append, "
                    f"one scan of the pool for all {gathered_rows} gathered descriptors (" +
                    ("foreign ones: inter-robot best match; own ones: intra-robot match with nb_best_matches = 10"
                     if intra else "inter-robot match: best old row + best new row") +
                    "), match records built on the device, one all-gather of the records, fetched by the host")
    # more than two query tiles: the scan is bound by the fp32 matrix pipe, not by HBM
    mfma_bound = gathered_rows > 32
    roofline = {"bound": "hbm", "kernel": scan_kernel, "achieved": achieved,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "frac_of_measured_copy_peak": achieved / 6290.0,
                "traffic": traffic, "traffic_source": traffic_source, "avg_kernel_ms": scan_ms.value,
                "launches_timed": n_l.value, "launches_sampled_every": prof_every,
                "algorithmic_bytes_per_launch": alg_bytes}
    if mfma_bound:
        tf = alg_flops / (scan_ms.value * 1e-3) / 1e12 if scan_ms.value > 0 else 0.0
        roofline.update({"bound": "mfma", "achieved": tf, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": tf / MFMA_F32_PEAK_TFLOPS, "algorithmic_flops_per_launch": alg_flops,
                         "hbm_gbs": achieved})
        roofline.pop("frac_of_measured_copy_peak", None)
        if mfma_busy is not None:      # matrix-pipe busy fraction of the scan launch, same PMC summary
            roofline["mfma_busy"] = mfma_busy
    result = {
        "metric": METRIC,
        "value": value,
        "unit": "candidates/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "settle_steps": settle,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": workload, "keyframes_per_gpu": N, "dim": D,
                   "queries_per_step_per_gpu": gathered_rows,
                   "evaluations_per_step_per_gpu": evals_per_rank, "k": 1,
                   "parallelism": f"robot-per-gpu x{world}"},
        "roofline": roofline,
    }
    if dist_info:
        result["dist"] = dist_info

    if rank == 0 and world == 1 and not cfg4 and not args.no_extras:
        result["extras"] = extras(torch, L, _lib, pool, dev, g, stream, N, D, X)
        # the second half of the headline metric ("Fiedler-solve ms"), BASELINE.json configs[2]
        if "fiedler_cold_device_ms" in result["extras"]:
            result["fiedler_solve_ms"] = result["extras"]["fiedler_cold_device_ms"]
            result["select_candidates_20fw_ms"] = result["extras"]["fw_subset_20iters_ms"]
            if "acm_select_candidates_ms" in result["extras"]:   # graph assembly + MAC, end to end
                result["acm_select_candidates_ms"] = result["extras"]["acm_select_candidates_ms"]
            result["roofline_fiedler"] = fiedler_roofline(result["extras"])
        for key in ("value_incl_h2d", "value_append_then_search"):
            if key in result["extras"]:
                result[key] = result["extras"][key]
    if rank == 0 and world == 1 and not cfg4 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(torch, X, new_local, out_idx, out_sim)

    if rank == 0:
        print(json.dumps(result), flush=True)
    if nb_robots > 1:
        shard.close()
        comm.close()
    if world > 1:
        dist.destroy_process_group()


def fiedler_roofline(ex):
    """Second half of the headline metric against ITS bound.  The cfg3 Laplacian
    with 500 active candidates is n = 50 000, nnz = n + 2 #E = 150 998: one SpMV reads
    nnz*12 + (n+1)*4 + 2n*8 B = 2.81 MB - L2-resident (32 MiB of L2, 4 MiB per XCD), so the solve is
    not HBM-bound but bound by the latency of the exchanges between the <= 64 co-resident
    workgroups of the persistent kernel: per iteration `poll_points` rounds in which every workgroup
    reads what the others published, each at the floor of one memory-side round trip
    (tools/exchange_probe.hip: 1.6-1.8 us per all-to-all round at 25-49 workgroups)."""
    n, nnz = 50_000, 50_000 + 2 * (49_998 + 1 + 500)
    spmv_bytes = nnz * 12 + (n + 1) * 4 + 2 * n * 8
    iters = ex.get("fiedler_cold_iters", 0)
    ms = ex.get("fiedler_cold_device_ms", 0.0)
    poll_points, round_trip_us = ex.get("fiedler_poll_points_per_iter", 2), 1.7
    achieved = spmv_bytes * iters / (ms * 1e-3) / 1e9 if ms else 0.0
    return {"bound": "latency (L2-resident working set; exchange rounds of the persistent kernel)",
            "algorithmic_bytes_per_iteration": spmv_bytes, "iterations": iters, "device_ms": ms,
            "achieved": achieved, "unit": "GB/s", "peak": HBM_PEAK_GBS, "frac": achieved / HBM_PEAK_GBS,
            "residency": "2.8 MB per SpMV + ~2 MB of solver vectors: resident in L2/registers for the whole solve, "
                         "HBM sees the operands once",
            "us_per_iteration": ms * 1e3 / iters if iters else None,
            "poll_floor_model": {"poll_points_per_iteration": poll_points, "round_trip_us": round_trip_us,
                                 "floor_ms": iters * poll_points * round_trip_us * 1e-3,
                                 "frac_of_floor": (iters * poll_points * round_trip_us * 1e-3) / ms if ms else None}}


def extras(torch, L, _lib, pool, dev, g, stream, N, D, X):
    """Secondary numbers of the same path (not the headline value)."""
    import numpy as np
    out = {}

    def timed(q, k, reps):
        oi = torch.empty((q.shape[0], k), dtype=torch.int32, device=dev)
        os_ = torch.empty((q.shape[0], k), dtype=torch.float32, device=dev)

        def run():
            _lib.check(L.cslam_pool_search_dev(pool.handle, ctypes.c_void_p(q.data_ptr()), D,
                                               q.shape[0], k, ctypes.c_void_p(oi.data_ptr()),
                                               ctypes.c_void_p(os_.data_ptr()),
                                               ctypes.c_void_p(stream)))
        # a kernel variant's first launches after another variant ran are slower than its steady
        # state (32 queries: 0.3175 ms after 3 warm-up calls, 0.300 after 10+): warm up with a third
        # of the repetitions, at least 3
        for _ in range(max(3, reps // 3)):
            run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            run()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    # MAC first: the solver is latency-bound, and the clocks sag for a while after the
    # 100-TFLOP/s bulk pass below
    try:
        from mac_bench_util import acm_select_bench, cfg1_bench, fiedler_bench
        out.update(fiedler_bench())
        out.update(acm_select_bench())
        out.update(cfg1_bench())     # BASELINE configs[0]: the reference's own CPU-runnable case
    except ImportError:
        pass
    q1 = torch.rand((1, D), device=dev, generator=g)
    q16 = torch.rand((16, D), device=dev, generator=g)
    q1024 = torch.rand((1024, D), device=dev, generator=g)

    # the same 16-descriptor step through the HOST-pointer entry point (cslam_pool_search):
    # H2D of the queries + D2H of the results included — what a caller without device buffers
    # pays (SURVEY §8d's "H2D included" rate; never the headline value)
    qh = np.ascontiguousarray(q16.cpu().numpy())
    idx_h = np.empty((16, 1), np.int32)
    sim_h = np.empty((16, 1), np.float32)
    cnt_h = np.empty(16, np.int32)

    def host_step():
        _lib.check(L.cslam_pool_search(pool.handle, _lib.ptr(qh), 16, 1, _lib.ptr(idx_h),
                                       _lib.ptr(sim_h), _lib.ptr(cnt_h)))
    for _ in range(3):
        host_step()
    t0 = time.perf_counter()
    for _ in range(50):
        host_step()
    th = (time.perf_counter() - t0) / 50
    out["host_pointer_16queries_ms"] = th * 1e3
    out["value_incl_h2d"] = 16 / th

    t = timed(q1, 1, 50)
    out["search_best_1query_ms"] = t * 1e3
    t = timed(q16, 10, 50)
    out["search_16queries_k10_ms"] = t * 1e3
    for mq in (32, 64, 80):  # mid-size batches: k_scan_mq with 2 / 4 / 5 query tiles (80 = cfg4's gathered block)
        tq = timed(torch.rand((mq, D), device=dev, generator=g), 1, 30)
        out[f"search_{mq}queries_ms"] = tq * 1e3
        out[f"search_{mq}queries_candidates_per_s"] = mq / tq
    # search with k > 1 at 80 queries (the sharded step's fused intra + inter search, K = block + nb_best_matches):
    # k_scan_mq<best> with dots + k_topk_from_parts since round 4
    out["search_80queries_k26_ms"] = timed(torch.rand((80, D), device=dev, generator=g), 26, 30) * 1e3
    t = timed(q1024, 1, 3)
    out["bulk_1024queries_ms"] = t * 1e3
    out["bulk_1024queries_candidates_per_s"] = 1024 / t
    out["bulk_1024queries_tflops"] = 2.0 * N * D * 1024 / t / 1e12

    # the reference's real per-keyframe sequence (lcsm.py:45-49): add_item, then search_best —
    # here for a block of 16: append the 16 rows to a pool of the same size, then match them.
    # Own pool with head-room so the 100k-row pool above keeps its size.
    try:
        from cslam_amd.nns_matching import NearestNeighborsMatching
        # 40 untimed steps first: the first launches over a freshly allocated pool run slow
        # (same transient as `settle_steps`); tools/append_cost.py: append + search = search + 7 us
        reps, warm = 100, 40
        p2 = NearestNeighborsMatching(dim=D, device_id=dev.index, capacity_hint=N + 16 * (reps + warm + 4))
        p2.add_items_device(X.data_ptr(), N, D, items=None, stream=stream)
        blocks = torch.rand((4, 16, D), device=dev, generator=g)
        oi = torch.empty((16, 1), dtype=torch.int32, device=dev)
        os_ = torch.empty((16, 1), dtype=torch.float32, device=dev)

        def append_search(i):
            b = blocks[i % 4]
            _lib.check(L.cslam_pool_add_dev(p2.handle, ctypes.c_void_p(b.data_ptr()), 16, D,
                                            ctypes.c_void_p(stream)))
            _lib.check(L.cslam_pool_search_dev(p2.handle, ctypes.c_void_p(b.data_ptr()), D, 16, 1,
                                               ctypes.c_void_p(oi.data_ptr()),
                                               ctypes.c_void_p(os_.data_ptr()), ctypes.c_void_p(stream)))
        for i in range(warm):
            append_search(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(reps):
            append_search(i)
        torch.cuda.synchronize()
        ta = (time.perf_counter() - t0) / reps
        out["append16_then_search_ms"] = ta * 1e3
        out["value_append_then_search"] = 16 / ta
        p2.close()
    except Exception as e:  # an extra, never fatal for the headline
        out["append_then_search_error"] = repr(e)
    return out


def cpu_baseline(torch, X, queries, gpu_idx, gpu_sim):
    """The oracle (CPU port of the reference's search) on this host, on a bounded sample of the
    SAME workload: all 16 queries of one step against the first `rows` pool rows, scaled to
    the full pool (the scan is linear in rows)."""
    import numpy as np
    from oracle import knn as oracle_knn
    rows = min(X.shape[0], 50_000)
    xs = X[:rows].cpu().numpy()
    q = queries.cpu().numpy()
    oracle_knn.similarities(xs[:1000], q[:1])  # warm the threads
    t0 = time.perf_counter()
    idx, sim, _ = oracle_knn.search(xs, q, 1)
    dt = time.perf_counter() - t0
    full = dt * X.shape[0] / rows
    out = {"value": q.shape[0] / full, "unit": "candidates/s",
           "cores": int(oracle_knn.num_threads()), "kind": "port",
           "sample": f"{q.shape[0]} queries x {rows} of {X.shape[0]} rows x {X.shape[1]}-d, "
                     f"k=1, OpenMP over rows; {dt:.2f} s measured, scaled x{X.shape[0] / rows:.1f} "
                     "to the full pool",
           "note": "strict fmaf chain of the numeric contract (cannot vectorise); the reference's "
                   "own torch-CPU search of the same pool measured 1.1 queries/s on 8 cores "
                   "(BASELINE.md, build container)"}
    # ... and the reference's own per-query arithmetic with BLAS-backed numpy in place of its torch-CPU calls
    # (oracle.knn.search_numpy: cosine_similarity over the whole pool, norms recomputed per call, full argsort -
    # cslam/nns_matching.py:57-63), 4 queries one by one against the same sample, all host cores BLAS gives it
    try:
        nq = min(4, q.shape[0])
        oracle_knn.search_numpy(xs[:1000], q[0], 1)
        t0 = time.perf_counter()
        same = 0
        for j in range(nq):
            ii, ss = oracle_knn.search_numpy(xs, q[j], 1)
            same += int(ii[0] == idx[j, 0])
        dt2 = time.perf_counter() - t0
        out["numpy_path"] = {"value": nq / (dt2 * X.shape[0] / rows), "unit": "candidates/s",
                             "sample": f"{nq} queries, one call each, x {rows} of {X.shape[0]} rows; {dt2:.2f} s measured, scaled "
                                       f"x{X.shape[0] / rows:.1f}; best row equal to the port's in {same} of {nq}",
                             "what": "the reference's per-query search with numpy + BLAS for its torch-CPU calls (sgemv, row norms "
                                     "recomputed per call, full argsort): the CPU path a user of the reference runs"}
    except Exception as e:
        out["numpy_path"] = {"error": repr(e)}
    # second half of the metric: one Fiedler solve of the cfg3 graph on the CPU path the reference
    # uses (networkx TraceMIN + SuperLU, restated in oracle/mac_oracle.py; single-threaded scipy)
    try:
        from mac_bench_util import cfg3_graph, greedy_w
        from oracle import mac_oracle
        n, fixed, cand = cfg3_graph()
        orc = mac_oracle.MacOracle([e.i for e in fixed], [e.j for e in fixed], [e.weight for e in fixed],
                                   [e.i for e in cand], [e.j for e in cand], [e.weight for e in cand], n)
        w0 = greedy_w(cand, 500)
        t0 = time.perf_counter()
        lam, _ = orc.evaluate_fiedler_pair(w0)
        out["fiedler_solve_ms"] = (time.perf_counter() - t0) * 1e3
        out["fiedler_lambda2"] = float(lam)
        out["fiedler_sample"] = "one cold TraceMIN-LU solve of the cfg3 graph (n=50000, 500 active candidates), 1 core"
    except Exception as e:  # the kNN baseline above stands on its own
        out["fiedler_error"] = repr(e)
    return out


if __name__ == "__main__":
    main()
