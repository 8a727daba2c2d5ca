"""Candidate selection by algebraic-connectivity maximisation — host mirror of
cslam/algebraic_connectivity_maximization.py:8-571.

Same class/method names, arguments and results as the reference (this is the object
`GlobalDescriptorLoopClosureDetection` and the reference's tests drive directly); the MAC
numerics run on the GPU (cslam_amd/mac/mac.py → csrc/mac.hip).  Bookkeeping is host Python
like the reference's, written around normalised edge keys so `remove_candidate_edges` is
O(k) instead of the reference's O(m*k) `__eq__` scan (acm.py:177-189) — results identical.

Reference quirks reproduced on purpose (pinned by tests/golden/acm_ref.json):
  * `add_match` looks the candidate up by its RAW key but stores it under the normalised key
    (acm.py:564-571 vs :84-89): with robot0_id > robot1_id the keep-max rule is bypassed.
  * `fill_odometry` and `total_nb_poses` include robots that are known but not included
    (acm.py:347-361, :500-502) → disconnected L → solver raises → retry ladder → greedy w_init.
  * `candidate_edges_to_fixed` overwrites the weight with `fixed_weight` (acm.py:198-202).
  * the constructor of LoopClosureSparseMatching never passes params, so the defaults below
    are what runs (loop_closure_sparse_matching.py:34-35).
"""
import itertools
from typing import NamedTuple

import numpy as np

from .mac.mac import MAC
from .mac.utils import Edge


class EdgeInterRobot(NamedTuple):
    """Inter-robot loop closure edge (acm.py:8-30).  Equality ignores weight and
    orientation; hashing is the plain tuple hash, as in the reference."""
    robot0_id: int
    robot0_keyframe_id: int
    robot1_id: int
    robot1_keyframe_id: int
    weight: float

    def _ends(self):
        a = (self.robot0_id, self.robot0_keyframe_id)
        b = (self.robot1_id, self.robot1_keyframe_id)
        return (a, b)

    def __eq__(self, other):
        a, b = self._ends()
        c, d = ((other.robot0_id, other.robot0_keyframe_id),
                (other.robot1_id, other.robot1_keyframe_id))
        return (a == c and b == d) or (a == d and b == c)

    __hash__ = tuple.__hash__


class _CandidateStore(dict):
    """`candidate_edges` — a dict keyed by the normalised edge key like the reference's
    (acm.py:45), which ALSO keeps (r0, kf0, r1, kf1) as int64 columns and the weight as a float64
    column in dict order: a slot per key (overwriting keeps the slot, like a dict keeps the
    position of an existing key; a re-inserted key goes to the end in both).  `select_candidates`
    at cfg3 scale (200k candidates) reads `live_columns()` instead of walking 200k Python objects."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        self._ids = np.empty((4, 1024), dtype=np.int64)
        self._w = np.empty(1024, dtype=np.float64)
        self._alive = np.zeros(1024, dtype=bool)
        self._slot = {}
        self._n = 0
        if args or kwargs:
            self.update(*args, **kwargs)

    def _grow(self):
        cap = 2 * len(self._alive)
        ids = np.empty((4, cap), dtype=np.int64)
        ids[:, :self._n] = self._ids[:, :self._n]
        w = np.empty(cap, dtype=np.float64)
        w[:self._n] = self._w[:self._n]
        alive = np.zeros(cap, dtype=bool)
        alive[:self._n] = self._alive[:self._n]
        self._ids, self._w, self._alive = ids, w, alive

    def _compact(self):
        keep = np.flatnonzero(self._alive[:self._n])
        self._ids[:, :len(keep)] = self._ids[:, keep]
        self._w[:len(keep)] = self._w[keep]
        self._alive[:len(keep)] = True
        self._alive[len(keep):self._n] = False
        self._n = len(keep)
        self._slot = {k: i for i, k in enumerate(self.keys())}   # dict order == slot order

    def __setitem__(self, key, edge):
        s = self._slot.get(key)
        if s is None:
            if self._n == len(self._alive):
                if 2 * len(self) < self._n:
                    self._compact()
                else:
                    self._grow()
            s = self._n
            self._n += 1
            self._slot[key] = s
            self._alive[s] = True
        self._ids[:, s] = edge[:4]
        self._w[s] = edge[4]
        super().__setitem__(key, edge)

    def __delitem__(self, key):
        super().__delitem__(key)
        self._alive[self._slot.pop(key)] = False

    _missing = object()

    def pop(self, key, default=_missing):
        if key in self:
            value = self[key]
            del self[key]
            return value
        if default is self._missing:
            raise KeyError(key)
        return default

    def popitem(self):
        key = next(reversed(self))
        return key, self.pop(key)

    def clear(self):
        super().clear()
        self._slot.clear()
        self._alive[:] = False
        self._n = 0

    def update(self, *args, **kwargs):
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    def setdefault(self, key, default=None):
        if key not in self:
            self[key] = default
        return self[key]

    def live_columns(self):
        """(r0, kf0, r1, kf1 as int64, weight as float64), each of len(self), in dict order
        (views of the store while nothing has been removed, compacted copies otherwise)."""
        n = self._n
        if len(self) == n:
            return (self._ids[0, :n], self._ids[1, :n], self._ids[2, :n], self._ids[3, :n],
                    self._w[:n])
        alive = self._alive[:n]
        return tuple(self._ids[c, :n][alive] for c in range(4)) + (self._w[:n][alive],)

    def live_rows(self):
        """float64 [len(self), 5] in dict order."""
        return np.stack([c.astype(np.float64) for c in self.live_columns()], axis=1) \
            if len(self) else np.zeros((0, 5))


class AlgebraicConnectivityMaximization(object):

    def __init__(self,
                 robot_id=0,
                 max_nb_robots=1,
                 max_iters=20,
                 fixed_weight=1.0,
                 extra_params={
                     "frontend.enable_sparsification": True,
                     "evaluation.enable_sparsification_comparison": False,
                 },
                 device_id=0):
        """
        Args:
            robot_id (int): ID of the local robot
            max_nb_robots (int): number of robots
            max_iters (int): Frank-Wolfe iterations (acm.py:38)
            fixed_weight (float): weight given to fixed measurements and odometry
            device_id (int): HIP device the MAC solver runs on
        """
        self.fixed_weight = fixed_weight
        self.params = extra_params
        self.max_iters = max_iters
        self.max_nb_robots = max_nb_robots
        self.robot_id = robot_id
        self.device_id = device_id

        self.fixed_edges = []
        self._candidate_edges = _CandidateStore()
        self.already_considered_matches = set()
        self.total_nb_poses = 0
        self.nb_poses = {r: 0 for r in range(max_nb_robots)}
        self.initial_fixed_edge_exists = {r: False for r in range(max_nb_robots)}
        self.offsets = {r: 0 for r in range(max_nb_robots)}
        self.log_greedy_edges = []
        self.log_mac_edges = []
        self.last_solver_exceptions = 0  # how many rungs of the retry ladder the last call used

    @property
    def candidate_edges(self):
        return self._candidate_edges

    @candidate_edges.setter
    def candidate_edges(self, mapping):
        """Callers may assign a plain dict (the reference's attribute is one)."""
        self._candidate_edges = mapping if isinstance(mapping, _CandidateStore) \
            else _CandidateStore(mapping)

    # ---- small helpers (acm.py:75-129) -----------------------------------------------------
    def edge_key(self, edge):
        """Orientation-free key: the lower robot id comes first (acm.py:75-89)."""
        a = (edge.robot0_id, edge.robot0_keyframe_id)
        b = (edge.robot1_id, edge.robot1_keyframe_id)
        if edge.robot0_id < edge.robot1_id:
            return a + b
        return b + a

    def replace_weight(self, edge, weight):
        """Copy of `edge` with another weight (acm.py:91-107)."""
        if type(edge) is EdgeInterRobot:
            return edge._replace(weight=weight)
        if type(edge) is Edge:
            return Edge(edge.i, edge.j, weight)
        return None

    def update_nb_poses(self, edge):
        """nb_poses[r] = 1 + largest keyframe id seen for r (acm.py:109-118)."""
        for r, kf in ((edge.robot0_id, edge.robot0_keyframe_id),
                      (edge.robot1_id, edge.robot1_keyframe_id)):
            if kf + 1 > self.nb_poses[r]:
                self.nb_poses[r] = kf + 1

    def update_initial_fixed_edge_exists(self, fixed_edge):
        """A fixed INTER-robot edge marks both robots as linked (acm.py:120-129)."""
        if fixed_edge.robot0_id != fixed_edge.robot1_id:
            self.initial_fixed_edge_exists[fixed_edge.robot0_id] = True
            self.initial_fixed_edge_exists[fixed_edge.robot1_id] = True

    # ---- graph edition (acm.py:131-202) ----------------------------------------------------
    def set_graph(self, fixed_edges, candidate_edges):
        self.fixed_edges = fixed_edges
        for e in self.fixed_edges:
            self.update_nb_poses(e)
            self.update_initial_fixed_edge_exists(e)
        for e in candidate_edges:
            self.update_nb_poses(e)
        for e in candidate_edges:
            self.candidate_edges[self.edge_key(e)] = e

    def add_fixed_edge(self, edge):
        self.fixed_edges.append(edge)
        self.update_nb_poses(edge)
        self.update_initial_fixed_edge_exists(edge)

    def add_candidate_edge(self, edge):
        key = self.edge_key(edge)
        if key in self.already_considered_matches:  # failed or already fixed (acm.py:169-170)
            return
        self.candidate_edges[key] = edge
        self.update_nb_poses(edge)

    def remove_candidate_edges(self, edges, failed=False):
        """Drop `edges` from the candidates and blacklist them (acm.py:177-189).
        EdgeInterRobot equality == equality of normalised keys, so a key lookup replaces the
        reference's scan."""
        keys = [self.edge_key(e) for e in edges]
        for e, key in zip(edges, keys):
            self.candidate_edges.pop(key, None)
            if e.robot0_id == e.robot1_id:
                # same-robot edges compare equal in either orientation (acm.py:17-30) but
                # their two orientations have different keys
                self.candidate_edges.pop(key[2:] + key[:2], None)
        self.already_considered_matches.update(keys)

    def candidate_edges_to_fixed(self, edges):
        """Successful candidates become fixed edges of weight `fixed_weight` (acm.py:191-202);
        the caller's list is rewritten in place like the reference does."""
        for i in range(len(edges)):
            edges[i] = self.replace_weight(edges[i], weight=self.fixed_weight)
            self.update_initial_fixed_edge_exists(edges[i])
        self.fixed_edges.extend(edges)
        self.remove_candidate_edges(edges)

    def add_match(self, match):
        """Keep the best-weight candidate per pair (acm.py:558-571) — raw-key lookup quirk."""
        raw = (match.robot0_id, match.robot0_keyframe_id, match.robot1_id,
               match.robot1_keyframe_id)
        known = self.candidate_edges.get(raw)
        if known is None or match.weight > known.weight:
            self.add_candidate_edge(match)

    # ---- initial guesses (acm.py:204-254) --------------------------------------------------
    def greedy_initialization(self, nb_candidates_to_choose, edges, weights=None):
        """One-hot vector of the heaviest edges; the very numpy call of acm.py:213-214 so that
        ties resolve identically.  `weights` (float64 array, same order) may replace `edges`."""
        if weights is None:
            weights = [e.weight for e in edges]
        w_init = np.zeros(len(weights))
        chosen = np.argpartition(weights, -nb_candidates_to_choose)[-nb_candidates_to_choose:]
        w_init[chosen] = 1.0
        return w_init

    def pseudo_greedy_initialization(self, nb_candidates_to_choose, nb_random, edges, weights=None):
        """Greedy for all but `nb_random` picks, the rest drawn with the GLOBAL numpy RNG
        (acm.py:219-243)."""
        w_init = self.greedy_initialization(nb_candidates_to_choose - nb_random, edges, weights)
        nb_edges = len(weights) if weights is not None else len(edges)
        drawn, trial, max_trials = 0, 0, 2 * nb_random
        while drawn < nb_random and trial < max_trials:
            j = int(np.random.rand() * nb_edges)
            if w_init[j] < 0.5:
                w_init[j] = 1.0
                drawn += 1
            trial += 1
        if trial >= max_trials:
            w_init = self.greedy_initialization(nb_candidates_to_choose, edges, weights)
        return w_init

    def random_initialization(self, nb_candidates_to_choose, edges):
        """Random weights then greedy (acm.py:245-254); mutates `edges` like the reference."""
        for e in range(len(edges)):
            edges[e] = self.replace_weight(edges[e], np.random.rand())
        return self.greedy_initialization(nb_candidates_to_choose, edges)

    def connection_biased_greedy_selection(self, nb_candidates_to_choose, edges,
                                           is_robot_included):
        """Greedy selection that first links every included robot lacking a fixed edge
        (acm.py:256-288)."""
        pool = edges.copy()
        forced = []
        for rid in [r for r in is_robot_included if is_robot_included[r]]:
            if self.initial_fixed_edge_exists[rid]:
                continue
            best, best_w = None, -1
            for i, e in enumerate(pool):
                if (e.robot0_id == rid or e.robot1_id == rid) and e.weight > best_w:
                    best, best_w = i, e.weight
            if best is not None:
                forced.append(best)
                pool[best] = self.replace_weight(pool[best], weight=0.0)
        w_init = np.zeros(len(edges))
        remaining = nb_candidates_to_choose - len(forced)
        if remaining > 0:
            w_init = self.greedy_initialization(remaining,
                                                self.rekey_edges(pool, is_robot_included))
        for i in forced:
            w_init[i] = 1.0
        return w_init

    # ---- multi-robot ↔ single-graph keys (acm.py:290-388) ----------------------------------
    def compute_offsets(self, is_robot_included):
        """Node-id offset of every included robot = poses of the included robots before it;
        excluded robots keep offset 0 (acm.py:290-309)."""
        self.offsets = {r: 0 for r in range(self.max_nb_robots)}
        running = 0
        for r in range(self.max_nb_robots):
            if is_robot_included[r]:
                self.offsets[r] = running
                running += self.nb_poses[r]

    def rekey_edges(self, edges, is_robot_included):
        """(robot, keyframe) pairs → single-graph node ids; edges touching an excluded robot
        are dropped (acm.py:311-333)."""
        out = []
        for e in edges:
            if is_robot_included[e.robot0_id] and is_robot_included[e.robot1_id]:
                out.append(Edge(self.offsets[e.robot0_id] + e.robot0_keyframe_id,
                                self.offsets[e.robot1_id] + e.robot1_keyframe_id, e.weight))
        return out

    def get_included_edges(self, edges, is_robot_included):
        return [e for e in edges
                if is_robot_included[e.robot0_id] and is_robot_included[e.robot1_id]]

    def fill_odometry(self):
        """Odometry chain k→k+1 of EVERY known robot, included or not (acm.py:347-361)."""
        odom = []
        for r in range(len(self.nb_poses)):
            base = self.offsets[r]
            for k in range(self.nb_poses[r] - 1):
                odom.append(Edge(base + k, base + k + 1, self.fixed_weight))
        return odom

    def recover_inter_robot_edges(self, edges, is_robot_included):
        """Inverse of rekey_edges (acm.py:363-388): the robot of a node id is the LAST included
        robot with a non-zero id whose offset does not exceed it (robot 0 by default)."""
        out = []
        for e in edges:
            r0 = r1 = 0
            for o in self.offsets:
                if o != 0 and is_robot_included[o]:
                    if e.i >= self.offsets[o]:
                        r0 = o
                    if e.j >= self.offsets[o]:
                        r1 = o
            out.append(EdgeInterRobot(r0, e.i - self.offsets[r0], r1, e.j - self.offsets[r1],
                                      e.weight))
        return out

    # ---- checks (acm.py:390-433) -----------------------------------------------------------
    def check_graph_disconnections(self, is_other_robot_considered):
        """A robot is included iff it is the local one or some fixed/candidate edge touches it
        while it is considered (acm.py:390-416)."""
        connected = {r: (r == self.robot_id) for r in range(self.max_nb_robots)}
        for e in list(self.fixed_edges) + list(self.candidate_edges.values()):
            if is_other_robot_considered[e.robot0_id]:
                connected[e.robot0_id] = True
            if is_other_robot_considered[e.robot1_id]:
                connected[e.robot1_id] = True
        return connected

    def check_initial_fixed_measurements_exists(self, is_robot_included):
        return all(self.initial_fixed_edge_exists[r] for r in is_robot_included
                   if is_robot_included[r])

    # ---- solver (acm.py:435-465) -----------------------------------------------------------
    def run_mac_solver(self, fixed_edges, candidate_edges, w_init, nb_candidates_to_choose,
                       mac=None, cand_weights=None):
        """MAC on the GPU with the reference's retry ladder: any exception (a disconnected
        L(w) in practice) re-draws the initial guess with growing randomness, at most
        `nb_candidates_to_choose` times, else the initial greedy guess is returned.
        `mac` / `cand_weights`: the array fast path of select_candidates hands over a MAC built
        from arrays and the candidate weights instead of edge lists."""
        if mac is None:
            mac = MAC(fixed_edges, candidate_edges, self.total_nb_poses, device_id=self.device_id)
        result = w_init.copy()
        trial = 0
        self.last_solver_exceptions = 0
        try:
            while trial < nb_candidates_to_choose:
                try:
                    result, _, _ = mac.fw_subset(w_init, nb_candidates_to_choose,
                                                 max_iters=self.max_iters)
                    break
                except Exception:  # the reference uses a bare except (acm.py:455)
                    trial += 1
                    self.last_solver_exceptions = trial
                    w_init = self.pseudo_greedy_initialization(nb_candidates_to_choose, trial,
                                                               candidate_edges, cand_weights)
        finally:
            mac.close()
        return result

    # ---- array fast path of select_candidates ------------------------------------------------
    @staticmethod
    def _inter_robot_rows(edges, count):
        """iterable of EdgeInterRobot → float64 [count, 5] (r0, kf0, r1, kf1, weight) in one pass."""
        if count == 0:
            return np.zeros((0, 5))
        return np.fromiter(itertools.chain.from_iterable(edges), dtype=np.float64,
                           count=5 * count).reshape(count, 5)

    def _select_candidates_arrays(self, nb_candidates_to_choose, is_other_robot_considered,
                                  greedy_initialization):
        """select_candidates for the common case (greedy initial guess, MAC branch, no comparison
        logs) without materialising one Python object per edge: at cfg3 scale (200k candidates)
        the list-based assembly below costs 0.5 s around a 25 ms solve.  Same graph, same edge
        order, same numpy calls for the initial guess, same recovery of the selected edges —
        returns None when the case does not apply and the list path must run."""
        if not greedy_initialization or self.params["evaluation.enable_sparsification_comparison"] \
                or not self.params["frontend.enable_sparsification"]:
            return None
        nr = self.max_nb_robots
        c_r0, c_k0, c_r1, c_k1, c_w = self.candidate_edges.live_columns()
        fixed = self._inter_robot_rows(self.fixed_edges, len(self.fixed_edges))
        f_r0, f_k0, f_r1, f_k1 = (fixed[:, c].astype(np.int64) for c in range(4))
        considered = np.array([bool(is_other_robot_considered[r]) for r in range(nr)])
        # check_graph_disconnections (acm.py:390-416)
        seen = np.zeros(nr, dtype=bool)
        for ids in (f_r0, f_r1, c_r0, c_r1):
            seen[ids] = True
        included = seen & considered
        included[self.robot_id] = True
        is_robot_included = {r: bool(included[r]) for r in range(nr)}
        if not self.check_initial_fixed_measurements_exists(is_robot_included):
            return None  # connection-biased greedy branch: list path
        self.compute_offsets(is_robot_included)
        off = np.array([self.offsets[r] for r in range(nr)], dtype=np.int64)

        def rekey(r0, k0, r1, k1, w):
            if not included[seen].all():             # some edge touches an excluded robot
                keep = included[r0] & included[r1]
                r0, k0, r1, k1, w = r0[keep], k0[keep], r1[keep], k1[keep], w[keep]
            return off[r0] + k0, off[r1] + k1, w

        fi, fj, fw = rekey(f_r0, f_k0, f_r1, f_k1, fixed[:, 4])
        ci, cj, cw = rekey(c_r0, c_k0, c_r1, c_k1, c_w)
        # fill_odometry (acm.py:347-361): EVERY known robot, included or not
        oi = [np.arange(self.offsets[r], self.offsets[r] + max(self.nb_poses[r] - 1, 0), dtype=np.int64)
              for r in range(len(self.nb_poses))]
        oi = np.concatenate(oi) if oi else np.zeros(0, np.int64)
        fi = np.concatenate([fi, oi])
        fj = np.concatenate([fj, oi + 1])
        fw = np.concatenate([fw, np.full(len(oi), float(self.fixed_weight))])

        nb = min(nb_candidates_to_choose, len(cw))
        if len(cw) == 0:
            return []
        self.total_nb_poses = sum(self.nb_poses[r] for r in range(len(self.nb_poses)))
        w_init = self.greedy_initialization(nb, None, weights=cw)
        mac = MAC.from_arrays(fi, fj, fw, ci, cj, cw, self.total_nb_poses, device_id=self.device_id)
        result = self.run_mac_solver(None, None, w_init, nb, mac=mac, cand_weights=cw)
        picked = np.nonzero(result.astype(int))[0]
        chosen = [Edge(int(ci[i]), int(cj[i]), float(cw[i])) for i in picked]
        inter_robot_edges = self.recover_inter_robot_edges(chosen, is_robot_included)
        self.remove_candidate_edges(inter_robot_edges)
        return inter_robot_edges

    # ---- entry point (acm.py:467-542) ------------------------------------------------------
    def select_candidates(self, nb_candidates_to_choose, is_other_robot_considered,
                          greedy_initialization=True):
        """Choose `nb_candidates_to_choose` candidate edges (budget) among the robots in range.

        Returns:
            list(EdgeInterRobot): selected edges; they leave the candidate set.
        """
        fast = self._select_candidates_arrays(nb_candidates_to_choose, is_other_robot_considered,
                                              greedy_initialization)
        if fast is not None:
            return fast
        is_robot_included = self.check_graph_disconnections(is_other_robot_considered)
        self.compute_offsets(is_robot_included)
        rekeyed_fixed = self.rekey_edges(self.fixed_edges, is_robot_included)
        rekeyed_fixed.extend(self.fill_odometry())
        rekeyed_cand = self.rekey_edges(self.candidate_edges.values(), is_robot_included)

        nb_candidates_to_choose = min(nb_candidates_to_choose, len(rekeyed_cand))
        if len(rekeyed_cand) == 0:
            return []

        self.total_nb_poses = sum(self.nb_poses[r] for r in range(len(self.nb_poses)))

        if greedy_initialization:
            w_init = self.greedy_initialization(nb_candidates_to_choose, rekeyed_cand)
        else:
            w_init = self.random_initialization(nb_candidates_to_choose, rekeyed_cand)

        if (self.params["frontend.enable_sparsification"]
                and self.check_initial_fixed_measurements_exists(is_robot_included)):
            result = self.run_mac_solver(rekeyed_fixed, rekeyed_cand, w_init,
                                         nb_candidates_to_choose)
        else:
            result = self.connection_biased_greedy_selection(
                nb_candidates_to_choose,
                self.get_included_edges(self.candidate_edges.values(), is_robot_included),
                is_robot_included)

        if self.params["evaluation.enable_sparsification_comparison"]:
            self.sparsification_comparison_logs(rekeyed_cand, is_robot_included, w_init, result)

        chosen = [rekeyed_cand[i] for i in np.nonzero(result.astype(int))[0]]
        inter_robot_edges = self.recover_inter_robot_edges(chosen, is_robot_included)
        self.remove_candidate_edges(inter_robot_edges)
        return inter_robot_edges

    def sparsification_comparison_logs(self, rekeyed_candidate_edges, is_robot_included,
                                       greedy_result, mac_result):
        """Greedy vs MAC selections for the evaluation logs (acm.py:544-556)."""
        pick = lambda mask: self.recover_inter_robot_edges(
            [rekeyed_candidate_edges[i] for i in np.nonzero(mask.astype(int))[0]],
            is_robot_included)
        self.log_greedy_edges = pick(greedy_result)
        self.log_mac_edges = pick(mac_result)
