"""Sparse matching for loop closure detection — host mirror of
cslam/loop_closure_sparse_matching.py:6-115.

Matches global descriptors against the local pool / the other robots' pools to generate
inter-robot loop-closure candidates, then selects candidates within the communication budget.
Same constructor and methods as the reference; the pools live in HBM and every search is a
HIP scan (cslam_amd/nns_matching.py).  `add_other_robot_global_descriptors` (plural) is an
addition that evaluates a whole `GlobalDescriptors` message in one launch — the serial loop
of cslam/global_descriptor_loop_closure_detection.py:396-404.
"""
import numpy as np

from .algebraic_connectivity_maximization import (AlgebraicConnectivityMaximization,
                                                  EdgeInterRobot)
from .nns_matching import NearestNeighborsMatching


class _Descriptor(object):
    """(robot_id, keyframe_id) of a descriptor that arrived without a ROS message."""
    __slots__ = ("robot_id", "keyframe_id")

    def __init__(self, robot_id, keyframe_id):
        self.robot_id = robot_id
        self.keyframe_id = keyframe_id


class LoopClosureSparseMatching(object):

    def __init__(self, params, node, device_id=0):
        """
        Args:
            params (dict): ROS 2 parameters (the flat dict built at
                cslam/loop_closure_detection_node.py:45-89)
            node: object with get_logger() (the ROS node)
            device_id (int): HIP device holding the pools
        """
        self.params = params
        self.node = node
        if self.params["frontend.sensor_type"] == "lidar":
            # ScanContext (cslam/lidar_pr) is a different modality: out of scope
            raise NotImplementedError("lidar / ScanContext matching is outside the hot path")
        self.local_nnsm = NearestNeighborsMatching(device_id=device_id)
        self.other_robots_nnsm = {}
        for i in range(self.params["max_nb_robots"]):
            if i != self.params["robot_id"]:
                self.other_robots_nnsm[i] = NearestNeighborsMatching(device_id=device_id)
        # the reference passes no params here, so the selector runs on its defaults
        # (lcsm.py:34-35; SURVEY.md §5 "dead" ROS param)
        self.candidate_selector = AlgebraicConnectivityMaximization(
            self.params["robot_id"], self.params["max_nb_robots"], device_id=device_id)

    # ---- lcsm.py:37-56 ---------------------------------------------------------------------
    def add_local_global_descriptor(self, embedding, keyframe_id):
        """Add a local keyframe and match it against every other robot's pool.

        Returns: list(EdgeInterRobot) — the candidates created."""
        matches = []
        vector = np.asarray(embedding).astype(np.float32)
        self.local_nnsm.add_item(vector, keyframe_id)
        me = self.params["robot_id"]
        for i in range(self.params["max_nb_robots"]):
            if i == me:
                continue
            kf, similarity = self.other_robots_nnsm[i].search_best(vector)
            if kf is not None and similarity >= self.params["frontend.similarity_threshold"]:
                match = EdgeInterRobot(me, keyframe_id, i, kf, similarity)
                self.candidate_selector.add_match(match)
                matches.append(match)
        return matches

    def add_local_global_descriptors_device(self, embeddings, keyframe_ids):
        """Batched, device-resident form of add_local_global_descriptor: `embeddings` is a CUDA
        float32 tensor [B, D] straight from the VPR head (cslam_amd.vpr), appended to the local
        pool device-to-device and matched against every other robot's pool with one scan per
        pool.  Equivalent to the singular method called in order (the other robots' pools do not
        change meanwhile); only the B (keyframe, similarity) results per pool reach the host.

        Returns: list(EdgeInterRobot) — the candidates created, in keyframe order per robot."""
        keyframe_ids = [int(k) for k in keyframe_ids]
        self.local_nnsm.add_items_tensor(embeddings, keyframe_ids)
        me = self.params["robot_id"]
        matches = []
        per_robot = {}
        for i in range(self.params["max_nb_robots"]):
            if i != me:
                per_robot[i] = self.other_robots_nnsm[i].search_best_batch_tensor(embeddings)
        for row, keyframe_id in enumerate(keyframe_ids):     # the reference's nesting order
            for i, (kfs, sims) in per_robot.items():
                kf, similarity = kfs[row], sims[row]
                if kf is not None and similarity >= self.params["frontend.similarity_threshold"]:
                    match = EdgeInterRobot(me, keyframe_id, i, kf, similarity)
                    self.candidate_selector.add_match(match)
                    matches.append(match)
        return matches

    # ---- lcsm.py:58-76 ---------------------------------------------------------------------
    def add_other_robot_global_descriptor(self, msg):
        """Add a keyframe descriptor received from another robot and match it against the
        local pool.  Returns the EdgeInterRobot created, or None."""
        vector = np.asarray(msg.descriptor).astype(np.float32)
        self.other_robots_nnsm[msg.robot_id].add_item(vector, msg.keyframe_id)
        kf, similarity = self.local_nnsm.search_best(vector)
        return self._candidate_from_best(msg, kf, similarity)

    def _candidate_from_best(self, msg, kf, similarity):
        if kf is None or not similarity >= self.params["frontend.similarity_threshold"]:
            return None
        self.node.get_logger().info(
            f"Found potential matching KF: ({kf},{msg.keyframe_id}): {similarity}")
        match = EdgeInterRobot(self.params["robot_id"], kf, msg.robot_id, msg.keyframe_id,
                               similarity)
        self.candidate_selector.add_match(match)
        return match

    def add_other_robot_global_descriptors(self, msgs):
        """Batched form of add_other_robot_global_descriptor: all descriptors of a message are
        appended to their pools and matched against the local pool in ONE scan.  Equivalent to
        calling the singular method in order (the local pool does not change meanwhile).

        Returns: list of (EdgeInterRobot or None), one per message."""
        msgs = list(msgs)
        if not msgs:
            return []
        vectors = np.stack([np.asarray(m.descriptor).astype(np.float32) for m in msgs])
        by_robot = {}
        for row, m in enumerate(msgs):
            by_robot.setdefault(m.robot_id, []).append(row)
        for rid, rows in by_robot.items():
            self.other_robots_nnsm[rid].add_items(vectors[rows], [msgs[r].keyframe_id for r in rows])
        kfs, sims = self.local_nnsm.search_best_batch(vectors)
        return [self._candidate_from_best(m, kf, s) for m, kf, s in zip(msgs, kfs, sims)]

    def add_other_robot_global_descriptors_device(self, robot_id, embeddings, keyframe_ids):
        """Device-resident form of add_other_robot_global_descriptors for descriptors of ONE
        other robot that already live in HBM (an all-gathered block, or a co-located robot's VPR
        output): CUDA float32 [B, D] appended to that robot's pool and matched against the local
        pool in one scan, only the B results reach the host.

        Returns: list of (EdgeInterRobot or None), one per row."""
        keyframe_ids = [int(k) for k in keyframe_ids]
        self.other_robots_nnsm[robot_id].add_items_tensor(embeddings, keyframe_ids)
        kfs, sims = self.local_nnsm.search_best_batch_tensor(embeddings)
        out = []
        for keyframe_id, kf, s in zip(keyframe_ids, kfs, sims):
            out.append(self._candidate_from_best(_Descriptor(robot_id, keyframe_id), kf, s))
        return out

    # ---- lcsm.py:78-97 ---------------------------------------------------------------------
    def match_local_loop_closures(self, descriptor, kf_id):
        """Intra-robot loop closure: best of the `frontend.nb_best_matches` neighbours that is
        far enough in time and similar enough.  Returns (kf or None, similarities)."""
        vector = np.asarray(descriptor).astype(np.float32)
        kfs, similarities = self.local_nnsm.search(vector, k=self.params["frontend.nb_best_matches"])
        if len(kfs) > 0 and kfs[0] == kf_id:  # the query itself, only if it ranks first
            kfs, similarities = kfs[1:], similarities[1:]
        if len(kfs) == 0:
            return None, similarities
        gap = self.params["frontend.intra_loop_min_inbetween_keyframes"]
        thr = self.params["frontend.similarity_threshold"]
        for kf, similarity in zip(kfs, similarities):
            if abs(kf - kf_id) < gap or similarity < thr:
                continue
            return kf, similarities
        return None, similarities

    # ---- lcsm.py:99-115 --------------------------------------------------------------------
    def select_candidates(self, number_of_candidates, is_neighbor_in_range,
                          greedy_initialization=True):
        """Select inter-robot loop closure candidates according to the budget.

        Args:
            number_of_candidates (int): inter-robot loop closure budget
            is_neighbor_in_range (dict(int, bool)): robots in communication range
            greedy_initialization (bool): greedy initial guess

        Returns:
            list(EdgeInterRobot): selected edges
        """
        return self.candidate_selector.select_candidates(int(number_of_candidates),
                                                         dict(is_neighbor_in_range),
                                                         greedy_initialization)
