"""a20 / N1 pinned by the reference itself: the scripted scenarios of tests/ros_scenarios.py were
run on the reference's own GlobalDescriptorLoopClosureDetection / NeighborManager /
NeighborMonitor (tests/golden/gen_golden.py::gen_gdlcd, build container) and their bus traces
committed as tests/golden/gdlcd_ref.json.  Here the same scripts drive cslam_amd's surface:

  * CPU (-m "not gpu"): surface + LoopClosureSparseMatching with the C oracle plugged in as the
    kNN matcher — every scenario that needs no MAC solve;
  * GPU (-m gpu): everything on the HIP path, including the broker scenario (MAC selection,
    vertex cover, LocalDescriptorsRequest emission, feedback).

Everything must be equal; similarities within 1e-6 (torch-CPU cosine vs the pinned fp32 chain,
DESIGN.md §3)."""
import json
import os

import numpy as np
import pytest

import ros_scenarios
from cslam_amd import ros_compat as rc
from cslam_amd.loop_closure_detection_node import PARAMETERS, flatten_parameters
from fake_ros import FakeBus, FakeNode

HERE = os.path.dirname(os.path.abspath(__file__))


def _fixture():
    return json.load(open(os.path.join(HERE, "golden", "gdlcd_ref.json")))


def _desc():
    return np.load(os.path.join(HERE, "golden", "lcsm_desc.npz"))["desc"]


def assert_same(got, ref, path="trace"):
    if isinstance(ref, dict):
        assert isinstance(got, dict) and sorted(got) == sorted(ref), f"{path}: keys {sorted(got)} != {sorted(ref)}"
        for k in ref:
            assert_same(got[k], ref[k], f"{path}.{k}")
    elif isinstance(ref, (list, tuple)):
        assert isinstance(got, (list, tuple)) and len(got) == len(ref), \
            f"{path}: length {len(got) if hasattr(got, '__len__') else got} != {len(ref)}\n got {got}\n ref {ref}"
        for i, (g, r) in enumerate(zip(got, ref)):
            assert_same(g, r, f"{path}[{i}]")
    elif isinstance(ref, float) and not isinstance(ref, bool):
        assert isinstance(got, (int, float)) and abs(got - ref) <= 1e-6, f"{path}: {got} != {ref}"
    else:
        assert got == ref and type(got) == type(ref), f"{path}: {got!r} != {ref!r}"


def _make_robot(bus, rid, nb, overrides):
    from cslam_amd.global_descriptor_loop_closure_detection import GlobalDescriptorLoopClosureDetection
    ros_params = dict(PARAMETERS)
    ros_params.update({"robot_id": rid, "max_nb_robots": nb})
    ros_params.update(overrides)
    node = FakeNode(bus, f"/r{rid}", ros_params)
    return node, GlobalDescriptorLoopClosureDetection(flatten_parameters(node), node)


# ------------------------------------------------------------------------------------------------
# CPU: the oracle stands in for the device pool
# ------------------------------------------------------------------------------------------------
class OracleNNS(object):
    """NearestNeighborsMatching's interface over oracle/knn_oracle.c (test double, CPU)."""

    def __init__(self, dim=None, device_id=0, capacity_hint=0):
        self.n = 0
        self.items = {}
        self.rows = None
        self.data = []

    def add_item(self, vector, item):
        self.add_items(np.asarray(vector, np.float32).reshape(1, -1), [item])

    def add_items(self, vectors, items):
        vectors = np.asarray(vectors, np.float32)
        self.rows = vectors.copy() if self.rows is None else np.vstack([self.rows, vectors])
        for it in items:
            self.items[self.n] = it
            self.n += 1
        self.data = self.rows

    def search(self, query, k):
        from oracle import knn as oracle_knn
        if self.n == 0:
            return [], []
        idx, sim, cnt = oracle_knn.search(self.rows, np.asarray(query, np.float32).reshape(1, -1), k)
        c = int(cnt[0])
        return [self.items[int(i)] for i in idx[0][:c]], sim[0][:c]

    def search_best(self, query):
        items, sims = self.search(query, 1)
        return (items[0], sims[0]) if len(items) else (None, None)

    def search_best_batch(self, queries):
        queries = np.asarray(queries, np.float32)
        if self.n == 0:
            return [None] * len(queries), [None] * len(queries)
        from oracle import knn as oracle_knn
        idx, sim, _ = oracle_knn.search(self.rows, queries, 1)
        return [self.items[int(i)] for i in idx[:, 0]], sim[:, 0]


@pytest.mark.parametrize("name", ["chunked_publication", "match_forwarding_three_robots",
                                  "intra_robot_and_neighbors"])
def test_surface_equals_reference_trace_cpu(monkeypatch, name):
    import cslam_amd.loop_closure_sparse_matching as lcsm
    monkeypatch.setattr(lcsm, "NearestNeighborsMatching", OracleNNS)
    got = ros_scenarios.SCENARIOS[name](_make_robot, rc, FakeBus, _desc())
    assert_same(json.loads(json.dumps(got)), _fixture()[name], name)


# ------------------------------------------------------------------------------------------------
# GPU: the product path
# ------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(ros_scenarios.SCENARIOS))
def test_surface_equals_reference_trace_hip(gpu, name):
    got = ros_scenarios.SCENARIOS[name](_make_robot, rc, FakeBus, _desc())
    assert_same(json.loads(json.dumps(got)), _fixture()[name], name)
