"""Edge containers of the MAC problem — host mirror of cslam/mac/utils.py:12,44-122.

`Edge(i, j, weight)` is the rekeyed single-graph edge the reference passes to `MAC`
(utils.py:13).  The Laplacian builders return scipy CSR matrices like the reference's
(utils.py:80-81, :121-122); they exist for API compatibility and parity tests — the device
solver works on an edge adjacency and never materialises L (csrc/mac.hip).
"""
from collections import namedtuple

import numpy as np

Edge = namedtuple("Edge", ["i", "j", "weight"])


def _csr_from_triplets(ei, ej, ew, n):
    from scipy.sparse import coo_matrix, csr_matrix
    ei = np.asarray(ei, dtype=np.int64).reshape(-1)
    ej = np.asarray(ej, dtype=np.int64).reshape(-1)
    ew = np.asarray(ew, dtype=np.float64).reshape(-1)
    rows = np.column_stack([ei, ej, ei, ej]).reshape(-1)
    cols = np.column_stack([ei, ej, ej, ei]).reshape(-1)
    vals = np.column_stack([ew, ew, -ew, -ew]).reshape(-1)
    return csr_matrix(coo_matrix((vals, (rows, cols)), shape=(n, n)))


def weight_graph_lap_from_edge_list(edges, num_vars):
    """Weighted graph Laplacian of a list of `Edge` (utils.py:44-81)."""
    return _csr_from_triplets([e.i for e in edges], [e.j for e in edges],
                              [e.weight for e in edges], num_vars)


def weight_graph_lap_from_edges(edges, weights, num_poses):
    """Weighted graph Laplacian from an (m,2) index array and a weight vector
    (utils.py:83-122)."""
    edges = np.asarray(edges).reshape(-1, 2)
    return _csr_from_triplets(edges[:, 0], edges[:, 1], weights, num_poses)


def split_measurements(measurements):
    """Odometry (|i-j| <= 1) vs loop-closure measurements (utils.py:125-141)."""
    odom = [m for m in measurements if abs(m.j - m.i) <= 1]
    lcs = [m for m in measurements if abs(m.j - m.i) > 1]
    return odom, lcs


def select_measurements(measurements, w):
    """Measurements whose selection weight is exactly one (utils.py:144-154)."""
    assert len(measurements) == len(w)
    return [m for m, wi in zip(measurements, w) if wi == 1.0]
