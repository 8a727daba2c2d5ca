"""MI355X-native NearestNeighborsMatching — host mirror of cslam/nns_matching.py:5-78.

Same class name, method names, argument meaning and return types as the reference; all
arithmetic runs in libcslam_hip.so (HIP, gfx950) through the C ABI of include/cslam_hip.h.
No torch on this path and no CPU fallback: constructing a matcher without the built library or
without a GPU raises.

Differences that are deliberate:
  * ties are broken by lowest row first (the reference's torch.argsort is unstable);
  * `search_batch` / `search_best_batch` are additions that evaluate a whole message of
    descriptors in one launch (the loop at
    cslam/global_descriptor_loop_closure_detection.py:396-404).
"""
import ctypes

import numpy as np

from . import _lib


def _to_numpy_f32(vector):
    """Accept np.ndarray (any float dtype), python lists or torch tensors
    (nns_matching.py:30-31, :58-59)."""
    if hasattr(vector, "detach") and hasattr(vector, "cpu"):  # torch tensor, without importing torch
        vector = vector.detach().cpu().numpy()
    return np.ascontiguousarray(np.asarray(vector), dtype=np.float32)


class _DataView(object):
    """What callers and tests do with `matcher.data` (nns_matching.py:19-21):
    `len(data)`, `data[row, col]`, `data[row]`, `data.cpu().numpy()[:n]`, `data.shape`."""

    def __init__(self, owner):
        self._owner = owner

    def _host(self):
        o = self._owner
        cap = o._capacity()
        out = np.zeros((cap, o.dim), dtype=np.float32)
        if cap > 0:
            _lib.check(o._lib.cslam_pool_read_rows(o._pool, 0, cap, _lib.ptr(out)))
        return out

    def __len__(self):
        return int(self._owner._capacity())

    @property
    def shape(self):
        return (int(self._owner._capacity()), int(self._owner.dim))

    def cpu(self):
        return self

    def numpy(self):
        return self._host()

    def __getitem__(self, key):
        o = self._owner
        if isinstance(key, (int, np.integer)) or (
                isinstance(key, tuple) and len(key) == 2
                and isinstance(key[0], (int, np.integer))):
            row = int(key if not isinstance(key, tuple) else key[0])
            if row < 0:
                row += len(self)
            out = np.zeros((1, o.dim), dtype=np.float32)
            _lib.check(o._lib.cslam_pool_read_rows(o._pool, row, 1, _lib.ptr(out)))
            return out[0] if not isinstance(key, tuple) else out[0][key[1]]
        return self._host()[key]

    def __array__(self, dtype=None, copy=None):
        h = self._host()
        return h if dtype is None else h.astype(dtype)


class NearestNeighborsMatching(object):
    """Nearest Neighbor matching of description vectors (device-resident pool)."""

    def __init__(self, dim=None, device_id=0, capacity_hint=0):
        """
        Args:
            dim (int, optional): Global descriptor size. Defaults to None (fixed by the first
                add_item, nns_matching.py:34-36).
            device_id (int): HIP device of the pool.
            capacity_hint (int): rows to reserve (0 → 1000 like the reference).
        """
        self._lib = _lib.load()
        if self._lib.cslam_device_count() <= 0:
            raise _lib.CslamHipError(_lib.CSLAM_E_NODEVICE,
                                     "no HIP device visible; cslam_amd has no CPU path")
        self.n = 0
        self.device_id = int(device_id)
        self.device = f"hip:{self.device_id}"
        self.dim = dim
        self.items = dict()
        self._pool = None
        self._capacity_hint = int(capacity_hint)
        self.data = []  # like the reference until the dimension is known
        self._one = {}  # k -> result buffers of the one-query calls (search_best, search)
        if dim is not None:
            self._create(int(dim))

    # -- lifetime ------------------------------------------------------------------------
    def _create(self, dim):
        handle = ctypes.c_void_p()
        _lib.check(self._lib.cslam_pool_create(dim, self._capacity_hint, self.device_id,
                                               ctypes.byref(handle)))
        self._pool = handle
        self.dim = dim
        self.data = _DataView(self)

    def _capacity(self):
        return self._lib.cslam_pool_capacity(self._pool) if self._pool else 0

    def close(self):
        if getattr(self, "_pool", None):
            self._lib.cslam_pool_destroy(self._pool)
            self._pool = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        """cslam_pool_t* for zero-copy callers (bench.py, multi_robot.py)."""
        return self._pool

    # -- add_item (nns_matching.py:23-42) ------------------------------------------------
    def add_item(self, vector, item):
        """Append one descriptor to the pool (one H2D row copy; capacity doubles when full).

        vector: 1-D float array (any float dtype; stored as fp32).
        item:   whatever the caller wants back from `search` for this row (a keyframe id).
        """
        if not (type(vector) is np.ndarray and vector.dtype == np.float32 and vector.flags["C_CONTIGUOUS"]):
            vector = _to_numpy_f32(vector)
        if self._pool is None:
            self._create(vector.size)
        if vector.size != self.dim:
            raise ValueError(f"descriptor size {vector.size} != pool dim {self.dim}")
        rc = self._lib.cslam_pool_add(self._pool, vector.ctypes.data, 1)
        if rc != 0:
            _lib.check(rc)
        self.items[self.n] = item
        self.n += 1

    def add_items(self, vectors, items):
        """Batched add_item: vectors [B, dim], items: sequence of B ids."""
        vectors = _to_numpy_f32(vectors)
        if vectors.ndim != 2:
            raise ValueError("add_items expects a [B, dim] array")
        if len(items) != vectors.shape[0]:
            raise ValueError("add_items: len(items) != number of rows")
        if vectors.shape[0] == 0:
            return
        if self._pool is None:
            self._create(vectors.shape[1])
        if vectors.shape[1] != self.dim:
            raise ValueError(f"descriptor size {vectors.shape[1]} != pool dim {self.dim}")
        _lib.check(self._lib.cslam_pool_add(self._pool, _lib.ptr(vectors), vectors.shape[0]))
        for it in items:
            self.items[self.n] = it
            self.n += 1

    def add_items_device(self, dev_ptr, n_rows, row_stride, items=None, stream=None):
        """Append rows that already live in HBM (e.g. an all-gathered block).  `items=None` maps
        every new row to its own row number (callers that only use row numbers)."""
        if self._pool is None:
            raise ValueError("pool dimension unknown; construct with dim=")
        n_rows, row_stride = int(n_rows), int(row_stride)
        if n_rows < 0 or (n_rows > 0 and row_stride < self.dim):
            raise ValueError(f"add_items_device: row_stride {row_stride} < dim {self.dim}")
        if items is not None and len(items) != n_rows:
            raise ValueError("add_items_device: len(items) != n_rows")
        if n_rows == 0:
            return
        _lib.check(self._lib.cslam_pool_add_dev(self._pool, ctypes.c_void_p(dev_ptr), n_rows,
                                                row_stride, ctypes.c_void_p(stream or 0)))
        first = self.n
        self.n = int(self._lib.cslam_pool_size(self._pool))   # the library is the source of truth
        if self.n != first + n_rows:
            raise RuntimeError("pool size out of step with the host mirror")
        if items is None:
            items = range(first, first + n_rows)
        for i, it in enumerate(items):
            self.items[first + i] = it

    def add_items_tensor(self, vectors, items):
        """Append the rows of a CUDA float32 tensor [B, dim] (e.g. descriptors straight out of the
        VPR head): device-to-device, on torch's current stream, no host round trip."""
        import torch
        if not (isinstance(vectors, torch.Tensor) and vectors.is_cuda):
            raise TypeError("add_items_tensor expects a CUDA tensor (use add_items for host data)")
        if vectors.dim() != 2 or vectors.dtype != torch.float32 or vectors.stride(1) != 1:
            raise ValueError("add_items_tensor expects float32 [B, dim] with unit inner stride")
        if len(items) != vectors.shape[0]:
            raise ValueError("add_items_tensor: len(items) != number of rows")
        if vectors.shape[0] == 0:
            return
        if self._pool is None:
            self._create(vectors.shape[1])
        if vectors.shape[1] != self.dim:
            raise ValueError(f"descriptor size {vectors.shape[1]} != pool dim {self.dim}")
        if vectors.device.index != self.device_id:
            raise ValueError(f"tensor lives on cuda:{vectors.device.index}, the pool on hip:{self.device_id}")
        self.add_items_device(vectors.data_ptr(), vectors.shape[0], vectors.stride(0), list(items),
                              stream=torch.cuda.current_stream().cuda_stream)

    def search_best_batch_tensor(self, queries):
        """search_best_batch for queries that live in HBM (CUDA float32 [M, dim]): only the M
        (row, similarity) pairs come back to the host."""
        import torch
        m = queries.shape[0]
        if self._pool is None or self.n == 0:
            return [None] * m, [None] * m
        if queries.dim() != 2 or queries.dtype != torch.float32 or queries.stride(1) != 1 \
                or queries.shape[1] != self.dim:
            raise ValueError("search_best_batch_tensor expects float32 [M, dim] with unit inner stride")
        if not queries.is_cuda or queries.device.index != self.device_id:
            raise ValueError(f"queries must live on cuda:{self.device_id} (the pool's device)")
        idx = torch.empty((m, 1), dtype=torch.int32, device=queries.device)
        sim = torch.empty((m, 1), dtype=torch.float32, device=queries.device)
        _lib.check(self._lib.cslam_pool_search_dev(
            self._pool, ctypes.c_void_p(queries.data_ptr()), queries.stride(0), m, 1,
            ctypes.c_void_p(idx.data_ptr()), ctypes.c_void_p(sim.data_ptr()),
            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
        rows = idx[:, 0].cpu().numpy()
        return [self.items[int(r)] for r in rows], sim[:, 0].cpu().numpy()

    # -- search (nns_matching.py:44-63) --------------------------------------------------
    def search_batch(self, queries, k):
        """Top-k of every query row in one pass.

        Returns (rows[M, kk] int32, sims[M, kk] float32) with kk = min(k, n)."""
        queries = _to_numpy_f32(queries)
        if queries.ndim == 1:
            queries = queries.reshape(1, -1)
        m = queries.shape[0]
        if self._pool is None or self.n == 0 or k <= 0 or m == 0:
            return np.zeros((m, 0), np.int32), np.zeros((m, 0), np.float32)
        if queries.shape[1] != self.dim:
            raise ValueError(f"query size {queries.shape[1]} != pool dim {self.dim}")
        k = int(k)
        idx = np.empty((m, k), np.int32)
        sim = np.empty((m, k), np.float32)
        cnt = np.empty(m, np.int32)
        _lib.check(self._lib.cslam_pool_search(self._pool, _lib.ptr(queries), m, k,
                                               _lib.ptr(idx), _lib.ptr(sim), _lib.ptr(cnt)))
        kk = int(cnt[0])
        return idx[:, :kk], sim[:, :kk]

    def _search_one(self, query, k):
        """One query, top-k, n > 0: (idx[k] int32, sim[k] float32, kk) in buffers kept per pool and k - the
        calls the reference makes per keyframe (`search_best`, `search(q, nb_best_matches)`): allocating the
        result arrays and their ctypes pointers per call was a third of the call's time at the reference's
        own pool sizes.  The buffers are overwritten by the next call."""
        buf = self._one.get(k)
        if buf is None:
            idx, sim, cnt = np.empty(k, np.int32), np.empty(k, np.float32), np.empty(1, np.int32)
            buf = self._one[k] = (idx, sim, _lib.ptr(idx), _lib.ptr(sim), _lib.ptr(cnt))
        q = query if (type(query) is np.ndarray and query.dtype == np.float32 and query.flags["C_CONTIGUOUS"]) \
            else _to_numpy_f32(query)
        if q.size != self.dim:
            raise ValueError(f"query size {q.size} != pool dim {self.dim}")
        rc = self._lib.cslam_pool_search(self._pool, q.ctypes.data, 1, k, buf[2], buf[3], buf[4])
        if rc != 0:
            _lib.check(rc)
        return buf[0], buf[1], (k if k < self.n else self.n)

    def search(self, query, k):
        """Cosine top-k of one query against rows [0, n).

        Returns (items of the min(k, n) most similar rows, best first; their similarities as
        np.float32[min(k, n)]); `([], [])` while the pool has no storage yet.
        """
        if len(self.data) == 0:
            return [], []
        k = int(k)
        if self.n == 0 or k <= 0 or k > 1024:
            rows, sims = self.search_batch(_to_numpy_f32(query).reshape(1, -1), k)
            return [self.items[int(r)] for r in rows[0]], sims[0]
        idx, sim, kk = self._search_one(query, k)
        items = self.items
        return [items[r] for r in idx[:kk].tolist()], sim[:kk].copy()

    def search_best(self, query):
        """`search(query, 1)` unpacked: (item, similarity) of the most similar row, or
        (None, None) while the pool has no storage yet."""
        if len(self.data) == 0:
            return None, None
        if self.n == 0:
            items, similarities = self.search(query, 1)
            return items[0], similarities[0]
        idx, sim, _ = self._search_one(query, 1)
        return self.items[int(idx[0])], sim[0]

    def search_best_batch(self, queries):
        """search_best for every row of `queries`: (items list, sims[M]); empty pool →
        ([None]*M, [None]*M)."""
        queries = _to_numpy_f32(queries)
        if queries.ndim == 1:
            queries = queries.reshape(1, -1)
        if len(self.data) == 0 or self.n == 0:
            return [None] * queries.shape[0], [None] * queries.shape[0]
        rows, sims = self.search_batch(queries, 1)
        return [self.items[int(r)] for r in rows[:, 0]], sims[:, 0]
