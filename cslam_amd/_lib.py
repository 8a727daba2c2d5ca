"""ctypes binding of libcslam_hip.so (C ABI declared in include/cslam_hip.h).

There is NO CPU fallback: if the shared library is missing or no gfx950 device is visible the
classes in this package raise.  Build with ``python -c 'import __graft_entry__ as g; g.build()'``
or ``make -C cslam_amd/csrc``.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcslam_hip.so")

CSLAM_OK = 0
CSLAM_E_INVALID = -1
CSLAM_E_HIP = -2
CSLAM_E_NODEVICE = -3
CSLAM_E_DISCONNECTED = -4
CSLAM_E_NOCONV = -5
CSLAM_E_COMM = -6
CSLAM_COMM_ID_BYTES = 128
CSLAM_COMM_OWN_STREAM = ctypes.c_void_p(-1)


class CslamHipError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"libcslam_hip error {code}: {message}")
        self.code = code


class GraphDisconnectedError(CslamHipError):
    """L(w) is the Laplacian of a disconnected graph (the reference's SuperLU raises
    'Factor is exactly singular' there; cslam/algebraic_connectivity_maximization.py:455)."""


_c = ctypes
_i32p = _c.POINTER(_c.c_int32)
_i64p = _c.POINTER(_c.c_int64)
_f32p = _c.POINTER(_c.c_float)
_f64p = _c.POINTER(_c.c_double)
_vp = _c.c_void_p

# name → (restype, argtypes); every symbol of include/cslam_hip.h
SIGNATURES = {
    "cslam_last_error": (_c.c_char_p, []),
    "cslam_abi_version": (_c.c_int, []),
    "cslam_device_count": (_c.c_int, []),
    "cslam_pool_create": (_c.c_int, [_c.c_int, _c.c_int64, _c.c_int, _c.POINTER(_vp)]),
    "cslam_pool_destroy": (_c.c_int, [_vp]),
    "cslam_pool_add": (_c.c_int, [_vp, _vp, _c.c_int64]),
    "cslam_pool_add_dev": (_c.c_int, [_vp, _vp, _c.c_int64, _c.c_int64, _vp]),
    "cslam_pool_size": (_c.c_int64, [_vp]),
    "cslam_pool_capacity": (_c.c_int64, [_vp]),
    "cslam_pool_dim": (_c.c_int, [_vp]),
    "cslam_pool_device": (_c.c_int, [_vp]),
    "cslam_pool_read_rows": (_c.c_int, [_vp, _c.c_int64, _c.c_int64, _vp]),
    "cslam_pool_read_inv_norms": (_c.c_int, [_vp, _c.c_int64, _c.c_int64, _vp]),
    "cslam_pool_device_view": (_c.c_int, [_vp, _c.POINTER(_vp), _i64p]),
    "cslam_pool_search": (_c.c_int, [_vp, _vp, _c.c_int, _c.c_int, _vp, _vp, _vp]),
    "cslam_pool_search_dev": (_c.c_int, [_vp, _vp, _c.c_int64, _c.c_int, _c.c_int, _vp, _vp, _vp]),
    "cslam_pool_search_range_dev": (_c.c_int, [_vp, _vp, _c.c_int64, _c.c_int, _c.c_int, _c.c_int64,
                                               _c.c_int64, _vp, _vp, _vp]),
    "cslam_pool_search_best_split_dev": (_c.c_int, [_vp, _vp, _c.c_int64, _c.c_int, _c.c_int64, _c.c_int64,
                                                    _vp, _vp, _vp, _vp, _vp]),
    "cslam_pool_similarities_dev": (_c.c_int, [_vp, _vp, _c.c_int64, _c.c_int, _vp, _c.c_int64, _vp]),
    "cslam_pool_profile_begin": (_c.c_int, [_vp, _c.c_int]),
    "cslam_pool_profile_begin_sampled": (_c.c_int, [_vp, _c.c_int, _c.c_int]),
    "cslam_pool_profile_end": (_c.c_int, [_vp, _f64p, _c.POINTER(_c.c_int)]),
    "cslam_pool_scan_kernel": (_c.c_char_p, [_vp]),
    "cslam_comm_unique_id": (_c.c_int, [_vp]),
    "cslam_comm_init": (_c.c_int, [_c.c_int, _c.c_int, _vp, _c.c_int, _c.POINTER(_vp)]),
    "cslam_comm_init_loopback": (_c.c_int, [_c.c_int, _c.c_int, _c.c_int, _c.POINTER(_vp)]),
    "cslam_comm_init_custom": (_c.c_int, [_c.c_int, _c.c_int, _c.c_int, _vp, _vp, _c.POINTER(_vp)]),
    "cslam_device_copy": (_c.c_int, [_vp, _vp, _c.c_int64, _c.c_int, _vp, _c.c_int]),
    "cslam_comm_destroy": (_c.c_int, [_vp]),
    "cslam_comm_rank": (_c.c_int, [_vp]),
    "cslam_comm_size": (_c.c_int, [_vp]),
    "cslam_comm_stream": (_vp, [_vp]),
    "cslam_comm_stats": (_c.c_int, [_vp, _i64p, _i64p, _i32p]),
    "cslam_comm_allgather": (_c.c_int, [_vp, _vp, _vp, _c.c_int64, _vp]),
    "cslam_allgather_descriptors": (_c.c_int, [_vp, _vp, _c.c_int64, _c.c_int32, _c.c_int32,
                                               _c.c_int32, _vp, _vp, _vp]),
    "cslam_shard_create": (_c.c_int, [_vp, _vp, _vp, _c.c_int, _c.c_int, _c.c_int, _c.c_double,
                                      _c.c_int, _c.c_int, _c.POINTER(_vp)]),
    "cslam_shard_destroy": (_c.c_int, [_vp]),
    "cslam_shard_set_keyframes": (_c.c_int, [_vp, _c.c_int64, _vp, _c.c_int64]),
    "cslam_shard_publish": (_c.c_int, [_vp, _vp, _c.c_int64, _vp, _c.c_int, _vp]),
    "cslam_shard_match": (_c.c_int, [_vp, _vp]),
    "cslam_shard_fetch": (_c.c_int, [_vp, _vp, _c.c_int64, _i64p, _i64p]),
    "cslam_shard_max_records": (_c.c_int64, [_vp]),
    "cslam_shard_pending": (_c.c_int, [_vp, _i64p, _i64p, _i64p]),
    "cslam_mac_create": (_c.c_int, [_c.c_int32, _vp, _vp, _vp, _c.c_int64, _vp, _vp, _vp,
                                    _c.c_int64, _c.c_int, _c.POINTER(_vp)]),
    "cslam_mac_destroy": (_c.c_int, [_vp]),
    "cslam_mac_laplacian": (_c.c_int, [_vp, _vp, _c.c_double, _i64p, _vp, _vp, _vp]),
    "cslam_mac_fiedler": (_c.c_int, [_vp, _vp, _c.c_double, _f64p, _vp, _i32p]),
    "cslam_mac_grad": (_c.c_int, [_vp, _vp, _vp]),
    "cslam_topk_mask": (_c.c_int, [_vp, _c.c_int64, _c.c_int64, _vp]),
    "cslam_mac_round_tiebreak": (_c.c_int, [_vp, _vp, _c.c_int64, _c.c_int, _vp]),
    "cslam_mac_fw_subset": (_c.c_int, [_vp, _vp, _c.c_int64, _c.c_int, _c.c_double, _vp, _vp,
                                       _f64p, _i32p, _vp]),
    "cslam_knobs_reload": (_c.c_int, []),
    "cslam_pool_flush_dev": (_c.c_int, [_vp]),
    "cslam_mac_fw_trace_enable": (_c.c_int, [_vp, _c.c_int]),
    "cslam_mac_fw_trace_read": (_c.c_int, [_vp, _c.c_int32, _vp, _i64p, _f64p, _f64p, _i32p]),
    "cslam_mac_stats": (_c.c_int, [_vp, _i64p, _f64p]),
    "cslam_ransac_rigid3d": (_c.c_int, [_vp, _vp, _c.c_int64, _c.c_int, _c.c_double, _c.c_uint64,
                                        _c.c_int, _vp, _vp, _vp, _i32p]),
}

# transport callback of cslam_comm_init_custom
ALLGATHER_FN = _c.CFUNCTYPE(_c.c_int, _vp, _vp, _vp, _c.c_int64, _vp)

_LIB = None


def _preload_shared_hip_runtime():
    """One process must hold ONE HIP runtime.  PyTorch-ROCm wheels bundle their own
    libamdhip64.so (SONAME libamdhip64.so.7, looked up as 'libamdhip64.so'), so if
    /opt/rocm's copy were mapped first a later `torch.cuda` init would map a second runtime
    and fail with 'No HIP GPUs are available'.  When torch is installed, map its copy first
    (without importing torch); libcslam_hip.so's NEEDED libamdhip64.so.7 then resolves to it by
    SONAME.  Without torch the system ROCm runtime is used."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)


def load():
    """Load libcslam_hip.so (once).  Raises if it has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is not built. "
            "Run `make -C cslam_amd/csrc` (there is no CPU fallback).")
    _preload_shared_hip_runtime()
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so is stale
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


def check(rc):
    if rc == CSLAM_OK:
        return
    msg = load().cslam_last_error().decode("utf-8", "replace")
    if rc == CSLAM_E_DISCONNECTED:
        raise GraphDisconnectedError(rc, msg)
    raise CslamHipError(rc, msg)


def ptr(a):
    """void* of a C-contiguous numpy array (None → NULL)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_vp)


def as_f32(a, ndim=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if ndim is not None and a.ndim != ndim:
        raise ValueError(f"expected {ndim}-d array, got shape {a.shape}")
    return a


def as_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def as_i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)
