"""Driver entry points: build() compiles everything, smoke() runs one small hot-path call."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def build() -> None:
    """Compile every HIP source for gfx950 into cslam_amd/libcslam_hip.so (hipcc cross-compiles
    without a GPU), compile the C oracle (the parity checker), and import the package."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "cslam_amd", "csrc"), "ARCH=gfx950"])
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    # No C/C++ reference to build into oracle/_ref: the reference's hot path is Python
    # (DESIGN.md §2); golden vectors come from tests/golden/gen_golden.py instead.
    import cslam_amd  # noqa: F401
    from cslam_amd import _lib
    lib = _lib.load()
    assert lib.cslam_abi_version() == 1


def smoke() -> None:
    """One small kNN search + one small MAC selection on cuda:0, checked against the oracle."""
    import numpy as np
    from cslam_amd import _lib
    from cslam_amd.nns_matching import NearestNeighborsMatching
    from oracle import knn as oracle_knn

    lib = _lib.load()
    assert lib.cslam_device_count() > 0, "smoke() needs a GPU"
    rs = np.random.RandomState(0)
    X = rs.rand(3000, 256).astype(np.float32)
    Q = rs.rand(5, 256).astype(np.float32)
    m = NearestNeighborsMatching()
    m.add_items(X, list(range(3000)))
    rows, sims = m.search_batch(Q, 10)
    o_idx, o_sim, _ = oracle_knn.search(X, Q, 10)
    assert np.array_equal(rows, o_idx), "kNN indices differ from the oracle"
    assert sims.tobytes() == o_sim.tobytes(), "kNN similarities differ from the oracle"
    m.close()
    try:
        from cslam_amd.mac.mac import smoke_check
    except ImportError:
        smoke_check = None
    if smoke_check is not None:
        smoke_check()
    print("smoke ok")


if __name__ == "__main__":
    build()
    if len(sys.argv) > 1 and sys.argv[1] == "smoke":
        smoke()
