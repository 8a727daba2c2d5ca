import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.join(ROOT, "tools") not in sys.path:     # workload builders shared with bench.py
    sys.path.insert(1, os.path.join(ROOT, "tools"))


# the library re-reads its measurement knobs (CSLAM_TOPK_WAVES, CSLAM_MQ_TAIL, ...) at every launch
# instead of once per process, so that a test can compare two settings in one process
os.environ.setdefault("CSLAM_KNOBS_LIVE", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        from cslam_amd import _lib
        return _lib.load().cslam_device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    """GPU tests fail loudly (never skip) when the HIP library or the device is missing."""
    from cslam_amd import _lib
    lib = _lib.load()
    assert lib.cslam_device_count() > 0, "no HIP device visible on a -m gpu run"
    return lib
