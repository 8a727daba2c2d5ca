import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.join(ROOT, "tools") not in sys.path:     # workload builders shared with bench.py
    sys.path.insert(1, os.path.join(ROOT, "tools"))


# The suite runs on the production path: measurement knobs (CSLAM_TOPK_WAVES, CSLAM_MQ_TAIL, ...) are read
# once and cached.  A test that compares two settings asks for the `knobs` fixture below.


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


@pytest.fixture
def knobs(monkeypatch):
    """`knobs.set("CSLAM_TOPK_WAVES", "0")`: change a measurement knob for the rest of THIS test; the
    library re-reads its knobs now and again when the environment is restored (cslam_knobs_reload)."""
    from cslam_amd import _lib
    lib = _lib.load()

    class Knobs(object):
        def __init__(self):
            self.touched = set()     # per fixture instance, i.e. per test

        def set(self, name, value):
            monkeypatch.setenv(name, str(value))
            self.touched.add(name)
            lib.cslam_knobs_reload()

        def only(self, env):
            """Exactly these knobs (of the ones this fixture has touched) are set, the others back at their defaults."""
            for name in list(self.touched):
                monkeypatch.delenv(name, raising=False)
            for name, value in env.items():
                monkeypatch.setenv(name, str(value))
                self.touched.add(name)
            lib.cslam_knobs_reload()

    yield Knobs()
    monkeypatch.undo()
    lib.cslam_knobs_reload()
