"""The C-ABI library builds, loads on a CPU-only host and exports every symbol that
include/cslam_hip.h declares (no compute calls here)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "cslam_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cslam_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    from cslam_amd import _lib
    lib = _lib.load()
    syms = _declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/cslam_hip.h but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature in cslam_amd/_lib.py"
    assert sorted(_lib.SIGNATURES) == syms
    assert lib.cslam_abi_version() == 1


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under cslam_amd/ may reference it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "cslam_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
                assert "liboracle" not in src, f


def test_no_gpu_means_loud_failure():
    from cslam_amd import _lib
    lib = _lib.load()
    if lib.cslam_device_count() > 0:
        return  # GPU box: covered by the -m gpu tests
    import pytest
    from cslam_amd.nns_matching import NearestNeighborsMatching
    with pytest.raises(_lib.CslamHipError):
        NearestNeighborsMatching(dim=8)
