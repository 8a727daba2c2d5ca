"""Static check of the built device assembly (cslam_amd/csrc/build/*.s, a by-product of `make`): no
compiler-generated instruction recycles a register that an in-flight inline-asm load still owes.
That is the cause found in round 3 for the GPU memory faults (HSA_STATUS_ERROR_MEMORY_APERTURE_
VIOLATION) of experimental scan variants in rounds 2 and 3 - see tools/check_asm_load_hazard.py and
DESIGN.md section 5, item 10.  The shipped kernels are clean; this keeps them so."""
import glob
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _asm(name):
    hits = glob.glob(os.path.join(ROOT, "cslam_amd", "csrc", "build", f"{name}-hip-amdgcn-amd-amdhsa-gfx950.s"))
    if not hits:
        pytest.skip("device assembly not kept (build with `make -C cslam_amd/csrc`)")
    return hits[0]


@pytest.mark.parametrize("unit,kernels", [("knn", "k_scan_dma"), ("scan_mq", "k_scan_mq")])
def test_no_register_of_an_inflight_asm_load_is_recycled(unit, kernels):
    import check_asm_load_hazard as chk
    findings = chk.check(_asm(unit), kernels, verbose=False)
    assert not findings, findings[:5]


def test_the_checker_sees_the_hazard(tmp_path):
    """The faulting pattern itself, reduced: an asm load of v[6:9] in flight, its covering wait jumped
    over, v[6:7] recycled as an address pair and used by an LDS-DMA."""
    import check_asm_load_hazard as chk
    p = tmp_path / "bad.s"
    p.write_text("\n".join([
        "_Z10k_scan_dmaTest: ; @_Z10k_scan_dmaTest",
        "\t;;#ASMSTART", "\tglobal_load_dwordx4 v[6:9], v[190:191], off offset:5*64", "\t;;#ASMEND",
        "\ts_cbranch_vccnz .LBB0_1", "\ts_waitcnt vmcnt(0)", ".LBB0_1:",
        "\tv_lshl_add_u64 v[6:7], s[2:3], 0, v[194:195]",
        "\tglobal_load_lds_dwordx4 v[6:7], off", "\ts_endpgm", ""]))
    assert len(chk.check(str(p), "k_scan_dma", verbose=False)) == 2
