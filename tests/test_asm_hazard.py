"""Static check of the built device assembly (cslam_amd/csrc/build/*.s, kept by `make`): no instruction
touches a register that an in-flight inline-asm load still owes - the cause of the GPU memory faults
(HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION) of experimental scan variants in rounds 2 and 3 and of a
stale-copy build in round 4; see tools/check_asm_load_hazard.py and DESIGN.md section 5, "The inline-asm load hazard".
`make` itself runs the check and fails on a finding; this test keeps the checker honest (its
self-tests) and fails - not skips - when the library was built without the assembly being kept."""
import glob
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _asm_files():
    return sorted(glob.glob(os.path.join(ROOT, "cslam_amd", "csrc", "build", "*-hip-amdgcn-amd-amdhsa-gfx950.s")))


def test_every_kernel_with_inline_asm_loads_is_clean():
    import check_asm_load_hazard as chk
    files = _asm_files()
    lib = os.path.join(ROOT, "cslam_amd", "libcslam_hip.so")
    if not files and not os.path.exists(lib):
        pytest.skip("library not built here")
    assert files, "libcslam_hip.so exists but cslam_amd/csrc/build/*.s does not: build with `make -C cslam_amd/csrc`"
    names = {os.path.basename(f).split("-hip-")[0] for f in files}
    assert {"knn", "scan_mq", "mac", "shard", "geom", "comm"} <= names, names
    checked = 0
    for f in files:
        findings = chk.check(f, verbose=False)
        assert not findings, (f, findings[:5])
        checked += sum(1 for k, items in chk.parse_kernels(f).items() if chk.check_kernel(k, items) is not None)
    assert checked >= 10   # the k_scan_dma family carries the asm loads


def _kernel(tmp_path, body, name="_Z10k_scan_dmaTest"):
    p = tmp_path / "k.s"
    p.write_text("\n".join([f"{name}: ; @{name}"] + body + ["\ts_endpgm", ".Lfunc_end0:", ""]))
    return str(p)


ASM = ["\t;;#ASMSTART", "\tglobal_load_dwordx4 v[6:9], v[190:191], off offset:5*64", "\t;;#ASMEND"]


def test_the_checker_sees_recycling_behind_a_skippable_wait(tmp_path):
    """The faulting pattern of round 3: the covering wait jumped over, v[6:7] recycled as an address."""
    import check_asm_load_hazard as chk
    f = _kernel(tmp_path, ASM + ["\ts_cbranch_vccnz .LBB0_1", "\ts_waitcnt vmcnt(0)", ".LBB0_1:",
                                 "\tv_lshl_add_u64 v[6:7], s[2:3], 0, v[194:195]",
                                 "\tglobal_load_lds_dwordx4 v[6:7], off"])
    assert len(chk.check(f, verbose=False)) == 2


def test_the_checker_sees_a_copy_of_an_owed_register(tmp_path):
    """Round 4's form: the compiler parks the not-yet-written fragment in an AccVGPR (a READ)."""
    import check_asm_load_hazard as chk
    f = _kernel(tmp_path, ASM + ["\tv_accvgpr_write_b32 a84, v6", "\ts_waitcnt vmcnt(0)", "\tv_mov_b32_e32 v1, v7"])
    found = chk.check(f, verbose=False)
    assert len(found) == 1 and "v_accvgpr_write" in found[0][2]
    f = _kernel(tmp_path, ASM + ["\tscratch_store_dwordx4 off, v[6:9], off offset:16", "\ts_waitcnt vmcnt(0)"])
    assert len(chk.check(f, verbose=False)) == 1


def test_counted_waits(tmp_path):
    import check_asm_load_hazard as chk
    younger = ["\tglobal_load_lds_dwordx4 v[20:21], off"] * 3
    # three younger operations: vmcnt(3) covers the load, vmcnt(4) does not
    f = _kernel(tmp_path, ASM + younger + ["\ts_waitcnt vmcnt(3)", "\tv_mov_b32_e32 v1, v6"])
    assert chk.check(f, verbose=False) == []
    f = _kernel(tmp_path, ASM + younger + ["\ts_waitcnt vmcnt(4)", "\tv_mov_b32_e32 v1, v6"])
    assert len(chk.check(f, verbose=False)) == 1
    f = _kernel(tmp_path, ASM + younger + ["\ts_waitcnt lgkmcnt(0)", "\tv_mov_b32_e32 v1, v6"])
    assert len(chk.check(f, verbose=False)) == 1


def test_loop_carried_loads_have_no_line_window(tmp_path):
    """A load issued at the bottom of a loop body is still owed at its top, however long the body."""
    import check_asm_load_hazard as chk
    filler = ["\tv_add_f32_e32 v100, v101, v102"] * 2000
    f = _kernel(tmp_path, [".LBB0_1:", "\tv_mov_b32_e32 v50, v6", "\ts_waitcnt vmcnt(0)"] + filler + ASM +
                ["\ts_cbranch_scc1 .LBB0_1", "\ts_waitcnt vmcnt(0)"])
    found = chk.check(f, verbose=False)
    assert len(found) == 1 and found[0][2].startswith("v_mov_b32_e32 v50")


def test_run_time_selector_markers(tmp_path):
    import check_asm_load_hazard as chk
    sel = lambda end: ["\t;;#ASMSTART", "\t; cslam.wait_dyn.begin", "\t;;#ASMEND", "\ts_cmp_eq_u32 s0, 0",
                       "\ts_cbranch_scc1 .LBB0_2", "\ts_waitcnt vmcnt(48)", "\ts_branch .LBB0_3", ".LBB0_2:",
                       "\ts_waitcnt vmcnt(0)", ".LBB0_3:", "\t;;#ASMSTART", "\t; " + end, "\t;;#ASMEND"]
    use = ["\tv_mov_b32_e32 v1, v6"]
    assert len(chk.check(_kernel(tmp_path, ASM + sel("cslam.wait_dyn.end") + use), verbose=False)) == 1
    assert chk.check(_kernel(tmp_path, ASM + sel("cslam.wait_dyn.end covers-asm-loads") + use), verbose=False) == []


def test_indirect_control_flow_is_a_finding(tmp_path):
    import check_asm_load_hazard as chk
    f = _kernel(tmp_path, ASM + ["\ts_setpc_b64 s[4:5]", "\ts_waitcnt vmcnt(0)"])
    assert len(chk.check(f, verbose=False)) == 1
