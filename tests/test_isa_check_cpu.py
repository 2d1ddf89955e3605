"""CPU tests of the assembly check the build runs on every device translation unit (tools/check_isa_exec_prologue.py,
DESIGN.md "Build-variant fragility: root cause"): the defect pattern is recognised and repaired, regular code is left alone, and
the reports the build left next to the shipped objects show nothing unrepaired."""
import glob
import importlib.util
import os

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
spec = importlib.util.spec_from_file_location("check_isa", os.path.join(ROOT, "tools", "check_isa_exec_prologue.py"))
chk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(chk)

# the shape of the defect as the ROCm 7.2 compiler emitted it in qp_solve_kernel<5,0> (-O2): the join block of a divergent
# region starts with the allocator's VGPR -> AGPR copy, the mask of the region that just ended is restored only after it
BAD = """
_Z6kernelv:
	s_mov_b64 s[0:1], exec
	v_readlane_b32 s18, v253, 41
	v_readlane_b32 s19, v253, 42
	s_and_b64 s[18:19], s[0:1], s[18:19]
	s_mov_b64 exec, s[18:19]
	s_cbranch_execz .LBB0_3
; %bb.1:
	v_mul_f64 v[6:7], v[2:3], -v[6:7]
.LBB0_2:
	s_or_b64 exec, exec, s[18:19]
.LBB0_3:
	v_accvgpr_write_b32 a52, v48
	s_or_b64 exec, exec, s[0:1]
	v_mul_f64 v[2:3], v[4:5], -v[16:17]
	s_endpgm
"""
# regular code: (a) restore first, copy after it; (b) the body of an else-branch ends with the restore of ITS region, opened by
# the block before it; (c) SGPR reloads by v_readlane (they ignore EXEC) ahead of the restore
GOOD = """
_Z6kernelv:
	s_and_saveexec_b64 s[0:1], vcc
	s_cbranch_execz .LBB0_3
; %bb.1:
	v_mul_f64 v[6:7], v[2:3], -v[6:7]
.LBB0_3:
	s_or_b64 exec, exec, s[0:1]
	v_accvgpr_write_b32 a52, v48
	s_andn2_saveexec_b64 s[12:13], s[12:13]
	s_cbranch_execz .LBB0_9
.LBB0_5:
	v_cmp_ne_u32_e32 vcc, 4, v1
	v_mov_b64_e32 v[26:27], v[10:11]
	s_or_b64 exec, exec, s[12:13]
	s_and_saveexec_b64 s[2:3], vcc
	s_cbranch_execz .LBB0_9
; %bb.6:
	v_accvgpr_write_b32 a1, v2
.LBB0_9:
	v_readlane_b32 s2, v254, 26
	v_readlane_b32 s3, v254, 27
	s_or_b64 exec, exec, s[2:3]
	v_accvgpr_read_b32 v2, a1
	s_endpgm
"""


def test_defect_is_flagged_and_repaired(tmp_path):
    p = tmp_path / "bad.s"
    p.write_text(BAD)
    hits = chk.scan(str(p))
    assert len(hits) == 1 and hits[0][1] == ".LBB0_3" and "v_accvgpr_write_b32 a52, v48" in hits[0][4][0][1] and chk.safe_to_move(hits[0])
    chk.repair(str(p), hits)
    fixed = [ln.strip() for ln in p.read_text().split("\n") if ln.strip()]
    i = fixed.index(".LBB0_3:")
    assert fixed[i + 1] == "s_or_b64 exec, exec, s[0:1]" and fixed[i + 2] == "v_accvgpr_write_b32 a52, v48"
    assert chk.scan(str(p)) == [] and len(fixed) == len([ln for ln in BAD.split("\n") if ln.strip()])


def test_regular_code_is_left_alone(tmp_path):
    p = tmp_path / "good.s"
    p.write_text(GOOD)
    assert chk.scan(str(p)) == []


def test_spill_code_counts_as_allocator_made(tmp_path):
    p = tmp_path / "spill.s"
    p.write_text(BAD.replace("v_accvgpr_write_b32 a52, v48", "scratch_load_dword v2, off, off offset:64"))
    assert len(chk.scan(str(p))) == 1


def _run_fix(path):
    import subprocess, sys
    return subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_isa_exec_prologue.py"), "--fix", str(path)], capture_output=True, text=True)


def test_only_the_proven_pattern_is_repaired_automatically(tmp_path):
    """--fix moves plain v_accvgpr_write copies that sit directly ahead of the restore and nothing else.  A spill reload (its
    s_waitcnt would be crossed), an s_waitcnt in the prologue, another vector instruction in the prologue (it could consume the
    moved copy's result, or run under the stale mask itself) or two copies into the same AGPR stop the build with exit code 2 and
    leave the file untouched."""
    ok = tmp_path / "ok.s"
    ok.write_text(BAD)
    r = _run_fix(ok)
    assert r.returncode == 0 and "moved behind it" in r.stdout and chk.scan(str(ok)) == []
    two = tmp_path / "two.s"
    two.write_text(BAD.replace("\tv_accvgpr_write_b32 a52, v48\n", "\tv_accvgpr_write_b32 a52, v48\n\tv_accvgpr_write_b32 a53, v49\n"))
    assert _run_fix(two).returncode == 0 and chk.scan(str(two)) == []
    cases = {
        "scratch_load": BAD.replace("v_accvgpr_write_b32 a52, v48", "scratch_load_dword v2, off, off offset:64"),
        "waitcnt": BAD.replace("\tv_accvgpr_write_b32 a52, v48\n", "\ts_waitcnt vmcnt(0)\n\tv_accvgpr_write_b32 a52, v48\n"),
        "consumer": BAD.replace("\tv_accvgpr_write_b32 a52, v48\n", "\tv_accvgpr_write_b32 a52, v48\n\tv_accvgpr_read_b32 v7, a52\n"),
        "other_valu": BAD.replace("\tv_accvgpr_write_b32 a52, v48\n", "\tv_accvgpr_write_b32 a52, v48\n\tv_add_f64 v[8:9], v[8:9], v[10:11]\n"),
        "same_dest": BAD.replace("\tv_accvgpr_write_b32 a52, v48\n", "\tv_accvgpr_write_b32 a52, v48\n\tv_accvgpr_write_b32 a52, v49\n"),
    }
    for name, text in cases.items():
        p = tmp_path / (name + ".s")
        p.write_text(text)
        r = _run_fix(p)
        assert r.returncode == 2 and "NOT the proven pattern" in r.stdout, (name, r.returncode, r.stdout)
        assert p.read_text() == text, name                      # untouched


def test_reports_of_the_built_library():
    """every object of the shipped library went through the check; a repaired unit says so, nothing is left flagged"""
    import pytest
    if not os.path.exists(os.path.join(ROOT, "fsae-mpc_amd", "lib", "libfsaempc.so")):
        pytest.skip("library not built in this checkout")
    logs = sorted(glob.glob(os.path.join(ROOT, "fsae-mpc_amd", "lib", "*.isa.log")))
    names = {os.path.basename(f)[:-8] for f in logs}
    need = {"capi", "ltv_build", "plant", "reference", "qp_solver_tu0", "qp_solver_tu1", "qp_solver_tu2", "qp_solver_tu3", "qp_solver_tu4",
            "qp_wg_1_5", "qp_wg_6_6", "qp_wg_7_7", "qp_wg_8_8", "qp_wg_9_9", "qp_wg_10_10", "qp_wg_11_11", "qp_wg_12_12"}
    assert need <= names, sorted(need - names)
    for f in logs:
        head = open(f).readline()
        assert (" 0 allocator-made" in head) or ("moved behind it" in head), (f, head)
