"""Static checks of the generated gfx950 code (no GPU): the ring kernels issue their LDS table reads as inline asm and count
the s_waitcnt themselves, so nothing but this scan protects a register between a read's issue and the wait that covers it
(tools/check_isa.py checks every translation unit; here: the scanner itself and the unit that holds the R = 50 kernels)."""
import importlib.util
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("check_isa", os.path.join(ROOT, "tools", "check_isa.py"))
check_isa = importlib.util.module_from_spec(spec)
spec.loader.exec_module(check_isa)


def test_scanner_finds_a_use_before_the_wait():
    asm = """_ZN4test:
	ds_read_b64 v[4:5], v1 offset:8
	ds_read_b64 v[6:7], v1 offset:16
	s_waitcnt lgkmcnt(1)
	v_min_f32_e32 v8, v4, v5
	v_min_f32_e32 v9, v6, v7
	s_waitcnt lgkmcnt(0)
	v_min_f32_e32 v9, v6, v7
	s_endpgm
"""
    hz = check_isa.lds_hazards(asm)
    assert len(hz) == 1 and "v6" in hz[0] and "line 6" in hz[0]
    spill = asm.replace("\tv_min_f32_e32 v9, v6, v7\n\ts_waitcnt lgkmcnt(0)", "\tscratch_store_dword off, v7, off\n\ts_waitcnt lgkmcnt(0)")
    assert len(check_isa.lds_hazards(spill)) == 1
    overwrite = asm.replace("\tv_min_f32_e32 v9, v6, v7\n\ts_waitcnt lgkmcnt(0)", "\tv_mov_b32_e32 v6, 0\n\ts_waitcnt lgkmcnt(0)")
    assert len(check_isa.lds_hazards(overwrite)) == 1


def test_ring_unit_has_no_inflight_use_and_no_fused_lds_ops(tmp_path):
    from neilpy_amd.build import CSRC, FLAGS, hipcc
    out = str(tmp_path / "ring_f32_p2.s")
    cmd = [hipcc()] + [f for f in FLAGS if f != "-fPIC"] + ["-DPART=2", "-DSMRF_F64=0", "--offload-device-only", "-S",
                                                           os.path.join(CSRC, "ring_part.hip"), "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    text = open(out).read()
    assert text.count("ds_read_b64") > 1000                # radii 2, 10, ..., 58: the scan has something to look at
    assert check_isa.lds_hazards(text) == []
    assert "ds_read2" not in text and "ds_write2" not in text
    # the in-place instances of this unit (csrc/ring_inpl.inc: R = 42 with 3 row pairs per batch, R = 50 with 2, both built for
    # 3 waves per SIMD = 168 VGPRs) keep the whole ring in registers: no scratch
    import re
    kernels = dict(re.findall(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.S))
    for r, np_ in ((42, 3), (50, 2)):
        for dil in (0, 1):
            name = "_ZN4smrf11ring_kernelIfLi%dELb%dELi256ELi%dEEEv8DiskArgsIT_E" % (r, dil, np_)
            assert name in kernels, name
            assert int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", kernels[name]).group(1)) == 0, name
            assert int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", kernels[name]).group(1)) <= 168, name


def test_chain_unit_has_no_inflight_use_no_scratch_loops(tmp_path):
    """the chained / table-free kernels (csrc/morph_chain.h) read their neighbours' cells with the same asm reads and
    counted waits: no use of a register in flight, no fused LDS ops, and no scratch beyond a few loop-invariant registers"""
    import re
    from neilpy_amd.build import CSRC, FLAGS, hipcc
    out = str(tmp_path / "chain.s")
    cmd = [hipcc()] + [f for f in FLAGS if f != "-fPIC"] + ["--offload-device-only", "-S", os.path.join(CSRC, "chain.hip"), "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    text = open(out).read()
    assert text.count("ds_read_b64") > 300
    assert check_isa.lds_hazards(text) == []
    assert "ds_read2" not in text and "ds_write2" not in text
    kernels = re.findall(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.S)
    assert len(kernels) == 18                               # 11 fp32 patterns + three fp64 chains + 4 fp64 singles
    for name, body in kernels:
        assert int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body).group(1)) <= 16, name


def test_instruction_budget_of_a_large_disk_instance(tmp_path):
    """tools/isa_budget.py (profiles/r04_isa_budget.md): the in-place R = 39 erosion instance compiled with
    phase marks - the consume phase's min / max per 64-cell row is the decomposition's count (R - 1 ring updates + K - 1
    width steps + 2 completed rows, a few more for the long first step), the table build and the turn-back are what the
    note says they are, and the instance holds its ring in registers"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_budget
    R = 39
    text = isa_budget.compile_one(R, False, 0, [], str(tmp_path))
    info, counts = isa_budget.analyse(text, False)
    rows = 2 * info["np"]
    hw, wk = isa_budget.disk(R)
    K = len(wk)
    assert info["np"] == 3 and info["scratch"] == 0 and info["vgpr"] <= 168
    consume = counts["consume"]["minmax"] / rows
    assert R + K - 1 <= consume <= R + K + 3, consume            # 63.5 against R + K = 63
    build = (counts["build"]["minmax"] + counts.get("build:halo", {}).get("minmax", 0)) / rows
    assert 2.5 <= build <= 4.5, build                           # levels 1, 2 over 256 + 2R staged cells per 256 outputs
    assert abs(counts["turn"]["mov"] / rows - (R + 1) / rows) < 0.5   # R + 1 v_mov per batch
    total_valu = sum(c.get(k, 0) for c in counts.values() for k in ("minmax", "mov", "valu_other")) / rows
    assert total_valu <= R + K + 22, total_valu                 # 82.8 measured; the note's budget table


def test_fused_window_steps_remove_the_hazard_nops(tmp_path):
    """round 5 (profiles/r05_scalar_budget.md): hipcc pads every asm statement that reads a register written by an earlier asm
    statement with `s_nop 0` unless a real instruction lies between - 13-17 per 64-cell row in the large-disk kernels.  Where
    csrc/ring_fuse.inc switches the fused window steps + late ring slots on (R = 47: mode 3) they are gone, the min / max count
    is unchanged, and the instance still holds its ring in registers; with the mode forced off the nops are back."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_budget
    R = 47
    res = {}
    for tag, defs in (("table", []), ("off", ["-DSMRF_RING_FUSE_MODE(T,R,INPL)=0"])):
        text = isa_budget.compile_one(R, False, 0, defs, str(tmp_path))
        info, counts = isa_budget.analyse(text, False)
        rows = 2 * info["np"]
        res[tag] = dict(nop=sum(c.get("sc:s_nop", 0) for c in counts.values()) / rows,
                        minmax=sum(c.get("minmax", 0) for c in counts.values()) / rows, scratch=info["scratch"], vgpr=info["vgpr"])
    assert res["table"]["scratch"] == 0 and res["table"]["vgpr"] <= 168
    assert res["table"]["minmax"] == res["off"]["minmax"]
    assert res["off"]["nop"] >= 12 and res["table"]["nop"] <= 4, res
