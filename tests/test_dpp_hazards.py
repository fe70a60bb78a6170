"""The hand-written `v_fmac_f64_dpp ... row_newbcast` instructions of the column kernels (mimsem_amd/csrc/column_dpp.inc) rely on a
two-wait-state rule hipcc cannot see through inline asm.  This compiles column_kernels.hip to gfx950 ISA (device side only, no GPU
needed: ~25 s) and re-checks the rule on what the compiler actually produced (scripts/check_dpp_hazards.py; a compiler or flag change
that slips a copy between the s_nop and the DPP read fails here instead of corrupting column solves silently).
The same ISA is checked for register spills hipcc placed where the exec mask drops lanes (scripts/check_spill_slots.py, round 4).
Limit of the check, stated: its look-back window restarts at labels, so a hazard across a loop back-edge is not seen; the source keeps
every DPP operand behind an `s_nop 1` in the SAME basic block for that reason."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def column_asm(tmp_path_factory):
    """column_kernels.hip compiled ONCE to gfx950 ISA (device side only: ~50 s) for every check of this module"""
    asm = tmp_path_factory.mktemp("isa") / "column_kernels.s"
    r = subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-Wno-unused-function",
                        "-Wno-unused-variable", os.path.join(ROOT, "mimsem_amd", "csrc", "column_kernels.hip"), "-o", str(asm)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return asm


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_no_dpp_hazard_in_the_generated_isa(column_asm):
    asm = column_asm
    c = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_dpp_hazards.py"), str(asm)], capture_output=True, text=True)
    assert c.returncode == 0, c.stdout[-2000:]
    n = int(c.stdout.split()[0])
    assert n > 10000 and "0 hazard(s)" in c.stdout, c.stdout          # the DPP kernels are really in there
    # round 4: no register spill in a divergent "Flow" block ahead of its exec restore, no reload of a slot some path leaves unwritten
    # (scripts/check_spill_slots.py: the pattern behind the wrong, run-to-run different bands of k_s3_sweep<4, box>)
    c = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_spill_slots.py"), str(asm)], capture_output=True, text=True)
    assert c.returncode == 0 and " 0 finding(s)" in c.stdout, c.stdout[-3000:]
    assert int(c.stdout.strip().split("\n")[-1].split()[0]) > 100, c.stdout[-300:]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_order4_walk_of_schur_3_runs_without_scratch(column_asm):
    """Round 5: k_s3_sweep<4> on the half-row block layout (dpp::RowsH: 16 VGPRs per block) must fit the register file -- round 4's 16-lane
    form ran with ~700 spilled dwords per lane (5.9 GB of scratch traffic per launch, and the spill pattern behind its wrong bands).
    From the code-object metadata: no scratch memory at all for both flavours of the half-row kernel; the order-3 walks stay spill-free."""
    import re
    s = column_asm.read_text()
    md = s[s.index("amdgpu_metadata"):]
    seen = {}
    for e in md.split("  - .agpr_count")[1:]:
        name = re.search(r"\.name:\s+(\S+)", e).group(1)
        if "k_s3_sweepILi" in name:
            g = lambda k: int(re.search(k + r":\s+(\d+)", e).group(1))
            seen[name] = (g(r"\.vgpr_spill_count"), g(r"\.private_segment_fixed_size"), g(r"\.vgpr_count"))
    half = {k: v for k, v in seen.items() if re.search(r"ILi4ELb[01]ELb1E", k)}       # k_s3_sweep<4, BOX, HALF = true>
    assert len(half) == 2, sorted(seen)
    for k, (spill, scratch, vgpr) in half.items():
        assert scratch == 0 and spill <= 8 and vgpr <= 512, (k, spill, scratch, vgpr)       # (a handful of "spills" into AGPRs cost no memory traffic)
    for k, (spill, scratch, vgpr) in seen.items():
        if "ILi3E" in k:
            assert scratch == 0, (k, spill, scratch)


def test_penta_solve_on_lane_major_bands_keeps_its_blocks_in_registers(column_asm):
    """Round 5: with run-time band strides every element of a block load carries its own 64-bit address; at order 4 those address registers
    pushed the two-sided block-pentadiagonal solve to 577 spilled dwords per lane (order 3: 114).  On the lane-major bands the fused path
    writes, the strides are compile-time constants (k_penta_dpp<N2, TWO, LM = true>): no scratch at order 3, and at order 4 only the few
    blocks parked across the refinement's outer loop."""
    import re
    s = column_asm.read_text()
    md = s[s.index("amdgpu_metadata"):]
    seen = {}
    for e in md.split("  - .agpr_count")[1:]:
        name = re.search(r"\.name:\s+(\S+)", e).group(1)
        m = re.search(r"k_penta_dppILi(\d+)ELb1ELb1E", name)
        if m:
            seen[int(m.group(1))] = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", e).group(1))
    assert set(seen) == {1, 4, 9, 16}, seen
    assert seen[1] == seen[4] == seen[9] == 0 and seen[16] <= 1280, seen


def test_spill_checker_sees_the_pattern_it_was_written_for(tmp_path):
    """scripts/check_spill_slots.py on two hand-written kernels: a spill store at the top of a structuriser "Flow" block (entered by
    s_cbranch_execz with the else-lanes still disabled) ahead of the s_or_saveexec that re-enables them -- the code hipcc produced in
    k_s3_sweep<4, box> -- is reported; the same store behind the exec restore is not"""
    bad = """
k_bad:
	s_and_saveexec_b64 s[0:1], vcc
	s_xor_b64 s[0:1], exec, s[0:1]
	s_cbranch_execz .LBB0_2
; %bb.1:
	v_mov_b32_e32 v2, 1
.LBB0_2:
	v_readlane_b32 s4, v254, 0
	scratch_store_dwordx2 off, v[88:89], off offset:8 ; 8-byte Folded Spill
	s_or_saveexec_b64 s[0:1], s[0:1]
	v_mov_b32_e32 v88, v2
	s_xor_b64 exec, exec, s[0:1]
	s_or_b64 exec, exec, s[0:1]
	scratch_load_dwordx2 v[0:1], off, off offset:8 ; 8-byte Folded Reload
	s_endpgm
.Lfunc_end0:
"""
    good = bad.replace("k_bad", "k_good").replace("""	scratch_store_dwordx2 off, v[88:89], off offset:8 ; 8-byte Folded Spill
	s_or_saveexec_b64 s[0:1], s[0:1]""", """	s_or_saveexec_b64 s[0:1], s[0:1]
	scratch_store_dwordx2 off, v[88:89], off offset:8 ; 8-byte Folded Spill""").replace("LBB0", "LBB1").replace("func_end0", "func_end1")
    script = os.path.join(ROOT, "scripts", "check_spill_slots.py")
    for name, text, rc in (("bad.s", bad, 1), ("good.s", good, 0)):
        f = tmp_path / name
        f.write_text(text)
        c = subprocess.run([sys.executable, script, str(f)], capture_output=True, text=True)
        assert c.returncode == rc, (name, c.stdout)
        assert ("ahead of its exec restore" in c.stdout) == (rc == 1), c.stdout
