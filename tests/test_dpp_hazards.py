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


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_no_dpp_hazard_in_the_generated_isa(tmp_path):
    asm = tmp_path / "column_kernels.s"
    r = subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-Wno-unused-function",
                        "-Wno-unused-variable", os.path.join(ROOT, "mimsem_amd", "csrc", "column_kernels.hip"), "-o", str(asm)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    c = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_dpp_hazards.py"), str(asm)], capture_output=True, text=True)
    assert c.returncode == 0, c.stdout[-2000:]
    n = int(c.stdout.split()[0])
    assert n > 10000 and "0 hazard(s)" in c.stdout, c.stdout          # the DPP kernels are really in there
    # round 4: no register spill in a divergent "Flow" block ahead of its exec restore, no reload of a slot some path leaves unwritten
    # (scripts/check_spill_slots.py: the pattern behind the wrong, run-to-run different bands of k_s3_sweep<4, box>)
    c = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_spill_slots.py"), str(asm)], capture_output=True, text=True)
    assert c.returncode == 0 and " 0 finding(s)" in c.stdout, c.stdout[-3000:]
    assert int(c.stdout.strip().split("\n")[-1].split()[0]) > 100, c.stdout[-300:]
