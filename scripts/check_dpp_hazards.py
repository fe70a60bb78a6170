#!/usr/bin/env python3
"""Static check of the hand-written DPP instructions in the gfx950 ISA of mimsem_amd/csrc/column_kernels.hip.

mimsem_amd/csrc/column_dpp.inc issues v_fmac_f64_dpp ... row_newbcast:m from inline asm.  The hardware needs TWO wait states
between a VALU write of a VGPR and a DPP read of it (and five after a VALU write of EXEC); hipcc's hazard recogniser inserts
them for instructions it knows, but inline asm is opaque to it.  The source keeps the rule by construction (every DPP source
goes through an `s_nop 1` asm that owns its registers), yet the register allocator is free to put a copy -- e.g. a
v_accvgpr_read_b32 when a kernel spills into the accumulator half of the register file -- between that s_nop and the DPP
instruction.  This script re-checks the rule on the ISA the compiler actually produced:

    for every v_fmac_f64_dpp / v_mov_b64_dpp: no VALU instruction in the two preceding wait states writes one of the
    DPP-source registers, and no v_cmpx (VALU write of EXEC) in the five preceding wait states.

usage: check_dpp_hazards.py file.s [...]     (exit code 1 on a violation; hipcc -S --cuda-device-only writes the .s)
"""
import re
import sys

REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs(tok):
    m = REG.search(tok)
    if not m:
        return set()
    if m.group(1) is not None:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return {int(m.group(3))}


def check(path):
    bad = 0
    ndpp = 0
    kernel = "?"
    window = []          # (wait_states, written vgprs, is_cmpx, text) of the preceding instructions, newest last
    for ln, line in enumerate(open(path), 1):
        code = line.split(";")[0].strip()
        if not code or code.startswith(".") or code.startswith("//"):
            continue
        if code.endswith(":"):
            if not code.startswith(".L") and not code.startswith("BB"):
                kernel = code[:-1]
            if not code.startswith(".L"):
                window = []                     # function entry: nothing precedes
            continue
        parts = code.split(None, 1)
        op = parts[0]
        ops = [t.strip() for t in parts[1].split(",")] if len(parts) > 1 else []
        if op in ("v_fmac_f64_dpp", "v_mov_b64_dpp"):
            ndpp += 1
            src0 = regs(ops[1])
            ws = 0
            for w_states, written, is_cmpx, text in reversed(window):
                if ws < 2 and written & src0:
                    print(f"{path}:{ln}: [{kernel}] DPP source {ops[1]} written {ws} wait state(s) earlier by: {text}")
                    bad += 1
                if ws < 5 and is_cmpx:
                    print(f"{path}:{ln}: [{kernel}] v_cmpx {ws} wait state(s) before a DPP instruction: {text}")
                    bad += 1
                ws += w_states
                if ws >= 5:
                    break
        # record this instruction
        if op == "s_nop":
            window.append((int(ops[0], 0) + 1, set(), False, code))
        else:
            written = set()
            if op.startswith("v_") and ops and not op.startswith("v_cmp") and not op.startswith("v_accvgpr_write") \
                    and not op.startswith("v_readlane") and not op.startswith("v_readfirstlane"):
                written = regs(ops[0]) if ops[0].lstrip().startswith("v") else set()
            window.append((1, written, op.startswith("v_cmpx"), code))
        if len(window) > 8:
            window.pop(0)
    return ndpp, bad


if __name__ == "__main__":
    total = fails = 0
    for p in sys.argv[1:]:
        n, b = check(p)
        total += n; fails += b
    print(f"{total} DPP f64 instructions checked, {fails} hazard(s)")
    sys.exit(1 if fails else 0)
