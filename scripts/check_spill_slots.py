#!/usr/bin/env python3
"""Static checks of the VGPR spills in hipcc's gfx950 assembly (round 4: how the wrong, run-to-run different bands of k_s3_sweep<4, box> were
found).  A spilled VGPR is saved lane by lane under the exec mask of the moment.  hipcc (ROCm 7.2) can place a spill store at the top of a
structuriser "Flow" block -- the join of a divergent if / else, entered by s_cbranch_execz with only the then-lanes enabled -- AHEAD of the
s_or_saveexec that re-enables the else-lanes, and overwrite the register for all lanes right after it: the else-lanes' value is in neither
place, a later reload hands them whatever the scratch slot held before.  Two rules, per kernel:

  A. no "Folded Spill" scratch_store between a label that an s_cbranch_execz targets and the next instruction that writes exec;
  B. every scratch_load is preceded, on EVERY path from the entry, by a scratch_store of the same bytes (forward must-analysis over the
     basic blocks; long branches through s_getpc / s_add / s_setpc are resolved from their label expressions; exec masks are not
     modelled: a store under a partial mask counts, so B alone does not prove a spill complete -- it did flag the faulty region).

The source-level cure is to keep divergent control flow out of kernels that spill heavily (k_s3_sweep: everything its branches depend on
is wave-uniform).  usage: check_spill_slots.py file.s [kernel-name-substring]     (exit code 1 on a finding)"""
import re
import sys


def kernel_text(path, sub):
    lines = open(path).read().split("\n")
    st = [i for i, l in enumerate(lines) if sub in l and re.match(r"^[_A-Za-z0-9.$]+:", l) and not l.startswith(".L")]
    out = []
    for s in st:
        en = next((i for i in range(s, len(lines)) if lines[i].startswith(".Lfunc_end")), None)
        if en is not None and lines[s + 1:en]:
            out.append((lines[s].split(":")[0], lines[s:en]))
    return out


def analyse(name, k, quiet=False):
    # basic blocks
    blocks, cur, label_of = [], None, {}
    def new(lbl=None):
        nonlocal cur
        cur = {"labels": [], "ins": [], "succ": [], "fall": True}
        blocks.append(cur)
    new()
    pend_target = None
    for ln, raw in enumerate(k):
        s = raw.split(";")[0].strip()
        if not s:
            continue
        m = re.match(r"^(\.L[A-Za-z0-9_]+):", s)
        if m:
            if cur["ins"]:
                new()
            cur["labels"].append(m.group(1)); label_of[m.group(1)] = cur
            continue
        if s.startswith("."):
            continue
        cur["ins"].append((ln, s + ("  ;Folded" if "Folded" in raw else "")))
        m = re.search(r"\((\.LBB\d+_\d+)-\.Lpost_getpc\d+\)&", s)
        if m:
            pend_target = m.group(1)
        op = s.split()[0]
        if op.startswith("s_cbranch"):
            cur["succ"].append(s.split()[1]); new()
        elif op == "s_branch":
            cur["succ"].append(s.split()[1]); cur["fall"] = False; new()
        elif op == "s_setpc_b64":
            cur["succ"].append(pend_target); cur["fall"] = False; new()
        elif op == "s_endpgm":
            cur["fall"] = False; new()
    for i, b in enumerate(blocks):
        b["s"] = [label_of[t] for t in b["succ"] if t in label_of]
        if b["fall"] and i + 1 < len(blocks):
            b["s"].append(blocks[i + 1])
        b["id"] = i
    preds = {b["id"]: [] for b in blocks}
    for b in blocks:
        for t in b["s"]:
            preds[t["id"]].append(b["id"])
    def slots(s):
        m = re.match(r"scratch_(store|load)_dword(x\d)?\s+(.*)", s)
        if not m or "Folded" not in s:                   # (register spills only: private arrays are the program's own business)
            return None
        off = re.search(r"offset:(\d+)", m.group(3)); off = int(off.group(1)) if off else 0
        w = {None: 1, "x2": 2, "x3": 3, "x4": 4}[m.group(2)]
        return m.group(1), {off + 4 * q for q in range(w)}
    ALL = set()
    for b in blocks:
        for _, s in b["ins"]:
            r = slots(s)
            if r:
                ALL |= r[1]
    IN = {b["id"]: set(ALL) for b in blocks}; OUT = {b["id"]: set(ALL) for b in blocks}
    IN[0] = set()
    changed = True
    while changed:
        changed = False
        for b in blocks:
            i = b["id"]
            if i:
                ps = [OUT[p] for p in preds[i]]
                inn = set.intersection(*ps) if ps else set(ALL)
            else:
                inn = set()
            o = set(inn)
            for _, s in b["ins"]:
                r = slots(s)
                if r and r[0] == "store":
                    o |= r[1]
            if inn != IN[i] or o != OUT[i]:
                IN[i], OUT[i] = inn, o; changed = True
    bad = 0
    for b in blocks:
        have = set(IN[b["id"]])
        if not preds[b["id"]] and b["id"]:
            continue                                     # unreachable filler
        for ln, s in b["ins"]:
            r = slots(s)
            if not r:
                continue
            if r[0] == "store":
                have |= r[1]
            elif not r[1] <= have:
                print(f"  {name}: line {ln}: {s}   <- slot(s) {sorted(r[1] - have)} not stored on every path")
                bad += 1
    if not quiet or bad:
        print(f"{name}: {len(blocks)} blocks, {len(ALL)} spill dwords, {bad} load(s) of possibly unwritten slots")
    return bad


def flow_block_spills(name, k):
    targets = set()
    for raw in k:
        s = raw.split(";")[0].strip()
        if s.startswith("s_cbranch_execz"):
            targets.add(s.split()[1])
    bad, cur = 0, None
    for ln, raw in enumerate(k):
        s = raw.split(";")[0].strip()
        m = re.match(r"^(\.L[A-Za-z0-9_]+):", s)
        if m:
            cur = m.group(1) if m.group(1) in targets else None
            continue
        if not cur or not s:
            continue
        if (re.match(r"s_\w+\s", s) and re.search(r"\bexec\b", s.split(None, 1)[1].split(",")[0])) or "saveexec" in s \
                or s.startswith(("s_cbranch", "s_branch", "s_setpc", "s_endpgm")):
            cur = None
            continue
        if s.startswith("scratch_store") and "Spill" in raw:
            print(f"  {name}: line {ln}: {s}   <- spill store in the flow block {cur} ahead of its exec restore")
            bad += 1
    return bad


if __name__ == "__main__":
    tot = nk = 0
    for name, k in kernel_text(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else ""):
        nk += 1
        tot += flow_block_spills(name, k)
        tot += analyse(name, k, quiet=len(sys.argv) <= 2)
    print(f"{nk} kernels checked, {tot} finding(s)")
    sys.exit(1 if tot else 0)
