#!/usr/bin/env python3
"""per-kernel summary of the rocprofv3 passes written by scripts/prof_umat_pmc.sh (gpurun_out/pmc_umat/<mode>_{s,a,b,f,w})"""
import collections
import csv
import glob
import sys
O = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_umat"
KEYS = ("k_apply_wave", "k_elem_apply", "k_gather")
for mode in ("wave", "twopass"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for part in ("a", "b", "f", "w"):
        for f in glob.glob(f"{O}/{mode}_{part}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if any(t in k for t in KEYS):
                    acc[(k.split("(")[0][-34:], int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = collections.defaultdict(list)
    for f in glob.glob(f"{O}/{mode}_s/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if any(t in k for t in KEYS):
                dur[(k.split("(")[0][-34:], int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    if not acc:
        continue
    print(f"== {mode}: per-launch averages over scripts/prof_umat.py ==")
    for k, cs in sorted(acc.items()):
        m = {n: sum(v) / len(v) for n, v in cs.items()}
        wc = m.get("SQ_WAVE_CYCLES", 0.0) or 1.0
        w = max(m.get("SQ_WAVES", 1), 1)
        d = dur.get(k, [0.0]); d = sorted(d)[len(d) // 2]
        print(f"{k[0]:34s} grid {k[1]:9d} {d:7.1f} us  waves {w:8.0f}  per wave: VALU {m.get('SQ_INSTS_VALU', 0) / w:6.0f} LDS {m.get('SQ_INSTS_LDS', 0) / w:5.0f} "
              f"VMEM {m.get('SQ_INSTS_VMEM', 0) / w:5.0f} SALU {m.get('SQ_INSTS_SALU', 0) / w:5.0f} cycles {wc / w:8.0f}\n"
              f"      of wave-cycles: VALU-active {100 * m.get('SQ_ACTIVE_INST_VALU', 0) / wc:5.1f}% LDS-active {100 * m.get('SQ_ACTIVE_INST_LDS', 0) / wc:5.1f}% "
              f"VMEM-active {100 * m.get('SQ_ACTIVE_INST_VMEM', 0) / wc:5.1f}% issue-stall {100 * m.get('SQ_WAIT_INST_ANY', 0) / wc:5.1f}% "
              f"LDS-stall {100 * m.get('SQ_WAIT_INST_LDS', 0) / wc:5.1f}% parked {100 * m.get('SQ_WAIT_ANY', 0) / wc:5.1f}% | busy-cycles {m.get('SQ_BUSY_CYCLES', 0):.3g} "
              f"bank-conflict {m.get('SQ_LDS_BANK_CONFLICT', 0):.3g}/{m.get('SQ_LDS_IDX_ACTIVE', 0):.3g} | FETCH_SIZE {m.get('FETCH_SIZE', 0):.5g} WRITE_SIZE {m.get('WRITE_SIZE', 0):.5g}")
