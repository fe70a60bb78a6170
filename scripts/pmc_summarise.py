#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter CSVs (FETCH_SIZE / WRITE_SIZE passes) into bytes per launch per kernel,
with the calibration factor measured on k_halo_pack's known byte count.  usage: pmc_summarise.py <dir> ..."""
import csv
import glob
import sys
from collections import defaultdict

rows = defaultdict(list)
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows[(r["Kernel_Name"].split("(")[0][-60:], r["Counter_Name"], int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
print("kernel | counter | grid | launches | mean value (KiB units per rocprof) | mean MB")
for (k, c, g), v in sorted(rows.items()):
    m = sum(v) / len(v)
    print(f"{k} | {c} | {g} | {len(v)} | {m:.1f} | {m * 1024 / 1e6:.2f}")
