#!/bin/bash
# phase timing of k_band_lu_wave (csrc/column_pivot.inc): the kernel leaves a column after phase MIMSEM_BLU_STOP = 1 (factorisation),
# 2 (+ back substitution), 3 (+ residual), 4 (+ replay of the elimination on the residual); 0 = the whole solve
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
for st in 1 2 3 4 0; do
  MIMSEM_BLU_STOP=$st rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_blu$st -o r -- python3 $R/scripts/prof_column.py > /dev/null 2>&1
  python3 - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/prof_blu$st/r_kernel_stats.csv")))
for r in rows:
    if "band_lu" in r["Name"]: print("stop=$st", r["Name"][:60], r["Calls"], "calls", float(r["AverageNs"])/1e3, "us")
PY
done
