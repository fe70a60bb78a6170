#!/bin/bash
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_newton -o r01 -- python3 $R/scripts/prof_newton.py > $R/gpurun_out/prof_newton.log 2>&1
grep -v amdgpu.ids $R/gpurun_out/prof_newton.log | tail -1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/prof_newton/r01_kernel_stats.csv")))
tot=sum(int(r["TotalDurationNs"]) for r in rows); calls=sum(int(r["Calls"]) for r in rows)
print("total kernel ms per iteration", tot/1e6/8, "launches per iteration", calls/8)
for r in rows[:18]:
    print("%-100s %5s calls %8.1f us avg %5s%%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
