#!/bin/bash
# kernel-level profile of one HorizSolve advection_rhs_ec + momentum_rhs_ec evaluation (scripts/prof_horiz.py) -> gpurun_out/prof_h/
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_h -o r02 -- python3 $R/scripts/prof_horiz.py > $R/gpurun_out/prof_h.log 2>&1
grep -v amdgpu.ids $R/gpurun_out/prof_h.log | tail -1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/prof_h/r02_kernel_stats.csv")))
tot=sum(int(r["TotalDurationNs"]) for r in rows); calls=sum(int(r["Calls"]) for r in rows)
mine=sum(int(r["TotalDurationNs"]) for r in rows if "at::native" not in r["Name"] and "rocclr" not in r["Name"])
print("total kernel ms", tot/1e6, "launches", calls, " share of GPU time in library kernels %.1f%%" % (100.0*mine/tot))
for r in rows[:26]:
    print("%-84s %6s calls %8.1f us avg %5s%%" % (r["Name"][:84], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
