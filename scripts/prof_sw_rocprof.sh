#!/bin/bash
# kernel-level profile of the config-3 shallow-water step (scripts/prof_sw.py) -> gpurun_out/prof_sw/
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_sw -o r01 -- python3 $R/scripts/prof_sw.py > $R/gpurun_out/prof_sw.log 2>&1
grep -v amdgpu.ids $R/gpurun_out/prof_sw.log | tail -2
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/prof_sw/r01_kernel_stats.csv")))
tot=sum(int(r["TotalDurationNs"]) for r in rows); calls=sum(int(r["Calls"]) for r in rows)
print("total kernel ms", tot/1e6, "launches", calls)
for r in rows[:22]:
    print("%-70s %6s calls %8.1f us avg %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
