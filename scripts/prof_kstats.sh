#!/bin/bash
# usage: prof_kstats.sh <tag> <script.py> [env assignments are inherited]: rocprofv3 --kernel-trace --stats summary of one workload script
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
TAG=$1; SCRIPT=$2
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o r -- python3 $R/$SCRIPT > $R/gpurun_out/prof_$TAG.log 2>&1
grep -v amdgpu.ids $R/gpurun_out/prof_$TAG.log | tail -3
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/prof_$TAG/r_kernel_stats.csv")))
for r in rows[:14]:
    print("%-110s %5s calls %9.1f us avg %6s%%" % (r["Name"][:110], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
