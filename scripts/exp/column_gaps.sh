#!/bin/bash
# timeline of one solve_schur_column_eta (kernel trace of scripts/prof_column.py): per-kernel durations and the idle time between consecutive
# kernels of a solve, with the pivoted fallback on (default) and off
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
for v in "DEFAULT=1" "MIMSEM_COLUMN_PIVOT_FALLBACK=0"; do
  d=gpurun_out/prof_gaps; rm -rf $d
  env $v rocprofv3 --kernel-trace --output-format csv -d $d -o p -- python3 scripts/prof_column.py > gpurun_out/prof_gaps.log 2>&1
  echo "== $v: $(grep 'ms per sweep' gpurun_out/prof_gaps.log)"
  python3 - "$d" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = lambda r: r["Kernel_Name"].split("(")[0].split("::")[-1][:28]
# the last solve: from the last k_schur_sweep on
idx = [i for i, r in enumerate(rows) if "k_schur_sweep" in r["Kernel_Name"]]
for start in idx[-2:]:
    prev_end = None; line = []
    for r in rows[start:start + 12]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if prev_end is not None:
            line.append("gap %.1f" % ((s - prev_end)/1e3))
        line.append("%s %.1f" % (names(r), (e - s)/1e3))
        prev_end = e
        if "k_schur_backsub" in r["Kernel_Name"]:
            break
    print("   " + " | ".join(line), "| total %.1f us" % ((prev_end - int(rows[start]["Start_Timestamp"]))/1e3))
PY
done
rm -rf gpurun_out/prof_gaps
