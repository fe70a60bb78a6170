#!/bin/bash
# kernels of ONE HorizSolve right-hand-side evaluation: the difference of two kernel-trace summaries (scripts/prof_horiz.py with 2 and 6
# timed evaluations) divided by 4 -- set-up (Lanczos, block inverses, fields) drops out.  -> gpurun_out/prof_he/per_eval.txt
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
for n in 2 6; do
  REPS=$n rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_he/n$n -o h -- python3 $R/scripts/prof_horiz.py > $R/gpurun_out/prof_he_$n.log 2>&1 || { tail -3 $R/gpurun_out/prof_he_$n.log; exit 1; }
done
python3 - <<PY | tee $R/gpurun_out/prof_he/per_eval.txt
import csv
def load(n):
    return {r["Name"]: (int(r["Calls"]), int(r["TotalDurationNs"])) for r in csv.DictReader(open("$R/gpurun_out/prof_he/n%d/h_kernel_stats.csv" % n))}
a, b = load(2), load(6)
rows = []
for k, (c6, t6) in b.items():
    c2, t2 = a.get(k, (0, 0))
    if c6 != c2:
        rows.append(((t6 - t2)/4e3, (c6 - c2)/4.0, k))
rows.sort(reverse=True)
print("per evaluation: %.1f us in %.0f launches" % (sum(r[0] for r in rows), sum(r[1] for r in rows)))
for t, c, k in rows:
    print("%9.1f us %6.1f calls %7.2f us each  %s" % (t, c, t/max(c, 1e-9), k[:130]))
PY
