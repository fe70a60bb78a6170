#!/bin/bash
# kernels of ONE HorizSolve evaluation driven from C++ (mimsem_amd/host/horiz_call): the difference of two kernel-trace summaries (2 and 6
# evaluations -- each evaluation runs eagerly and as a recorded graph, with and without reuse: the program's own loop) -> gpurun_out/prof_hc/
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
cd $R && python scripts/exp/write_horiz_case.py gpurun_out/horiz_case.arr || exit 1
cd /tmp
for n in 2 6; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_hc/n$n -o h -- $R/mimsem_amd/host/horiz_call $R/gpurun_out/horiz_case.arr $n > $R/gpurun_out/prof_hc_$n.log 2>&1 || { tail -3 $R/gpurun_out/prof_hc_$n.log; exit 1; }
done
tail -1 $R/gpurun_out/prof_hc_6.log | cut -c1-600
python3 - <<PY | tee $R/gpurun_out/prof_hc/per_eval.txt
import csv
def load(n):
    return {r["Name"]: (int(r["Calls"]), int(r["TotalDurationNs"])) for r in csv.DictReader(open("$R/gpurun_out/prof_hc/n%d/h_kernel_stats.csv" % n))}
a, b = load(2), load(6)
rows = []
for k, (c6, t6) in b.items():
    c2, t2 = a.get(k, (0, 0))
    if c6 != c2:
        rows.append(((t6 - t2)/4e3, (c6 - c2)/4.0, k))
rows.sort(reverse=True)
print("per repetition of the program's loop: %.1f us in %.0f launches" % (sum(r[0] for r in rows), sum(r[1] for r in rows)))
for t, c, k in rows:
    print("%9.1f us %6.1f calls %7.2f us each  %s" % (t, c, t/max(c, 1e-9), k[:130]))
PY
rm -f $R/gpurun_out/horiz_case.arr
