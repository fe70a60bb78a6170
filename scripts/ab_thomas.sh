#!/bin/bash
# The Helmholtz solve inside solve_schur_column_eta (3 456 columns x 30 levels): one-sided sweep (MIMSEM_THOMAS2=0) against the two-sided
# sweep at one wavefront per SIMD (MIMSEM_THOMAS_WPS=1, round 3) and at two (default, round 5); wall clock per solve and rocprofv3 kernel averages of the three kernels of the solve.
export MIMSEM_EXPERIMENTS=1      # (the switches below belong to closed experiments: DESIGN 9.1)
EXP_LIB="$(cd "$(dirname "$0")/.." && pwd)/build_ab/libmimsem_hip_exp.so"; [ -z "$MIMSEM_LIB" ] && [ -f "$EXP_LIB" ] && export MIMSEM_LIB="$EXP_LIB"      # (the variants are compiled in only with -DMIMSEM_WITH_EXPERIMENTS: scripts/build_variant.sh exp "-DMIMSEM_WITH_EXPERIMENTS")
out=gpurun_out/ab_thomas.log; : > $out
cd /tmp && export TMPDIR=/tmp
for v in "MIMSEM_THOMAS2=0" "MIMSEM_THOMAS_WPS=1" "DEFAULT=1"; do
  echo "== $v" >> $GRAFT_REPO_ROOT/$out
  d=$GRAFT_REPO_ROOT/gpurun_out/prof_thomas_$(echo $v | tr -c 'A-Za-z0-9' '_')
  env $v python3 $GRAFT_REPO_ROOT/scripts/prof_column.py 2>/dev/null | tail -1 >> $GRAFT_REPO_ROOT/$out
  env $v rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python3 $GRAFT_REPO_ROOT/scripts/prof_column.py > /dev/null 2>&1
  python3 - "$d" >> $GRAFT_REPO_ROOT/$out <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if any(k in row["Name"] for k in ("thomas", "schur_sweep", "schur_backsub")):
            print("   %-60s calls %4s avg %8.1f us" % (row["Name"][:60], row["Calls"], float(row["AverageNs"]) / 1e3))
PY
done
cat $GRAFT_REPO_ROOT/$out
