#!/bin/bash
# Round-6 evidence in one call on one box: the default bench line (with its in-run PMC passes: families_cold, box_p4 cold), the same
# command under rocprofv3 --kernel-trace --stats, hot / cold launches of the headline step separately, the column / schur_3 / box-p4 /
# Newton / HorizSolve / SW kernel summaries and the SQ + traffic counters of the column solves.  Outputs under gpurun_out/final6/
# (copied into profiles/r06_* afterwards).  Default switches: no MIMSEM_EXPERIMENTS.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final6; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd $R
echo "[1] bench default"; python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -5 $O/bench_default.err; exit 1; }
cp bench_extras.json $O/bench_default_extras.json
echo "[2] bench extras"; python bench.py --no-cpu --no-pmc --no-sweep --horiz --pcie --steps 300 --warmup 30 > $O/bench_extras_line.json 2> $O/bench_extras.err || { tail -5 $O/bench_extras.err; exit 1; }
cp bench_extras.json $O/bench_extras.json
cd /tmp
echo "[3] rocprofv3 stats of the bench command"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o r06 -- python3 $R/bench.py --no-cpu --no-pmc --no-sw --no-column --no-families --no-sweep --steps 300 --warmup 30 > $O/bench_under_rocprof.json 2> $O/rocprof.err || { tail -5 $O/rocprof.err; exit 1; }
echo "[4] hot / cold launches separately"
for w in hot cold; do
  ONLY=$w REPS=40 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$w -o r06 -- python3 $R/scripts/prof_umat.py > $O/stats_$w.log 2>&1 || { tail -5 $O/stats_$w.log; exit 1; }
done
echo "[5] column / schur_3 / box p4 / Newton / HorizSolve / SW kernel summaries"
for s in column column3 column_box_p4 newton horiz sw; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/k_$s -o r06 -- python3 $R/scripts/prof_$s.py > $O/k_$s.log 2>&1 || { tail -3 $O/k_$s.log; }
done
echo "[6] SQ + traffic counters of the column solves"
cd $R
bash scripts/prof_pmc_generic.sh f6col scripts/prof_column.py k_schur,k_thomas,k_band_lu > $O/column_pmc_eta.txt 2>&1
bash scripts/prof_pmc_generic.sh f6col3 scripts/prof_column3.py k_s3,k_penta > $O/column_pmc_s3.txt 2>&1
bash scripts/prof_pmc_generic.sh f6box scripts/prof_column_box_p4.py k_schur,k_thomas,k_penta,k_s3,k_newton,k_diag,k_band_lu > $O/column_pmc_box.txt 2>&1
cat $O/column_pmc_eta.txt $O/column_pmc_s3.txt $O/column_pmc_box.txt | cut -c1-330
echo "[7] the C++-hosted SW step (<= 20 steps: the profiler's limit, profiles/r06_rocprof_graph_sigsegv.txt)"
bash scripts/prof_sw_cpp.sh > $O/sw_cpp.txt 2>&1; tail -16 $O/sw_cpp.txt | cut -c1-200
echo "[8] the kernels of ONE HorizSolve evaluation, Python host and C++ host (differences of two kernel-trace summaries)"
bash scripts/prof_horiz_per_eval.sh > $O/horiz_per_eval.log 2>&1; cp gpurun_out/prof_he/per_eval.txt $O/horiz_per_eval.txt; head -6 $O/horiz_per_eval.txt | cut -c1-160
bash scripts/prof_horiz_cpp_per_eval.sh > $O/horiz_cpp_per_eval.log 2>&1; cp gpurun_out/prof_hc/per_eval.txt $O/horiz_cpp_per_eval.txt; head -6 $O/horiz_cpp_per_eval.txt | cut -c1-160
echo final profiles done
