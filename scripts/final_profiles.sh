#!/bin/bash
# Round-end evidence: the default bench line, the same command under rocprofv3 --kernel-trace --stats, and the two PMC passes.
# Outputs under gpurun_out/final/ (copied into profiles/ afterwards).
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
export TMPDIR=/tmp
cd $R
python bench.py --steps 300 --warmup 30 > $O/bench_default.json 2> $O/bench_default.err || exit 1
python bench.py --no-cpu --no-pmc --families --column --box --sw --horiz --pcie > $O/bench_extras.json 2> $O/bench_extras.err || exit 1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o r02 -- python3 $R/bench.py --no-cpu --no-pmc --no-sw --no-column --steps 300 --warmup 30 > $O/bench_under_rocprof.json 2> $O/rocprof.err || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o r02 -- python3 $R/scripts/pmc_traffic.py > $O/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o r02 -- python3 $R/scripts/pmc_traffic.py > $O/pmc_write.log 2>&1 || exit 1
cd $R
python3 scripts/pmc_to_json.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json > $O/pmc_traffic.txt 2>&1
python3 scripts/pmc_summarise.py $O/pmc_fetch $O/pmc_write > $O/pmc_summary.txt 2>&1
tail -c 1500 $O/bench_default.json
# the other kernel-stats summaries of the round (column, column-3, Newton, HorizSolve, SW) and the PMC passes of the Umat step
cd $R
for s in prof_column_rocprof.sh prof_column3_rocprof.sh prof_newton_rocprof.sh prof_horiz_rocprof.sh prof_sw_rocprof.sh; do
  [ -f scripts/$s ] && bash scripts/$s > $O/${s%.sh}.log 2>&1
done
MODES="wave twopass" bash scripts/prof_umat_pmc.sh > $O/umat_pmc.txt 2>&1
echo final profiles done
