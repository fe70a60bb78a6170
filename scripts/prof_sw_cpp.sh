#!/bin/bash
# kernel trace of the C++-hosted shallow-water step (mimsem_amd/host/sw_call, config-3 sphere, recorded Picard iterations): busy time of the
# kernels against the wall time of a step -> what part of a graph node's ~5 us is the kernel and what part the hand-over between nodes
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
python scripts/exp/write_sw_case3.py gpurun_out/sw_case3.bin 20    # (20 steps: on the 200-step case rocprofv3 --kernel-trace dies with a SIGSEGV inside librocprofiler-sdk.so's HSA packet interception, below hipGraphLaunch -- the tool, not this library: the backtraced run is profiles/r06_rocprof_graph_sigsegv.txt, scripts/exp/rocprof_graph_200.sh)
./mimsem_amd/host/sw_call gpurun_out/sw_case3.bin 3 | cut -c1-260
rm -rf gpurun_out/prof_swcpp
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_swcpp -o p --output-format csv -- ./mimsem_amd/host/sw_call gpurun_out/sw_case3.bin 3 > gpurun_out/prof_swcpp.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_swcpp/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows); calls = sum(int(r["Calls"]) for r in rows)
print("all kernels of the process (both modes, warm-up and set-up included): %d launches, %.1f ms busy, %.2f us per launch" % (calls, tot/1e6, tot/calls/1e3))
for r in rows[:14]:
    print("  %-64s calls %6s avg %7.2f us  %5.1f %%" % (r["Name"][:64], r["Calls"], float(r["AverageNs"])/1e3, float(r["Percentage"])))
PY
rm -f gpurun_out/sw_case3.bin; find gpurun_out/prof_swcpp -name '*trace.csv' -delete
