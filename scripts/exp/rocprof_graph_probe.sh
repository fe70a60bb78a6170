#!/bin/bash
# does rocprofv3 --kernel-trace survive hipGraphLaunch of graphs recorded through mimsem_graph_*?  (bench_call: 60 kernel nodes; sw_call: ~225)
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
python scripts/exp/write_sw_case3.py gpurun_out/sw_case3.bin 20
for cmd in "./mimsem_amd/host/bench_call 12 50" "./mimsem_amd/host/sw_call gpurun_out/sw_case3.bin 2"; do
  rm -rf gpurun_out/prof_probe
  timeout -k 10 120 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_probe -o p --output-format csv -- $cmd > gpurun_out/prof_probe.log 2>&1
  echo "== $cmd -> exit $?"; grep -c "Segmentation\|SIGSEGV" gpurun_out/prof_probe.log; tail -2 gpurun_out/prof_probe.log | cut -c1-200
done
rm -f gpurun_out/sw_case3.bin; rm -rf gpurun_out/prof_probe
