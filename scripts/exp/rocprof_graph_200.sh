#!/bin/bash
# ONE profiled 200-step run of the C++-hosted shallow-water step (the run length at which round 5 recorded a SIGSEGV under rocprofv3
# --kernel-trace), with the program's own fatal-signal backtrace on (MIMSEM_BACKTRACE=1): which module do the frames of the fault belong to?
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
python scripts/exp/write_sw_case3.py gpurun_out/sw_case3_200.bin 200
rm -rf gpurun_out/prof_g200
MIMSEM_BACKTRACE=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_g200 -o p --output-format csv -- ./mimsem_amd/host/sw_call gpurun_out/sw_case3_200.bin 5 > gpurun_out/prof_g200.log 2>&1
echo "exit $?" >> gpurun_out/prof_g200.log
grep -v "^W2026\|^E2026" gpurun_out/prof_g200.log | cut -c1-400 | tail -40
ls -la gpurun_out/prof_g200 2>/dev/null | head
find gpurun_out/prof_g200 -name '*trace.csv' -delete 2>/dev/null
rm -f gpurun_out/sw_case3_200.bin
