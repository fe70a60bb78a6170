#!/bin/bash
# C++-hosted shallow-water step (mimsem_amd/host/sw_call) on the config-3 sphere: the legacy default stream against a stream of the context's own
set -e
cd "$(dirname "$0")/.."
python scripts/exp/write_sw_case3.py gpurun_out/sw_case3.bin 200
for i in 1 2; do
echo "default stream:"; MIMSEM_SW_DEFAULT_STREAM=1 ./mimsem_amd/host/sw_call gpurun_out/sw_case3.bin 5
echo "own stream:"; ./mimsem_amd/host/sw_call gpurun_out/sw_case3.bin 5
done
rm -f gpurun_out/sw_case3.bin
