#!/bin/bash
# C++-hosted shallow-water step (mimsem_amd/host/sw_call) on the config-3 sphere: A/B of host-side switches
#   MIMSEM_SW_DEFAULT_STREAM=1   the context on the legacy default stream (graph recorded on a blocking stream) against a stream of its own
#   MIMSEM_SW_STEP2=1            the [u|h] Chebyshev step in two launches (the gather epilogue in the next element pass) against three
export MIMSEM_EXPERIMENTS=1      # (the switches below belong to closed experiments: DESIGN 9.1)
EXP_LIB="$(cd "$(dirname "$0")/.." && pwd)/build_ab/libmimsem_hip_exp.so"; [ -z "$MIMSEM_LIB" ] && [ -f "$EXP_LIB" ] && export MIMSEM_LIB="$EXP_LIB"      # (the variants are compiled in only with -DMIMSEM_WITH_EXPERIMENTS: scripts/build_variant.sh exp "-DMIMSEM_WITH_EXPERIMENTS")
set -e
cd "$(dirname "$0")/.."
python scripts/exp/write_sw_case3.py gpurun_out/sw_case3.bin 200
for i in 1 2; do
for v in "DEFAULT=1" "MIMSEM_SW_STEP2=1" "MIMSEM_SW_DEFAULT_STREAM=1"; do
echo "$v:"; env $v ./mimsem_amd/host/sw_call gpurun_out/sw_case3.bin 5 | cut -c1-215
done
done
rm -f gpurun_out/sw_case3.bin
