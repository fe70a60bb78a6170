#!/bin/bash
# A/B of the q solve as a parallel branch of the recorded Picard iteration (MIMSEM_SW_FORK=1: second stream + second context of the same mesh)
# against the single chain (0): config-3 steps/s (scripts/exp/sw_steps.py), then the SW parity tests with the branch on
export MIMSEM_EXPERIMENTS=1      # (the switches below belong to closed experiments: DESIGN 9.1)
EXP_LIB="$(cd "$(dirname "$0")/.." && pwd)/build_ab/libmimsem_hip_exp.so"; [ -z "$MIMSEM_LIB" ] && [ -f "$EXP_LIB" ] && export MIMSEM_LIB="$EXP_LIB"      # (the variants are compiled in only with -DMIMSEM_WITH_EXPERIMENTS: scripts/build_variant.sh exp "-DMIMSEM_WITH_EXPERIMENTS")
cd "$(dirname "$0")/.."
for e in 0 1 0 1; do
  TAG="fork=$e" MIMSEM_SW_FORK=$e python3 scripts/exp/sw_steps.py 2>&1 | tail -1
done
MIMSEM_SW_FORK=1 python3 -m pytest tests/test_gpu_sweqn.py -x -q 2>&1 | tail -3
