#!/bin/bash
export MIMSEM_EXPERIMENTS=1      # (the switches below belong to closed experiments: DESIGN 9.1)
EXP_LIB="$(cd "$(dirname "$0")/.." && pwd)/build_ab/libmimsem_hip_exp.so"; [ -z "$MIMSEM_LIB" ] && [ -f "$EXP_LIB" ] && export MIMSEM_LIB="$EXP_LIB"      # (the variants are compiled in only with -DMIMSEM_WITH_EXPERIMENTS: scripts/build_variant.sh exp "-DMIMSEM_WITH_EXPERIMENTS")
for c in 8 10 12; do echo "MIMSEM_SW_CHUNK=$c"; MIMSEM_SW_CHUNK=$c python scripts/prof_sw.py 2>&1 | grep -v amdgpu.ids; done
