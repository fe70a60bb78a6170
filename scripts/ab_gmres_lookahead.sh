#!/bin/bash
# A/B: GraphedGMRES lookahead (Arnoldi steps queued per host synchronisation) on the config-3 SW step
export MIMSEM_EXPERIMENTS=1      # (the switches below belong to closed experiments: DESIGN 9.1)
EXP_LIB="$(cd "$(dirname "$0")/.." && pwd)/build_ab/libmimsem_hip_exp.so"; [ -z "$MIMSEM_LIB" ] && [ -f "$EXP_LIB" ] && export MIMSEM_LIB="$EXP_LIB"      # (the variants are compiled in only with -DMIMSEM_WITH_EXPERIMENTS: scripts/build_variant.sh exp "-DMIMSEM_WITH_EXPERIMENTS")
for la in 1 4 8 16; do
  echo "MIMSEM_GMRES_LOOKAHEAD=$la"
  MIMSEM_GMRES_LOOKAHEAD=$la python scripts/prof_sw.py
done
