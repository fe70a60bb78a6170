#!/bin/bash
# A/B of the 1-form mass solves of HorizSolve's right-hand sides (scripts/prof_horiz.py: 7 solves of 14 steps on 3 456 elements x 30 levels):
#   sweeps : a call per Chebyshev step (mimsem_block_chebyshev_sweep: 3 launches each, x and p cleared first)          MIMSEM_CHEB_WHOLE=0
#   whole  : one call per solve, the first step without operator pass (mimsem_block_chebyshev_solve, the default)      MIMSEM_CHEB_WHOLE=1
#   pend   : the same entry in two launches per step (the update folded into the next element pass; experiments build) MIMSEM_CHEB_PEND=1
export MIMSEM_EXPERIMENTS=1
cd "$(dirname "$0")/.."
EXP_LIB="$(pwd)/build_ab/libmimsem_hip_exp.so"; [ -f "$EXP_LIB" ] || { echo "build the experiments library first: scripts/build_variant.sh exp -DMIMSEM_WITH_EXPERIMENTS"; exit 1; }
export MIMSEM_LIB="$EXP_LIB"
for rep in 1 2; do
  echo -n "sweeps: "; MIMSEM_CHEB_WHOLE=0 python scripts/prof_horiz.py 2>/dev/null | tail -1
  echo -n "whole : "; MIMSEM_CHEB_WHOLE=1 python scripts/prof_horiz.py 2>/dev/null | tail -1
  echo -n "pend  : "; MIMSEM_CHEB_WHOLE=1 MIMSEM_CHEB_PEND=1 python scripts/prof_horiz.py 2>/dev/null | tail -1
done
