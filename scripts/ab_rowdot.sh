#!/bin/bash
# A/B of the one-launch rowdot (round 6: the last block of a row reduces its partial sums, k_rowdot_fused) against the two launches
# (k_rowdot_partial + k_rowdot_final) on the shallow-water step of the Python host, experiments build, same box.  Results are the same bits
# (tests/test_gpu_next_rows.py compares both with the oracle's dot); what changes is 6 launches per Picard iteration.
export MIMSEM_EXPERIMENTS=1
cd "$(dirname "$0")/.."
EXP_LIB="$(pwd)/build_ab/libmimsem_hip_exp.so"; [ -f "$EXP_LIB" ] || { echo "build the experiments library first: scripts/build_variant.sh exp -DMIMSEM_WITH_EXPERIMENTS"; exit 1; }
export MIMSEM_LIB="$EXP_LIB"
for rep in 1 2; do
  for d in 0 1; do
    echo -n "Python host, MIMSEM_ROWDOT_TWO=$d: "; MIMSEM_ROWDOT_TWO=$d python scripts/exp/galewsky_long.py 480 | python3 -c "
import json,sys
rows=[json.loads(l) for l in sys.stdin if l.startswith('{\"step\"')]
print(' '.join('%.4f' % r['ms_per_step'] for r in rows[1:]), 'ms/step per 120 steps (after the first 120)')"
  done
done
