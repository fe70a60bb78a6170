#!/bin/bash
# A/B of the shared-launch solves of a shallow-water Picard iteration (round 6: mimsem_sw_dual_chebyshev -- the mass-flux solve and the
# potential-vorticity solve issued as ONE grid per launch index) against the two solves one after the other (round 5), same binary, same box:
# the C++ host (mimsem_amd/host/sw_call, config 3, 240 steps) and the Python host (scripts/exp/sw_steps.py).  MIMSEM_SW_DUAL is a host-level switch.
export MIMSEM_EXPERIMENTS=1
cd "$(dirname "$0")/.."
python scripts/exp/write_sw_case3.py gpurun_out/sw_case3.bin 240
for rep in 1 2 3; do
  for d in 1 0; do
    echo -n "C++ host, MIMSEM_SW_DUAL=$d: "; MIMSEM_SW_DUAL=$d ./mimsem_amd/host/sw_call gpurun_out/sw_case3.bin 5 | python3 -c "import json,sys; d=json.load(sys.stdin)['graph']; print('%.1f steps/s  %.4f ms/step' % (d['steps_per_s'], d['ms_per_step']))"
  done
done
for d in 1 0; do
  echo -n "Python host, MIMSEM_SW_DUAL=$d: "; MIMSEM_SW_DUAL=$d python scripts/exp/galewsky_long.py 480 | python3 -c "
import json,sys
rows=[json.loads(l) for l in sys.stdin if l.startswith('{\"step\"')]
print(' '.join('%.4f' % r['ms_per_step'] for r in rows[1:]), 'ms/step per 120 steps (after the first 120)')"
done
rm -f gpurun_out/sw_case3.bin
