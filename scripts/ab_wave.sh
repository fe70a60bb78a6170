#!/bin/bash
# A/B of the wave-level fused 1-form kernel (k_apply_wave + k_gather_perim) against the two-pass form, and of its work-item order
# (MIMSEM_WAVE_ORDER bit 0: XCD-contiguous blocks, bit 1: group-major items) and level chunk; bench.py hot (103 680 units) and cold.
out=gpurun_out/ab_wave.log; : > $out
run() { echo "== $*" >> $out; env "$@" python bench.py --no-cpu --no-sw --no-column 2>>gpurun_out/ab_wave.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['roofline_cold']
print('value %.3e ms/step %.4f | hot k1 %.2f us op %.2f us | cold k1 %.2f us op %.2f us value %.3e' % (d['value'], d['ms_per_step'], r['avg_kernel_us'], r['whole_operator']['avg_us'], c['avg_kernel_us'], c['whole_operator']['avg_us'], c['value']))" >> $out; }
run MIMSEM_WAVE=0
run MIMSEM_WAVE_ORDER=3


cat $out
