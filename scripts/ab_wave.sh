#!/bin/bash
# A/B of the wave-level fused 1-form kernel (k_apply_wave + k_wave_perim) against the two-pass form (MIMSEM_WAVE=0) and of its knobs:
# MIMSEM_WAVE_CPP = chunks of 8 levels per wavefront (default: balanced parts leaving >= 1536 wavefronts), MIMSEM_WAVE_ORDER, MIMSEM_WAVE_LCH.
# bench.py hot (103 680 units) and cold (829 440 units); all variants in ONE run (boxes of the pool differ by ~10 %).
export MIMSEM_EXPERIMENTS=1      # (the switches below belong to closed experiments: DESIGN 9.1)
EXP_LIB="$(cd "$(dirname "$0")/.." && pwd)/build_ab/libmimsem_hip_exp.so"; [ -z "$MIMSEM_LIB" ] && [ -f "$EXP_LIB" ] && export MIMSEM_LIB="$EXP_LIB"      # (the variants are compiled in only with -DMIMSEM_WITH_EXPERIMENTS: scripts/build_variant.sh exp "-DMIMSEM_WITH_EXPERIMENTS")
out=gpurun_out/ab_wave.log; : > $out
run() { echo "== $*" >> $out; env "$@" python bench.py --no-cpu --no-pmc --no-sw --no-column 2>>gpurun_out/ab_wave.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['roofline_cold']
print('value %.3e ms/step %.4f | hot k1 %.2f us op %.2f us | cold k1 %.2f us op %.2f us value %.3e' % (d['value'], d['ms_per_step'], r['avg_kernel_us'], r['whole_operator']['avg_us'], c['avg_kernel_us'], c['whole_operator']['avg_us'], c['value']))" >> $out; }
run MIMSEM_WAVE=0
run DEFAULT=1
for c in ${CPPS:-1 2 4}; do run MIMSEM_WAVE_CPP=$c; done
run DEFAULT=1
cat $out
