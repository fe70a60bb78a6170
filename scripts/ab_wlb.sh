#!/bin/bash
# A/B: levels a wavefront of k_apply_wave takes in lock-step: 2 (default build) against 4 (build_ab/${ABLIB:-libmimsem_hip_lb4.so}, -DMIMSEM_WLB=4);
# bench.py hot and cold, both variants twice in ONE run.
export MIMSEM_EXPERIMENTS=1      # (the switches below belong to closed experiments: DESIGN 9.1)
EXP_LIB="$(cd "$(dirname "$0")/.." && pwd)/build_ab/libmimsem_hip_exp.so"; [ -z "$MIMSEM_LIB" ] && [ -f "$EXP_LIB" ] && export MIMSEM_LIB="$EXP_LIB"      # (the variants are compiled in only with -DMIMSEM_WITH_EXPERIMENTS: scripts/build_variant.sh exp "-DMIMSEM_WITH_EXPERIMENTS")
out=gpurun_out/ab_wlb.log; : > $out
run() { echo "== $*" >> $out; env "$@" python bench.py --no-cpu --no-pmc --no-sw --no-column 2>>gpurun_out/ab_wlb.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['roofline_cold']
print('value %.3e ms/step %.4f | hot k1 %.2f us op %.2f us | cold k1 %.2f us op %.2f us value %.3e' % (d['value'], d['ms_per_step'], r['avg_kernel_us'], r['whole_operator']['avg_us'], c['avg_kernel_us'], c['whole_operator']['avg_us'], c['value']))" >> $out; }
V=$PWD/build_ab/${ABLIB:-libmimsem_hip_lb4.so}
run MIMSEM_LIB=$V
run DEFAULT=1
run MIMSEM_LIB=$V
run DEFAULT=1
cat $out
