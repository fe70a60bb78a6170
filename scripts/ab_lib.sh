#!/bin/bash
# A/B of a library variant (build_ab/libmimsem_hip_$1.so, scripts/build_variant.sh) against the default: bench.py hot and cold, twice each
export MIMSEM_EXPERIMENTS=1      # (the switches below belong to closed experiments: DESIGN 9.1)
EXP_LIB="$(cd "$(dirname "$0")/.." && pwd)/build_ab/libmimsem_hip_exp.so"; [ -z "$MIMSEM_LIB" ] && [ -f "$EXP_LIB" ] && export MIMSEM_LIB="$EXP_LIB"      # (the variants are compiled in only with -DMIMSEM_WITH_EXPERIMENTS: scripts/build_variant.sh exp "-DMIMSEM_WITH_EXPERIMENTS")
out=gpurun_out/ab_lib_$1.log; : > $out
run() { echo "== $*" >> $out; env "$@" python bench.py --no-cpu --no-pmc --no-sw --no-column --no-families --no-sweep 2>>gpurun_out/ab_lib.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['roofline_cold']
print('value %.3e ms/step %.4f | hot k1 %.2f us op %.2f us | cold k1 %.2f us op %.2f us value %.3e' % (d['value'], d['ms_per_step'], r['avg_kernel_us'], r['whole_operator']['avg_us'], c['avg_kernel_us'], c['whole_operator']['avg_us'], c['value']))" >> $out; }
V=$PWD/build_ab/libmimsem_hip_$1.so
run MIMSEM_LIB=$V
run DEFAULT=1
run MIMSEM_LIB=$V
run DEFAULT=1
cat $out
