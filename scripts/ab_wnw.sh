#!/bin/bash
# A/B on the GPU box: wavefronts per workgroup of k_apply_wave (rebuilt in the box's ephemeral copy) and kernel arguments in device memory
export MIMSEM_EXPERIMENTS=1      # (the switches below belong to closed experiments: DESIGN 9.1)
EXP_LIB="$(cd "$(dirname "$0")/.." && pwd)/build_ab/libmimsem_hip_exp.so"; [ -z "$MIMSEM_LIB" ] && [ -f "$EXP_LIB" ] && export MIMSEM_LIB="$EXP_LIB"      # (the variants are compiled in only with -DMIMSEM_WITH_EXPERIMENTS: scripts/build_variant.sh exp "-DMIMSEM_WITH_EXPERIMENTS")
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" python bench.py --no-cpu --no-pmc --no-sw --no-column 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['roofline_cold']
print('value %.3e ms/step %.4f | hot k1 %.2f us op %.2f us | cold k1 %.2f us op %.2f us value %.3e' % (d['value'], d['ms_per_step'], r['avg_kernel_us'], r['whole_operator']['avg_us'], c['avg_kernel_us'], c['whole_operator']['avg_us'], c['value']))"; }
run X=1
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
for n in 2 8 16; do
  (cd mimsem_amd/csrc && rm -f elem_kernels.o && make CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -DMIMSEM_WNW=$n" > /tmp/b.log 2>&1) || { tail -3 /tmp/b.log; continue; }
  run WNW=$n
done
