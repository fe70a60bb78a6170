#!/bin/bash
# operator families (bench.py --families), wave-level kernels on / off for the 2-form-valued ones (MIMSEM_WAVE2=0) and altogether (MIMSEM_WAVE=0)
export MIMSEM_EXPERIMENTS=1      # (the switches below belong to closed experiments: DESIGN 9.1)
EXP_LIB="$(cd "$(dirname "$0")/.." && pwd)/build_ab/libmimsem_hip_exp.so"; [ -z "$MIMSEM_LIB" ] && [ -f "$EXP_LIB" ] && export MIMSEM_LIB="$EXP_LIB"      # (the variants are compiled in only with -DMIMSEM_WITH_EXPERIMENTS: scripts/build_variant.sh exp "-DMIMSEM_WITH_EXPERIMENTS")
for v in "X=1" "MIMSEM_WAVE2=0" "MIMSEM_WAVE=0" "X=1"; do echo "== $v"; env $v python bench.py --no-cpu --no-pmc --no-sw --no-column --cold 0 --families 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:'%.2e'%v for k,v in d['families'].items()})"; done
