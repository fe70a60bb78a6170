#!/bin/bash
export MIMSEM_EXPERIMENTS=1      # (the switches below belong to closed experiments: DESIGN 9.1)
EXP_LIB="$(cd "$(dirname "$0")/.." && pwd)/build_ab/libmimsem_hip_exp.so"; [ -z "$MIMSEM_LIB" ] && [ -f "$EXP_LIB" ] && export MIMSEM_LIB="$EXP_LIB"      # (the variants are compiled in only with -DMIMSEM_WITH_EXPERIMENTS: scripts/build_variant.sh exp "-DMIMSEM_WITH_EXPERIMENTS")
R=$GRAFT_REPO_ROOT; cd $R
run() {
  env "$@" python3 bench.py --steps 200 --warmup 20 --no-families --no-column --no-sweep --no-sw --no-cpu --no-pmc > /dev/null 2> /dev/null
  python3 - <<PY
import json
d = json.load(open("bench_extras.json"))
print("$*", " | ".join("%s k1 %.1f whole %.1f us" % (k, d[k]["avg_kernel_us"], d[k]["whole_operator"]["avg_us"]) for k in ("roofline", "roofline_cold")))
PY
}
run MIMSEM_WAVE_TILE=0
run MIMSEM_WAVE_TILE=0 MIMSEM_WAVE_ORDER=1
run MIMSEM_WAVE_TILE=1
run MIMSEM_WAVE_TILE=1 MIMSEM_WAVE_CPP=1
run MIMSEM_WAVE_TILE=1 MIMSEM_WAVE_CPP=4
