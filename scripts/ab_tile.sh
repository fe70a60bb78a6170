#!/bin/bash
# A/B of the TILE mode of k_apply_wave (MIMSEM_WAVE_TILE=1: four wave-groups per workgroup, inner partial sums through LDS behind one barrier per
# work item) against the default two-launch form: headline step (cache resident) and the 8-sphere HBM-resident workload, kernel times from the
# context's HIP events, PMC traffic from the bench's child rocprofv3 passes
export MIMSEM_EXPERIMENTS=1      # (the switches below belong to closed experiments: DESIGN 9.1)
EXP_LIB="$(cd "$(dirname "$0")/.." && pwd)/build_ab/libmimsem_hip_exp.so"; [ -z "$MIMSEM_LIB" ] && [ -f "$EXP_LIB" ] && export MIMSEM_LIB="$EXP_LIB"      # (the variants are compiled in only with -DMIMSEM_WITH_EXPERIMENTS: scripts/build_variant.sh exp "-DMIMSEM_WITH_EXPERIMENTS")
R=$GRAFT_REPO_ROOT; cd $R
for t in 0 1 0 1; do
  MIMSEM_WAVE_TILE=$t MIMSEM_VERBOSE=1 python3 bench.py --steps 200 --warmup 20 --no-families --no-column --no-sweep --no-sw --no-cpu > /tmp/ab_tile_$t.json 2> /tmp/ab_tile_$t.err
  grep -m1 "wave plan" /tmp/ab_tile_$t.err | cut -c1-220
  python3 - <<PY
import json
d = json.load(open("bench_extras.json"))
for key in ("roofline", "roofline_cold"):
    r = d[key]; w = r["whole_operator"]
    print("tile=$t %-14s k1 %.1f us frac %.3f | whole %.1f us frac %.3f traffic/compulsory %s | value %.3e" % (key, r["avg_kernel_us"], r["frac"], w["avg_us"], w["frac"],
          w.get("traffic_over_compulsory"), d["value"] if key == "roofline" else r["value"]))
PY
done
