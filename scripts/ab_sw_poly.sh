#!/bin/bash
# A/B of the polynomial preconditioner of the [u|h] solve (mimsem_amd/sweqn.py, MIMSEM_SW_POLY = d Richardson steps on the coupled element
# blocks per application) -> steps/s, Krylov counts, error norms, drifts on configs 2 and 3
export MIMSEM_EXPERIMENTS=1      # (the switches below belong to closed experiments: DESIGN 9.1)
EXP_LIB="$(cd "$(dirname "$0")/.." && pwd)/build_ab/libmimsem_hip_exp.so"; [ -z "$MIMSEM_LIB" ] && [ -f "$EXP_LIB" ] && export MIMSEM_LIB="$EXP_LIB"      # (the variants are compiled in only with -DMIMSEM_WITH_EXPERIMENTS: scripts/build_variant.sh exp "-DMIMSEM_WITH_EXPERIMENTS")
R=$GRAFT_REPO_ROOT; cd $R
for e in 1 2 3 4; do
  MIMSEM_SW_POLY=$e python3 bench.py --no-families --no-column --no-sweep --cold 0 --no-pmc --no-cpu > /dev/null 2> /dev/null
  python3 - <<PY
import json
d = json.load(open("bench_extras.json"))["sw"]
for k, v in d.items():
    print("poly=$e", k, "steps/s %.1f" % v["steps_per_s"], "picard/step %.1f" % v["picard_iterations_per_step"], "its", v["krylov_iterations_last"],
          "drift", {a: "%.3e" % b for a, b in v["relative_drift_over_timed_steps"].items()},
          "errs", None if not v["williamson2_error_norms_L1_L2_Linf"] else {a: ["%.12e" % x for x in b] for a, b in v["williamson2_error_norms_L1_L2_Linf"].items()})
PY
done
