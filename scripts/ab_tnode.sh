#!/bin/bash
# A/B on one box (round 4): thickInv of k_apply_wave from the [element][point] pair table (default) or from the NODAL pair table
# (MIMSEM_WAVE_TNODE=1): kernel averages of the headline step, cache-resident (103 680 units) and HBM-resident (8 spheres)
export MIMSEM_EXPERIMENTS=1      # (the switches below belong to closed experiments: DESIGN 9.1)
EXP_LIB="$(cd "$(dirname "$0")/.." && pwd)/build_ab/libmimsem_hip_exp.so"; [ -z "$MIMSEM_LIB" ] && [ -f "$EXP_LIB" ] && export MIMSEM_LIB="$EXP_LIB"      # (the variants are compiled in only with -DMIMSEM_WITH_EXPERIMENTS: scripts/build_variant.sh exp "-DMIMSEM_WITH_EXPERIMENTS")
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_tnode; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
for round in 1 2; do
for v in 0 1; do
  export MIMSEM_WAVE_TNODE=$v
  for w in hot cold; do
    ONLY=$w REPS=40 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t${v}_${w}_$round -o r -- python3 $R/scripts/prof_umat.py > $O/t${v}_${w}_$round.log 2>&1 || { tail -n 3 $O/t${v}_${w}_$round.log; exit 1; }
    echo "tnode $v $w $round: $(grep -E 'k_apply_wave|k_wave_perim' $O/t${v}_${w}_$round/r_kernel_stats.csv | sed 's/(anonymous namespace):://g; s/void //' | awk -F'","|",' '{print substr($1,2,14), $4}' | tr '\n' ' ')"
  done
done
done
