#!/bin/bash
# A/B of the task mapping of the chunked column walks on one box (round 4): build_ab/libmimsem_hip_oldmap.so (-DMIMSEM_TASKMAP_OLD: (column,
# chunk) pairs in task order) against the in-tree library (four columns of one chunk per wavefront).  Kernel totals (ns over the script's calls).
export MIMSEM_EXPERIMENTS=1      # (the switches below belong to closed experiments: DESIGN 9.1)
EXP_LIB="$(cd "$(dirname "$0")/.." && pwd)/build_ab/libmimsem_hip_exp.so"; [ -z "$MIMSEM_LIB" ] && [ -f "$EXP_LIB" ] && export MIMSEM_LIB="$EXP_LIB"      # (the variants are compiled in only with -DMIMSEM_WITH_EXPERIMENTS: scripts/build_variant.sh exp "-DMIMSEM_WITH_EXPERIMENTS")
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_taskmap; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
for round in 1 2; do
for v in default oldmap; do
  if [ $v = default ]; then unset MIMSEM_LIB; else export MIMSEM_LIB=$R/build_ab/libmimsem_hip_$v.so; fi
  for s in column column3 column_box_p4; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/${v}_${s}_$round -o r -- python3 $R/scripts/prof_$s.py > $O/${v}_${s}_$round.log 2>&1 || { tail -n 3 $O/${v}_${s}_$round.log; exit 1; }
    echo "$v $s $round: $(grep -E 'k_s3_sweep|k_s3_backsub|k_schur_sweep|k_schur_backsub' $O/${v}_${s}_$round/r_kernel_stats.csv | sed 's/(anonymous namespace):://g; s/void //' | cut -d'(' -f1,2 | cut -d, -f1,2,3 | tr '\n' ' ')"
  done
done
done
