#!/bin/bash
# levels per task of k_s3_sweep<4> on config 5's columns, one box (round 4): MIMSEM_SWEEP_CHUNK = 16 | 22 | 32 | 64, kernel averages
export MIMSEM_EXPERIMENTS=1      # (the switches below belong to closed experiments: DESIGN 9.1)
EXP_LIB="$(cd "$(dirname "$0")/.." && pwd)/build_ab/libmimsem_hip_exp.so"; [ -z "$MIMSEM_LIB" ] && [ -f "$EXP_LIB" ] && export MIMSEM_LIB="$EXP_LIB"      # (the variants are compiled in only with -DMIMSEM_WITH_EXPERIMENTS: scripts/build_variant.sh exp "-DMIMSEM_WITH_EXPERIMENTS")
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_s3chunk; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
for round in 1 2; do
for c in 16 22 32 64; do
  export MIMSEM_SWEEP_CHUNK=$c
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/c${c}_$round -o r -- python3 $R/scripts/prof_column_box_p4.py > $O/c${c}_$round.log 2>&1 || { tail -n 3 $O/c${c}_$round.log; exit 1; }
  echo "chunk $c round $round: $(grep -E 'k_s3_sweep|k_s3_backsub|k_schur_sweep|k_schur_backsub' $O/c${c}_$round/r_kernel_stats.csv | sed 's/(anonymous namespace):://g; s/void //' | cut -d'(' -f1,2 | cut -d, -f1,4 | tr '\n' ' ')"
done
done
