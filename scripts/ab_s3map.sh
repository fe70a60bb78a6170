#!/bin/bash
# A/B of k_s3_sweep's task mapping / index pinning on one box (round 4): build_ab/libmimsem_hip_{nokeep,oldmap}.so against the in-tree library
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_s3map; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
for round in 1 2; do
for v in default nokeep oldmap; do
  if [ $v = default ]; then unset MIMSEM_LIB; else export MIMSEM_LIB=$R/build_ab/libmimsem_hip_$v.so; fi
  for s in column3 column_box_p4; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/${v}_${s}_$round -o r -- python3 $R/scripts/prof_$s.py > $O/${v}_${s}_$round.log 2>&1 || { tail -n 3 $O/${v}_${s}_$round.log; exit 1; }
    echo "$v $s $round: $(grep -E 'k_s3_sweep' $O/${v}_${s}_$round/r_kernel_stats.csv | cut -d, -f2-4 | tr '\n' ' ')"
  done
done
done
