export MIMSEM_EXPERIMENTS=1      # (the switches below belong to closed experiments: DESIGN 9.1)
EXP_LIB="$(cd "$(dirname "$0")/.." && pwd)/build_ab/libmimsem_hip_exp.so"; [ -z "$MIMSEM_LIB" ] && [ -f "$EXP_LIB" ] && export MIMSEM_LIB="$EXP_LIB"      # (the variants are compiled in only with -DMIMSEM_WITH_EXPERIMENTS: scripts/build_variant.sh exp "-DMIMSEM_WITH_EXPERIMENTS")
python -m pytest tests -m gpu -q -x > gpurun_out/t16.log 2>&1; tail -3 gpurun_out/t16.log
for v in 1 0; do
  echo "FUSE=$v"
  if [ $v = 1 ]; then export MIMSEM_FUSE=1; else unset MIMSEM_FUSE; fi
  python bench.py --no-cpu --no-sw --no-column --steps 400 --warmup 40 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print(d['value']/1e9, d['ms_per_step']*1e3, d['roofline']['avg_kernel_us'], d['roofline_op']['avg_us'])"
done
unset MIMSEM_FUSE
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_fuse -o r01 -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu --no-sw --no-column > /dev/null 2>&1; grep -E "k_elem|k_gather" $R/gpurun_out/prof_fuse/r01_kernel_stats.csv
