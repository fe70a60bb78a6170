export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
for v in 0 1; do
  MIMSEM_NOSWZ=$v rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_swz$v -o r01 -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu --no-sw --no-column > $R/gpurun_out/prof_swz$v.json 2> /dev/null
  echo "NOSWZ=$v"; grep -E "k_elem|k_gather" $R/gpurun_out/prof_swz$v/r01_kernel_stats.csv
  python3 -c "import json; d=json.load(open('$R/gpurun_out/prof_swz$v.json')); print(d['value']/1e9, d['ms_per_step']*1e3, d['roofline']['avg_kernel_us'], d['roofline_op']['avg_us'])"
done
