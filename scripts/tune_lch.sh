python -m pytest tests -m gpu -q -x > gpurun_out/t5.log 2>&1; tail -3 gpurun_out/t5.log
for l in 0 1 2 3 5 8 15 30; do
  echo "LCH=$l"; MIMSEM_LCH=$l python bench.py --no-cpu --no-sw --no-column --steps 300 --warmup 30 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print(d['value']/1e9, d['ms_per_step']*1e3, d['roofline']['avg_kernel_us'], d['roofline_op']['avg_us'])"
done
