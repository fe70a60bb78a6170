python -m pytest tests -m gpu -q -x > gpurun_out/t6.log 2>&1; tail -2 gpurun_out/t6.log
for e in 1 8 1000000; do
  echo "PROF_EVERY=$e"; MIMSEM_BENCH_PROF_EVERY=$e python bench.py --no-cpu --no-sw --no-column --steps 400 --warmup 40 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print(d['value']/1e9, d['ms_per_step']*1e3, d.get('roofline',{}).get('avg_kernel_us'), d.get('roofline_op',{}).get('avg_us'))"
done
