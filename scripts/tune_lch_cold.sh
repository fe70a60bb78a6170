export MIMSEM_EXPERIMENTS=1      # (the switches below belong to closed experiments: DESIGN 9.1)
for l in 8 10 15 30; do MIMSEM_LCH=$l python bench.py --no-cpu --no-sw --no-column --steps 50 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['roofline_cold']; r=d['roofline']; print('LCH=$l cold k1 %.1f us op %.1f us frac %.3f | resident k1 %.1f us step %.2f us' % (c['avg_kernel_us'], c['whole_operator']['avg_us'], c['frac'], r['avg_kernel_us'], d['ms_per_step']*1e3))"; done
