export MIMSEM_EXPERIMENTS=1      # (the switches below belong to closed experiments: DESIGN 9.1)
for w in 1 0 1 0; do echo "== MIMSEM_WAVE=$w"; MIMSEM_WAVE=$w python bench.py --no-cpu --no-pmc --no-sw --no-column --cold 0 --box 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['box_p4'])"; done
