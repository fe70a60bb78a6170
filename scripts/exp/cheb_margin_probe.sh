export MIMSEM_EXPERIMENTS=1
for pt in "" 1; do
for m in 0.90,1.05 0.95,1.03 0.97,1.02 0.99,1.01 1.0,1.0; do
  echo -n "PERTURB=$pt margin $m: "; PERTURB=$pt MIMSEM_CHEB_MARGIN=$m python scripts/prof_horiz.py 2>/dev/null | tail -1
done; done
