#!/bin/bash
# A/B: GraphedGMRES lookahead (Arnoldi steps queued per host synchronisation) on the config-3 SW step
export MIMSEM_EXPERIMENTS=1      # (the switches below belong to closed experiments: DESIGN 9.1)
for la in 1 4 8 16; do
  echo "MIMSEM_GMRES_LOOKAHEAD=$la"
  MIMSEM_GMRES_LOOKAHEAD=$la python scripts/prof_sw.py
done
