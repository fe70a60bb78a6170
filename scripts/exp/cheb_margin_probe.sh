#!/bin/bash
# How wide must the safety margins around the Ritz interval of P M1 be?  scripts/prof_horiz.py (HorizSolve's right-hand sides, 3 456 elements x 30
# levels) with fixed factors for the lower / upper end (MIMSEM_CHEB_MARGIN, experiments), on the bench's smooth thickness field and on one that varies
# by ~1 % from point to point (PERTURB=1): steps after calibration, interval, residual bounds of the Ritz values, checks, ms per evaluation.
# -> profiles/r06_cheb_margin_probe.txt; the default since then: krylov.ritz_margins (as wide as the estimate is uncertain, at least 1 %).
cd "$(dirname "$0")/../.."
export MIMSEM_EXPERIMENTS=1
for pt in "" 1; do
for m in 0.90,1.05 0.95,1.03 0.97,1.02 0.99,1.01 1.0,1.0; do
  echo -n "PERTURB=$pt margin $m: "; PERTURB=$pt MIMSEM_CHEB_MARGIN=$m python scripts/prof_horiz.py 2>/dev/null | tail -1
done; done
