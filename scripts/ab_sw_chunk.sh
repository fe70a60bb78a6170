#!/bin/bash
export MIMSEM_EXPERIMENTS=1      # (the switches below belong to closed experiments: DESIGN 9.1)
for c in 8 10 12; do echo "MIMSEM_SW_CHUNK=$c"; MIMSEM_SW_CHUNK=$c python scripts/prof_sw.py 2>&1 | grep -v amdgpu.ids; done
