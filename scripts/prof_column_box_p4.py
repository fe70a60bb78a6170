#!/usr/bin/env python3
"""config 5's column half (p = 4, 32 x 32 box x 64 levels): the workload of bench.py's column_box_p4 extra, alone (for rocprofv3 and A/B runs:
MIMSEM_SCHUR_FUSED=0 MIMSEM_NEWTON_FUSED=0 MIMSEM_SCHUR3_WIDE_FACTORS=1 selects round 1's un-fused chain)"""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
print(json.dumps(bench.column_box_p4_extras(0, np.random.default_rng(20241024), torch), indent=1))
