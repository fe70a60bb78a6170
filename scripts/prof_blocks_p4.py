#!/usr/bin/env python3
"""The one GEMM-shaped product of the element path at p = 4 (BASELINE config 5: "MFMA element mat-vec path"): the dense 40 x 40 element
block times the gathered residuals of every level -- the block pass of the Chebyshev / Richardson sweeps -- on the config-5 grid
(32 x 32 periodic box x 64 levels, 65 536 units): register-row kernel (default) or k_blocks_residual_mfma (MIMSEM_BLOCKS_MFMA=1:
v_mfma_f64_16x16x4 over 16 levels).  Run under rocprofv3 by scripts/ab_mfma_p4.sh (kernel time, SQ_INSTS_VALU_MFMA_MOPS_F64,
SQ_VALU_MFMA_BUSY_CYCLES)."""
import os, sys, time
os.environ.setdefault("MIMSEM_EXPERIMENTS", "1")      # (closed-experiment switches are read only under this master switch: DESIGN 9.1)
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from mimsem_amd.device import DeviceMesh, Engine
from mimsem_amd.geom import BoxGeom
from mimsem_amd.mesh import PeriodicBox, box_coords
from mimsem_amd.topo import Topo
NK = 64
bx = PeriodicBox(4, 32, 4); bc = box_coords(4, 32, 1000.0)
bt = [Topo(bx, p, NK) for p in range(4)]; bg = [BoxGeom(t, bx, bc, NK, 1000.0) for t in bt]
for g in bg:
    g.set_levels(np.repeat(np.linspace(0.0, 1500.0, NK + 1)[:, None], g.n0, axis=1))
dm = DeviceMesh(bt, bg, nk=NK, numbering="global"); eng = Engine(dm)
r = np.random.default_rng(3)
nd = 2 * eng.n1e
B = eng.tensor(r.standard_normal((dm.nEl, nd, nd)) / nd)
b = eng.tensor(r.standard_normal((NK, dm.n1))); x = eng.tensor(r.standard_normal((NK, dm.n1))); p = eng.tensor(r.standard_normal((NK, dm.n1)))
es = eng.tensor(r.uniform(0.5, 1.5, (NK, dm.nEl)))
sweep = lambda: eng.block_chebyshev_sweep("UMAT", B, x, b, p, 0.7, 0.3, elem_scale=es, scale=1e8, flags=1)
for _ in range(3):
    sweep()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(20):
    sweep()
torch.cuda.synchronize()
print("us per Chebyshev sweep (3 launches: element pass, block pass, gather + update): %.1f   units %d" % ((time.perf_counter() - t) / 20 * 1e6, dm.nEl * NK))
