#!/usr/bin/env python3
"""The bench.py step (Umat apply over the config-4 sphere x 30 levels) alone, hot (103 680 units) and cold (8 spheres), for rocprofv3
passes: scripts/prof_umat_pmc.sh.  MIMSEM_WAVE=0 selects the two-pass form."""
import os
os.environ.setdefault("MIMSEM_EXPERIMENTS", "1")      # (closed-experiment switches are read only under this master switch: DESIGN 9.1)
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from mimsem_amd.device import DeviceMesh, Engine
from mimsem_amd.geom import Geom
from mimsem_amd.mesh import CubedSphere, sphere_coords
from mimsem_amd.topo import Topo
from mimsem_amd.workloads import z_levels

PN, NE, NK = 3, 24, 30
cs = CubedSphere(PN, NE, 24); coords = sphere_coords(PN, NE)
topos = [Topo(cs, p, NK) for p in range(24)]
geoms = [Geom(t, cs, coords, NK) for t in topos]
for g in geoms:
    g.set_levels(z_levels(NK, g.n0))
dm = DeviceMesh(topos, geoms, nk=NK, numbering="global")
rng = np.random.default_rng(1)
reps = int(os.environ.get("REPS", "10"))
only = os.environ.get("ONLY", "")                       # "hot" | "cold": one workload per process, so that a rocprofv3 summary has one row per size
for R in [r for r, tag in ((1, "hot"), (int(os.environ.get("COLD", "8")), "cold")) if not only or only == tag]:
    d = dm if R == 1 else bench.replicate(dm, R)
    eng = Engine(d, device=0)
    x = eng.tensor(rng.standard_normal((NK, d.n1))); y = eng.zeros(NK, d.n1)
    call, _ = eng.prepare_apply("UMAT", x, lev0=0, scale=1e8, flags=1, out=y)
    for _ in range(reps):
        call()
    torch.cuda.synchronize()
    del eng
print("done")
