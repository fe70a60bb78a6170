#!/usr/bin/env python3
"""The config-5 element step alone (Umat apply, p = 4, 32 x 32 periodic box x 64 levels, 65 536 units) for rocprofv3 passes."""
import os, sys
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
x = eng.tensor(r.standard_normal((NK, dm.n1))); y = eng.zeros(NK, dm.n1)
call, _ = eng.prepare_apply(os.environ.get("OP", "UMAT"), x, lev0=0, scale=1e8, flags=1, out=y)
for _ in range(int(os.environ.get("REPS", "20"))):
    call()
torch.cuda.synchronize()
eng.set_profiling(1)
for _ in range(30):
    call()
torch.cuda.synchronize()
c1, c2, cn = eng.profile_read()
print("kernel us: %.2f + %.2f = %.2f   (MIMSEM_WAVE_CPP=%s MIMSEM_WAVE_LCH=%s)" % (c1/cn*1e3, c2/cn*1e3, (c1+c2)/cn*1e3, os.environ.get("MIMSEM_WAVE_CPP"), os.environ.get("MIMSEM_WAVE_LCH")))
