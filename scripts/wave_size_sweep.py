#!/usr/bin/env python3
"""k_apply_wave / perimeter-pass duration against the number of wavefronts (sphere sizes ne = 6, 12, 24, 48 x 30 levels): what part of
the 103 680-unit launch is ramp (dispatch) and what part is a wavefront's own lifetime."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mimsem_amd.device import DeviceMesh, Engine
from mimsem_amd.geom import Geom
from mimsem_amd.mesh import CubedSphere, sphere_coords
from mimsem_amd.topo import Topo
from mimsem_amd.workloads import z_levels
NK = 30
for ne in (6, 12, 24, 48):
    cs = CubedSphere(3, ne, 6); coords = sphere_coords(3, ne)
    topos = [Topo(cs, p, NK) for p in range(6)]
    geoms = [Geom(t, cs, coords, NK) for t in topos]
    for g in geoms:
        g.set_levels(z_levels(NK, g.n0))
    dm = DeviceMesh(topos, geoms, nk=NK, numbering="global")
    eng = Engine(dm, device=0)
    rng = np.random.default_rng(1)
    x = eng.tensor(rng.standard_normal((NK, dm.n1))); y = eng.zeros(NK, dm.n1)
    call, _ = eng.prepare_apply("UMAT", x, lev0=0, scale=1e8, flags=1, out=y)
    for _ in range(5):
        call()
    torch.cuda.synchronize(); eng.set_profiling(1); t = time.perf_counter()
    for _ in range(50):
        call()
    torch.cuda.synchronize(); wall = (time.perf_counter() - t)/50
    a, b, n = eng.profile_read(); eng.set_profiling(0)
    print("ne %2d  units %7d  k1 %.2f us  k2 %.2f us  wall %.2f us/step" % (ne, dm.nEl*NK, a/n*1e3, b/n*1e3, wall*1e6), flush=True)
    del eng
