"""small launches (the reference's single-level assemble + MatMult granularity): Umat apply wall time per call, default two-launch form
against the in-kernel finishing phase (MIMSEM_WAVE_FIN=1) -- where the second LAUNCH, not the bytes, is the cost"""
import os, sys, time
os.environ.setdefault("MIMSEM_EXPERIMENTS", "1")      # (closed-experiment switches are read only under this master switch: DESIGN 9.1)
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from mimsem_amd.device import DeviceMesh, Engine
from mimsem_amd.geom import Geom
from mimsem_amd.mesh import CubedSphere, sphere_coords
from mimsem_amd.topo import Topo
from mimsem_amd.workloads import SCALE, z_levels
rng = np.random.default_rng(1)
for ne, nk in ((8, 1), (16, 1), (24, 1), (24, 4), (24, 8), (24, 30)):
    cs = CubedSphere(3, ne, 6); coords = sphere_coords(3, ne)
    topos = [Topo(cs, p, nk) for p in range(6)]; geoms = [Geom(t, cs, coords, nk) for t in topos]
    for g in geoms: g.set_levels(z_levels(nk, g.n0))
    dm = DeviceMesh(topos, geoms, nk=nk, numbering="global")
    row = []
    for env in ({}, {"MIMSEM_WAVE_FIN": "1"}):
        os.environ.update(env)
        try: eng = Engine(dm)
        finally:
            for k in env: del os.environ[k]
        x = eng.tensor(rng.standard_normal((nk, dm.n1))); y = eng.zeros(nk, dm.n1)
        call, _ = eng.prepare_apply("UMAT", x, lev0=0, scale=SCALE, flags=1, out=y)
        for _ in range(20): call()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(500): call()
        torch.cuda.synchronize(); wall = (time.perf_counter() - t)/500*1e6
        eng.set_profiling(1)
        for _ in range(50): call()
        torch.cuda.synchronize(); c1, c2, cn = eng.profile_read(); eng.set_profiling(0)
        row.append((wall, c1/cn*1e3, c2/cn*1e3))
        del eng
    print("%2dx%2dx6 x %2d levels (%7d units): default %.2f us/call (kernels %.2f + %.2f) | finishing phase %.2f us/call (kernel %.2f)" %
          (ne, ne, nk, dm.nEl*nk, row[0][0], row[0][1], row[0][2], row[1][0], row[1][1]), flush=True)
