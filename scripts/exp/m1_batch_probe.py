"""How does one batched 1-form mass solve scale with the number of level rows?  (30 / 60 / 90 rows on the config-4 sphere: if a 60-row solve
costs well under two 30-row solves, HorizSolve's independent ksp1 solves -- grad(Pi), grad(theta), the mass flux -- are worth batching
through a context whose levels repeat the column's)"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from mimsem_amd.device import DeviceMesh, Engine
from mimsem_amd.geom import Geom
from mimsem_amd.krylov import MassSolver
from mimsem_amd.mesh import CubedSphere, sphere_coords
from mimsem_amd.topo import Topo
from mimsem_amd.workloads import z_levels
pn, ne = 3, 24
cs = CubedSphere(pn, ne, 24); coords = sphere_coords(pn, ne)
for rep in (1, 2, 3):
    nk = 30 * rep
    topos = [Topo(cs, p, nk) for p in range(24)]
    geoms = [Geom(t, cs, coords, nk) for t in topos]
    base = z_levels(30, geoms[0].n0)                                   # [31, n0] interface heights
    thick = np.diff(base, axis=0)
    lev = np.concatenate([base[:1], base[:1] + np.cumsum(np.tile(thick, (rep, 1)), axis=0)])      # the 30 thicknesses repeated `rep` times
    for g in geoms:
        g.set_levels(lev)
    dm = DeviceMesh(topos, geoms, nk=nk, numbering="global")
    eng = Engine(dm)
    ms = MassSolver(eng, 1.0e8, True)
    b = eng.tensor(np.random.default_rng(1).standard_normal((nk, dm.n1)))
    for _ in range(3):
        x, its = ms.solve(b)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        x, its = ms.solve(b)
    torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 20
    print("rows %3d: %.3f ms per solve (%s steps), %.4f ms per 30 rows" % (nk, 1e3 * el, its, 1e3 * el / rep))
    del ms, eng
