#!/usr/bin/env python3
"""workload for rocprofv3 --kernel-trace --stats of the column path: solve_schur_eta on the 24x24x6 x 30 grid"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from mimsem_amd.device import DeviceMesh, Engine
from mimsem_amd.geom import Geom
from mimsem_amd.mesh import CubedSphere, sphere_coords
from mimsem_amd.topo import Topo
from mimsem_amd.workloads import z_levels
NK = bench.NK
cs = CubedSphere(3, 24, 24); coords = sphere_coords(3, 24)
topos = [Topo(cs, p, NK) for p in range(24)]; geoms = [Geom(t, cs, coords, NK) for t in topos]
for g in geoms: g.set_levels(z_levels(NK, g.n0))
dm = DeviceMesh(topos, geoms, nk=NK); eng = Engine(dm)
rng = np.random.default_rng(0)
nEl, n2 = dm.nEl, eng.n2e
area = float(dm.det.mean())*4.0/n2; dz = float(dm.thick.mean())
lev = lambda nl, lo, hi: eng.tensor(rng.uniform(lo, hi, (nEl, nl*n2))*area*dz)
theta, rho, eta, pi = lev(NK, 280, 320), lev(NK, 0.5, 1.2), lev(NK, 5, 6), lev(NK, 700, 1000)
F = [eng.tensor(rng.standard_normal((nEl, n*n2))*1e8) for n in (NK-1, NK, NK, NK)]
for _ in range(3):
    eng.solve_schur_eta(75.0, theta, rho, eta, pi, *[f.clone() for f in F])
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5):
    eng.solve_schur_eta(75.0, theta, rho, eta, pi, *[f.clone() for f in F])
torch.cuda.synchronize(); print("ms per sweep", (time.perf_counter()-t)/5*1e3)
