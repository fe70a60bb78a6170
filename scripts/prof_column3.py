#!/usr/bin/env python3
"""workload for the pentadiagonal column solve: solve_schur_column_3 (eul/VertSolve.cpp:504-675) on the 24x24x6 x 30 grid"""
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
# interface fields (theta, velz) carry no thickness: tests/test_gpu_column.py::_col_fields (rounds 1-3 multiplied theta by dz too: unphysical)
theta, rho, rt, pi = lev(NK + 1, 280, 320)/dz, lev(NK, 0.5, 1.2), lev(NK, 250, 400), lev(NK, 700, 1000)
velz = lev(NK - 1, -1.0, 1.0)/dz
F = [eng.tensor(rng.standard_normal((nEl, n*n2))*1e8) for n in (NK-1, NK, NK, NK)]
for _ in range(2):
    eng.solve_schur_3(75.0, theta, velz, rho, rt, pi, *[f.clone() for f in F])
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5):
    eng.solve_schur_3(75.0, theta, velz, rho, rt, pi, *[f.clone() for f in F])
torch.cuda.synchronize(); print("ms per sweep (solve_schur_column_3)", (time.perf_counter()-t)/5*1e3, "unconverged columns", eng.solve_status()[0])
