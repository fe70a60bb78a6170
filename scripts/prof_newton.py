#!/usr/bin/env python3
"""workload for rocprofv3: Newton iterations of the vertical implicit solve (VertSolve.solve_schur_eta) on the 24x24x6 x 30 grid"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from mimsem_amd.device import DeviceMesh, Engine
from mimsem_amd.geom import Geom, gll_points
from mimsem_amd.mesh import CubedSphere, sphere_coords
from mimsem_amd.topo import Topo
from mimsem_amd.vertsolve import VertSolve
from mimsem_amd.workloads import z_levels
NK = bench.NK
cs = CubedSphere(3, 24, 24); coords = sphere_coords(3, 24)
topos = [Topo(cs, p, NK) for p in range(24)]; geoms = [Geom(t, cs, coords, NK) for t in topos]
for g in geoms: g.set_levels(z_levels(NK, g.n0))
dm = DeviceMesh(topos, geoms, nk=NK); eng = Engine(dm)
rng = np.random.default_rng(0)
nEl, n2, nk = dm.nEl, eng.n2e, NK
wd = np.diff(gll_points(3)); wj = np.outer(wd, wd).ravel()
cell = dm.det.mean(axis=1)[:, None, None] * dm.thick.mean(axis=2).T[:, :, None] * wj[None, None, :]
zl = np.mean([g.levs.mean(axis=1) for g in dm.geoms], axis=0); zm = 0.5 * (zl[:-1] + zl[1:])
th_v = 300.0 + 0.004 * zm
pi_v = 1004.5 - (9.80616 / 0.004) * np.log(th_v / 300.0)
rho_v = (1.0e5 / 287.0) * (pi_v / 1004.5) ** (717.5 / 287.0) / th_v
colv = lambda v: eng.tensor((cell * v[None, :, None]).reshape(nEl, nk * n2) * (1.0 + 1e-4 * rng.standard_normal((nEl, nk * n2))))
vs = VertSolve(eng, 75.0)
levs = np.zeros((nk + 1, dm.nq))
for g in dm.geoms:
    levs[:, np.searchsorted(dm.gidq, g.loc0[np.arange(g.n0)])] = g.levs
zv = vs.init_gz(levs)
st = (eng.zeros(nEl, (nk - 1) * n2), colv(rho_v), colv(rho_v * th_v), colv(pi_v))
vs.solve_schur_eta(*st, zv, maxit=2, tol=0.0)
torch.cuda.synchronize(); t0 = time.perf_counter()
vs.solve_schur_eta(*st, zv, maxit=6, tol=0.0)
torch.cuda.synchronize(); print("ms per Newton iteration", (time.perf_counter() - t0) / 6 * 1e3, vs.history[-1])
