"""the HorizSolve case of scripts/prof_horiz.py as an array file for mimsem_amd/host/horiz_call (C++ host): usage write_horiz_case.py out.arr"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mimsem_amd.device import DeviceMesh
from mimsem_amd.geom import Geom
from mimsem_amd.mesh import CubedSphere, sphere_coords
from mimsem_amd.topo import Topo
from mimsem_amd.workloads import mesh_arrays, write_arrays, z_levels

PN, NE, NK = 3, 24, 30
cs = CubedSphere(PN, NE, 24); coords = sphere_coords(PN, NE)
topos = [Topo(cs, p, NK) for p in range(24)]
geoms = [Geom(t, cs, coords, NK) for t in topos]
for g in geoms:
    g.set_levels(z_levels(NK, g.n0))
dm = DeviceMesh(topos, geoms, nk=NK, numbering="global")
rng = np.random.default_rng(1)
xq = np.zeros((dm.nq, 3))
for g in geoms:
    xq[g.loc0] = coords[g.loc0]
omega = 7.292e-5
xg = xq[dm.gidq]
fg = np.broadcast_to(2.0 * omega * xg[:, 2] / np.linalg.norm(xg, axis=1), (NK, dm.n0)).copy()
area = float(dm.det.mean()) * 4.0 / 9; dz = float(dm.thick.mean()); ln = area ** 0.5
u1 = rng.standard_normal((NK, dm.n1)) * 20.0 * ln * dz
h1 = rng.uniform(0.8, 1.2, (NK, dm.n2)) * area * dz
arr = mesh_arrays(dm)
arr.update(fg=fg, u1=u1, u2=u1 * 1.01, h1=h1, h2=h1 * 1.001, theta=rng.uniform(290, 310, (NK, dm.n2)) * area * dz,
           Pi=rng.uniform(900, 1000, (NK, dm.n2)) * area * dz, velz=rng.standard_normal((NK - 1, dm.n2)) * area,
           dudz=rng.standard_normal((NK - 1, dm.n1)) * 1e-3 * ln)
write_arrays(sys.argv[1], arr)
