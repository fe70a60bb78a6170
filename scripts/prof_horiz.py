"""Profile target: HorizSolve advection_rhs_ec + momentum_rhs_ec over 30 levels (run under rocprofv3 --kernel-trace --stats)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mimsem_amd.device import DeviceMesh, Engine
from mimsem_amd.geom import Geom
from mimsem_amd.horizsolve import HorizSolve
from mimsem_amd.mesh import CubedSphere, sphere_coords
from mimsem_amd.topo import Topo
from mimsem_amd.workloads import z_levels

PN, NE, NK = 3, 24, 30
cs = CubedSphere(PN, NE, 24); coords = sphere_coords(PN, NE)
topos = [Topo(cs, p, NK) for p in range(24)]
geoms = [Geom(t, cs, coords, NK) for t in topos]
for g in geoms:
    g.set_levels(z_levels(NK, g.n0, rng=np.random.default_rng(5) if os.environ.get("PERTURB") else None))      # PERTURB=1: the layer thickness varies by ~1 % from point to point
dm = DeviceMesh(topos, geoms, nk=NK, numbering="global")
eng = Engine(dm)
rng = np.random.default_rng(1)
xq = np.zeros((dm.nq, 3))
for g in geoms:
    xq[g.loc0] = coords[g.loc0]
hs = HorizSolve(eng, quad_coords=xq[dm.gidq])
area = float(dm.det.mean()) * 4.0 / 9; dz = float(dm.thick.mean()); ln = area ** 0.5
u1 = eng.tensor(rng.standard_normal((NK, dm.n1)) * 20.0 * ln * dz); u2 = u1 * 1.01
h1 = eng.tensor(rng.uniform(0.8, 1.2, (NK, dm.n2)) * area * dz); h2 = h1 * 1.001
th = eng.tensor(rng.uniform(290, 310, (NK, dm.n2)) * area * dz); Pi = eng.tensor(rng.uniform(900, 1000, (NK, dm.n2)) * area * dz)
vz = eng.tensor(rng.standard_normal((NK - 1, dm.n2)) * area); dudz = eng.tensor(rng.standard_normal((NK - 1, dm.n1)) * 1e-3 * ln)
hs.m1.fixed_its = 14
def rhs():
    dF, dG, Fk, Gk = hs.advection_rhs_ec(u1, u2, h1, h2, th)
    return hs.momentum_rhs_ec(th, dudz, dudz, vz, vz, Pi, u1, u2, h1, h2, Fx=Fk, Fk=Fk)
rhs(); torch.cuda.synchronize(); t0 = time.perf_counter()
REPS = int(os.environ.get("REPS", "3"))
for _ in range(REPS):
    rhs()
torch.cuda.synchronize()
print("ms/eval %.4f  steps %d  interval [%.4f, %.4f]  ritz errors %s  checks ok %s  worst %.2e" % ((time.perf_counter() - t0) / REPS * 1e3, hs.m1._cheb.steps, hs.m1._cheb.lmin, hs.m1._cheb.lmax, getattr(hs.m1, "ritz_errors", None), hs.verify(), hs.m1.worst_check))
