"""how good are the extrapolated initial guesses of the SW step's solves?  prints, per step and Picard iteration, the GMRES iteration count of
the [u|h] solve and the relative size of the initial residual (MIMSEM_SW_EXTRAPOLATE = 0 | 1 | 2)"""
import os, sys
os.environ.setdefault("MIMSEM_EXPERIMENTS", "1")      # (closed-experiment switches are read only under this master switch: DESIGN 9.1)
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from mimsem_amd.device import DeviceMesh, Engine
from mimsem_amd.geom import Geom
from mimsem_amd.mesh import CubedSphere, sphere_coords
from mimsem_amd.sweqn import SWEqn, galewsky
from mimsem_amd.topo import Topo
import mimsem_amd.krylov as K
ne, dt = 24, 360.0
cs = CubedSphere(3, ne, 6); coords = sphere_coords(3, ne)
topos = [Topo(cs, p, 1) for p in range(6)]; geoms = [Geom(t, cs, coords, 1, signed_det=True) for t in topos]
for g in geoms: g.set_levels(np.stack([np.zeros(g.n0), np.ones(g.n0)]))
dm = DeviceMesh(topos, geoms, nk=1, numbering="global"); eng = Engine(dm)
xq = np.zeros((dm.nq, 3))
for g in geoms: xq[g.loc0] = coords[g.loc0]
S = SWEqn(eng, xq[dm.gidq])
uq, hq = galewsky(torch.as_tensor(xq[dm.gidq], device=eng.device))
u, h = S.init1(uq), S.init2(hq)
orig = K.GraphedGMRES.solve
def spy(self, apply_A, b, precond, x0=None, **kw):
    if x0 is not None:
        r0 = float(torch.linalg.vector_norm(precond(b - apply_A(x0)))) / float(torch.linalg.vector_norm(precond(b)))
    else:
        r0 = 1.0
    out = orig(self, apply_A, b, precond, x0=x0, **kw)
    print("   gmres n=%d its %d  |r0|/|Pb| %.2e" % (b.numel(), out[1], r0))
    return out
K.GraphedGMRES.solve = spy
for step in range(6):
    print("step", step)
    u, h = S.solve(u, h, dt, nits=2, q_exact=False)
    print("   its", dict(S.its))
