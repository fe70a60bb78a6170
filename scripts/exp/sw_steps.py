"""config-3 SW steps/s (Galewsky, 24x24x6, dt = 360 s, 2 Picard iterations) over 30 timed steps, for A/B runs of the solver switches"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from mimsem_amd.device import DeviceMesh, Engine
from mimsem_amd.geom import Geom
from mimsem_amd.mesh import CubedSphere, sphere_coords
from mimsem_amd.sweqn import SWEqn, galewsky
from mimsem_amd.topo import Topo
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
for _ in range(4): u, h = S.solve(u, h, dt, nits=2, q_exact=False)
best = []
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): u, h = S.solve(u, h, dt, nits=2, q_exact=False)
    torch.cuda.synchronize(); best.append(10 / (time.perf_counter() - t0))
print("%s steps/s %s  its %s" % (os.environ.get("TAG", ""), " ".join("%.1f" % b for b in best), dict(S.its)))
