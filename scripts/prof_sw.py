"""Profile target: a few shallow-water Picard steps on the config-3 grid (run under rocprofv3 --kernel-trace --stats)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mimsem_amd.device import DeviceMesh, Engine
from mimsem_amd.geom import Geom
from mimsem_amd.mesh import CubedSphere, sphere_coords
from mimsem_amd.sweqn import SWEqn, williamson2
from mimsem_amd.topo import Topo

ne = int(os.environ.get("SW_NE", "24")); graphs = os.environ.get("SW_GRAPHS", "1") == "1"
cs = CubedSphere(3, ne, 6); coords = sphere_coords(3, ne)
topos = [Topo(cs, p, 1) for p in range(6)]
geoms = [Geom(t, cs, coords, 1, signed_det=True) for t in topos]
for g in geoms:
    g.set_levels(np.stack([np.zeros(g.n0), np.ones(g.n0)]))
dm = DeviceMesh(topos, geoms, nk=1, numbering="global")
eng = Engine(dm)
xq = np.zeros((dm.nq, 3))
for g in geoms:
    xq[g.loc0] = coords[g.loc0]
S = SWEqn(eng, xq[dm.gidq], use_graphs=graphs)
uq, hq = williamson2(torch.as_tensor(xq[dm.gidq], device=eng.device), alpha=0.0)
u, h = S.init1(uq), S.init2(hq)
for _ in range(3):                       # warm-up: graph captures, adaptive sweep counts settle
    u, h = S.solve(u, h, 360.0, nits=2, q_exact=False)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    u, h = S.solve(u, h, 360.0, nits=2, q_exact=False)
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) / 5 * 1e3, S.its)
