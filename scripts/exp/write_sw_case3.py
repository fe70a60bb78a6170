"""write the config-3 shallow-water case (24x24x6 p=3 sphere, Galewsky jet) for the C++ hosts: write_sw_case3.py <out.bin> <nsteps>"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mimsem_amd.device import DeviceMesh, Engine
from mimsem_amd.geom import Geom
from mimsem_amd.mesh import CubedSphere, sphere_coords
from mimsem_amd.sweqn import SWEqn, galewsky
from mimsem_amd.topo import Topo
from mimsem_amd.workloads import write_sw_case

ne = 24
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
S = SWEqn(eng, xq[dm.gidq])
uq, hq = galewsky(torch.as_tensor(xq[dm.gidq], device=eng.device))
u, h = S.init1(uq), S.init2(hq)
write_sw_case(sys.argv[1], dm, S.fg[0].cpu().numpy(), u[0].cpu().numpy(), h[0].cpu().numpy(), 360.0, int(sys.argv[2]), 2, False)
