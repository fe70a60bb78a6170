#!/usr/bin/env python3
"""Experiment: does plain preconditioned Richardson x += P(b - A x) contract on the shallow-water operator A with the coupled
element-block preconditioner (config 3, Galewsky state)?  Prints the residual history and the GMRES iteration count beside it."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mimsem_amd.device import DeviceMesh, Engine
from mimsem_amd.geom import Geom
from mimsem_amd.mesh import CubedSphere, sphere_coords
from mimsem_amd.sweqn import SWEqn, galewsky
from mimsem_amd.topo import Topo
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
for dt in (360.0, 600.0):
    u1, h1 = S.solve(u, h, dt, nits=2, q_exact=False)
    print("dt", dt, "gmres its", S.its)
    qi = S.diagnose_q(dt, u, h)
    f = S.assemble_residual(u, h, u, h, dt, False, None, qi=qi, qj=qi)
    body = S._krylov_body(dt)
    with eng.space("uh"):
        Pb = S.precond_A(-f, dt)
        x = torch.zeros_like(Pb)
        n0 = float(eng.norm(Pb))
        hist = []
        for k in range(40):
            r = Pb - body(x)
            hist.append(float(eng.norm(r)) / n0)
            x = x + r
    print("richardson residual history:", " ".join("%.2e" % v for v in hist))
    print("contraction (geometric mean of the last 10 ratios):", (hist[-1] / hist[-11]) ** 0.1)
