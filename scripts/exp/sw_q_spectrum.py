"""Ritz values of diag(M0h)^-1 M0h_up (the upwinded lumped 0-form mass of the potential-vorticity solve, src/SWEqn_Picard.cpp:322-341) -- is a
Chebyshev semi-iteration with fixed bounds possible there too?  (the operator changes every step with h and u)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from mimsem_amd.device import DeviceMesh, Engine
from mimsem_amd.geom import Geom
from mimsem_amd.krylov import arnoldi_ritz
from mimsem_amd.mesh import CubedSphere, sphere_coords
from mimsem_amd.sweqn import SWEqn, galewsky, UP_TAU
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
for step in range(0, 9):
    if step in (0, 4, 8):
        m0h = eng.pvec(0, 1, 1.0, h2=h)
        body = lambda v: eng.apply_up("PHMAT_UP", v, h, u, fac=UP_TAU, dt=dt) / m0h
        ev = arnoldi_ritz(body, dm.n0, 40, eng.device)
        print("step %d: Re in [%.4f, %.4f], |Im| max %.4f, |1 - lambda| max %.4f" % (step, ev.real.min(), ev.real.max(), np.abs(ev.imag).max(), np.abs(1 - ev).max()))
    u, h = S.solve(u, h, dt, nits=2, q_exact=False)
