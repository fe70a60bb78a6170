"""Where a config-3 shallow-water Picard iteration spends its time: residual assembly (its Krylov solves) vs the packed [u,h] GMRES."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mimsem_amd.device import DeviceMesh, Engine
from mimsem_amd.geom import Geom
from mimsem_amd.mesh import CubedSphere, sphere_coords
from mimsem_amd.sweqn import SWEqn, williamson2
from mimsem_amd.topo import Topo

ne = int(os.environ.get("SW_NE", "24"))
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
uq, hq = williamson2(torch.as_tensor(xq[dm.gidq], device=eng.device), alpha=0.0)
u, h = S.init1(uq), S.init2(hq)
dt = 360.0
u1, h1 = S.solve(u, h, dt, nits=2, q_exact=False)


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


f = S.assemble_residual(u, h, u1, h1, dt)
print("assemble_residual      %.3f ms" % timeit(lambda: S.assemble_residual(u, h, u1, h1, dt)))
print("  diagnose_F           %.3f ms  (cg its %d)" % (timeit(lambda: S.diagnose_F(u, u1, h, h1)), S.its["F"]))
print("  diagnose_Phi         %.3f ms" % timeit(lambda: S.diagnose_Phi(u, u1, h, h1)))
print("  diagnose_q (upwind)  %.3f ms  (gmres its %d)" % (timeit(lambda: S.diagnose_q(dt, u, h)), S.its["q"]))
F = S.diagnose_F(u, u1, h, h1); q = S.diagnose_q(dt, u, h)
print("  R_up                 %.3f ms" % timeit(lambda: S.R_up(q, u, dt, F)))
g = S._gA[1]
res = g.solve(lambda v: S.apply_A(v, dt), -f, lambda r: S.precond_A(r, dt), rtol=S.rtol, maxit=1000)
print("gmres A                %.3f ms  (its %d)" % (timeit(lambda: g.solve(lambda v: S.apply_A(v, dt), -f, lambda r: S.precond_A(r, dt), rtol=S.rtol, maxit=1000)), res[1]))
x = S.pack(u, h)
print("  apply_A              %.3f ms" % timeit(lambda: S.apply_A(x, dt), 20))
print("  precond_A            %.3f ms" % timeit(lambda: S.precond_A(x, dt), 20))
print("  graph replay j=30    %.3f ms" % timeit(lambda: g._graph(30).replay(), 20))
print("  graph replay j=5     %.3f ms" % timeit(lambda: g._graph(5).replay(), 20))
print("  graph replay j=55    %.3f ms" % timeit(lambda: g._graph(55).replay(), 20))
