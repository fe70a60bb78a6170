"""Ritz values of P A (the coupled element-block preconditioner times the [u|h] operator of the SW Picard step) from an Arnoldi process:
where does the spectrum lie (real interval -> Chebyshev; vertical segment around 1 -> gravity-wave pairs)?"""
import os, sys
os.environ.setdefault("MIMSEM_EXPERIMENTS", "1")      # (closed-experiment switches are read only under this master switch: DESIGN 9.1)
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
os.environ["MIMSEM_SW_POLY"] = "1"
from mimsem_amd.device import DeviceMesh, Engine
from mimsem_amd.geom import Geom
from mimsem_amd.mesh import CubedSphere, sphere_coords
from mimsem_amd.sweqn import SWEqn, galewsky
from mimsem_amd.topo import Topo
ne, dt = int(os.environ.get("SW_NE", "24")), float(os.environ.get("SW_DT", "360"))
cs = CubedSphere(3, ne, 6); coords = sphere_coords(3, ne)
topos = [Topo(cs, p, 1) for p in range(6)]; geoms = [Geom(t, cs, coords, 1, signed_det=True) for t in topos]
for g in geoms: g.set_levels(np.stack([np.zeros(g.n0), np.ones(g.n0)]))
dm = DeviceMesh(topos, geoms, nk=1, numbering="global"); eng = Engine(dm)
xq = np.zeros((dm.nq, 3))
for g in geoms: xq[g.loc0] = coords[g.loc0]
S = SWEqn(eng, xq[dm.gidq])
uq, hq = galewsky(torch.as_tensor(xq[dm.gidq], device=eng.device))
u, h = S.init1(uq), S.init2(hq)
u, h = S.solve(u, h, dt, nits=2, q_exact=False)
body = S._krylov_body1(dt)
n = S.n1 + S.n2
m = 60
rng = torch.Generator(device=eng.device); rng.manual_seed(1)
V = torch.zeros(m + 1, n, dtype=torch.float64, device=eng.device); H = np.zeros((m + 1, m))
v = torch.randn(1, n, dtype=torch.float64, device=eng.device, generator=rng); V[0] = (v / torch.linalg.vector_norm(v))[0]
for j in range(m):
    w = body(V[j:j + 1].contiguous()).reshape(-1)
    for _ in range(2):
        hh = V[:j + 1] @ w; w = w - hh @ V[:j + 1]; H[:j + 1, j] += hh.cpu().numpy()
    H[j + 1, j] = float(torch.linalg.vector_norm(w)); V[j + 1] = w / H[j + 1, j]
ev = np.linalg.eigvals(H[:m, :m])
print("Ritz values of P A (m = %d): Re in [%.3f, %.3f], |Im| max %.3f" % (m, ev.real.min(), ev.real.max(), np.abs(ev.imag).max()))
print("largest |1 - lambda| %.3f; count with |Im| > 0.05: %d of %d" % (np.abs(1 - ev).max(), int((np.abs(ev.imag) > 0.05).sum()), m))
order = np.argsort(-np.abs(1 - ev))
print("outermost:", ", ".join("%.3f%+.3fi" % (ev[i].real, ev[i].imag) for i in order[:12]))
