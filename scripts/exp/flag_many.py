"""How to get MORE THAN 64 columns flagged by the unpivoted block sweep (status 1) yet solvable by the pivoted band LU -- the input the ownership
test of k_band_lu_wave needs (tests/test_gpu_fullsize.py).  Tries a few ways of worsening the conditioning of 200 chosen columns."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from tests.helpers import z_levels
from mimsem_amd.device import DeviceMesh, Engine
from mimsem_amd.geom import Geom
from mimsem_amd.mesh import CubedSphere, sphere_coords
from mimsem_amd.topo import Topo
PN, NE, NK, NPATCH = 3, 24, 30, 24
cs = CubedSphere(PN, NE, NPATCH); coords = sphere_coords(PN, NE)
topos = [Topo(cs, p, NK) for p in range(NPATCH)]
geoms = [Geom(t, cs, coords, NK) for t in topos]
for g in geoms:
    g.set_levels(z_levels(NK, g.n0))
dm = DeviceMesh(topos, geoms, nk=NK, numbering="global")
eng = Engine(dm)
n2, nEl = eng.n2e, dm.nEl
area = float(dm.det.mean()) * 4.0 / n2; dz = float(dm.thick.mean())
chosen = np.arange(5, nEl, 17)[:200]
for name, fn in (("none", None),
                 ("rho_lev5_x1e-6", lambda a: a["rho"].__setitem__((chosen, slice(5 * n2, 6 * n2)), a["rho"][chosen, 5 * n2:6 * n2] * 1e-6)),
                 ("rho_lev5_x1e-9", lambda a: a["rho"].__setitem__((chosen, slice(5 * n2, 6 * n2)), a["rho"][chosen, 5 * n2:6 * n2] * 1e-9)),
                 ("theta_lev5_x1e4", lambda a: a["theta"].__setitem__((chosen, slice(5 * n2, 6 * n2)), a["theta"][chosen, 5 * n2:6 * n2] * 1e4)),
                 ("pi_lev5_x1e-6", lambda a: a["pi"].__setitem__((chosen, slice(5 * n2, 6 * n2)), a["pi"][chosen, 5 * n2:6 * n2] * 1e-6)),
                 ("pi_all_x1e3", lambda a: a["pi"].__setitem__((chosen, slice(None)), a["pi"][chosen] * 1e3)),
                 ("rho_dof0_x1e-7", lambda a: a["rho"].__setitem__((chosen, slice(5 * n2, 5 * n2 + 1)), a["rho"][chosen, 5 * n2:5 * n2 + 1] * 1e-7))):
    rng = np.random.default_rng(5)
    lev = lambda nl, lo, hi: rng.uniform(lo, hi, (nEl, nl * n2)) * area * dz
    a = {"theta": lev(NK, 280, 320), "rho": lev(NK, 0.5, 1.2), "eta": lev(NK, 5, 6), "pi": lev(NK, 700, 1000)}
    F0 = [rng.standard_normal((nEl, n * n2)) * 1e8 for n in (NK - 1, NK, NK, NK)]
    if fn:
        fn(a)
    t = {k: eng.tensor(v) for k, v in a.items()}
    for mode in (0, 1):
        eng.set_pivot_fallback(mode)
        eng.solve_schur_eta(75.0, t["theta"], t["rho"], t["eta"], t["pi"], *[eng.tensor(x) for x in F0])
        nb, st, ratio = eng.solve_status()
        print(name, "fallback", mode, "unresolved", nb, "status counts", {int(k): int((st == k).sum()) for k in np.unique(st)},
              "flagged among chosen", int(np.isin(np.nonzero(st != 0)[0], chosen).sum()), flush=True)
# ---- the fields of the naturally flagged columns copied into 200 others (the geometry of the target element differs)
rng = np.random.default_rng(5)
lev = lambda nl, lo, hi: rng.uniform(lo, hi, (nEl, nl * n2)) * area * dz
a = {"theta": lev(NK, 280, 320), "rho": lev(NK, 0.5, 1.2), "eta": lev(NK, 5, 6), "pi": lev(NK, 700, 1000)}
F0 = [rng.standard_normal((nEl, n * n2)) * 1e8 for n in (NK - 1, NK, NK, NK)]
eng.set_pivot_fallback(0)
eng.solve_schur_eta(75.0, *[eng.tensor(a[k]) for k in ("theta", "rho", "eta", "pi")], *[eng.tensor(x) for x in F0])
nb, st, ratio = eng.solve_status()
src = np.nonzero(st == 1)[0]
print("naturally flagged:", src.tolist())
for scale_geom in (False, True):
    b = {k: v.copy() for k, v in a.items()}; G0 = [x.copy() for x in F0]
    for i, e in enumerate(chosen):
        s_ = src[i % len(src)]
        fac = (dm.det[e].mean() / dm.det[s_].mean()) if scale_geom else 1.0      # fields are integrals over the element: rescale to the target's area
        for k in b:
            b[k][e] = a[k][s_] * fac
        for x, y in zip(G0, F0):
            x[e] = y[s_]
    for mode in (0, 1):
        eng.set_pivot_fallback(mode)
        eng.solve_schur_eta(75.0, *[eng.tensor(b[k]) for k in ("theta", "rho", "eta", "pi")], *[eng.tensor(x) for x in G0])
        nb, st, ratio = eng.solve_status()
        print("copied fields (area-scaled %s) fallback %d unresolved %d status counts %s flagged among chosen %d" %
              (scale_geom, mode, nb, {int(k): int((st == k).sum()) for k in np.unique(st)}, int(np.isin(np.nonzero(st != 0)[0], chosen).sum())), flush=True)
eng.set_pivot_fallback(1)
