"""Experiment (round 6, late): would a RESTRICTED block preconditioner for the 1-form mass matrix -- every shared edge takes the local solve of ONE
of its two elements instead of the weighted sum of both -- keep the spectrum of P M1 tight enough for the fixed-length Chebyshev iteration?
It would remove the gather epilogue of a sweep (15 of 52 us: the block pass could write z, p and x of the slots its element owns directly).
Dense matrices of a small sphere (p = 3, 4 x 4 x 6 elements, one level) from the device's element matrices; numpy eigenvalues on the host."""
import math, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mimsem_amd.device import DeviceMesh, Engine
from mimsem_amd.geom import Geom
from mimsem_amd.mesh import CubedSphere, sphere_coords
from mimsem_amd.topo import Topo
from mimsem_amd.workloads import SCALE, z_levels

PN, NE, NK = 3, int(os.environ.get("NE", "4")), 1
cs = CubedSphere(PN, NE, 6); coords = sphere_coords(PN, NE)
topos = [Topo(cs, p, NK) for p in range(6)]
geoms = [Geom(t, cs, coords, NK) for t in topos]
for g in geoms:
    g.set_levels(z_levels(NK, g.n0))
dm = DeviceMesh(topos, geoms, nk=NK, numbering="global")
eng = Engine(dm)
n1e, nEl, n1 = eng.n1e, eng.nEl, dm.n1
em = eng.element_matrices("UMAT", lev=0, scale=SCALE, flags=0).view(nEl, 2, 2, n1e, n1e).permute(0, 1, 3, 2, 4).reshape(nEl, 2*n1e, 2*n1e).cpu().numpy()
idx = np.concatenate([np.asarray(dm.inds1x), np.asarray(dm.inds1y)], axis=1)          # [nEl, 24] global slots
M = np.zeros((n1, n1))
for e in range(nEl):
    M[np.ix_(idx[e], idx[e])] += em[e]
mult = np.zeros(n1); np.add.at(mult, idx.reshape(-1), 1.0)
owner = np.full(n1, -1)
for e in range(nEl):
    for s in idx[e]:
        if owner[s] < 0:
            owner[s] = e


def build(local, win, wout):
    """P = sum_e Rout_e^T local_e diag(win) R_e ; wout[e]: weights of the rows the element writes"""
    P = np.zeros((n1, n1))
    for e in range(nEl):
        L = local[e]*win[e][None, :]
        P[np.ix_(idx[e], idx[e])] += wout[e][:, None]*L
    return P


d = 1.0/mult[idx]                                               # [nEl, 24]
own = (owner[idx] == np.arange(nEl)[:, None]).astype(float)     # 1 on the rows the element owns
one = np.ones_like(d)
Minv = np.linalg.inv(em)
Ainv = np.stack([np.linalg.inv(M[np.ix_(idx[e], idx[e])]) for e in range(nEl)])
cases = {
    "today: sum_e R^T D Me^-1 D R (weighted additive, element matrices)": build(Minv, d, d),
    "restricted: owner rows of Me^-1 D R": build(Minv, d, own),
    "restricted: owner rows of Me^-1 R": build(Minv, one, own),
    "restricted, sqrt weights in: owner rows of Me^-1 D^1/2 R": build(Minv, np.sqrt(d), own),
    "classic additive Schwarz: sum_e R^T Ae^-1 R (assembled sub-blocks)": build(Ainv, one, one),
    "classic weighted: sum_e R^T D Ae^-1 R": build(Ainv, one, d),
    "classic RAS: owner rows of Ae^-1 R": build(Ainv, one, own),
}
for name, P in cases.items():
    ev = np.linalg.eigvals(P @ M)
    re, im = ev.real, np.abs(ev.imag)
    lo, hi = re.min(), re.max()
    if lo <= 0:
        print("%-72s  Re in [%.3f, %.3f]  |Im| <= %.3f   NOT positive" % (name, lo, hi, im.max())); continue
    # steps of a Chebyshev iteration on the smallest ellipse (centre d, real foci) around the spectrum for 1e-14: asymptotic rate of the ellipse
    dd, cc = 0.5*(hi + lo), 0.5*(hi - lo)
    a = cc; bax = im.max()
    # ellipse with foci +-c' on the real axis through (a,0) and (0,b): c'^2 = a^2 - b^2 (if b < a)
    if bax < a:
        cf = math.sqrt(a*a - bax*bax)
        rate = (a + bax)/(dd + math.sqrt(dd*dd - cf*cf))
    else:
        rate = float("nan")
    steps = math.ceil(math.log(2e14)/math.log(1.0/rate)) if rate == rate and rate < 1 else -1
    print("%-72s  Re in [%.3f, %.3f]  |Im| <= %.1e  kappa %.2f  rate %.3f  steps(1e-14) %d" % (name, lo, hi, im.max(), hi/lo, rate, steps))
