#!/usr/bin/env python3
"""Distribution of the componentwise backward error  w = max_i |f - L d|_i / (|L| |d| + |f|)_i  of solve_schur_column_eta's Helmholtz solve on
the bench's 3 456 rough random columns, WITHOUT refinement and after 1 / 2 steps: how many columns leave the unpivoted block elimination at
round-off level already (LAPACK's xGERFS would not refine them) -- the question behind skipping the refinement's second solve per column"""
import os, sys
os.environ.setdefault("MIMSEM_EXPERIMENTS", "1")      # (closed-experiment switches are read only under this master switch: DESIGN 9.1)
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from mimsem_amd.device import DeviceMesh, Engine
from mimsem_amd.geom import Geom
from mimsem_amd.mesh import CubedSphere, sphere_coords
from mimsem_amd.topo import Topo
from mimsem_amd.workloads import z_levels
NK = 30
cs = CubedSphere(3, 24, 24); coords = sphere_coords(3, 24)
topos = [Topo(cs, p, NK) for p in range(24)]; geoms = [Geom(t, cs, coords, NK) for t in topos]
for g in geoms: g.set_levels(z_levels(NK, g.n0))
dm = DeviceMesh(topos, geoms, nk=NK); eng = Engine(dm)
nEl, n2 = dm.nEl, eng.n2e
area = float(dm.det.mean())*4.0/n2; dz = float(dm.thick.mean())
rng = np.random.default_rng(77)
lev = lambda nl, lo, hi: eng.tensor(rng.uniform(lo, hi, (nEl, nl*n2))*area*dz)
theta, rho, eta, pi = lev(NK, 280, 320), lev(NK, 0.5, 1.2), lev(NK, 5, 6), lev(NK, 700, 1000)
F0 = [rng.standard_normal((nEl, n*n2))*1e8 for n in (NK-1, NK, NK, NK)]
L = eng.helmholtz_blocks(75.0, theta, rho, eta, pi).view(nEl, NK, 3, n2, n2)
def band(M, v):
    y = torch.einsum("ekij,ekj->eki", M[:, :, 1], v)
    y[:, 1:] += torch.einsum("ekij,ekj->eki", M[:, 1:, 0], v[:, :-1])
    y[:, :-1] += torch.einsum("ekij,ekj->eki", M[:, :-1, 2], v[:, 1:])
    return y
ref = None
for nref in (0, 1, 2, 4):
    os.environ["MIMSEM_REFINE"] = str(nref)
    os.environ["MIMSEM_COLUMN_PIVOT_FALLBACK"] = "0"
    F = [eng.tensor(f) for f in F0]
    out = eng.solve_schur_eta(75.0, theta, rho, eta, pi, *F)
    d = out[3].view(nEl, NK, n2); rhs = F[3].view(nEl, NK, n2)          # F_pi as the sweep left it = the Helmholtz right-hand side
    r = (rhs - band(L, d)).abs()
    s = band(L.abs(), d.abs()) + rhs.abs()
    w = (r/s).amax(dim=(1, 2)).cpu().numpy()
    q = np.percentile(w, [10, 50, 90, 99, 100])
    eps = np.finfo(float).eps
    msg = "refine %d: backward error percentiles 10/50/90/99/100 %% = %s ; columns below 2 eps: %d, 4 eps: %d, 16 eps: %d of %d" % (
        nref, " ".join("%.1e" % v for v in q), (w <= 2*eps).sum(), (w <= 4*eps).sum(), (w <= 16*eps).sum(), nEl)
    if nref == 4:
        ref = d.clone()
    print(msg)
# forward error of the un-refined solution against the refined one
os.environ["MIMSEM_REFINE"] = "0"
F = [eng.tensor(f) for f in F0]
d0 = eng.solve_schur_eta(75.0, theta, rho, eta, pi, *F)[3].view(nEl, NK, n2)
fe = (torch.linalg.vector_norm(d0 - ref, dim=(1, 2))/torch.linalg.vector_norm(ref, dim=(1, 2))).cpu().numpy()
print("un-refined vs 4x refined solution, relative difference percentiles 10/50/90/99/100 %% = " + " ".join("%.1e" % v for v in np.percentile(fe, [10, 50, 90, 99, 100])))

# ---- round 6: WHICH property of the block elimination sets that backward error?  The same block-Thomas recurrence in numpy (true divisions
# throughout) on 400 of the columns, with the diagonal blocks handled three ways: (a) explicit inverse by UNPIVOTED Gauss-Jordan (the kernel's
# arithmetic), (b) explicit inverse by LU with partial pivoting (numpy.linalg.inv = LAPACK getrf + getri): "pivot inside the diagonal block",
# (c) no explicit inverse: every product with D^-1 is a pivoted-LU SOLVE (getrf + getrs) -- the backward-stable form.
sel = np.arange(0, nEl, nEl//400)[:400]
Ls = L[sel].cpu().numpy(); fs = F[3].view(nEl, NK, n2)[sel].cpu().numpy()     # (F[3] of the last call: the right-hand side the sweep left)
def gj_unpivoted(A):
    A = A.copy(); n = A.shape[-1]; X = np.broadcast_to(np.eye(n), A.shape).copy()
    for p in range(n):
        piv = A[:, p, p][:, None]
        rowA, rowX = A[:, p, :]/piv, X[:, p, :]/piv
        fac = A[:, :, p].copy(); fac[:, p] = 0.0
        A -= fac[:, :, None]*rowA[:, None, :]; X -= fac[:, :, None]*rowX[:, None, :]
        A[:, p, :], X[:, p, :] = rowA, rowX
    return X
def thomas(kind):
    nc = Ls.shape[0]
    G = np.zeros((nc, NK, n2, n2)); g = np.zeros((nc, NK, n2))
    Dp = Ls[:, 0, 1].copy(); fp = fs[:, 0].copy()
    for k in range(NK):
        if kind == "solve":
            sol = np.linalg.solve(Dp, np.concatenate([Ls[:, k, 2], fp[:, :, None]], axis=2))
            G[:, k], g[:, k] = sol[:, :, :n2], sol[:, :, n2]
        else:
            inv = gj_unpivoted(Dp) if kind == "gj" else np.linalg.inv(Dp)
            G[:, k] = inv @ Ls[:, k, 2]; g[:, k] = np.einsum("eij,ej->ei", inv, fp)
        if k + 1 < NK:
            Dp = Ls[:, k + 1, 1] - Ls[:, k + 1, 0] @ G[:, k]
            fp = fs[:, k + 1] - np.einsum("eij,ej->ei", Ls[:, k + 1, 0], g[:, k])
    d = np.zeros((nc, NK, n2)); d[:, NK - 1] = g[:, NK - 1]
    for k in range(NK - 2, -1, -1):
        d[:, k] = g[:, k] - np.einsum("eij,ej->ei", G[:, k], d[:, k + 1])
    return d
def bw(d):
    def bandn(M, v):
        y = np.einsum("ekij,ekj->eki", M[:, :, 1], v)
        y[:, 1:] += np.einsum("ekij,ekj->eki", M[:, 1:, 0], v[:, :-1]); y[:, :-1] += np.einsum("ekij,ekj->eki", M[:, :-1, 2], v[:, 1:])
        return y
    r = np.abs(fs - bandn(Ls, d)); s = bandn(np.abs(Ls), np.abs(d)) + np.abs(fs)
    return (r/s).max(axis=(1, 2))
conds = np.linalg.cond(Ls[:, :, 1].reshape(-1, n2, n2))
print("condition numbers of the raw diagonal blocks: percentiles 50/90/100 %% = " + " ".join("%.1e" % v for v in np.percentile(conds, [50, 90, 100])))
for kind, what in (("gj", "(a) explicit inverse, unpivoted Gauss-Jordan"), ("inv", "(b) explicit inverse, LU with partial pivoting"), ("solve", "(c) pivoted-LU solves, no explicit inverse")):
    w = bw(thomas(kind))
    print("numpy block-Thomas %-52s backward error percentiles 10/50/90/100 %% = %s ; <= 4 eps: %d of %d" %
          (what, " ".join("%.1e" % v for v in np.percentile(w, [10, 50, 90, 100])), (w <= 4*eps).sum(), len(w)))
