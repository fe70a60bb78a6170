#!/usr/bin/env python3
"""true residual |F - L d| / |F| of solve_schur_column_3's pentadiagonal system per column: fused path vs round 2's chain + super-block sweep,
on bench.py's random column workload and on a hydrostatic config-4 state"""
import os, sys
os.environ.setdefault("MIMSEM_EXPERIMENTS", "1")      # (closed-experiment switches are read only under this master switch: DESIGN 9.1)
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from mimsem_amd.device import DeviceMesh, Engine
from mimsem_amd.geom import Geom, gll_points
from mimsem_amd.mesh import CubedSphere, sphere_coords
from mimsem_amd.topo import Topo
from mimsem_amd.workloads import z_levels
NK = bench.NK
cs = CubedSphere(3, 24, 24); coords = sphere_coords(3, 24)
topos = [Topo(cs, p, NK) for p in range(24)]; geoms = [Geom(t, cs, coords, NK) for t in topos]
for g in geoms: g.set_levels(z_levels(NK, g.n0))
dm = DeviceMesh(topos, geoms, nk=NK); eng = Engine(dm)
rng = np.random.default_rng(0)
nEl, n2 = dm.nEl, eng.n2e
def resid(tag, theta, velz, rho, rt, pi, F):
    for env in ({}, {"MIMSEM_SCHUR3_CHAIN": "1", "MIMSEM_SCHUR3_SUPERBLOCKS": "1"}):
        os.environ.update(env)
        Fc = [f.clone() for f in F]
        d_u, d_rho, d_rt, d_pi, L = eng.solve_schur_3(75.0, theta, velz, rho, rt, pi, *Fc, want_L=True)
        for k in env: del os.environ[k]
        d = d_rt.view(nEl, NK, n2); f = Fc[2].view(nEl, NK, n2)
        Ld = torch.zeros_like(d)
        for b in range(5):
            off = b - 2
            lo, hi = max(0, -off), min(NK, NK - off)
            Ld[:, lo:hi] += torch.einsum("ekij,ekj->eki", L[:, lo:hi, b], d[:, lo + off:hi + off])
        res = (torch.linalg.norm((f - Ld).reshape(nEl, -1), dim=1)/torch.linalg.norm(f.reshape(nEl, -1), dim=1)).cpu().numpy()
        q = np.quantile(res, [0.5, 0.9, 0.99, 1.0])
        print("%s %-22s residual p50 %.1e p90 %.1e p99 %.1e max %.1e" % (tag, "chain+superblocks" if env else "fused+penta", *q), flush=True)
area = float(dm.det.mean())*4.0/n2; dz = float(dm.thick.mean())
lev = lambda nl, lo, hi: eng.tensor(rng.uniform(lo, hi, (nEl, nl*n2))*area*dz)
F = [eng.tensor(rng.standard_normal((nEl, n*n2))*1e8) for n in (NK-1, NK, NK, NK)]
resid("bench-random ", lev(NK + 1, 280, 320), eng.tensor(rng.standard_normal((nEl, (NK - 1)*n2))*0.1*area), lev(NK, 0.5, 1.2), lev(NK, 150, 350), lev(NK, 700, 1000), F)
resid("test-random  ", lev(NK + 1, 280, 320)/dz, lev(NK - 1, -1, 1)/dz, lev(NK, 0.5, 1.2), lev(NK, 250, 400), lev(NK, 700, 1000), F)
# hydrostatic
wd = np.diff(gll_points(3)); wj = np.outer(wd, wd).ravel()
detm = dm.det.mean(axis=1); thm = dm.thick.mean(axis=2).T
zi = np.mean([g.levs.mean(axis=1) for g in geoms], axis=0); zm = 0.5*(zi[1:] + zi[:-1])
th_v = 300.0 + 0.004*zm; thI_v = 300.0 + 0.004*zi
pi_v = 1004.5 - (9.80616/0.004)*np.log(th_v/300.0)
rho_v = (1.0e5/287.0)*(pi_v/1004.5)**(717.5/287.0)/th_v
pert = lambda nl: 1.0 + 1e-2*rng.standard_normal((nEl, nl*n2))
levh = lambda v: eng.tensor((detm[:, None, None]*thm[:, :, None]*v[None, :, None]*wj[None, None, :]).reshape(nEl, NK*n2)*pert(NK))
itf = lambda v, nl: (detm[:, None, None]*v[None, :nl, None]*wj[None, None, :]).reshape(nEl, nl*n2)*pert(nl)
resid("hydrostatic  ", eng.tensor(itf(thI_v, NK + 1)), eng.tensor(itf(np.ones(NK + 1), NK - 1)*0.5*rng.standard_normal((nEl, (NK - 1)*n2))), levh(rho_v), levh(rho_v*th_v), levh(pi_v), F)
nbad, st, ratio = eng.solve_status()
