#!/usr/bin/env python3
"""how the refinement of the block-pentadiagonal solve converges: |last correction| / |solution| per column after 1..4 allowed steps,
on the rough random columns of bench.py's `column` extra (config 4 grid) and on the hydrostatic columns of config 5"""
import os, sys
os.environ.setdefault("MIMSEM_EXPERIMENTS", "1")      # (closed-experiment switches are read only under this master switch: DESIGN 9.1)
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from mimsem_amd.device import DeviceMesh, Engine
from mimsem_amd.geom import Geom
from mimsem_amd.mesh import CubedSphere, sphere_coords
from mimsem_amd.topo import Topo
from mimsem_amd.workloads import z_levels
which = sys.argv[1] if len(sys.argv) > 1 else "both"
def report(tag, eng, call):
    for nref in (1, 2, 3, 4):
        os.environ["MIMSEM_REFINE"] = str(nref)
        call()
        nbad, st, ratio = eng.solve_status()
        q = np.quantile(ratio, [0.5, 0.9, 0.99, 1.0])
        print("%s refine<=%d: unconverged %d of %d; ratio p50 %.1e p90 %.1e p99 %.1e max %.1e" % (tag, nref, nbad, ratio.size, *q), flush=True)
    del os.environ["MIMSEM_REFINE"]
if which in ("both", "c4"):
    NK = bench.NK
    cs = CubedSphere(3, 24, 24); coords = sphere_coords(3, 24)
    topos = [Topo(cs, p, NK) for p in range(24)]; geoms = [Geom(t, cs, coords, NK) for t in topos]
    for g in geoms: g.set_levels(z_levels(NK, g.n0))
    dm = DeviceMesh(topos, geoms, nk=NK); eng = Engine(dm)
    rng = np.random.default_rng(0)
    nEl, n2 = dm.nEl, eng.n2e
    area = float(dm.det.mean())*4.0/n2; dz = float(dm.thick.mean())
    lev = lambda nl, lo, hi: eng.tensor(rng.uniform(lo, hi, (nEl, nl*n2))*area*dz)
    theta, rho, rt, pi, eta = lev(NK + 1, 280, 320), lev(NK, 0.5, 1.2), lev(NK, 150, 350), lev(NK, 700, 1000), lev(NK, 5, 6)
    velz = eng.tensor(rng.standard_normal((nEl, (NK - 1)*n2))*0.1*area)
    F = [eng.tensor(rng.standard_normal((nEl, n*n2))*1e8) for n in (NK-1, NK, NK, NK)]
    report("config4/random schur_3  ", eng, lambda: eng.solve_schur_3(75.0, theta, velz, rho, rt, pi, *[f.clone() for f in F]))
    report("config4/random schur_eta", eng, lambda: eng.solve_schur_eta(75.0, theta[:, :NK*n2].contiguous(), rho, eta, pi, *[f.clone() for f in F]))
if which in ("both", "c5"):
    engb, dmb, levs, fld, F, _ = bench.box_column_workload(0, np.random.default_rng(20241024))
    report("config5/hydrostatic schur_3  ", engb, lambda: engb.solve_schur_3(0.5, fld["theta"], fld["velz"], fld["rho"], fld["rt"], fld["pi"], *[f.clone() for f in F], flags=3))
    report("config5/hydrostatic schur_eta", engb, lambda: engb.solve_schur_eta(0.5, fld["thetaL"], fld["rho"], fld["eta"], fld["pi"], *[f.clone() for f in F]))
