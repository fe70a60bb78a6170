#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE in SEPARATE runs, MI355X_MICROARCH.md
section HBM): a calibration launch with a known byte count in this code's own access width (8 B per lane:
k_halo_pack with an identity index list over a 512 MiB vector), then the benchmark step (Umat apply on the
24x24x6 x 30-level sphere) and its larger-than-cache variant (8 replicas).  scripts/pmc_summarise.py turns the
counter CSVs into bytes per launch."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mimsem_amd.device import DeviceMesh, Engine  # noqa: E402
from mimsem_amd.geom import Geom  # noqa: E402
from mimsem_amd.mesh import CubedSphere, sphere_coords  # noqa: E402
from mimsem_amd.topo import Topo  # noqa: E402
from mimsem_amd.workloads import z_levels  # noqa: E402

PN, NE, NPATCH, NK = bench.PN, bench.NE, bench.NPATCH, bench.NK
cs = CubedSphere(PN, NE, NPATCH)
coords = sphere_coords(PN, NE)
topos = [Topo(cs, p, NK) for p in range(NPATCH)]
geoms = [Geom(t, cs, coords, NK) for t in topos]
for g in geoms:
    g.set_levels(z_levels(NK, g.n0))
dm = DeviceMesh(topos, geoms, nk=NK, numbering="global")
eng = Engine(dm)
rng = np.random.default_rng(1)

# calibration: 64 Mi doubles read through an identity index list, written contiguously
n = 64 * 1024 * 1024
v = torch.randn(n, dtype=torch.float64, device="cuda")
idx = torch.arange(n, dtype=torch.int32, device="cuda")
for _ in range(3):
    buf = eng.halo_pack(idx, v)
torch.cuda.synchronize()
del buf, v, idx

x = eng.tensor(rng.standard_normal((NK, dm.n1))); y = eng.zeros(NK, dm.n1)
for _ in range(10):
    eng.apply("UMAT", x, lev0=0, scale=bench.SCALE, flags=1, out=y)
torch.cuda.synchronize()

R = 8
dmc = bench.replicate(dm, R)
engc = Engine(dmc)
xc = engc.tensor(rng.standard_normal((NK, dmc.n1))); yc = engc.zeros(NK, dmc.n1)
for _ in range(6):
    engc.apply("UMAT", xc, lev0=0, scale=bench.SCALE, flags=1, out=yc)
torch.cuda.synchronize()
# the other families of SURVEY 8(d) on the same two engines (bench.py families / families_cold), a few launches each
for e_, dm_, xx, yy in ((eng, dm, x, y), (engc, dmc, xc, yc)):
    x2 = e_.tensor(rng.standard_normal((NK, dm_.n2))); hh = e_.tensor(rng.uniform(1, 2, (NK, dm_.n2))*1e3)
    q0 = e_.tensor(rng.standard_normal((NK, dm_.n0))*1e-4); y2 = e_.zeros(NK, dm_.n2)
    for row, op, fin, fcf, fl in bench.FAMILIES:
        if op == "UMAT":
            continue
        xin = xx if fin == 1 else x2
        f = {None: None, 0: q0, 1: xx, 2: hh}[fcf]
        o = yy if op in ("UHMAT", "ROTMAT") else y2
        for _ in range(4):
            e_.apply(op, xin, f=f, lev0=0, scale=bench.SCALE, flags=fl, out=o)
    torch.cuda.synchronize()
del engc, eng
# config 5's element half: p = 4 box, cache resident and on 8 boxes
from mimsem_amd.geom import BoxGeom  # noqa: E402
from mimsem_amd.mesh import PeriodicBox, box_coords  # noqa: E402
bx = PeriodicBox(4, 32, 4); bc = box_coords(4, 32, 1000.0); nkb = 64
bt = [Topo(bx, p, nkb) for p in range(4)]; bg = [BoxGeom(t, bx, bc, nkb, 1000.0) for t in bt]
for g in bg:
    g.set_levels(np.repeat(np.linspace(0.0, 1500.0, nkb + 1)[:, None], g.n0, axis=1))
dmb = DeviceMesh(bt, bg, nk=nkb, numbering="global")
for dm_ in (dmb, bench.replicate(dmb, 8)):
    e_ = Engine(dm_)
    xb = e_.tensor(rng.standard_normal((nkb, dm_.n1))); yb = e_.zeros(nkb, dm_.n1)
    for _ in range(5):
        e_.apply("UMAT", xb, lev0=0, scale=bench.SCALE, flags=1, out=yb)
    torch.cuda.synchronize()
    del e_
print("pmc workload done")
