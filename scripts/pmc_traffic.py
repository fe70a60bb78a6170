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
from tests.helpers import z_levels  # noqa: E402

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
print("pmc workload done")
