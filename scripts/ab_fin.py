"""A/B + stress of the in-kernel finishing phase of k_apply_wave (round 3 experiment, MIMSEM_WAVE_FIN=1) against the default perimeter pass:
  * bitwise equality of y over many launches whose inputs CHANGE from launch to launch (a stale partial sum would show), plain and
    accumulate form, on the config-4 sphere (cache resident) and on R spheres (HBM resident, every XCD pairing occurs);
  * HIP-event time of the operator's kernels, both forms.
usage: python scripts/ab_fin.py [R=8] [rounds=60]"""
import os
os.environ.setdefault("MIMSEM_EXPERIMENTS", "1")      # (closed-experiment switches are read only under this master switch: DESIGN 9.1)
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402
from mimsem_amd.device import DeviceMesh, Engine  # noqa: E402
from mimsem_amd.geom import Geom  # noqa: E402
from mimsem_amd.mesh import CubedSphere, sphere_coords  # noqa: E402
from mimsem_amd.topo import Topo  # noqa: E402
from mimsem_amd.workloads import SCALE, z_levels  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 60
PN, NE, NPATCH, NK = 3, 24, 24, 30


def engines(dm):
    os.environ["MIMSEM_WAVE_FIN"] = "1"
    try:
        fin = Engine(dm)
    finally:
        del os.environ["MIMSEM_WAVE_FIN"]
    return fin, Engine(dm)


def timed(eng, call, steps=40):
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    eng.set_profiling(1); t = time.perf_counter()
    for _ in range(steps):
        call()
    torch.cuda.synchronize(); wall = (time.perf_counter() - t) / steps
    c1, c2, cn = eng.profile_read(); eng.set_profiling(0)
    return c1 / cn * 1e3, c2 / cn * 1e3, wall * 1e6


cs = CubedSphere(PN, NE, NPATCH); coords = sphere_coords(PN, NE)
topos = [Topo(cs, p, NK) for p in range(NPATCH)]
geoms = [Geom(t, cs, coords, NK) for t in topos]
for g in geoms:
    g.set_levels(z_levels(NK, g.n0))
dm1 = DeviceMesh(topos, geoms, nk=NK, numbering="global")
bad = 0
for name, dm in (("1 sphere", dm1), ("%d spheres" % R, bench.replicate(dm1, R))):
    fin, per = engines(dm)
    rng = np.random.default_rng(3)
    xs = [fin.tensor(rng.standard_normal((NK, dm.n1))) for _ in range(3)]
    yf, yp = fin.zeros(NK, dm.n1), per.zeros(NK, dm.n1)
    for it in range(ROUNDS):
        x = xs[it % 3] * float(1 + it)
        for flags in (1, 3):                                   # plain, accumulate
            if flags == 3:
                yf.copy_(xs[(it + 1) % 3]); yp.copy_(xs[(it + 1) % 3])
            nl = NK if it % 4 else 1 + it % NK                 # ragged level counts too
            fin.apply("UMAT", x[:nl], lev0=0, scale=SCALE, flags=flags, alpha=0.5 if flags == 3 else 1.0, out=yf[:nl])
            per.apply("UMAT", x[:nl], lev0=0, scale=SCALE, flags=flags, alpha=0.5 if flags == 3 else 1.0, out=yp[:nl])
            if not torch.equal(yf[:nl], yp[:nl]):
                d = (yf[:nl] != yp[:nl]).sum().item()
                bad += 1
                print("MISMATCH", name, "round", it, "flags", flags, "nlev", nl, "entries", d, flush=True)
    h = fin.tensor(rng.uniform(1, 2, (NK, dm.n2)) * 1e3)
    for op, f, fl in (("UHMAT", h, 1), ("UTMAT", None, 0)):
        nl = NK - 1 if op == "UTMAT" else NK
        a = fin.apply(op, xs[0][:nl], f=None if f is None else f[:nl], lev0=0, scale=SCALE, flags=fl)
        b = per.apply(op, xs[0][:nl], f=None if f is None else f[:nl], lev0=0, scale=SCALE, flags=fl)
        if not torch.equal(a, b):
            bad += 1; print("MISMATCH", name, op, flush=True)
    print(name, "equality rounds done, mismatches so far:", bad, flush=True)
    for label, eng, y in (("finishing phase", fin, yf), ("perimeter pass", per, yp)):
        call, _ = eng.prepare_apply("UMAT", xs[0], lev0=0, scale=SCALE, flags=1, out=y)
        k1, k2, wall = timed(eng, call)
        print("%-10s %-16s kernel %.2f us + %.2f us = %.2f us   wall/step %.2f us" % (name, label, k1, k2, k1 + k2, wall), flush=True)
    del fin, per, xs, yf, yp
    torch.cuda.empty_cache()
print("RESULT", "OK" if bad == 0 else "FAILED (%d)" % bad)
sys.exit(0 if bad == 0 else 1)
