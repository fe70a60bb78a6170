"""Config 3 over the reference driver's run length (src/Galewsky.cpp:83-152: dt = 360 s, 2 Picard iterations, upwinded q; 4 800 steps = 20 days,
the jet rolls up after day 4): how long does the fixed-length mode stay engaged, where do its checks miss, what do the re-estimates cost?
    python scripts/exp/galewsky_long.py [nsteps=1440] [ne=24]
Prints one line per 120 steps (half a day): mode counters, the regions in force, conservation drifts, wall time per step."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from mimsem_amd.device import DeviceMesh, Engine  # noqa: E402
from mimsem_amd.geom import Geom  # noqa: E402
from mimsem_amd.mesh import CubedSphere, sphere_coords  # noqa: E402
from mimsem_amd.sweqn import SWEqn, galewsky  # noqa: E402
from mimsem_amd.topo import Topo  # noqa: E402


def build(ne, pn=3):
    cs = CubedSphere(pn, ne, 6); coords = sphere_coords(pn, ne)
    topos = [Topo(cs, p, 1) for p in range(6)]
    geoms = [Geom(t, cs, coords, 1, signed_det=True) for t in topos]
    for g in geoms:
        g.set_levels(np.stack([np.zeros(g.n0), np.ones(g.n0)]))
    dm = DeviceMesh(topos, geoms, nk=1, numbering="global")
    eng = Engine(dm)
    xq = np.zeros((dm.nq, 3))
    for g in geoms:
        xq[g.loc0] = coords[g.loc0]
    return dm, eng, xq[dm.gidq]


def main():
    nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 1440
    ne = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    dm, eng, xq = build(ne)
    S = SWEqn(eng, xq)
    uq, hq = galewsky(torch.as_tensor(xq, device=eng.device))
    u, h = S.init1(uq), S.init2(hq)
    c0 = S.conservation(u, h)
    misses = []
    torch.cuda.synchronize(); t0 = time.perf_counter(); tl = t0
    for n in range(1, nsteps + 1):
        r0 = S.recalibrations
        u, h = S.solve(u, h, 360.0, nits=2, q_exact=False)
        if S.recalibrations != r0:
            misses.append((n, S.last_miss, dict(S._pg.regions) if getattr(S._pg, "regions", None) else None))
        if n % 120 == 0 or n == nsteps:
            torch.cuda.synchronize(); t1 = time.perf_counter()
            c = S.conservation(u, h)
            vn = eng.interp_quad(1, u)[0]
            w = S.curl(u)
            print(json.dumps({"step": n, "day": n * 360.0 / 86400.0, "ms_per_step": 1e3 * (t1 - tl) / (120 if n % 120 == 0 else n % 120),
                              "fixed_iterations": S.fixed_iterations, "adaptive_iterations": S.adaptive_iterations, "recalibrations": S.recalibrations,
                              "its": dict(S.its), "drift": {k: (c[k] - c0[k]) / abs(c0[k]) for k in ("mass", "energy", "enstrophy")},
                              "vorticity_total": c["vorticity"], "max_vorticity": float(w.abs().max()), "max_v": float(vn[..., 1].abs().max()),
                              "max_u": float(vn[..., 0].max()), "finite": bool(torch.isfinite(vn).all())}), flush=True)
            torch.cuda.synchronize(); tl = time.perf_counter()
    print(json.dumps({"total_s": time.perf_counter() - t0, "misses": [(n, m, r) for n, m, r in misses][:40], "n_misses": len(misses)}, default=str), flush=True)


if __name__ == "__main__":
    main()
