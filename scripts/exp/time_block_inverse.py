"""mimsem_block_inverse on the 3 456 coupled [u|h] blocks (33 x 33) of the config-3 SW preconditioner and on the 24 x 24 M1 blocks:
one wavefront per block (round 5) against the thread-per-block kernel (MIMSEM_INV_THREAD=1)"""
import os, sys, time
os.environ.setdefault("MIMSEM_EXPERIMENTS", "1")      # (closed-experiment switches are read only under this master switch: DESIGN 9.1)
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from mimsem_amd.device import DeviceMesh, Engine
from mimsem_amd.geom import Geom
from mimsem_amd.mesh import CubedSphere, sphere_coords
from mimsem_amd.topo import Topo
cs = CubedSphere(3, 4, 6); coords = sphere_coords(3, 4)
topos = [Topo(cs, p, 1) for p in range(6)]; geoms = [Geom(t, cs, coords, 1) for t in topos]
for g in geoms: g.set_levels(np.stack([np.zeros(g.n0), np.ones(g.n0)]))
eng = Engine(DeviceMesh(topos, geoms, nk=1, numbering="global"))
r = np.random.default_rng(0)
for n in (24, 33, 40, 56):
    G = r.standard_normal((3456, n, n)); B = G @ G.transpose(0, 2, 1) + n * np.eye(n)
    Bt = eng.tensor(B)
    for env in ("0", "1"):
        os.environ["MIMSEM_INV_THREAD"] = env
        eng.block_inverse_status(Bt.clone()); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): eng.block_inverse(Bt.clone())
        torch.cuda.synchronize()
        print("n = %d, %s: %.3f ms per 3 456 blocks" % (n, "thread per block" if env == "1" else "wavefront per block", (time.perf_counter() - t0) / 5 * 1e3))
del os.environ["MIMSEM_INV_THREAD"]
