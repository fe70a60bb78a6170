"""Timing probe (round 6): what would k_sw_blocks_apply (the block pass of the [u|h] Chebyshev step, 33 x 33 doubles per element = 30 MB per launch on the
config-3 sphere) cost with 4-byte block entries?  Run against the default library and against a probe build whose kernel reads the block array as floats
(wrong numbers, right traffic: the kernel line `const float* Be = (const float*)B + ...`).  Measured: 9.8 us -> 8.1 us per pass + gather, back to back:
at most 5 % of the SW step; declined."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from mimsem_amd.device import DeviceMesh, Engine
from mimsem_amd.geom import Geom
from mimsem_amd.mesh import CubedSphere, sphere_coords
from mimsem_amd.topo import Topo
cs = CubedSphere(3, 24, 6); coords = sphere_coords(3, 24)
topos = [Topo(cs, p, 1) for p in range(6)]
geoms = [Geom(t, cs, coords, 1, signed_det=True) for t in topos]
for g in geoms: g.set_levels(np.stack([np.zeros(g.n0), np.ones(g.n0)]))
dm = DeviceMesh(topos, geoms, nk=1, numbering="global"); eng = Engine(dm)
nd = 2*eng.n1e + eng.n2e
B = torch.randn(dm.nEl, nd, nd, dtype=torch.float64, device=eng.device)*1e-3
x = torch.randn(1, dm.n1 + dm.n2, dtype=torch.float64, device=eng.device)
y = torch.zeros_like(x)
for _ in range(20): eng.sw_blocks_apply(B, x, out=y)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(500): eng.sw_blocks_apply(B, x, out=y)
e1.record(); torch.cuda.synchronize()
print(os.environ.get("MIMSEM_LIB", "default"), "sw_blocks_apply (block pass + gather): %.2f us per call" % (e0.elapsed_time(e1)*1e3/500))
