import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mimsem_amd.device import DeviceMesh, Engine, check
from mimsem_amd.geom import BoxGeom
from mimsem_amd.mesh import PeriodicBox, box_coords
from mimsem_amd.topo import Topo
pn, nk = 4, 2
bx = PeriodicBox(pn, 2, 1); bc = box_coords(pn, 2, 1000.0)
t = Topo(bx, 0, nk); g = BoxGeom(t, bx, bc, nk, 1000.0)
g.set_levels(np.repeat(np.linspace(0.0, 100.0, nk + 1)[:, None], g.n0, axis=1))
eng = Engine(DeviceMesh([t], [g], nk=nk, numbering="global"))
ntask = 2
A = np.zeros((ntask, 16, 16)); 
for k in range(ntask):
    A[k] = np.arange(256).reshape(16, 16) + 1000 * k
B = np.tile(np.eye(16), (ntask, 1, 1))
x = np.tile(np.eye(1, 16, 0)[0], (ntask, 1)); x[1] = np.eye(1, 16, 9)[0]     # unit vectors: y = column 0 (task 0), column 9 (task 1)
cq = np.ones((ntask, 25))
out = eng.zeros(ntask, 825)
tA, tB, tx, tc = (eng.tensor(v) for v in (A, B, x, cq))
check(eng.L.mimsem_selftest_rows_half(eng.ctx, ntask, tA.data_ptr(), tB.data_ptr(), tx.data_ptr(), tc.data_ptr(), out.data_ptr()), "selftest")
o = out.cpu().numpy()
np.set_printoptions(linewidth=200, precision=4, suppress=True)
for k in range(ntask):
    print("task", k, "y     ", o[k, 768:784]); print("   want ", A[k] @ x[k]); print("   probe", o[k, 809:825])
