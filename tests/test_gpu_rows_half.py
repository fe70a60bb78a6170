"""The half-row block algebra of order 4 (csrc/column_dpp.inc, dpp::RowsH): a 16 x 16 block row split over two lanes of a wavefront (8 columns
each, the halves exchanged with v_permlane16_swap) -- the layout k_s3_sweep<4> runs on so that its ~20 live blocks fit the register file.
Every primitive against numpy on random blocks (mimsem_selftest_rows_half): a failure here is the algebra's, not a kernel's."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_rows_half_primitives_match_numpy(oracle):
    from mimsem_amd.device import DeviceMesh, Engine, check
    from mimsem_amd.geom import BoxGeom
    from mimsem_amd.mesh import PeriodicBox, box_coords
    from mimsem_amd.topo import Topo
    pn, nk = 4, 2
    bx = PeriodicBox(pn, 2, 1); bc = box_coords(pn, 2, 1000.0)
    t = Topo(bx, 0, nk); g = BoxGeom(t, bx, bc, nk, 1000.0)
    g.set_levels(np.repeat(np.linspace(0.0, 100.0, nk + 1)[:, None], g.n0, axis=1))
    eng = Engine(DeviceMesh([t], [g], nk=nk, numbering="global"))
    rng = np.random.default_rng(44)
    ntask = 7                                                            # (odd: the last wavefront carries one task twice)
    A = rng.standard_normal((ntask, 16, 16)); A = A @ A.transpose(0, 2, 1) + 16.0 * np.eye(16)      # SPD: the unpivoted sweep's domain
    A *= 10.0 ** rng.uniform(-3, 8, (ntask, 1, 1))                       # (mass blocks carry SCALE = 1e8)
    B = rng.standard_normal((ntask, 16, 16))
    x = rng.standard_normal((ntask, 16)); cq = rng.uniform(0.5, 2.0, (ntask, 25))
    out = eng.zeros(ntask, 825)
    tA, tB, tx, tc = (eng.tensor(v) for v in (A, B, x, cq))
    check(eng.L.mimsem_selftest_rows_half(eng.ctx, ntask, tA.data_ptr(), tB.data_ptr(), tx.data_ptr(), tc.data_ptr(), out.data_ptr()), "selftest")
    o = out.cpu().numpy()
    W = np.array(oracle.tables(pn, pn)["W"]).reshape(25, 16)             # W[q][j] = E[qx][jx] E[qy][jy]   (ElMats.cpp:105-110)
    rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
    for k in range(ntask):
        C, Ai, As, y, fq, probe = o[k, :256].reshape(16, 16), o[k, 256:512].reshape(16, 16), o[k, 512:768].reshape(16, 16), o[k, 768:784], o[k, 784:809], o[k, 809:825]
        assert rel(C, A[k] @ B[k] + B[k] @ A[k]) < 1e-13, k
        assert rel(y, A[k] @ x[k]) < 1e-13, k
        assert rel(Ai, np.linalg.inv(A[k])) < 1e-11, k
        assert rel(fq, W @ x[k]) < 1e-13, k
        assert rel(As, W.T @ (cq[k][:, None] * W)) < 1e-13, k
        # getH<3>: lanes of half 0 read x[3], half 1 x[11] -- the probe is stored by both halves of a row (the second store wins: half 1);
        # xsum of {1 in half 0, 2 in half 1} = 3 in both
        assert set(np.round(probe - 300.0, 12)) <= {round(x[k][3], 12), round(x[k][11], 12)}, k
