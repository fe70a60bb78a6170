"""The block pass of the Chebyshev / Richardson sweeps on the matrix cores (k_blocks_residual_mfma: the (2 n1e x 2 n1e) element block
times the residuals of 16 levels as v_mfma_f64_16x16x4 products; opt-in, MIMSEM_BLOCKS_MFMA=1) against the default register-row form of
the same library (itself checked against dense algebra in test_gpu_next_rows.py) and against a dense restatement of one sweep."""
import os

import numpy as np
import pytest

from tests.helpers import SCALE, rel_l2, z_levels

pytestmark = pytest.mark.gpu


def _engines(pn, ne, nk):
    from mimsem_amd.device import DeviceMesh, Engine
    from mimsem_amd.geom import Geom
    from mimsem_amd.mesh import CubedSphere, sphere_coords
    from mimsem_amd.topo import Topo
    cs = CubedSphere(pn, ne, 6); coords = sphere_coords(pn, ne)
    topos = [Topo(cs, p, nk) for p in range(6)]
    geoms = [Geom(t, cs, coords, nk) for t in topos]
    for g in geoms:
        g.set_levels(z_levels(nk, g.n0))
    dm = DeviceMesh(topos, geoms, nk=nk, numbering="global")
    engs = []
    for flag in ("0", "1"):                              # explicit both ways: the default differs by order (p = 4: matrix cores)
        os.environ["MIMSEM_BLOCKS_MFMA"] = flag
        try:
            engs.append(Engine(dm))
        finally:
            del os.environ["MIMSEM_BLOCKS_MFMA"]
    rr, mf = engs
    return dm, mf, rr


@pytest.mark.parametrize("pn,ne,nk", [(3, 4, 30), (3, 3, 17), (2, 4, 16), (4, 2, 7)], ids=["p3_30lev", "p3_17lev", "p2_16lev", "p4_7lev"])
def test_block_sweeps_on_the_matrix_cores(pn, ne, nk):
    import torch
    dm, mf, rr = _engines(pn, ne, nk)
    r = np.random.default_rng(3)
    nd = 2 * mf.n1e
    B = r.standard_normal((dm.nEl, nd, nd)) / nd
    b = r.standard_normal((nk, dm.n1)); x0 = r.standard_normal((nk, dm.n1)); p0 = r.standard_normal((nk, dm.n1))
    es = r.uniform(0.5, 1.5, (nk, dm.nEl))
    out = []
    for eng in (mf, rr):
        x, p = eng.tensor(x0), eng.tensor(p0); upd = torch.zeros_like(x)
        eng.block_chebyshev_sweep("UMAT", eng.tensor(B), x, eng.tensor(b), p, 0.7, 0.3, elem_scale=eng.tensor(es), scale=SCALE, flags=1, upd=upd)
        x2 = eng.tensor(x0)
        eng.block_richardson_sweep("UMAT", eng.tensor(B), x2, eng.tensor(b), scale=SCALE, flags=1)
        out.append([t.cpu().numpy() for t in (x, p, upd, x2)])
    for a, c, name in zip(out[0], out[1], ("x", "p", "z", "x_richardson")):
        assert rel_l2(a, c) < 1e-13, name
        for k in range(nk):                                             # level by level: a wrong level would hide in the norm
            assert rel_l2(a[k], c[k]) < 1e-12, (name, k)
    # dense restatement of the preconditioned residual z = sum_e R_e^T es_e B_e R_e (b - M x) on three levels
    y = mf.apply("UMAT", mf.tensor(x0), lev0=0, scale=SCALE, flags=1).cpu().numpy()
    slots = np.concatenate([dm.inds1x, dm.inds1y], axis=1)              # [nEl, nd]
    for k in (0, nk // 2, nk - 1):
        res = b[k] - y[k]
        z = np.zeros(dm.n1)
        for e in range(dm.nEl):
            # blocks are column-major per element: entry (row i, column c) at [e, c, i]
            np.add.at(z, slots[e], es[k, e] * (B[e].T @ res[slots[e]]))
        assert rel_l2(out[0][2][k], z) < 1e-12, k
    # the whole solve as one call (mimsem_block_chebyshev_solve: its first step has no operator result -- that block pass always runs on the
    # register-row form) gives the bits of the sweep calls on either engine
    coef = [(0.9, 0.0), (0.8, 0.2), (0.85, 0.15), (0.8, 0.1)]
    for eng in (mf, rr):
        x = torch.zeros(nk, dm.n1, dtype=torch.float64, device=eng.device); p = torch.zeros_like(x)
        for al, be in coef:
            eng.block_chebyshev_sweep("UMAT", eng.tensor(B), x, eng.tensor(b), p, al, be, elem_scale=eng.tensor(es), scale=SCALE, flags=1)
        y1 = eng.block_chebyshev_solve("UMAT", eng.tensor(B), eng.tensor(b), coef, elem_scale=eng.tensor(es), scale=SCALE, flags=1)
        assert torch.equal(y1, x)
    # run-to-run reproducible
    x, p = mf.tensor(x0), mf.tensor(p0); upd = torch.zeros_like(x)
    mf.block_chebyshev_sweep("UMAT", mf.tensor(B), x, mf.tensor(b), p, 0.7, 0.3, elem_scale=mf.tensor(es), scale=SCALE, flags=1, upd=upd)
    assert np.array_equal(x.cpu().numpy(), out[0][0]) and np.array_equal(upd.cpu().numpy(), out[0][2])
