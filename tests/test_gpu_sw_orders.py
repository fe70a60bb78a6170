"""The shallow-water engine kernels at element orders other than the benchmark's p = 3: fused [u,h] operator, coupled element-block
preconditioner, Richardson sweeps and a whole Picard step.  Reference here = the composition of the individual engine operators,
which tests/test_gpu_horizontal.py pins against the oracle for every order."""
import numpy as np
import pytest

from tests.helpers import rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=[2, 4, 5], ids=lambda p: "p%d" % p)
def sw_p(request):
    import torch
    from mimsem_amd.device import DeviceMesh, Engine
    from mimsem_amd.geom import Geom
    from mimsem_amd.mesh import CubedSphere, sphere_coords
    from mimsem_amd.sweqn import SWEqn, williamson2
    from mimsem_amd.topo import Topo
    pn, ne = request.param, 3
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
    S = SWEqn(eng, xq[dm.gidq])
    uq, hq = williamson2(torch.as_tensor(xq[dm.gidq], device=eng.device), alpha=0.0)
    return pn, eng, S, uq, hq


def test_fused_operator_all_orders(sw_p):
    import torch
    pn, eng, S, uq, hq = sw_p
    dm = eng.mesh
    r = np.random.default_rng(pn)
    x = eng.tensor(np.concatenate([r.standard_normal(dm.n1), 30.0 * r.standard_normal(dm.n2)]).reshape(1, -1))
    got, ref = S.apply_A(x, 900.0), S.apply_A_composed(x, 900.0)
    assert rel_l2(got.cpu().numpy(), ref.cpu().numpy()) < 1e-13
    assert torch.equal(got, S.apply_A(x, 900.0))


def test_coupled_blocks_and_sweeps_all_orders(sw_p):
    import torch
    from mimsem_amd._lib import MimsemError
    pn, eng, S, uq, hq = sw_p
    dm = eng.mesh
    r = np.random.default_rng(10 + pn)
    x = eng.tensor(r.standard_normal(dm.n1 + dm.n2).reshape(1, -1))
    if pn <= 4:
        C = S._coupled_element_blocks(900.0)
        idx = torch.cat([torch.as_tensor(dm.inds1x), torch.as_tensor(dm.inds1y), torch.as_tensor(dm.inds2) + dm.n1], dim=1).long().to(eng.device)
        zz = torch.einsum("ecr,ec->er", C, x[0][idx])
        ref = torch.zeros_like(x[0]); ref.index_add_(0, idx.reshape(-1), zz.reshape(-1))
        assert rel_l2(eng.sw_blocks_apply(C, x)[0].cpu().numpy(), ref.cpu().numpy()) < 1e-13
    else:                                   # 2 n1e + n2e > 64 rows: refused loudly, SWEqn keeps the block-diagonal preconditioner
        nd = 2 * eng.n1e + eng.n2e
        with pytest.raises(MimsemError):
            eng.sw_blocks_apply(torch.zeros(eng.nEl, nd, nd, dtype=torch.float64, device=eng.device), x)
    u = x[:, :dm.n1].contiguous(); b = eng.tensor(r.standard_normal(dm.n1).reshape(1, -1))
    cm = S.m1_pre.transpose(1, 2).contiguous()
    ref = S.precond_M1(b - S.M1(u))
    u1 = u.clone(); upd = torch.zeros_like(u)
    eng.block_richardson_sweep("UMAT", cm, u1, b, upd=upd)
    assert rel_l2(upd.cpu().numpy(), ref.cpu().numpy()) < 1e-13 and torch.equal(u1, u + upd)


def test_picard_step_all_orders(sw_p):
    """a whole SWEqn::solve step on the steady Williamson-2 state: converges, conserves mass to round-off, stays near the state"""
    pn, eng, S, uq, hq = sw_p
    u, h = S.init1(uq), S.init2(hq)
    c0 = S.conservation(u, h)
    u1, h1 = S.solve(u, h, 600.0, nits=3, q_exact=False)
    c1 = S.conservation(u1, h1)
    assert abs(c1["mass"] - c0["mass"]) < 1e-13 * abs(c0["mass"])
    assert S.history[-1] < S.history[0]
    tol = {2: 5e-2, 4: 1e-3, 5: 1e-3}[pn]               # distance from the steady state = truncation error of the coarse mesh
    assert rel_l2(h1[0].cpu().numpy(), h[0].cpu().numpy()) < tol
