"""GPU parity: HIP engine (through the C ABI) vs the CPU oracle on the same seeded inputs.
Bar (BASELINE.json north_star): fields within 1e-10 relative L2; integer topology bit-exact."""
import numpy as np
import pytest

from tests.helpers import SCALE, make_patch, rel_l2

pytestmark = pytest.mark.gpu
TOL = 1e-10


@pytest.fixture(scope="module", params=[(3, 4, 6, 0), (3, 4, 24, 13), (4, 2, 6, 3), (2, 3, 6, 5), (5, 1, 6, 2), (6, 1, 6, 0), (7, 1, 6, 4)],
                ids=lambda p: "p%d_ne%d_np%d_pi%d" % p)
def setup(request, oracle):
    import torch
    from mimsem_amd.device import DeviceMesh, Engine
    pn, ne, nprocs, pi = request.param
    cs, topo, geom, P, rng = make_patch(oracle, pn, ne, nprocs, pi, nk=3, seed=pn * 100 + pi)
    eng = Engine(DeviceMesh([topo], [geom], nk=3, numbering="local"))
    return eng, P, rng


def _fields(P, rng):
    return dict(h2=rng.uniform(0.5, 1.5, P.n2) * 1e6, u1=rng.standard_normal(P.n1) * 1e3, q0=rng.standard_normal(P.n0) * 1e-4,
                x0=rng.standard_normal(P.n0), x1=rng.standard_normal(P.n1), x2=rng.standard_normal(P.n2))


CASES = [  # op, flag, coefficient-field key, input key
    ("UMAT", 1, None, "x1"), ("UMAT", 0, None, "x1"), ("WMAT", 1, None, "x2"), ("WMAT", 0, None, "x2"),
    ("UHMAT", 1, "h2", "x1"), ("UHMAT", 0, "h2", "x1"), ("PMAT", 0, None, "x0"), ("PHMAT", 0, "h2", "x0"),
    ("WTQUMAT", 0, "u1", "x1"), ("ROTMAT", 0, "q0", "x1"), ("WHMAT", 1, "h2", "x2"), ("WHMAT", 0, "h2", "x2"),
    ("UTMAT", 0, None, "x1"), ("UTMAT_H", 0, "h2", "x1"), ("UTQWMAT", 0, "u1", "x2"), ("WTQDUDZ", 0, "u1", "x1"),
    ("WMATINV", 0, None, "x2"), ("WHMATINV", 0, "h2", "x2"),
]


@pytest.mark.parametrize("op,flag,fkey,xkey", CASES, ids=[f"{c[0]}_{c[1]}" for c in CASES])
def test_apply_matches_oracle(setup, op, flag, fkey, xkey):
    eng, P, rng = setup
    F = _fields(P, np.random.default_rng(11))
    lev = 1
    f = F[fkey] if fkey else None
    want = P.apply(op, F[xkey], lev=lev, scale=SCALE, flag=flag, f1=f)
    got = eng.apply(op, eng.tensor(F[xkey]), f=eng.tensor(f) if f is not None else None, lev0=lev, scale=SCALE, flags=flag)
    assert rel_l2(got.cpu().numpy(), want) < TOL, op


@pytest.mark.parametrize("op,flag,fkey", [(c[0], c[1], c[2]) for c in CASES], ids=[f"{c[0]}_{c[1]}" for c in CASES])
def test_element_matrices_match_oracle(setup, op, flag, fkey):
    """the dense blocks the reference hands to MatSetValues"""
    eng, P, rng = setup
    F = _fields(P, np.random.default_rng(11))
    f = F[fkey] if fkey else None
    want = P.op_elmats(op, 1, SCALE, flag, f)
    got = eng.element_matrices(op, f=eng.tensor(f) if f is not None else None, lev=1, scale=SCALE, flags=flag)
    assert got.shape == want.shape
    assert rel_l2(got.cpu().numpy(), want) < TOL, op


def test_level_batch_equals_single_levels(setup):
    """one launch over all levels == per-level launches (bitwise: same kernel, same order)"""
    import torch
    eng, P, rng = setup
    x = eng.tensor(np.random.default_rng(3).standard_normal((2, P.n1)))
    h = eng.tensor(np.random.default_rng(4).uniform(1, 2, (2, P.n2)))
    yb = eng.apply("UHMAT", x, f=h, lev0=0, scale=SCALE, flags=1)
    for k in range(2):
        yk = eng.apply("UHMAT", x[k], f=h[k], lev0=k, scale=SCALE, flags=1)
        assert torch.equal(yb[k], yk)
        want = P.apply("UHMAT", x[k].cpu().numpy(), lev=k, scale=SCALE, flag=1, f1=h[k].cpu().numpy())
        assert rel_l2(yk.cpu().numpy(), want) < TOL


def test_run_to_run_bitwise_reproducible(setup):
    import torch
    eng, P, rng = setup
    x = eng.tensor(np.random.default_rng(5).standard_normal(P.n1))
    q = eng.tensor(np.random.default_rng(6).standard_normal(P.n0))
    a = eng.apply("ROTMAT", x, f=q, lev0=2, scale=SCALE)
    for _ in range(3):
        assert torch.equal(a, eng.apply("ROTMAT", x, f=q, lev0=2, scale=SCALE))


def test_uvec_family_and_accumulate(setup):
    """Uvec::assemble / assemble_hu / assemble_wxu == UMAT/UHMAT/ROTMAT applied to the velocity (alpha = fac),
    and zero_and_scatter=false accumulation (eul/HorizSolve.cpp:300-303)"""
    eng, P, rng = setup
    r = np.random.default_rng(8)
    vel, vel2 = r.standard_normal(P.n1), r.standard_normal(P.n1)
    rho, vort = r.uniform(1, 2, P.n2) * 1e6, r.standard_normal(P.n0) * 1e-4
    tv, tv2, trho, tvort = (eng.tensor(a) for a in (vel, vel2, rho, vort))
    assert rel_l2(eng.apply("UMAT", tv, lev0=1, scale=SCALE, flags=1).cpu().numpy(), P.uvec(1, SCALE, vel)) < TOL
    assert rel_l2(eng.apply("ROTMAT", tv, f=tvort, lev0=1, scale=SCALE).cpu().numpy(), P.uvec_wxu(1, SCALE, vel, vort)) < TOL
    y = eng.apply("UHMAT", tv, f=trho, lev0=1, scale=SCALE, flags=1, alpha=1.0 / 3.0)
    eng.apply("UHMAT", tv2, f=trho, lev0=1, scale=SCALE, flags=1 | 2, alpha=1.0 / 6.0, out=y)
    want = P.uvec_hu(1, SCALE, vel, rho, 1.0 / 3.0) + P.uvec_hu(1, SCALE, vel2, rho, 1.0 / 6.0)
    assert rel_l2(y.cpu().numpy(), want) < TOL


def test_wvec_family(setup):
    """B18 Wvec::assemble / assemble_K (eul/Assembly.cpp:2457-2545) against the restatement of the reference's loops with the
    transpose table filled (the reference leaves Wt uninitialised; its call sites are commented out) -- and that restatement
    against Wmat / WtQUmat, which is what the two methods compute"""
    eng, P, rng = setup
    r = np.random.default_rng(12)
    rho = r.uniform(1, 2, P.n2) * 1e6; vel1, vel2 = r.standard_normal(P.n1) * 10.0, r.standard_normal(P.n1) * 10.0
    for vs in (True, False):
        want = P.wvec(1, SCALE, vs, rho)
        assert rel_l2(eng.wvec(eng.tensor(rho), lev0=1, scale=SCALE, vert_scale=vs).cpu().numpy(), want) < TOL
        assert rel_l2(P.apply("WMAT", rho, lev=1, scale=SCALE, flag=int(vs)), want) < 1e-13
    want = P.wvec_K(2, SCALE, vel1, vel2)
    assert rel_l2(eng.wvec_K(eng.tensor(vel1), eng.tensor(vel2), lev0=2, scale=SCALE).cpu().numpy(), want) < TOL
    assert rel_l2(P.apply("WTQUMAT", vel1, lev=2, scale=SCALE, f1=vel2), want) < 1e-13


def test_pvec_phvec(setup):
    eng, P, rng = setup
    h = np.random.default_rng(9).uniform(1, 2, P.n2) * 1e6
    assert rel_l2(eng.pvec(1, 1, SCALE)[0].cpu().numpy(), P.pvec(1, SCALE)) < TOL
    assert rel_l2(eng.pvec(2, 1, SCALE, h2=eng.tensor(h[None, :]))[0].cpu().numpy(), P.phvec(2, SCALE, h)) < TOL


def test_incidence(setup):
    """E10/E21 stencils are +-1 sums: bit-exact; E21.E10 = 0; E12 = -E21^T, E01 = -E10^T"""
    import torch
    eng, P, rng = setup
    r = np.random.default_rng(10)
    x0, x1, x2 = r.standard_normal(P.n0), r.standard_normal(P.n1), r.standard_normal(P.n2)
    y1 = eng.incidence("E10", eng.tensor(x0)); y2 = eng.incidence("E21", eng.tensor(x1))
    assert np.array_equal(y1.cpu().numpy(), P.e10(x0))
    assert np.array_equal(y2.cpu().numpy(), P.e21(x1))
    # transposes: <E12 a, b> = -<a, E21 b>
    a2, b1 = eng.tensor(x2), eng.tensor(x1)
    lhs = torch.dot(eng.incidence("E12", a2), b1).item(); rhs = -torch.dot(a2, eng.incidence("E21", b1)).item()
    assert abs(lhs - rhs) <= 1e-12 * max(1.0, abs(rhs))
    a1, b0 = eng.tensor(x1), eng.tensor(x0)
    lhs = torch.dot(eng.incidence("E01", a1), b0).item(); rhs = -torch.dot(a1, eng.incidence("E10", b0)).item()
    assert abs(lhs - rhs) <= 1e-12 * max(1.0, abs(rhs))


@pytest.mark.parametrize("which,op", [(0, "PHMAT_UP"), (1, "ROTMAT_UP")])
def test_upwinded_sw_operators(setup, which, op):
    """Phmat::assemble_up / RotMat_up::assemble (src/Assembly.cpp:499-567, :1784-1853): the 0-form evaluated at
    departure points x_q - tau*u; shift kept at ~0.2 of the reference element as in the SW runs (UP_TAU=0.5)"""
    eng, P, rng = setup
    r = np.random.default_rng(21)
    fac, dt = 0.5, 600.0
    ul = r.uniform(-1, 1, P.n1) * P.det.mean() * 0.2 / (fac * dt)
    if which == 0:
        f1 = r.uniform(0.5, 1.5, P.n2) * 1e4; x = r.standard_normal(P.n0)
    else:
        f1 = r.standard_normal(P.n0) * 1e-4; x = r.standard_normal(P.n1)
    want, _ = P.apply_up(which, x, fac, dt, f1, ul)
    got = eng.apply_up(op, eng.tensor(x), eng.tensor(f1), eng.tensor(ul), fac, dt, lev0=0)
    assert rel_l2(got.cpu().numpy(), want) < TOL


@pytest.mark.parametrize("which,op", [(0, "WTQ"), (1, "PTQ"), (2, "UTQ")])
def test_quad_grid_projections(setup, which, op):
    """B7: WtQmat / PtQmat / UtQmat (eul/Assembly.cpp:707-902) applied to a quad-point-grid field"""
    eng, P, rng = setup
    r = np.random.default_rng(31)
    xq = r.standard_normal(P.n0q * (2 if which == 2 else 1))
    want = P.project_from_quad(which, xq)
    got = eng.apply(op, eng.tensor(xq), lev0=0, scale=1.0)
    assert rel_l2(got.cpu().numpy(), want) < TOL


@pytest.mark.parametrize("which", [0, 1, 2])
def test_eul_upwinded_test_functions(setup, which):
    """B2 Umat::assemble_up, B4 Uhmat::assemble_up, B17 Uvec::assemble_hu_up (eul/Assembly.cpp:156-279, 477-560, 2281-2373)"""
    eng, P, rng = setup
    r = np.random.default_rng(41)
    lev, tau = 1, 75.0
    vscale = P.det.mean() / P.thickInv[lev].mean() * 0.3 / tau        # shift ~ 0.3 of the reference element
    u1 = r.uniform(-1, 1, P.n1) * vscale; u2 = r.uniform(-1, 1, P.n1) * vscale
    x = r.standard_normal(P.n1); h = r.uniform(0.5, 1.5, P.n2) * 1e6
    t = eng.tensor
    if which == 0:
        want = P.apply_testup(0, x, lev, SCALE, tau, u1, u2)
        got = eng.apply_up("UMAT_UP", t(x), t(u1), t(u2), lev0=lev, scale=SCALE, tau=tau)
    elif which == 1:
        want = P.apply_testup(1, x, lev, SCALE, tau, h, u1 * P.det.mean() ** 0)   # u1 is Piola-mapped inside
        got = eng.apply_up("UHMAT_UP", t(x), t(h), t(u1), lev0=lev, scale=SCALE, tau=tau)
    else:
        vel = u1
        want = P.uvec_hu_up(lev, SCALE, vel, h, 1.0 / 3.0, tau, u2)
        got = eng.apply_up("UVEC_HU_UP", t(vel), t(h), t(u2), lev0=lev, scale=SCALE, tau=tau, alpha=1.0 / 3.0)
    assert rel_l2(got.cpu().numpy(), want) < TOL
    if which < 2:      # MT = MatTranspose(M) (Assembly.cpp:261, :559): the transposed apply of the same operator
        wantT = P.apply_testup(which, x, lev, SCALE, tau, u1 if which == 0 else h, u2 if which == 0 else u1, transpose=True)
        gotT = (eng.apply_up("UMAT_UP", t(x), t(u1), t(u2), lev0=lev, scale=SCALE, tau=tau, flags=4) if which == 0 else
                eng.apply_up("UHMAT_UP", t(x), t(h), t(u1), lev0=lev, scale=SCALE, tau=tau, flags=4))
        assert rel_l2(gotT.cpu().numpy(), wantT) < TOL


def _exner_fields(P, r, lev):
    """2-form DoFs whose point values (after /det * thickInv) are Exner pressures cp*sigma^(R/cp), sigma in [0.5,1] per element"""
    n = P.n
    dx = np.diff(P.arr("nx", (n + 1,)))
    cell = np.outer(dx, dx).reshape(-1)                                       # integral of 1 over each face of the reference element
    i2 = P.elinds("n2")
    iq = P.elinds("q")
    out = []
    for (k, lo, hi) in ((lev, 0.5, 1.0), (0, 0.98, 1.02)):
        f = np.zeros(P.n2)
        for e in range(P.nEl):
            sig = r.uniform(lo, hi)
            val = 1004.5 * sig ** (287.0 / 1004.5)
            scale_e = P.det[e].mean() / P.thickInv[k][iq[e]].mean()
            f[i2[e]] = val * cell * scale_e * (1.0 + 1e-3 * r.standard_normal(n * n))
        out.append(f)
    return out


def test_umat_ray_held_suarez_friction(setup):
    """B16 Umat_ray::assemble (eul/Assembly.cpp:1876-1979): apply, element blocks, level batching"""
    eng, P, rng = setup
    r = np.random.default_rng(97)
    lev, dt = 2, 120.0
    ek, es = _exner_fields(P, r, lev)
    x = r.standard_normal(P.n1)
    want, em = P.umat_ray(x, lev, SCALE, dt, ek, es)
    assert np.abs(em).max() > 0                       # sigma > 0.7 somewhere (and the k_v = 0 branch elsewhere)
    t = eng.tensor
    got = eng.apply_ray(t(x), t(ek), t(es), dt, lev0=lev, scale=SCALE)
    assert rel_l2(got.cpu().numpy(), want) < TOL
    gm = eng.element_matrices_ray(t(ek), t(es), dt, lev=lev, scale=SCALE)
    assert rel_l2(gm.cpu().numpy(), em) < TOL
    # M1 + M1ray as the reference forms it with MatAXPY (eul/Euler_2.cpp:1448): accumulate into the Umat apply
    import torch
    base = eng.apply("UMAT", t(x), lev0=lev, scale=SCALE, flags=1)
    both = base.clone()
    from mimsem_amd._lib import FLAG_ACCUM
    eng.apply_ray(t(x), t(ek), t(es), dt, lev0=lev, scale=SCALE, flags=FLAG_ACCUM, out=both)
    assert rel_l2(both.cpu().numpy(), base.cpu().numpy() + want) < TOL
    # all levels in one call
    eks = [_exner_fields(P, r, k)[0] for k in range(3)]
    xs = r.standard_normal((3, P.n1))
    got3 = eng.apply_ray(t(xs), t(np.stack(eks)), t(es), dt, lev0=0, scale=SCALE).cpu().numpy()
    for k in range(3):
        wk, _ = P.umat_ray(xs[k], k, SCALE, dt, eks[k], es)
        assert rel_l2(got3[k], wk) < TOL


def test_element_blocks_apply(setup):
    """mimsem_elem_blocks_apply: MatMult with caller-supplied element blocks (the reference's MatSetValues blocks), all forms"""
    eng, P, rng = setup
    import torch
    r = np.random.default_rng(63)
    n1e = P.n1e
    x = r.standard_normal((2, P.n1))
    em = eng.element_matrices("UMAT", lev=1, scale=SCALE, flags=1).view(P.nEl, 2, 2, n1e, n1e)
    B = em.permute(0, 1, 3, 2, 4).reshape(P.nEl, 2 * n1e, 2 * n1e).contiguous()
    want = eng.apply("UMAT", eng.tensor(x), lev0=1, scale=SCALE, flags=1)            # level 1 geometry on both rows
    want1 = eng.apply("UMAT", eng.tensor(x[1]), lev0=1, scale=SCALE, flags=1)
    got = eng.blocks_apply(1, B, eng.tensor(x))
    assert rel_l2(got[1].cpu().numpy(), want1.cpu().numpy()) < TOL
    idx = {0: P.elinds("n0"), 2: P.elinds("n2"), 1: np.concatenate([P.elinds("n1x"), P.elinds("n1y")], axis=1)}
    for form, nd, nv in ((0, P.n0e, P.n0), (1, 2 * n1e, P.n1), (2, P.n2e, P.n2)):
        Bl = r.standard_normal((2, P.nEl, nd, nd))                                   # different blocks per level, nonsymmetric
        xv = r.standard_normal((2, nv))
        for tr in (False, True):
            ref = np.zeros((2, nv))
            for lev in range(2):
                for e in range(P.nEl):
                    Be = Bl[lev, e].T if tr else Bl[lev, e]
                    ref[lev, idx[form][e]] += Be @ xv[lev, idx[form][e]]
            got = eng.blocks_apply(form, eng.tensor(Bl), eng.tensor(xv), transpose=tr, alpha=0.5).cpu().numpy()
            assert rel_l2(got, 0.5 * ref) < 1e-12, (form, tr)
    y0 = eng.tensor(r.standard_normal((2, P.n2))); keep = y0.clone()
    B2 = eng.tensor(r.standard_normal((P.nEl, P.n2e, P.n2e))); x2 = eng.tensor(r.standard_normal((2, P.n2)))
    eng.blocks_apply(2, B2, x2, accum=True, out=y0)
    assert rel_l2((y0 - keep).cpu().numpy(), eng.blocks_apply(2, B2, x2).cpu().numpy()) < 1e-12


def test_element_blocks_apply_level_sweep(setup):
    """level-independent blocks kept in LDS across the level sweep, with the per-(level, element) scale"""
    eng, P, rng = setup
    r = np.random.default_rng(64)
    idx = {0: P.elinds("n0"), 2: P.elinds("n2"), 1: np.concatenate([P.elinds("n1x"), P.elinds("n1y")], axis=1)}
    for form, nd, nv in ((0, P.n0e, P.n0), (1, 2 * P.n1e, P.n1), (2, P.n2e, P.n2)):
        B = r.standard_normal((P.nEl, nd, nd)); xv = r.standard_normal((3, nv)); sc = r.uniform(0.5, 2.0, (3, P.nEl))
        for tr in (False, True):
            for scale in (None, sc):
                ref = np.zeros((3, nv))
                for lev in range(3):
                    for e in range(P.nEl):
                        Be = (B[e].T if tr else B[e]) * (1.0 if scale is None else scale[lev, e])
                        ref[lev, idx[form][e]] += Be @ xv[lev, idx[form][e]]
                got = eng.blocks_apply(form, eng.tensor(B), eng.tensor(xv), transpose=tr,
                                       elem_scale=None if scale is None else eng.tensor(scale)).cpu().numpy()
                assert rel_l2(got, ref) < 1e-12, (form, tr, scale is not None)


@pytest.mark.parametrize("env,val", [("MIMSEM_DIRECT", "1"), ("MIMSEM_FUSE", "1"), ("MIMSEM_WAVE", "0"), ("MIMSEM_WAVE_ORDER", "0"),
                                     ("MIMSEM_WAVE_LCH", "2"), ("MIMSEM_WAVE_SINGLES", "1")])
def test_opt_in_scatter_variants_agree_with_default(setup, env, val, monkeypatch):
    """the alternative scatter-add organisations kept behind environment switches (direct single-contributor writes; LDS group sums
    per workgroup + perimeter pass; the two-pass form instead of the wave-level fused default; other work-item orders / level chunks
    of the wave-level kernel) produce the default result"""
    import torch
    from mimsem_amd.device import Engine
    eng, P, rng = setup
    monkeypatch.setenv(env, val)
    alt = Engine(eng.mesh)                                  # the switches are read at context creation
    monkeypatch.delenv(env)
    r = np.random.default_rng(8)
    x1 = r.standard_normal((3, P.n1)); x0 = r.standard_normal((3, P.n0)); h = r.uniform(0.5, 1.5, (3, P.n2)) * 1e6
    q0 = r.standard_normal((3, P.n0)) * 1e-4
    for op, x, f, fl in (("UMAT", x1, None, 1), ("UHMAT", x1, h, 1), ("ROTMAT", x1, q0, 0), ("UTMAT_H", x1, h, 0), ("PMAT", x0, None, 0)):
        a = eng.apply(op, eng.tensor(x), f=None if f is None else eng.tensor(f), lev0=0, scale=SCALE, flags=fl)
        b = alt.apply(op, alt.tensor(x), f=None if f is None else alt.tensor(f), lev0=0, scale=SCALE, flags=fl)
        if env in ("MIMSEM_WAVE_ORDER", "MIMSEM_WAVE_LCH"):
            assert torch.equal(a, b), op                    # same arithmetic, different work-item order / chunking
        else:
            assert rel_l2(b.cpu().numpy(), a.cpu().numpy()) < 1e-13, op
    base = eng.tensor(r.standard_normal((3, P.n1)))
    ya, yb = base.clone(), base.clone()
    eng.apply("UMAT", eng.tensor(x1), lev0=0, scale=SCALE, flags=3, alpha=0.25, out=ya)      # accumulate
    alt.apply("UMAT", alt.tensor(x1), lev0=0, scale=SCALE, flags=3, alpha=0.25, out=yb)
    assert rel_l2(yb.cpu().numpy(), ya.cpu().numpy()) < 1e-13


@pytest.mark.parametrize("kind,form,push", [("0", 0, True), ("1l", 1, False), ("1g", 1, True), ("2l", 2, False), ("2g", 2, True)])
def test_interp_quad_matches_oracle_points(setup, kind, form, push):
    """row A7: one launch gives what Geom::interp* (eul/Geom.cpp:328-417) gives point by point"""
    eng, P, rng = setup
    r = np.random.default_rng(31)
    x = r.standard_normal((2, (P.n0, P.n1, P.n2)[form]))
    got = eng.interp_quad(form, eng.tensor(x), push_forward=push).cpu().numpy()
    mp1 = P.mp1 if hasattr(P, "mp1") else eng.mesh.m + 1
    nex = P.nElsX
    nc = 2 if form == 1 else 1
    want = np.zeros((2, nex * nex, mp1 * mp1, nc))
    for lev in range(2):
        for ey in range(nex):
            for ex in range(nex):
                for q in range(mp1 * mp1):
                    want[lev, ey * nex + ex, q] = P.interp(kind, ex, ey, q % mp1, q // mp1, x[lev])[:nc]
    got = got.reshape(want.shape)
    assert rel_l2(got, want) < 1e-13, kind
    one = eng.interp_quad(form, eng.tensor(x[1]), push_forward=push).cpu().numpy().reshape(want.shape[1:])
    assert np.array_equal(one, got[1])
