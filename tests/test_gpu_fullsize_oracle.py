"""Oracle parity AT BASELINE.json's full sizes, where the oracle can in fact run (round-1 VERDICT, "What's weak" 3):

* the column oracle works on ONE column (a 270 x 270 dense problem at p=3, nk=30; 1024 x 1024 at p=4, nk=64), so a sample of
  columns of the config-4 and config-5 grids is compared at their own nk: solve_schur_column_eta / _3, the Helmholtz operator,
  the theta diagnoses, the EOS vectors, one Newton iteration of the vertical solve;
* the horizontal oracle works on ONE patch (12 x 12 elements of the 24-patch sphere), so the applies of the full 103 680-unit
  launch are compared patch by patch on every DoF that no other patch touches (2-form results: every DoF);
* config 3 (24x24x6, shallow water, signed det) the same way for the upwinded operators, config 1 (8x8x6, 6 patches) for every
  operator family and its 6-rank halo (tests/test_halo_gloo.py holds the gloo half).

Tolerances: 1e-10 (north_star) for operators and vectors; for the Schur solves see tests/helpers.py::solve_error_budget -- each
side is held to its OWN system through an extended-precision solve, and the two solutions may differ by what the
conditioning of L_pi makes of the round-off difference between the two assembled operators."""
import numpy as np
import pytest

from tests.helpers import SCALE, dense_from_band, ld_solve, rel_l2, z_levels

pytestmark = pytest.mark.gpu
TOL = 1e-10


# ---------------------------------------------------------------------------------------------------------------------------
def _sphere(pn, ne, npatch, nk, signed_det=False, flat_levels=False):
    from mimsem_amd.device import DeviceMesh, Engine
    from mimsem_amd.geom import Geom
    from mimsem_amd.mesh import CubedSphere, sphere_coords
    from mimsem_amd.topo import Topo
    cs = CubedSphere(pn, ne, npatch); coords = sphere_coords(pn, ne)
    topos = [Topo(cs, p, nk) for p in range(npatch)]
    geoms = [Geom(t, cs, coords, nk, signed_det=signed_det) for t in topos]
    for g in geoms:
        g.set_levels(np.stack([np.zeros(g.n0), np.ones(g.n0)]) if flat_levels else z_levels(nk, g.n0))
    dm = DeviceMesh(topos, geoms, nk=nk, numbering="global")
    return cs, coords, topos, geoms, dm, Engine(dm)


def _oracle_patch(oracle, cs, coords, geom, pi, nk, abs_det=True):
    P = oracle.Patch(cs.pn, cs.pn, cs.nel, nk)
    P.set_sphere_geometry(coords[cs.patches[pi].loc0], abs_det=abs_det)
    P.set_levels(geom.levs)
    return P


class PatchView:
    """slot maps of patch number `i` (position in the DeviceMesh's patch list) inside a global-numbering DeviceMesh, and the masks
    of its local 0/1-form slots that no OTHER patch contributes to (the patch oracle holds their complete sums)"""

    def __init__(self, dm, topos, i):
        t = topos[i]
        self.t = t
        self.s0 = np.searchsorted(dm.gid0, t.loc0); self.s1 = np.searchsorted(dm.gid1, t.loc1)
        assert np.array_equal(dm.gid0[self.s0], t.loc0) and np.array_equal(dm.gid1[self.s1], t.loc1)
        e0 = sum(tt.nElsX ** 2 for tt in topos[:i]); n2e = t.elOrd ** 2
        self.el0 = e0
        self.s2 = e0 * n2e + np.arange(t.n2)
        c1 = np.bincount(np.concatenate([dm.inds1x.ravel(), dm.inds1y.ravel()]), minlength=dm.n1)
        l1 = np.bincount(np.concatenate([t.all_inds1x_l().ravel(), t.all_inds1y_l().ravel()]), minlength=t.n1)
        self.int1 = (l1 > 0) & (l1 == c1[self.s1])
        c0 = np.bincount(dm.inds0.ravel(), minlength=dm.n0)
        l0 = np.bincount(t.all_inds0_l().ravel(), minlength=t.n0)
        self.int0 = (l0 > 0) & (l0 == c0[self.s0])
        self.slots = {0: self.s0, 1: self.s1, 2: self.s2}
        self.mask = {0: self.int0, 1: self.int1, 2: np.ones(t.n2, dtype=bool)}


SPACES = dict(UMAT=(1, None, 1), UHMAT=(1, 2, 1), ROTMAT=(1, 0, 1), WTQUMAT=(1, 1, 2), WMAT=(2, None, 2), WHMAT=(2, 2, 2),
              PMAT=(0, None, 0), PHMAT=(0, 2, 0), UTQWMAT=(2, 1, 1), WTQDUDZ=(1, 1, 2), UTMAT=(1, None, 1))


def _compare_patch_applies(oracle, cs, coords, topos, geoms, dm, eng, nk, patches, levels, cases, seed):
    rng = np.random.default_rng(seed)
    sizes = {0: dm.n0, 1: dm.n1, 2: dm.n2}
    X = {0: rng.standard_normal((nk, dm.n0)), 1: rng.standard_normal((nk, dm.n1)), 2: rng.standard_normal((nk, dm.n2))}
    Fld = {0: rng.standard_normal((nk, dm.n0)) * 1e-4, 1: rng.standard_normal((nk, dm.n1)) * 1e3, 2: rng.uniform(0.5, 1.5, (nk, dm.n2)) * 1e6}
    views = {i: PatchView(dm, topos, i) for i in patches}
    orcs = {i: _oracle_patch(oracle, cs, coords, geoms[i], topos[i].pi, nk) for i in patches}
    worst = {}
    for op, flag in cases:
        sin, sf, sout = SPACES[op]
        nl = nk - 1 if op == "UTMAT" else nk                   # Ut_mat needs thick[lev + 1]
        y = eng.apply(op, eng.tensor(X[sin][:nl]), f=eng.tensor(Fld[sf][:nl]) if sf is not None else None, lev0=0, scale=SCALE, flags=flag).cpu().numpy()
        assert y.shape == (nl, sizes[sout])
        for i in patches:
            v, P = views[i], orcs[i]
            assert v.mask[sout].sum() > 0.7 * v.mask[sout].size          # a 12x12 patch: the large majority of its DoFs are its own
            for k in levels:
                k = min(k, nl - 1)
                want = P.apply(op, X[sin][k, v.slots[sin]], lev=k, scale=SCALE, flag=flag, f1=Fld[sf][k, v.slots[sf]] if sf is not None else None)
                got = y[k, v.slots[sout]]
                m = v.mask[sout]
                err = rel_l2(got[m], want[m])
                worst[op] = max(worst.get(op, 0.0), err)
                assert err < TOL, (op, i, k, err)
    return worst


# ---- config 4: p=3, 24x24x6 sphere x 30 levels -----------------------------------------------------------------------------
@pytest.fixture(scope="module")
def cfg4():
    out = _sphere(3, 24, 24, 30)
    assert out[4].nEl * 30 == 103680
    return out


def test_config4_patch_applies_match_oracle(cfg4, oracle):
    """the 103 680-unit launch of every operator family vs the patch oracle (eul/Assembly.cpp:66-153, 416-474, 933-986, 1030-1083,
    1243-1299, 2004-2098, 1490-1640) on three of the 24 patches and three of the 30 levels"""
    cs, coords, topos, geoms, dm, eng = cfg4
    cases = [("UMAT", 1), ("UHMAT", 1), ("ROTMAT", 0), ("WTQUMAT", 0), ("WMAT", 1), ("WHMAT", 1), ("PMAT", 0), ("PHMAT", 0),
             ("UTQWMAT", 0), ("WTQDUDZ", 0), ("UTMAT", 0)]
    worst = _compare_patch_applies(oracle, cs, coords, topos, geoms, dm, eng, 30, patches=(2, 13, 21), levels=(0, 16, 29), cases=cases, seed=404)
    print("config 4 worst relative L2 per operator:", {k: "%.1e" % v for k, v in worst.items()})


def _col_fields(dm, eng, nk, rng, ranges):
    n2, nEl = eng.n2e, dm.nEl
    area = float(dm.det.mean()) * 4.0 / n2; dz = float(dm.thick.mean())
    lev = lambda nl, lo, hi: rng.uniform(lo, hi, (nEl, nl * n2)) * area * dz
    return dict(rho=lev(nk, *ranges["rho"]), rt=lev(nk, *ranges["rt"]), theta=lev(nk + 1, *ranges["theta"]) / dz,
                pi=lev(nk, *ranges["pi"]), eta=lev(nk, *ranges["eta"]), velz=lev(nk - 1, -1.0, 1.0) / dz, thetaL=lev(nk, *ranges["theta"]))


def _hydrostatic_fields(dm, geoms, eng, nk, pn, rng, noise):
    """an EOS-consistent hydrostatic column (theta = 300 K + 4 K/km, dPi/dz = -g/theta) as 2-form DoFs (value x sub-cell area x det
    x thickness; interface fields without the thickness) with `noise` relative perturbations: the regime the solves run in"""
    from mimsem_amd.geom import gll_points
    nEl, n2 = dm.nEl, eng.n2e
    wd = np.diff(gll_points(pn)); wj = np.outer(wd, wd).ravel()
    detm = dm.det.mean(axis=1)                                           # [nEl]
    thm = dm.thick.mean(axis=2).T                                        # [nEl, nk]
    zi = np.mean([g.levs.mean(axis=1) for g in geoms], axis=0); zm = 0.5 * (zi[1:] + zi[:-1])
    th_v = 300.0 + 0.004 * zm; thI_v = 300.0 + 0.004 * zi
    pi_v = 1004.5 - (9.80616 / 0.004) * np.log(th_v / 300.0)
    rho_v = (1.0e5 / 287.0) * (pi_v / 1004.5) ** (717.5 / 287.0) / th_v
    pert = lambda nl: 1.0 + noise * rng.standard_normal((nEl, nl * n2))
    lev = lambda v: (detm[:, None, None] * thm[:, :, None] * v[None, :, None] * wj[None, None, :]).reshape(nEl, nk * n2) * pert(nk)
    itf = lambda v, nl: (detm[:, None, None] * v[None, :nl, None] * wj[None, None, :]).reshape(nEl, nl * n2) * pert(nl)
    return dict(rho=lev(rho_v), rt=lev(rho_v * th_v), pi=lev(pi_v), thetaL=lev(th_v), eta=lev(np.log(th_v)),
                theta=itf(thI_v, nk + 1), velz=itf(np.ones(nk + 1), nk - 1) * 0.5 * rng.standard_normal((nEl, (nk - 1) * n2)))


RANGES4 = dict(rho=(0.5, 1.2), rt=(250.0, 400.0), theta=(280.0, 320.0), pi=(700.0, 1000.0), eta=(5.0, 6.0))


def _sample_columns(topos, per_patch, patches, rng):
    """(global element, patch list position, local element) triples"""
    out = []
    for i in patches:
        nel = topos[i].nElsX ** 2
        e0 = sum(t.nElsX ** 2 for t in topos[:i])
        for le in sorted(rng.choice(nel, per_patch, replace=False)):
            out.append((e0 + int(le), i, int(le)))
    return out


def _check_schur_eta(eng, P_of, cols, F, Fs, dt, nk, n2, report):
    """solve_schur_column_eta on the sampled columns: operator, solution, updated right-hand sides; error budget in helpers"""
    from tests.helpers import solve_error_budget
    t = eng.tensor
    L = eng.helmholtz_blocks(dt, t(F["thetaL"]), t(F["rho"]), t(F["eta"]), t(F["pi"]))
    dF = [t(f) for f in Fs]
    d = eng.solve_schur_eta(dt, t(F["thetaL"]), t(F["rho"]), t(F["eta"]), t(F["pi"]), *dF)
    names = ("d_u", "d_rho", "d_eta", "d_pi")
    for e, i, le in cols:
        P = P_of(i)
        ex, ey = le % P.nElsX, le // P.nElsX
        ref = P.solve_schur_column_eta(ex, ey, dt, F["thetaL"][e], F["rho"][e], F["eta"][e], F["pi"][e], *[f[e] for f in Fs])
        Ld = dense_from_band(L[e].cpu().numpy(), nk, n2, lo=1)
        rhs_hip = dF[3][e].cpu().numpy()
        b = solve_error_budget(Ld, rhs_hip, d[3][e].cpu().numpy(), ref["L_pi"], ref["F_pi"], ref["d_pi"])
        report.append(b)
        assert b["rel_L"] < TOL, ("L_pi", e, b)
        assert rel_l2(rhs_hip, ref["F_pi"]) < TOL
        assert b["hip_vs_own_system"] < 1e-10, b          # the block-Thomas + refinement solves ITS system to the north-star level
        assert b["diff"] < max(TOL, 4.0 * b["diff_of_exact_solutions"] + b["hip_vs_own_system"] + b["oracle_vs_own_system"]), b
        for name, got in zip(names[:3], d[:3]):
            # back substitutions inherit d_pi's error (amplified by at most the norms of G_pi / DIV): same budget
            assert rel_l2(got[e].cpu().numpy(), ref[name]) < max(TOL, 10.0 * b["diff"]), (name, e, b)
        for name, got in zip(("F_u", "F_rho", "F_eta"), dF[:3]):
            assert rel_l2(got[e].cpu().numpy(), ref[name]) < max(TOL, 10.0 * b["diff"]), (name, e, b)


def test_config4_sampled_columns_schur_eta(cfg4, oracle):
    """20 of the 3 456 columns at nk = 30 against orc_solve_schur_column_eta (eul/VertSolve.cpp:677-823)"""
    cs, coords, topos, geoms, dm, eng = cfg4
    nk, n2 = 30, eng.n2e
    rng = np.random.default_rng(4401)
    F = _col_fields(dm, eng, nk, rng, RANGES4)
    Fs = [rng.standard_normal((dm.nEl, n * n2)) * 1e8 for n in (nk - 1, nk, nk, nk)]
    cols = _sample_columns(topos, 5, (0, 9, 14, 23), rng)
    cache = {}
    P_of = lambda i: cache.setdefault(i, _oracle_patch(oracle, cs, coords, geoms[i], topos[i].pi, nk))
    report = []
    _check_schur_eta(eng, P_of, cols, F, Fs, 75.0, nk, n2, report)
    print("config 4 schur_eta budget (max over %d columns):" % len(cols), {k: "%.1e" % max(r[k] for r in report) for k in report[0]})


def test_config4_sampled_columns_schur_3_theta_eos(cfg4, oracle):
    """the same sample: solve_schur_column_3 (eul/VertSolve.cpp:504-675), diagTheta2 / diagTheta_L2 (:289-352), the EOS vectors
    (eul/VertOps.cpp:732-787, 987-1047, 1204-1305) and three block operators at nk = 30"""
    from tests.helpers import solve_error_budget
    cs, coords, topos, geoms, dm, eng = cfg4
    nk, n2 = 30, eng.n2e
    rng = np.random.default_rng(4402)
    F = _col_fields(dm, eng, nk, rng, RANGES4)
    N, Nm = nk * n2, (nk - 1) * n2
    Fs = [rng.standard_normal((dm.nEl, n)) * 1e8 for n in (Nm, N, N, N)]
    cols = _sample_columns(topos, 4, (1, 8, 15, 22), rng)
    cache = {}
    P_of = lambda i: cache.setdefault(i, _oracle_patch(oracle, cs, coords, geoms[i], topos[i].pi, nk))
    t = eng.tensor
    dt = 75.0
    dF = [t(f) for f in Fs]
    d_u, d_rho, d_rt, d_pi, L = eng.solve_schur_3(dt, t(F["theta"]), t(F["velz"]), t(F["rho"]), t(F["rt"]), t(F["pi"]), *dF, want_L=True)
    th0 = eng.diag_theta(0, t(F["rho"]), t(F["rt"])); th1 = eng.diag_theta(1, t(F["rho"]), t(F["rt"]))
    eos = [eng.column_eos(0, t(F["rt"]), t(F["pi"])), eng.column_eos(1, t(F["rt"]), None, 1004.5 * (287.0 / 1e5) ** (287.0 / 717.5), 287.0 / 717.5),
           eng.column_eos(2, t(F["thetaL"]), t(F["eta"])), eng.column_eos(3, t(F["rho"]), t(F["eta"] * 1e-3))]
    blk = {op: eng.colop_blocks(op, f1=t(F[k]) if k else None, flags=fl) for op, k, fl in (("CONST_RHO", "rho", 0), ("LINEAR_RT", "rt", 1), ("EOS_BLOCK", "rt", 0))}
    report = []
    for e, i, le in cols:
        P = P_of(i)
        ex, ey = le % P.nElsX, le // P.nElsX
        ref = P.solve_schur_column_3(ex, ey, dt, F["theta"][e], F["velz"][e], F["rho"][e], F["rt"][e], F["pi"][e], *[f[e] for f in Fs])
        Ld = dense_from_band(L[e].cpu().numpy(), nk, n2, lo=2)
        b = solve_error_budget(Ld, dF[2][e].cpu().numpy(), d_rt[e].cpu().numpy(), ref["L"], ref["F_rt"], ref["d_rt"])
        report.append(b)
        assert b["rel_L"] < 1e-9, ("L_rt_rt", e, b)               # a product of ten factors with three explicit inverses in it
        assert b["hip_vs_own_system"] < 1e-10, b
        assert b["diff"] < max(TOL, 4.0 * b["diff_of_exact_solutions"] + b["hip_vs_own_system"] + b["oracle_vs_own_system"]), b
        for name, got in (("d_u", d_u), ("d_pi", d_pi), ("d_rho", d_rho)):
            # back substitutions: products of the solution with G / D / M^-1 factors (cancellation amplifies the solve errors of both sides)
            assert rel_l2(got[e].cpu().numpy(), ref[name]) < max(TOL, 50.0 * (b["diff"] + b["hip_vs_own_system"] + b["oracle_vs_own_system"])), (name, e, b)
        # theta diagnoses: block-diagonal systems, each block inverted -- conditioning of one 9x9 mass block only
        assert rel_l2(th0[e].cpu().numpy(), P.diag_theta_L2(ex, ey, F["rho"][e], F["rt"][e])) < TOL
        assert rel_l2(th1[e].cpu().numpy(), P.diag_theta2(ex, ey, F["rho"][e], F["rt"][e])) < TOL
        want = [P.eos_residual(ex, ey, F["rt"][e], F["pi"][e]), P.eos_rhs(ex, ey, F["rt"][e], 1004.5 * (287.0 / 1e5) ** (287.0 / 717.5), 287.0 / 717.5),
                P.const_log_theta_plus_eta(ex, ey, F["thetaL"][e], F["eta"][e]), P.const_rho_exp_eta(ex, ey, F["rho"][e], F["eta"][e] * 1e-3)]
        for g, w in zip(eos, want):
            assert rel_l2(g[e].cpu().numpy(), w) < TOL
        for (op, k, fl) in (("CONST_RHO", "rho", 0), ("LINEAR_RT", "rt", 1), ("EOS_BLOCK", "rt", 0)):
            D = P.colop_dense(op, ex, ey, flag=fl, f1=F[k][e])
            got = blk[op][e].cpu().numpy()
            for r in range(got.shape[0]):
                assert rel_l2(got[r], D[r*n2:(r+1)*n2, r*n2:(r+1)*n2]) < TOL, (op, e, r)
    print("config 4 schur_3 budget (max over %d columns):" % len(cols), {k: "%.1e" % max(r[k] for r in report) for k in report[0]})


def test_config4_one_newton_iteration_on_sampled_columns(cfg4, oracle):
    """one iteration of VertSolve::solve_schur_eta (eul/VertSolve.cpp:1721-1973) for all 3 456 columns; 16 sampled columns against
    the column-by-column restatement oracle/vert_oracle.py at nk = 30"""
    from mimsem_amd.geom import gll_points
    from mimsem_amd.vertsolve import VertSolve
    from oracle import vert_oracle
    cs, coords, topos, geoms, dm, eng = cfg4
    nEl, n2, nk = dm.nEl, eng.n2e, 30
    rng = np.random.default_rng(4403)
    wd = np.diff(gll_points(3)); wj = np.outer(wd, wd).ravel()
    cell = dm.det.mean(axis=1)[:, None, None] * dm.thick.mean(axis=2).T[:, :, None] * wj[None, None, :]
    zl = np.mean([g.levs.mean(axis=1) for g in geoms], axis=0); zm = 0.5 * (zl[:-1] + zl[1:])
    th_v = 300.0 + 0.004 * zm
    pi_v = 1004.5 - (9.80616 / 0.004) * np.log(th_v / 300.0)
    rho_v = (1.0e5 / 287.0) * (pi_v / 1004.5) ** (717.5 / 287.0) / th_v
    colv = lambda v: (cell * v[None, :, None]).reshape(nEl, nk * n2) * (1.0 + 1e-4 * rng.standard_normal((nEl, nk * n2)))
    dt = 75.0
    vs = VertSolve(eng, dt)
    levs = np.zeros((nk + 1, dm.nq))
    for g in geoms:
        levs[:, np.searchsorted(dm.gidq, g.loc0[np.arange(g.n0)])] = g.levs
    zv = vs.init_gz(levs)
    velz, rho, rt, exner = np.zeros((nEl, (nk - 1) * n2)), colv(rho_v), colv(rho_v * th_v), colv(pi_v)
    t = eng.tensor
    got = vs.solve_schur_eta(t(velz), t(rho), t(rt), t(exner), zv, maxit=1, tol=0.0)
    zv_h = zv.cpu().numpy()
    for i in (3, 12, 17, 20):
        P = _oracle_patch(oracle, cs, coords, geoms[i], topos[i].pi, nk)
        e0 = sum(tt.nElsX ** 2 for tt in topos[:i]); ne = P.nEl
        sel = sorted(rng.choice(ne, 4, replace=False))
        sl = slice(e0, e0 + ne)
        want = vert_oracle.solve_schur_eta(P, dt, velz[sl], rho[sl], rt[sl], exner[sl], zv_h[sl], 1, columns=sel)
        for a, b, name in zip(got, want[:4], ("velz", "rho", "rt", "exner")):
            for le in sel:
                ref = b[le]
                assert np.all(np.isfinite(ref)), name
                # the state moves by ~1e-4 relative in this iteration; the comparison is on the new state
                assert rel_l2(a[e0 + le].cpu().numpy(), ref) < (1e-8 if name == "velz" else TOL), (name, i, le)


# ---- config 5: p=4, 32x32 periodic box x 64 levels ------------------------------------------------------------------------
@pytest.fixture(scope="module")
def cfg5():
    from mimsem_amd.device import DeviceMesh, Engine
    from mimsem_amd.geom import BoxGeom
    from mimsem_amd.mesh import PeriodicBox, box_coords
    from mimsem_amd.topo import Topo
    pn, ne, npr, nk, lx = 4, 32, 4, 64, 1000.0
    bx = PeriodicBox(pn, ne, npr); bc = box_coords(pn, ne, lx)
    topos = [Topo(bx, p, nk) for p in range(npr)]
    geoms = [BoxGeom(t, bx, bc, nk, lx) for t in topos]
    rng = np.random.default_rng(5500)
    dz = 1500.0 / nk                                                       # box/Bubble: uniform levels; here +-5 % of dz per quad point
    levs = np.repeat(np.linspace(0.0, 1500.0, nk + 1)[:, None], geoms[0].n0, axis=1)
    levs[1:-1] += 0.05 * dz * rng.uniform(-1, 1, (nk - 1, geoms[0].n0))
    for g in geoms:
        g.set_levels(levs)
    dm = DeviceMesh(topos, geoms, nk=nk, numbering="global")
    assert dm.nEl * nk == 65536
    return bx, topos, geoms, dm, Engine(dm), levs


def _box_patch(oracle, bx, geom, levs, nk):
    P = oracle.Patch(bx.pn, bx.pn, bx.nel, nk)
    P.set_metric(geom.det, geom.J); P.set_levels(levs)
    return P


def test_config5_patch_applies_match_oracle(cfg5, oracle):
    """p = 4 (25 quadrature points, 40 x 40 element blocks), 64 levels in one launch vs the patch oracle on two of the four patches"""
    bx, topos, geoms, dm, eng, levs = cfg5
    nk = 64
    rng = np.random.default_rng(5501)
    X = {0: rng.standard_normal((nk, dm.n0)), 1: rng.standard_normal((nk, dm.n1)), 2: rng.standard_normal((nk, dm.n2))}
    Fld = {0: rng.standard_normal((nk, dm.n0)) * 1e-4, 1: rng.standard_normal((nk, dm.n1)) * 10.0, 2: rng.uniform(0.5, 1.5, (nk, dm.n2)) * 1e3}
    for op, flag in (("UMAT", 1), ("UHMAT", 1), ("ROTMAT", 0), ("WTQUMAT", 0), ("WHMAT", 1), ("PHMAT", 0)):
        sin, sf, sout = SPACES[op]
        y = eng.apply(op, eng.tensor(X[sin]), f=eng.tensor(Fld[sf]) if sf is not None else None, lev0=0, scale=SCALE, flags=flag).cpu().numpy()
        for i in (0, 3):
            v = PatchView(dm, topos, i); P = _box_patch(oracle, bx, geoms[i], levs, nk)
            for k in (0, 31, 63):
                want = P.apply(op, X[sin][k, v.slots[sin]], lev=k, scale=SCALE, flag=flag, f1=Fld[sf][k, v.slots[sf]] if sf is not None else None)
                m = v.mask[sout]
                assert m.sum() > 0.7 * m.size
                assert rel_l2(y[k, v.slots[sout]][m], want[m]) < TOL, (op, i, k)


def test_config5_sampled_columns_match_oracle(cfg5, oracle):
    """8 of the 1 024 columns at p = 4, nk = 64 (16 x 16 blocks, 1 024 x 1 024 per column): solve_schur_column_eta, the box twin of
    solve_schur_column_3 (box/VertSolve.cpp:879-1058), theta diagnoses and EOS vectors"""
    from tests.helpers import solve_error_budget
    bx, topos, geoms, dm, eng, levs = cfg5
    nk, n2 = 64, eng.n2e
    rng = np.random.default_rng(5502)
    F = _hydrostatic_fields(dm, geoms, eng, nk, 4, rng, noise=1e-2)
    N, Nm = nk * n2, (nk - 1) * n2
    Fs = [rng.standard_normal((dm.nEl, n)) * 1e8 for n in (Nm, N, N, N)]
    cols = _sample_columns(topos, 2, (0, 1, 2, 3), rng)
    cache = {}
    P_of = lambda i: cache.setdefault(i, _box_patch(oracle, bx, geoms[i], levs, nk))
    report = []
    _check_schur_eta(eng, P_of, cols, F, Fs, 0.5, nk, n2, report)
    print("config 5 schur_eta budget (max over %d columns):" % len(cols), {k: "%.1e" % max(r[k] for r in report) for k in report[0]})
    t = eng.tensor
    dF = [t(f) for f in Fs]
    dt = 0.5
    d_u, d_rho, d_rt, d_pi, L = eng.solve_schur_3(dt, t(F["theta"]), t(F["velz"]), t(F["rho"]), t(F["rt"]), t(F["pi"]), *dF, want_L=True, flags=3)
    th0 = eng.diag_theta(0, t(F["rho"]), t(F["rt"])); th1 = eng.diag_theta(1, t(F["rho"]), t(F["rt"]))
    eos = eng.column_eos(0, t(F["rt"]), t(F["pi"]))
    rep3 = []
    for e, i, le in cols[::2]:
        P = P_of(i)
        ex, ey = le % P.nElsX, le // P.nElsX
        ref = P.solve_schur_column_3(ex, ey, dt, F["theta"][e], F["velz"][e], F["rho"][e], F["rt"][e], F["pi"][e], *[f[e] for f in Fs], flags=3)
        Ld = dense_from_band(L[e].cpu().numpy(), nk, n2, lo=2)
        b = solve_error_budget(Ld, dF[2][e].cpu().numpy(), d_rt[e].cpu().numpy(), ref["L"], ref["F_rt"], ref["d_rt"])
        rep3.append(b)
        assert b["rel_L"] < 1e-9 and b["hip_vs_own_system"] < 1e-10, b
        assert b["diff"] < max(TOL, 4.0 * b["diff_of_exact_solutions"] + b["hip_vs_own_system"] + b["oracle_vs_own_system"]), b
        for name, got in (("d_u", d_u), ("d_pi", d_pi), ("d_rho", d_rho)):
            assert rel_l2(got[e].cpu().numpy(), ref[name]) < max(TOL, 50.0 * (b["diff"] + b["hip_vs_own_system"] + b["oracle_vs_own_system"])), (name, e, b)
        assert rel_l2(th0[e].cpu().numpy(), P.diag_theta_L2(ex, ey, F["rho"][e], F["rt"][e])) < TOL
        assert rel_l2(th1[e].cpu().numpy(), P.diag_theta2(ex, ey, F["rho"][e], F["rt"][e])) < TOL
        assert rel_l2(eos[e].cpu().numpy(), P.eos_residual(ex, ey, F["rt"][e], F["pi"][e])) < TOL
    print("config 5 schur_3 (box) budget:", {k: "%.1e" % max(r[k] for r in rep3) for k in rep3[0]})


def test_config5_pivoted_band_lu_at_the_size_limit(cfg5):
    """mimsem_column_set_pivot_fallback(2) on config 5's columns: 1 024 unknowns per column (the limit of column_pivot.inc), scalar
    half-bandwidth 31 (Helmholtz system) and 47 (box solve_schur_column_3) -- every column through the band LU with partial pivoting,
    status 3, the solutions of the block sweep reproduced"""
    bx, topos, geoms, dm, eng, levs = cfg5
    nk, n2 = 64, eng.n2e
    rng = np.random.default_rng(5503)
    F = _hydrostatic_fields(dm, geoms, eng, nk, 4, rng, noise=1e-2)
    N, Nm = nk * n2, (nk - 1) * n2
    Fs = [rng.standard_normal((dm.nEl, n)) * 1e8 for n in (Nm, N, N, N)]
    t = eng.tensor
    runs = (lambda: eng.solve_schur_eta(0.5, t(F["thetaL"]), t(F["rho"]), t(F["eta"]), t(F["pi"]), *[t(f) for f in Fs]),
            lambda: eng.solve_schur_3(0.5, t(F["theta"]), t(F["velz"]), t(F["rho"]), t(F["rt"]), t(F["pi"]), *[t(f) for f in Fs], flags=3))
    for run in runs:
        ref = run()
        assert eng.solve_status()[0] == 0
        eng.set_pivot_fallback(2)
        try:
            out = run()
            nbad, st, ratio = eng.solve_status()
        finally:
            eng.set_pivot_fallback(1)
        assert nbad == 0 and (st == 3).all() and ratio.max() < 1e-10, (nbad, np.unique(st), float(ratio.max()))
        for a, b in zip(out, ref):
            assert rel_l2(a.cpu().numpy(), b.cpu().numpy()) < 1e-9


# ---- config 3: p=3, 24x24x6 sphere, shallow water (signed det, no thickness) -------------------------------------------
def test_config3_upwinded_operators_match_oracle(oracle):
    """Phmat::assemble_up / RotMat_up::assemble (src/Assembly.cpp:499-567, 1784-1853) and the plain src-flavour Umat / RotMat on the
    3 456-element single-level sphere of configs 2/3 (6 patches of 24 x 24) vs the patch oracle with signed det"""
    cs, coords, topos, geoms, dm, eng = _sphere(3, 24, 6, 1, signed_det=True, flat_levels=True)
    rng = np.random.default_rng(3301)
    fac, dt = 0.5, 360.0
    ug = rng.uniform(-1, 1, dm.n1) * float(np.abs(dm.det).mean()) * 0.2 / (fac * dt)
    hg = rng.uniform(0.5, 1.5, dm.n2) * 1e4; qg = rng.standard_normal(dm.n0) * 1e-4
    x0, x1 = rng.standard_normal(dm.n0), rng.standard_normal(dm.n1)
    t = eng.tensor
    y_ph = eng.apply_up("PHMAT_UP", t(x0), t(hg), t(ug), fac, dt, lev0=0).cpu().numpy()
    y_ro = eng.apply_up("ROTMAT_UP", t(x1), t(qg), t(ug), fac, dt, lev0=0).cpu().numpy()
    y_um = eng.apply("UMAT", t(x1), lev0=0, scale=1.0, flags=0).cpu().numpy()
    y_r = eng.apply("ROTMAT", t(x1), f=t(qg), lev0=0, scale=1.0).cpu().numpy()
    for i in (1, 4):
        v = PatchView(dm, topos, i)
        P = _oracle_patch(oracle, cs, coords, geoms[i], topos[i].pi, 1, abs_det=False)
        ul = ug[v.s1]
        want, _ = P.apply_up(0, x0[v.s0], fac, dt, hg[v.s2], ul)
        assert rel_l2(y_ph[v.s0][v.int0], want[v.int0]) < TOL
        want, _ = P.apply_up(1, x1[v.s1], fac, dt, qg[v.s0], ul)
        assert rel_l2(y_ro[v.s1][v.int1], want[v.int1]) < TOL
        assert rel_l2(y_um[v.s1][v.int1], P.apply("UMAT", x1[v.s1], lev=0, scale=1.0, flag=0)[v.int1]) < TOL
        assert rel_l2(y_r[v.s1][v.int1], P.apply("ROTMAT", x1[v.s1], lev=0, scale=1.0, flag=0, f1=qg[v.s0])[v.int1]) < TOL


# ---- config 1: p=3, 8x8x6 sphere, 6 patches (the reference's own CPU-runnable decomposition) -----------------------------
def test_config1_grid_operator_parity(oracle):
    """BASELINE config 1 grid (p=3, 8 x 8 elements per face, 6 ranks = 6 patches, one level): every operator family on every patch
    against the patch oracle, in the reference's per-rank LOCAL layout (what each of the 6 MPI ranks holds) and in the global one"""
    from mimsem_amd.device import DeviceMesh, Engine
    cs, coords, topos, geoms, dm, eng = _sphere(3, 8, 6, 1)
    assert dm.nEl == 384 and cs.nDofs1G == 6912 and cs.nDofs0G == 3458 and cs.nDofs2G == 3456
    cases = [("UMAT", 1), ("UHMAT", 1), ("ROTMAT", 0), ("WTQUMAT", 0), ("WMAT", 1), ("WHMAT", 1), ("PMAT", 0), ("PHMAT", 0), ("UTQWMAT", 0)]
    _compare_patch_applies(oracle, cs, coords, topos, geoms, dm, eng, 1, patches=range(6), levels=(0,), cases=cases, seed=101)
    # one rank's view: local (ghosted) numbering, complete comparison incl. the ghost rows the rank computes partial sums for
    rng = np.random.default_rng(102)
    for pi in (0, 5):
        engl = Engine(DeviceMesh([topos[pi]], [geoms[pi]], nk=1, numbering="local"))
        P = _oracle_patch(oracle, cs, coords, geoms[pi], pi, 1)
        x1 = rng.standard_normal(P.n1); h = rng.uniform(0.5, 1.5, P.n2) * 1e6; q = rng.standard_normal(P.n0) * 1e-4
        for op, f, flag in (("UMAT", None, 1), ("UHMAT", h, 1), ("ROTMAT", q, 0)):
            got = engl.apply(op, engl.tensor(x1), f=engl.tensor(f) if f is not None else None, lev0=0, scale=SCALE, flags=flag).cpu().numpy()
            assert rel_l2(got, P.apply(op, x1, lev=0, scale=SCALE, flag=flag, f1=f)) < TOL, (op, pi)
