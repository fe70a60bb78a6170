"""The wave-level fused form of the 1-form -> 1-form operators (k_apply_wave + k_wave_perim, mimsem_amd/csrc/elem_wave.inc) on small
spheres in the reference's global numbering (where its slot-pair plan exists): against the two-pass form of the same library
(MIMSEM_WAVE=0, itself checked against the oracle operator by operator in test_gpu_horizontal.py) and, at p = 3, against the oracle
directly.  The full-size oracle comparisons of test_gpu_fullsize_oracle.py run the wave form too (config 1, 3, 4, 5 grids)."""
import ctypes as C

import numpy as np
import pytest

from tests.helpers import SCALE, rel_l2, z_levels

pytestmark = pytest.mark.gpu
NK = 11                                   # 8 + 3: a full chunk and a ragged one


def _mesh(pn, ne):
    from mimsem_amd.device import DeviceMesh
    from mimsem_amd.geom import Geom
    from mimsem_amd.mesh import CubedSphere, sphere_coords
    from mimsem_amd.topo import Topo
    cs = CubedSphere(pn, ne, 6); coords = sphere_coords(pn, ne)
    topos = [Topo(cs, p, NK) for p in range(6)]
    geoms = [Geom(t, cs, coords, NK) for t in topos]
    for g in geoms:
        g.set_levels(z_levels(NK, g.n0))
    return cs, topos, geoms, DeviceMesh(topos, geoms, nk=NK, numbering="global")


@pytest.fixture(scope="module", params=[(2, 4), (3, 4), (4, 2), (3, 3), (2, 5)], ids=lambda p: "p%d_ne%d" % p)
def engines(request):
    import os
    from mimsem_amd.device import Engine
    pn, ne = request.param
    cs, topos, geoms, dm = _mesh(pn, ne)
    wave = Engine(dm)
    st = (C.c_int * 5)()
    assert wave.L.mimsem_op_wave_stats(wave.ctx, NK, st) == 1 and st[0] > 0 and st[4] == 8, list(st)
    os.environ["MIMSEM_WAVE"] = "0"
    try:
        two = Engine(dm)
    finally:
        del os.environ["MIMSEM_WAVE"]
    os.environ["MIMSEM_WAVE2"] = "2"                     # Wmat too on the DPP kernel (off by default: slower there)
    try:
        wave_all = Engine(dm)
    finally:
        del os.environ["MIMSEM_WAVE2"]
    wave.wave_all = wave_all
    os.environ["MIMSEM_WAVE_FIN"] = "1"                  # opt-in: the perimeter slots finished inside k_apply_wave (no second launch)
    try:
        wave.fin = Engine(dm)
    finally:
        del os.environ["MIMSEM_WAVE_FIN"]
    os.environ["MIMSEM_WAVE_TILE"] = "1"                 # opt-in (round 5): four wave-groups per workgroup, their shared slots summed in LDS
    try:
        wave.tile = Engine(dm)
    finally:
        del os.environ["MIMSEM_WAVE_TILE"]
    assert two.L.mimsem_op_wave_stats(two.ctx, NK, st) == 0
    return pn, dm, wave, two, (cs, topos, geoms)


CASES = [("UMAT", 1), ("UMAT", 0), ("UHMAT", 1), ("UHMAT", 0), ("ROTMAT", 0), ("UTMAT", 0), ("UTMAT_H", 0)]


@pytest.mark.parametrize("op,fl", CASES, ids=["%s_%d" % c for c in CASES])
def test_wave_form_equals_two_pass_form(engines, op, fl):
    import torch
    pn, dm, wave, two, _ = engines
    r = np.random.default_rng(17)
    x = r.standard_normal((NK, dm.n1))
    f = {"UHMAT": r.uniform(0.5, 1.5, (NK, dm.n2)) * 1e6, "UTMAT_H": r.uniform(0.5, 1.5, (NK, dm.n2)) * 1e6,
         "ROTMAT": r.standard_normal((NK, dm.n0)) * 1e-4}.get(op)
    nl = NK - 1 if op == "UTMAT" else NK
    for lev0, nlev in ((0, nl), (2, 3), (5, 1)):
        xs = x[:nlev]; fs = None if f is None else f[:nlev]
        a = wave.apply(op, wave.tensor(xs), f=None if fs is None else wave.tensor(fs), lev0=lev0, scale=SCALE, flags=fl)
        b = two.apply(op, two.tensor(xs), f=None if fs is None else two.tensor(fs), lev0=lev0, scale=SCALE, flags=fl)
        assert rel_l2(a.cpu().numpy(), b.cpu().numpy()) < 1e-13, (op, lev0, nlev)
        for k in range(nlev):                                           # level by level too: a wrong level would hide in the norm
            assert rel_l2(a[k].cpu().numpy(), b[k].cpu().numpy()) < 1e-12, (op, lev0, k)
    base = r.standard_normal((nl, dm.n1))
    ya, yb = wave.tensor(base), two.tensor(base)
    wave.apply(op, wave.tensor(x[:nl]), f=None if f is None else wave.tensor(f[:nl]), lev0=0, scale=SCALE, flags=fl | 2, alpha=-0.375, out=ya)
    two.apply(op, two.tensor(x[:nl]), f=None if f is None else two.tensor(f[:nl]), lev0=0, scale=SCALE, flags=fl | 2, alpha=-0.375, out=yb)
    assert rel_l2(ya.cpu().numpy(), yb.cpu().numpy()) < 1e-13, op
    # run-to-run reproducible bit for bit (fixed summation order, no atomics)
    a1 = wave.apply(op, wave.tensor(x[:nl]), f=None if f is None else wave.tensor(f[:nl]), lev0=0, scale=SCALE, flags=fl)
    a2 = wave.apply(op, wave.tensor(x[:nl]), f=None if f is None else wave.tensor(f[:nl]), lev0=0, scale=SCALE, flags=fl)
    assert torch.equal(a1, a2)


@pytest.mark.parametrize("op,fl", CASES, ids=["%s_%d" % c for c in CASES])
def test_finishing_phase_equals_perimeter_pass(engines, op, fl):
    """Round 3 experiment (MIMSEM_WAVE_FIN=1): the perimeter slots finished inside k_apply_wave by whichever wave-group
    reaches a side second -- bit for bit what the default form (k_wave_perim, second launch) writes: both add the lower group's partial
    sum first.  Inputs change from launch to launch, so a partial sum read stale (from an earlier launch) cannot go unnoticed."""
    import torch
    pn, dm, per, two, _ = engines
    wave = per.fin
    r = np.random.default_rng(29)
    nl = NK - 1 if op == "UTMAT" else NK
    f = {"UHMAT": r.uniform(0.5, 1.5, (NK, dm.n2)) * 1e6, "UTMAT_H": r.uniform(0.5, 1.5, (NK, dm.n2)) * 1e6,
         "ROTMAT": r.standard_normal((NK, dm.n0)) * 1e-4}.get(op)
    ft = None if f is None else wave.tensor(f)
    ya, yb = wave.zeros(nl, dm.n1), per.zeros(nl, dm.n1)
    for it in range(12):
        x = wave.tensor(r.standard_normal((nl, dm.n1)) * (1.0 + it))
        for lev0, nlev in ((0, nl), (1, 3), (4, 1), (0, 8)):
            fs = None if ft is None else ft[:nlev]
            a = wave.apply(op, x[:nlev], f=fs, lev0=lev0, scale=SCALE, flags=fl)
            b = per.apply(op, x[:nlev], f=fs, lev0=lev0, scale=SCALE, flags=fl)
            assert torch.equal(a, b), (op, it, lev0, nlev, int((a != b).sum()))
        base = wave.tensor(r.standard_normal((nl, dm.n1)))
        ya.copy_(base); yb.copy_(base)
        wave.apply(op, x, f=None if ft is None else ft[:nl], lev0=0, scale=SCALE, flags=fl | 2, alpha=0.25, out=ya)
        per.apply(op, x, f=None if ft is None else ft[:nl], lev0=0, scale=SCALE, flags=fl | 2, alpha=0.25, out=yb)
        assert torch.equal(ya, yb), (op, it, "accumulate", int((ya != yb).sum()))


@pytest.mark.parametrize("op,fl", CASES, ids=["%s_%d" % c for c in CASES])
def test_tile_mode_equals_perimeter_pass(engines, op, fl):
    """Round 5 experiment (MIMSEM_WAVE_TILE=1): the four wavefronts of a workgroup take the four wave-groups of a tile; the
    partial sums of the slots those groups share meet in the workgroup's LDS behind ONE barrier per work item and go straight into y,
    the perimeter pass finishes the tile's outer perimeter only.  Bit for bit the default form (a sum of two parts either way), for
    whole and ragged level ranges, single levels, the accumulate form and inputs that change from launch to launch."""
    import torch
    pn, dm, per, two, _ = engines
    wave = per.tile
    if not wave.L.mimsem_build_has_experiments():
        pytest.skip("tile mode is a closed experiment: compiled in only with -DMIMSEM_WITH_EXPERIMENTS (MIMSEM_LIB=build_ab/libmimsem_hip_exp.so)")
    st, st0 = (C.c_int * 5)(), (C.c_int * 5)()
    assert wave.L.mimsem_op_wave_stats(wave.ctx, NK, st) == 1 and per.L.mimsem_op_wave_stats(per.ctx, NK, st0) == 1
    if pn in (3, 4):
        assert st[3] < st0[3], (list(st), list(st0))                    # fewer slots left to the perimeter pass: the tiles exist
    r = np.random.default_rng(31)
    nl = NK - 1 if op == "UTMAT" else NK
    f = {"UHMAT": r.uniform(0.5, 1.5, (NK, dm.n2)) * 1e6, "UTMAT_H": r.uniform(0.5, 1.5, (NK, dm.n2)) * 1e6,
         "ROTMAT": r.standard_normal((NK, dm.n0)) * 1e-4}.get(op)
    ft = None if f is None else wave.tensor(f)
    ya, yb = wave.zeros(nl, dm.n1), per.zeros(nl, dm.n1)
    for it in range(4):
        x = wave.tensor(r.standard_normal((nl, dm.n1)) * (1.0 + it))
        for lev0, nlev in ((0, nl), (1, 3), (4, 1), (0, 8), (2, 9)):
            nlev = min(nlev, nl - lev0)
            fs = None if ft is None else ft[:nlev]
            a = wave.apply(op, x[:nlev], f=fs, lev0=lev0, scale=SCALE, flags=fl)
            b = per.apply(op, x[:nlev], f=fs, lev0=lev0, scale=SCALE, flags=fl)
            assert torch.equal(a, b), (op, it, lev0, nlev, int((a != b).sum()))
        base = wave.tensor(r.standard_normal((nl, dm.n1)))
        ya.copy_(base); yb.copy_(base)
        wave.apply(op, x, f=None if ft is None else ft[:nl], lev0=0, scale=SCALE, flags=fl | 2, alpha=0.25, out=ya)
        per.apply(op, x, f=None if ft is None else ft[:nl], lev0=0, scale=SCALE, flags=fl | 2, alpha=0.25, out=yb)
        assert torch.equal(ya, yb), (op, it, "accumulate", int((ya != yb).sum()))


CASES2 = [("WMAT", 1), ("WMAT", 0), ("WHMAT", 1), ("WHMAT", 0), ("WTQUMAT", 0), ("WTQDUDZ", 0)]


@pytest.mark.parametrize("op,fl", CASES2, ids=["%s_%d" % c for c in CASES2])
def test_two_form_valued_operators_on_the_wave_kernel(engines, op, fl):
    """p = 3: Wmat, Whmat, WtQUmat, WtQdUdz_mat through k_apply_wave2 (one launch, all of the element algebra through DPP) against the
    k_elem_apply form of the same library (MIMSEM_WAVE=0)"""
    import torch
    pn, dm, wave, two, _ = engines
    if pn != 3:
        pytest.skip("k_apply_wave2 exists at p = 3")
    if op == "WMAT":
        wave = wave.wave_all
    r = np.random.default_rng(23)
    x = r.standard_normal((NK, dm.n2 if op in ("WMAT", "WHMAT") else dm.n1))
    f = {"WHMAT": r.uniform(0.5, 1.5, (NK, dm.n2)) * 1e6, "WTQUMAT": r.standard_normal((NK, dm.n1)) * 1e3, "WTQDUDZ": r.standard_normal((NK, dm.n1)) * 1e3}.get(op)
    for lev0, nlev in ((0, NK), (3, 2), (6, 1)):
        xs = x[:nlev]; fs = None if f is None else f[:nlev]
        a = wave.apply(op, wave.tensor(xs), f=None if fs is None else wave.tensor(fs), lev0=lev0, scale=SCALE, flags=fl)
        b = two.apply(op, two.tensor(xs), f=None if fs is None else two.tensor(fs), lev0=lev0, scale=SCALE, flags=fl)
        assert rel_l2(a.cpu().numpy(), b.cpu().numpy()) < 1e-13, (op, lev0, nlev)
        for k in range(nlev):
            assert rel_l2(a[k].cpu().numpy(), b[k].cpu().numpy()) < 1e-12, (op, lev0, k)
    base = r.standard_normal((NK, dm.n2))
    ya, yb = wave.tensor(base), two.tensor(base)
    wave.apply(op, wave.tensor(x), f=None if f is None else wave.tensor(f), lev0=0, scale=SCALE, flags=fl | 2, alpha=0.75, out=ya)
    two.apply(op, two.tensor(x), f=None if f is None else two.tensor(f), lev0=0, scale=SCALE, flags=fl | 2, alpha=0.75, out=yb)
    assert rel_l2(ya.cpu().numpy(), yb.cpu().numpy()) < 1e-13, op
    a1 = wave.apply(op, wave.tensor(x), f=None if f is None else wave.tensor(f), lev0=0, scale=SCALE, flags=fl)
    assert torch.equal(a1, wave.apply(op, wave.tensor(x), f=None if f is None else wave.tensor(f), lev0=0, scale=SCALE, flags=fl))
    single = wave.apply(op, wave.tensor(x[4]), f=None if f is None else wave.tensor(f[4]), lev0=4, scale=SCALE, flags=fl)
    assert torch.equal(single, a1[4])                                  # level batch == single level, bit for bit


def test_wave_form_against_the_oracle(engines, oracle):
    """p = 3: every patch of the small sphere against oracle.Patch (eul/Assembly.cpp coefficient loops restated)"""
    pn, dm, wave, two, (cs, topos, geoms) = engines
    if pn != 3:
        pytest.skip("oracle comparison at p = 3")
    from mimsem_amd.mesh import sphere_coords
    from tests.test_gpu_fullsize_oracle import PatchView, _oracle_patch
    coords = sphere_coords(pn, cs.ne)
    r = np.random.default_rng(5)
    x = r.standard_normal((NK, dm.n1)); h = r.uniform(0.5, 1.5, (NK, dm.n2)) * 1e6; q = r.standard_normal((NK, dm.n0)) * 1e-4
    got = {("UMAT", 1): wave.apply("UMAT", wave.tensor(x), lev0=0, scale=SCALE, flags=1).cpu().numpy(),
           ("UHMAT", 1): wave.apply("UHMAT", wave.tensor(x), f=wave.tensor(h), lev0=0, scale=SCALE, flags=1).cpu().numpy(),
           ("ROTMAT", 0): wave.apply("ROTMAT", wave.tensor(x), f=wave.tensor(q), lev0=0, scale=SCALE, flags=0).cpu().numpy()}
    for pi in range(6):
        v = PatchView(dm, topos, pi); v.pid = pi
        P = _oracle_patch(oracle, cs, coords, geoms[pi], pi, NK)
        for (op, fl), y in got.items():
            for lev in (0, NK - 1):
                f1 = {"UMAT": None, "UHMAT": h[lev][v.s2], "ROTMAT": q[lev][v.s0]}[op]
                want = P.apply(op, x[lev][v.s1], lev=lev, scale=SCALE, flag=fl, f1=f1)
                assert rel_l2(y[lev][v.s1][v.int1], want[v.int1]) < 1e-10, (op, lev, v.pid)


@pytest.mark.parametrize("pn,ne", [(3, 8), (2, 4)], ids=["p3", "p2"])
def test_paired_local_layout_gets_the_wave_plan(pn, ne):
    """One patch in a rank-LOCAL layout, as a reference rank holds it.  The reference's own local numbering (eul/Topo.cpp:215-240) has
    no slot-pair plan (two-pass kernels); the co-located one, Topo(paired=True), has -- same operator, vectors permuted."""
    from mimsem_amd.device import DeviceMesh, Engine
    from mimsem_amd.geom import Geom
    from mimsem_amd.mesh import CubedSphere, sphere_coords
    from mimsem_amd.topo import Topo
    cs = CubedSphere(pn, ne, 6); coords = sphere_coords(pn, ne)
    engs = []
    for paired in (False, True):
        t = Topo(cs, 2, NK, paired=paired)
        g = Geom(t, cs, coords, NK); g.set_levels(z_levels(NK, g.n0))
        engs.append((t, Engine(DeviceMesh([t], [g], nk=NK, numbering="local"))))
    (tr, ref), (tp, par) = engs
    st = (C.c_int * 5)()
    assert pn < 3 or ref.L.mimsem_op_wave_stats(ref.ctx, NK, st) == 0        # (p = 2: few enough slots per group for a plan either way)
    assert par.L.mimsem_op_wave_stats(par.ctx, NK, st) == 1 and st[0] > 0, list(st)
    # slot of every reference-layout entry in the paired layout
    slot = np.empty(tr.n1, dtype=np.int64); slot[0::2] = np.arange(0, tr.n1, 2); slot[1::2] = tp._yslot.ravel()
    assert np.array_equal(tp.loc1[slot], tr.loc1)
    r = np.random.default_rng(41)
    x = r.standard_normal((NK, tr.n1)); xp = np.empty_like(x); xp[:, slot] = x
    h = r.uniform(0.5, 1.5, (NK, tr.n2)) * 1e6; q = r.standard_normal((NK, tr.n0)) * 1e-4
    for op, f, fl in (("UMAT", None, 1), ("UMAT", None, 0), ("UHMAT", h, 1), ("ROTMAT", q, 0), ("UTMAT_H", h, 0)):
        a = ref.apply(op, ref.tensor(x), f=None if f is None else ref.tensor(f), lev0=0, scale=SCALE, flags=fl).cpu().numpy()
        b = par.apply(op, par.tensor(xp), f=None if f is None else par.tensor(f), lev0=0, scale=SCALE, flags=fl).cpu().numpy()
        assert rel_l2(b[:, slot], a) < 1e-13, op
    if pn == 3:                                                        # 1-form in, 2-form out on the DPP kernel
        u = r.standard_normal((NK, tr.n1)) * 1e3; up = np.empty_like(u); up[:, slot] = u
        a = ref.apply("WTQUMAT", ref.tensor(x), f=ref.tensor(u), lev0=0, scale=SCALE, flags=0).cpu().numpy()
        b = par.apply("WTQUMAT", par.tensor(xp), f=par.tensor(up), lev0=0, scale=SCALE, flags=0).cpu().numpy()
        assert rel_l2(b, a) < 1e-13
