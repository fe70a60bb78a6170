"""Error behaviour of the C ABI: every entry point returns a negative MIMSEM_ERR_* code (never crashes, never launches) on
null / out-of-range arguments, unsupported orders and wrong entry points; mimsem_strerror names them."""
import ctypes as C

import numpy as np
import pytest

from tests.helpers import SCALE, make_patch

pytestmark = pytest.mark.gpu
ERR_ARG, ERR_UNSUPPORTED = -1, -2


@pytest.fixture(scope="module")
def small(oracle):
    from mimsem_amd.device import DeviceMesh, Engine
    cs, topo, geom, P, rng = make_patch(oracle, 3, 2, 6, 0, nk=4, seed=5)
    dm = DeviceMesh([topo], [geom], nk=4, numbering="local")
    return Engine(dm), dm, P


def test_argument_errors(small):
    from mimsem_amd._lib import COLOPS, OPS, MimsemError
    eng, dm, P = small
    L, ctx = eng.L, eng.ctx
    x = eng.tensor(np.ones((4, dm.n1))); y = eng.zeros(4, dm.n1); h = eng.tensor(np.ones((4, dm.n2)))
    px, py, ph = x.data_ptr(), y.data_ptr(), h.data_ptr()
    ap = lambda op, lev0, nlev, f, xx, yy: L.mimsem_op_apply(ctx, op, lev0, nlev, SCALE, 0, f, dm.n2, xx, dm.n1, yy, dm.n1, 1.0)
    assert ap(OPS["UMAT"], 0, 4, None, px, py) == 0
    assert ap(999, 0, 1, None, px, py) == ERR_ARG                       # unknown operator
    assert ap(OPS["UMAT"], 0, 5, None, px, py) == ERR_ARG               # levels beyond nk
    assert ap(OPS["UMAT"], -1, 1, None, px, py) == ERR_ARG
    assert ap(OPS["UMAT"], 0, 1, None, None, py) == ERR_ARG             # null input
    assert ap(OPS["UHMAT"], 0, 1, None, px, py) == ERR_ARG              # coefficient field missing
    assert ap(OPS["UTMAT"], 0, 4, None, px, py) == ERR_ARG              # needs thick[lev+1]
    assert ap(OPS["UMAT_UP"], 0, 1, px, px, py) == ERR_ARG              # wrong entry point: needs mimsem_op_apply_up
    assert L.mimsem_op_apply_up(ctx, OPS["UMAT"], 0, 1, SCALE, 1.0, 0, px, dm.n1, px, dm.n1, px, dm.n1, py, dm.n1, 1.0) == ERR_ARG
    assert L.mimsem_op_apply_up(ctx, OPS["UMAT_UP"], 0, 1, SCALE, 1.0, 4, px, dm.n1, None, dm.n1, px, dm.n1, py, dm.n1, 1.0) == ERR_ARG
    assert L.mimsem_op_apply(None, OPS["UMAT"], 0, 1, SCALE, 0, None, 0, px, dm.n1, py, dm.n1, 1.0) == ERR_ARG      # null context
    assert L.mimsem_op_elmat_size(ctx, 999) == ERR_ARG
    assert L.mimsem_incidence_apply(ctx, 7, 1, px, dm.n1, py, dm.n1) == ERR_ARG
    assert L.mimsem_colop_nblocks(ctx, 999) == ERR_ARG
    assert L.mimsem_colop_blocks(ctx, COLOPS["CONST_RHO"], 0, None, None, None) == ERR_ARG
    assert L.mimsem_colop_blocks_ex(ctx, COLOPS["LINCON2_UP"], 0, 1.0, None, None, None, 0, py) == ERR_ARG         # velocity missing
    assert L.mimsem_column_diag_theta(ctx, 5, ph, ph, ph) == ERR_ARG
    assert L.mimsem_column_incidence(ctx, 3, ph, ph) == ERR_ARG
    assert L.mimsem_elem_blocks_apply(ctx, 4, 1, 0, px, 0, None, 0, px, dm.n1, py, dm.n1, 1.0) == ERR_ARG
    assert L.mimsem_halo_segments(ctx, px, 65, px, 0, 1, 1, 0, px, py, dm.n1) == ERR_ARG                            # too many segments
    # the entry points added with the shallow-water step
    p0 = eng.zeros(1, dm.n0).data_ptr(); pk = eng.zeros(1, dm.n1 + dm.n2).data_ptr(); pk2 = eng.zeros(1, dm.n1 + dm.n2).data_ptr()
    assert L.mimsem_interp_quad(ctx, 3, 0, 1, px, dm.n1, py, dm.n1) == ERR_ARG                                       # no such form
    assert L.mimsem_interp_quad(ctx, 1, 2, 1, px, dm.n1, py, dm.n1) == ERR_ARG                                       # unknown flag
    assert L.mimsem_interp_quad(ctx, 1, 1, 1, None, dm.n1, py, dm.n1) == ERR_ARG
    assert L.mimsem_sw_operator_apply(ctx, 1, 1.0, 9.8, 1e4, p0, 0, pk, 0, pk, 0) == ERR_ARG                         # in place
    assert L.mimsem_sw_operator_apply(ctx, 1, 1.0, 9.8, 1e4, None, 0, pk, 0, pk2, 0) == ERR_ARG
    assert L.mimsem_sw_operator_apply(ctx, 2, 1.0, 9.8, 1e4, p0, 0, pk, 5, pk2, 5) == ERR_ARG                        # rows overlap
    assert L.mimsem_sw_blocks_apply(ctx, 1, None, pk, 0, pk2, 0) == ERR_ARG
    assert L.mimsem_op_richardson_sweep(ctx, OPS["WMAT"], 0, 1, 1.0, 0.0, 0, None, 0, None, 0, ph, 0, ph, 0, ph, 0, None, 0) == ERR_ARG     # 2-form result: no gather pass
    assert L.mimsem_op_richardson_sweep(ctx, OPS["WTQUMAT"], 0, 1, 1.0, 0.0, 0, px, 0, None, 0, px, 0, px, 0, py, 0, None, 0) == ERR_ARG   # not square
    assert L.mimsem_op_richardson_sweep(ctx, OPS["PHMAT_UP"], 0, 1, 1.0, 1.0, 0, ph, 0, None, 0, p0, 0, p0, 0, p0, 0, None, 0) == ERR_ARG  # velocity missing
    assert L.mimsem_block_richardson_sweep(ctx, OPS["PMAT"], 0, 1, 1.0, 0, None, 0, px, p0, 0, p0, 0, None, 0) == ERR_ARG                  # 1-forms only
    assert L.mimsem_block_richardson_sweep(ctx, OPS["UMAT"], 0, 1, 1.0, 0, None, 0, None, px, 0, py, 0, None, 0) == ERR_ARG
    assert L.mimsem_krylov_normalize(ctx, 0, px, py, 0, None, None, py, 0) == ERR_ARG
    assert L.mimsem_krylov_orthogonalize(ctx, 2, dm.n1, px, 5, -1.0, py, py) == ERR_ARG                              # ldv < n
    for code in (-1, -2, -3, -4, -5):
        assert len(L.mimsem_strerror(code)) > 5
    with pytest.raises(AssertionError):                                  # vector lengths are checked by the host layer
        eng.incidence("E10", x)
    with pytest.raises(AssertionError):
        eng.apply("UHMAT", x, f=h[:, :5].contiguous())
    with pytest.raises(MimsemError):
        eng.apply("UTMAT", x, lev0=0, scale=SCALE)                        # needs thick[lev+1]: the ABI's error code surfaces as an exception
    # nothing above poisoned the context
    assert float(eng.apply("UMAT", x, lev0=0, scale=SCALE).abs().sum()) > 0


def test_unsupported_configurations(small):
    from mimsem_amd._lib import MeshDesc
    eng, dm, P = small
    d = dm.desc()
    ctx = C.c_void_p()
    d.quadOrd = d.elOrd + 1                                                # quadrature order != element order
    assert eng.L.mimsem_ctx_create(C.byref(d), 0, C.byref(ctx)) == ERR_UNSUPPORTED
    d.quadOrd = d.elOrd = 8                                                # no GLL table beyond 7 (eul/Basis.cpp:31-89)
    assert eng.L.mimsem_ctx_create(C.byref(d), 0, C.byref(ctx)) == ERR_UNSUPPORTED
    d = dm.desc(); d.nEl = -1
    assert eng.L.mimsem_ctx_create(C.byref(d), 0, C.byref(ctx)) == ERR_ARG
    assert eng.L.mimsem_ctx_create(None, 0, C.byref(ctx)) == ERR_ARG
    # the p = 7 limit of the test-upwinded operators and of the pentadiagonal Schur solve is reported, not crashed on
    assert not ctx.value


def test_empty_inputs_are_no_ops(small):
    """zero levels / zero-length halo lists: success, nothing written"""
    import torch
    eng, dm, P = small
    x0 = torch.zeros(0, dm.n1, dtype=torch.float64, device=eng.device)
    y = eng.apply("UMAT", x0, lev0=0, scale=SCALE)
    assert y.shape == (0, dm.n1)
    y1 = eng.tensor(np.ones((2, dm.n1))); keep = y1.clone()
    idx = torch.zeros(0, dtype=torch.int32, device=eng.device)
    buf = eng.halo_pack(idx, y1)
    assert buf.shape == (2, 0)
    eng.halo_unpack(idx, buf, y1, add=True)
    assert torch.equal(y1, keep)
    assert eng.L.mimsem_incidence_apply(eng.ctx, 1, 0, y1.data_ptr(), dm.n1, y1.data_ptr(), dm.n1) == 0
    assert eng.L.mimsem_krylov_mdot(eng.ctx, 0, dm.n1, y1.data_ptr(), dm.n1, y1.data_ptr(), y1.data_ptr()) == 0


def test_column_wrappers_reject_wrong_shapes(small):
    """the column entry points take raw pointers: the host mirror checks every array length and raises MimsemError (also under
    python -O: no assert) instead of letting a short tensor become an out-of-bounds device access"""
    import torch
    from mimsem_amd._lib import MimsemError
    eng, dm, P = small
    nk, n2, nEl = 4, eng.n2e, dm.nEl
    col = lambda sl: eng.zeros(nEl, sl * n2) + 1.0
    bad = [lambda: eng.colop_apply("CONST_RHO", col(nk), f1=col(nk)),                       # nout_slots missing (was a TypeError)
           lambda: eng.colop_apply("CONST_RHO", col(nk), f1=col(nk - 1), nout_slots=nk),      # short coefficient field
           lambda: eng.colop_apply("CONLIN_W", col(nk), f1=col(nk - 1), nout_slots=nk),       # x must have nk-1 slots
           lambda: eng.colop_apply("CONLIN_W", col(nk - 1), f1=col(nk - 1), nout_slots=nk - 1),   # rows are nk
           lambda: eng.colop_apply("LINEAR", col(nk - 1)[:-1], nout_slots=nk - 1),            # a column missing
           lambda: eng.colop_blocks("LINEAR_THETA", f1=col(nk)),                            # theta lives on nk+1 interfaces
           lambda: eng.column_eos(0, col(nk), None),                                        # exner missing
           lambda: eng.column_eos(7, col(nk), col(nk)),
           lambda: eng.diag_theta(0, col(nk), col(nk + 1)),
           lambda: eng.column_incidence("V01", col(nk - 1)),
           lambda: eng.solve_schur_eta(75.0, col(nk), col(nk), col(nk), col(nk), col(nk), col(nk), col(nk), col(nk)),      # F_u has nk-1 slots
           lambda: eng.solve_schur_3(75.0, col(nk), col(nk - 1), col(nk), col(nk), col(nk), col(nk - 1), col(nk), col(nk), col(nk)),   # theta on nk+1
           lambda: eng.l2_vert_to_horiz(col(nk), nk + 1),
           lambda: eng.temp_forcing_hs(eng.zeros(nEl, 3), col(nk), col(nk + 1), col(nk)),
           lambda: eng.colop_apply("CONST", col(nk).cpu(), nout_slots=nk),                    # host tensor
           lambda: eng.colop_apply("CONST", col(nk).float(), nout_slots=nk)]                  # wrong dtype
    for i, fn in enumerate(bad):
        with pytest.raises(MimsemError):
            fn()
    # and the well-formed calls still work
    y = eng.colop_apply("CONLIN_W", col(nk - 1), f1=col(nk - 1), nout_slots=nk)
    yt = eng.colop_apply("CONLIN_W", col(nk), f1=col(nk - 1), nout_slots=nk - 1, transpose=True)
    assert y.shape == (nEl, nk * n2) and yt.shape == (nEl, (nk - 1) * n2) and bool(torch.isfinite(y).all())


def test_round2_entry_points_reject_bad_arguments(small):
    """mimsem_op_apply_part, mimsem_ctx_set_halo_slots, mimsem_op_wave_stats, mimsem_vec_combine, mimsem_interface_average,
    mimsem_halo_*: negative codes, never a launch"""
    from mimsem_amd._lib import OPS
    eng, dm, P = small
    L, ctx = eng.L, eng.ctx
    x = eng.tensor(np.ones((4, dm.n1))); y = eng.zeros(4, dm.n1)
    px, py = x.data_ptr(), y.data_ptr()
    part = lambda op, p: L.mimsem_op_apply_part(ctx, op, 0, 4, SCALE, 0, None, 0, px, dm.n1, py, dm.n1, 1.0, p)
    assert part(OPS["UMAT"], 1) == 0 and part(OPS["UMAT"], 2) == 0       # always a valid sequence (here: whole op, then nothing)
    assert part(OPS["UMAT"], 3) == ERR_ARG and part(OPS["UMAT"], -1) == ERR_ARG
    assert part(OPS["UMAT_UP"], 1) == ERR_ARG                             # upwinded operators have their own entry point
    bad = np.array([dm.n1], np.int32)
    assert L.mimsem_ctx_set_halo_slots(ctx, 1, bad.ctypes.data, 1) == ERR_ARG
    assert L.mimsem_ctx_set_halo_slots(ctx, 0, bad.ctypes.data, 0) == ERR_ARG      # 1-forms only
    assert L.mimsem_ctx_set_halo_slots(None, 1, None, 0) == ERR_ARG
    st = (C.c_int * 5)()
    assert L.mimsem_op_wave_stats(ctx, 0, st) == ERR_ARG and L.mimsem_op_wave_stats(None, 4, st) == ERR_ARG
    assert L.mimsem_vec_combine(ctx, 4, dm.n1, 1.0, px, dm.n1, 3, px, dm.n1, 0.0, None, 0, py, dm.n1) == ERR_ARG     # unknown op
    assert L.mimsem_vec_combine(ctx, 4, dm.n1, 1.0, px, dm.n1, 1, None, 0, 0.0, None, 0, py, dm.n1) == ERR_ARG       # a*b without b
    assert L.mimsem_vec_combine(ctx, 4, dm.n1, 1.0, None, 0, 0, None, 0, 0.0, None, 0, py, dm.n1) == ERR_ARG
    assert L.mimsem_vec_combine(ctx, 0, dm.n1, 1.0, None, 0, 0, None, 0, 0.0, None, 0, None, 0) == 0                   # empty: no-op
    assert L.mimsem_interface_average(ctx, 0, dm.n1, px, dm.n1, py, dm.n1) == ERR_ARG
    out = C.c_void_p()
    rk = np.array([0], np.int32); off_bad = np.array([3, 1], np.int32); idx = np.array([0, 1, 2], np.int32)
    assert L.mimsem_halo_create(ctx, 1, rk.ctypes.data, idx.ctypes.data, off_bad.ctypes.data, idx.ctypes.data, off_bad.ctypes.data, dm.n1, 1, C.byref(out)) == ERR_ARG
    assert L.mimsem_halo_create(ctx, 65, rk.ctypes.data, idx.ctypes.data, off_bad.ctypes.data, idx.ctypes.data, off_bad.ctypes.data, dm.n1, 1, C.byref(out)) == ERR_ARG
    assert L.mimsem_halo_create(ctx, 0, None, None, None, None, None, dm.n1, 0, C.byref(out)) == ERR_ARG               # max_nlev >= 1
    assert L.mimsem_halo_begin(None, 1, 1, py, dm.n1) == ERR_ARG and L.mimsem_halo_end(None) == ERR_ARG
    # combine through the wrapper: shape mismatch is a MimsemError, not a launch
    from mimsem_amd._lib import MimsemError
    with pytest.raises(MimsemError):
        eng.combine(x, 1.0, "mul", eng.zeros(3, dm.n1))
    # and it computes what it says
    import torch
    b = eng.tensor(np.full((4, dm.n1), 2.0)); c = eng.tensor(np.full((4, dm.n1), 3.0))
    assert torch.equal(eng.combine(x, 2.0, "div", b, beta=-1.0, c=c), torch.full_like(x, 2.0 * 0.5 - 3.0))
    a = eng.tensor(np.arange(3 * 5, dtype=float).reshape(3, 5))
    want = torch.stack([0.5 * a[0], 0.5 * (a[0] + a[1]), 0.5 * (a[1] + a[2]), 0.5 * a[2]])
    assert torch.equal(eng.interface_average(a, 4), want)


def test_set_halo_slots_refused_inside_a_capture(small):
    """a set-up call that re-derives device tables must not run on a capturing stream (MIMSEM_ERR_STATE), and leaves the context usable"""
    import torch
    eng, dm, P = small
    sl = np.array([0, 1], np.int32)
    x = eng.tensor(np.ones((4, dm.n1))); y = eng.zeros(4, dm.n1)
    rcs = []

    def fn():
        rcs.append(eng.L.mimsem_ctx_set_halo_slots(eng.ctx, 1, sl.ctypes.data, 2))
        return eng.apply("UMAT", x, lev0=0, scale=SCALE, flags=1, out=y)
    g, out = eng.capture(fn)                       # runs fn once outside the capture (warm-up), once inside
    assert rcs == [0, -4], rcs
    g.replay(); torch.cuda.synchronize()
    assert torch.isfinite(out).all() and torch.equal(out, eng.apply("UMAT", x, lev0=0, scale=SCALE, flags=1))


def test_argument_errors_of_the_round_6_solver_entries(small):
    """mimsem_block_chebyshev_solve, mimsem_krylov_chebyshev_start, mimsem_krylov_axpy_dots, mimsem_sw_dual_chebyshev: nothing launches on bad
    arguments; empty inputs are a clean no-op"""
    from mimsem_amd._lib import OPS
    eng, dm, P = small
    L, ctx = eng.L, eng.ctx
    nd = 2 * eng.n1e
    b = eng.tensor(np.ones((4, dm.n1))); x = eng.zeros(4, dm.n1); B = eng.zeros(dm.nEl, nd, nd)
    coef = (C.c_double * 4)(1.0, 0.0, 1.0, 0.1)
    pb, px, pB = b.data_ptr(), x.data_ptr(), B.data_ptr()
    solve = lambda op, lev0, nlev, flags, blocks, rhs, nsteps, cf, out: L.mimsem_block_chebyshev_solve(
        ctx, op, lev0, nlev, SCALE, flags, None, 0, blocks, None, 0, rhs, dm.n1, nsteps, cf, out, dm.n1, None, 0, None, 0)
    assert solve(OPS["UMAT"], 0, 4, 1, pB, pb, 2, coef, px) == 0
    assert solve(OPS["UHMAT"], 0, 4, 1, pB, pb, 2, coef, px) == ERR_UNSUPPORTED        # the mass operator only
    assert solve(OPS["UMAT"], 0, 4, 2, pB, pb, 2, coef, px) == ERR_ARG                 # no accumulate form
    assert solve(OPS["UMAT"], 0, 5, 1, pB, pb, 2, coef, px) == ERR_ARG                 # levels beyond nk
    assert solve(OPS["UMAT"], 0, 4, 1, None, pb, 2, coef, px) == ERR_ARG
    assert solve(OPS["UMAT"], 0, 4, 1, pB, pb, 0, coef, px) == ERR_ARG                 # no steps
    assert solve(OPS["UMAT"], 0, 4, 1, pB, pb, 2, None, px) == ERR_ARG
    assert solve(OPS["UMAT"], 0, 4, 1, pB, pb, 2, coef, pb) == ERR_ARG                 # in place on the right-hand side
    assert solve(OPS["UMAT"], 0, 0, 1, None, None, 2, coef, None) == 0                 # empty batch
    r = eng.zeros(1, 100); d = eng.zeros(1, 100); v = eng.tensor(np.ones((1, 100))); out = eng.zeros(2)
    st = lambda th, c, rr, dd, xx: L.mimsem_krylov_chebyshev_start(ctx, 1, 100, 1.0, th, c, 100, rr, 100, dd, 100, xx, 100)
    assert st(0.8, v.data_ptr(), r.data_ptr(), d.data_ptr(), x.data_ptr()) == 0
    assert st(0.0, v.data_ptr(), r.data_ptr(), d.data_ptr(), x.data_ptr()) == ERR_ARG   # theta = 0
    assert st(0.8, v.data_ptr(), r.data_ptr(), r.data_ptr(), x.data_ptr()) == ERR_ARG   # outputs alias
    assert st(0.8, None, r.data_ptr(), d.data_ptr(), x.data_ptr()) == ERR_ARG
    assert L.mimsem_krylov_axpy_dots(ctx, 100, v.data_ptr(), v.data_ptr(), out.data_ptr()) == ERR_ARG      # in place
    assert L.mimsem_krylov_axpy_dots(ctx, 100, v.data_ptr(), r.data_ptr(), None) == ERR_ARG
    assert L.mimsem_krylov_axpy_dots(ctx, -1, v.data_ptr(), r.data_ptr(), out.data_ptr()) == ERR_ARG
    out.fill_(7.0)
    assert L.mimsem_krylov_axpy_dots(ctx, 0, v.data_ptr(), r.data_ptr(), out.data_ptr()) == 0 and not bool(out.any())   # empty: both norms zero
    px_ = lambda yy, bb, dd, pp, xx: L.mimsem_krylov_chebyshev_px(ctx, 1, 100, 0.5, 0.1, yy, 100, bb, 100, dd, 100, pp, 100, xx, 100, None, 0)
    assert px_(v.data_ptr(), None, None, r.data_ptr(), d.data_ptr()) == 0
    assert px_(v.data_ptr(), None, v.data_ptr(), r.data_ptr(), d.data_ptr()) == ERR_ARG     # a diagonal without a right-hand side
    assert px_(v.data_ptr(), None, None, r.data_ptr(), r.data_ptr()) == ERR_ARG             # p and x alias
    assert px_(None, None, None, r.data_ptr(), d.data_ptr()) == ERR_ARG
    p1 = eng.zeros(1, dm.n1); x1 = eng.zeros(1, dm.n1); p0 = eng.zeros(1, dm.n0); x0 = eng.zeros(1, dm.n0); b0 = eng.zeros(1, dm.n0); h = eng.zeros(1, dm.n2)
    dual = lambda nA, upd1, pb1: L.mimsem_sw_dual_chebyshev(ctx, nA, C.addressof(coef), pB, pb, p1.data_ptr(), x1.data_ptr(), upd1, pb1,
                                                           2, C.addressof(coef), 1.0, h.data_ptr(), pb, b0.data_ptr(), b0.data_ptr(), p0.data_ptr(), x0.data_ptr(), None, None)
    assert dual(0, None, None) == ERR_ARG                                               # no steps
    assert dual(1, r.data_ptr(), d.data_ptr()) == ERR_ARG                               # a one-step chain has one preconditioned residual
