"""Regression test for the GPU fault recorded in DESIGN.md 9.0 (round 1): a hipGraph captured around engine calls has the
addresses of the context's workspaces (d_ye: element-local results, d_col: column workspace, d_kry: Krylov partial sums) baked
into its kernel arguments; a later call that needs a LARGER workspace used to free the old buffer, and the next replay of the
graph faulted (MEMORY_APERTURE_VIOLATION).  Outgrown workspaces are now retired until mimsem_ctx_destroy
(mimsem_amd/csrc/api.hip ensure_ye / ensure_col / ensure_kry), and a workspace refuses to grow while its stream is capturing.

Run once per suite: capture -> grow every workspace through larger requests -> scribble -> replay -> compare with eager."""
import numpy as np
import pytest

from mimsem_amd.workloads import SCALE, z_levels

pytestmark = pytest.mark.gpu


def _engine(nk=4, pn=3, ne=2):
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
    return dm, Engine(dm)


def test_graph_replay_survives_workspace_growth():
    import torch
    from mimsem_amd.krylov import GraphedRichardson
    dm, eng = _engine()
    nk, n2, nEl = 4, eng.n2e, dm.nEl
    rng = np.random.default_rng(90)
    t = eng.tensor
    ws = lambda: eng.L.mimsem_ctx_workspace_bytes(eng.ctx)

    # (1) the graph that faulted in round 1: preconditioned Richardson sweeps on the 1-form mass matrix (element pass -> d_ye ->
    #     block pass -> gather with update) captured by GraphedRichardson, exactly as SWEqn.solve_M1 sets it up
    n1e = eng.n1e
    em = eng.element_matrices("UMAT").view(nEl, 2, 2, n1e, n1e)
    B = em.permute(0, 1, 3, 2, 4).reshape(nEl, 2 * n1e, 2 * n1e)
    idx = torch.cat([torch.as_tensor(dm.inds1x, device=eng.device), torch.as_tensor(dm.inds1y, device=eng.device)], dim=1).long()
    mult = torch.zeros(dm.n1, dtype=torch.float64, device=eng.device)
    mult.index_add_(0, idx.reshape(-1), torch.ones(idx.numel(), dtype=torch.float64, device=eng.device))
    d = 1.0 / mult[idx]
    pre = (d[:, :, None] * torch.linalg.inv(B) * d[:, None, :]).contiguous()
    cm = pre.transpose(1, 2).contiguous()
    precond = lambda r: eng.blocks_apply(1, pre, r, transpose=True)
    xs = t(rng.standard_normal((nk, dm.n1)))
    b = eng.apply("UMAT", xs)
    gr = GraphedRichardson(eng, (nk, dm.n1), chunk=6, sweep=lambda x, rhs, upd: eng.block_richardson_sweep("UMAT", cm, x, rhs, upd=upd))
    sol = gr.solve(b, precond, rtol=1e-13)
    assert sol is not None, "Richardson sweeps on P^-1 M1 must contract"
    want_rich = sol[0].clone()
    assert float(torch.linalg.vector_norm(want_rich - xs) / torch.linalg.vector_norm(xs)) < 1e-10
    # (2) plain captures: an operator apply (d_ye), a column operator (d_col), a multi-dot (d_kry)
    x1 = t(rng.standard_normal((nk, dm.n1))); h = t(rng.uniform(1, 2, (nk, dm.n2)) * 1e6); y1 = eng.zeros(nk, dm.n1)
    rho = t(np.abs(rng.standard_normal((nEl, nk * n2))) + 1.0); xc = t(rng.standard_normal((nEl, nk * n2)))
    V = t(rng.standard_normal((6, 4096))); w = t(rng.standard_normal(4096)); hdot = eng.zeros(6)
    out = {}

    def body():
        eng.apply("UHMAT", x1, f=h, lev0=0, scale=SCALE, flags=1, out=y1)
        out["col"] = eng.colop_apply("CONST_RHO", xc, f1=rho, nout_slots=nk)
        eng.mdot(V, w, out=hdot)
        return out["col"]
    graph, gcol = eng.capture(body)
    graph.replay(); torch.cuda.synchronize()
    want_y, want_col, want_h = y1.clone(), gcol.clone(), hdot.clone()
    bytes0 = ws()

    # (3) grow every workspace through LARGER requests than anything seen so far
    nrow = 6 * nk                                                    # packed [u | h] rows: d_ye grows (2*per*nlev + nrow*nlev doubles)
    f0 = t(rng.standard_normal(dm.n0) * 1e-4)
    xu = t(rng.standard_normal((nrow, dm.n1 + dm.n2)))
    eng.sw_operator(0.5 * 360.0, 9.80616, 1.0e4, f0, xu)
    area = float(dm.det.mean()) * 4.0 / n2; dz = float(dm.thick.mean())
    lev = lambda nl, lo, hi: t(rng.uniform(lo, hi, (nEl, nl * n2)) * area * dz)
    F = [t(rng.standard_normal((nEl, n * n2)) * 1e8) for n in (nk - 1, nk, nk, nk)]
    eng.solve_schur_3(75.0, lev(nk + 1, 280, 320) / dz, lev(nk - 1, -1, 1) / dz, lev(nk, 0.5, 1.2), lev(nk, 250, 400), lev(nk, 700, 1000), *F)   # d_col grows
    Vb = t(rng.standard_normal((40, 1 << 20))); wb = t(rng.standard_normal(1 << 20))
    eng.mdot(Vb, wb)                                                 # d_kry grows
    torch.cuda.synchronize()
    assert ws() > bytes0, "the larger requests did not grow any workspace: the test no longer tests anything"

    # (4) scribble over the NEW workspaces with different inputs, then replay the OLD graphs
    eng.apply("UMAT", t(rng.standard_normal((nk, dm.n1))), lev0=0, scale=SCALE, flags=1)
    eng.colop_apply("CONST_RHO", t(rng.standard_normal((nEl, nk * n2))), f1=rho, nout_slots=nk)
    y1.zero_(); gcol.zero_(); hdot.zero_()
    graph.replay(); torch.cuda.synchronize()
    assert torch.equal(y1, want_y) and torch.equal(gcol, want_col) and torch.equal(hdot, want_h)
    # eager on the grown workspaces gives the same bits as the replay on the retired ones
    assert torch.equal(eng.apply("UHMAT", x1, f=h, lev0=0, scale=SCALE, flags=1), want_y)
    assert torch.equal(eng.colop_apply("CONST_RHO", xc, f1=rho, nout_slots=nk), want_col)
    sol = gr.solve(b, precond, rtol=1e-13)              # replays the graphs captured BEFORE the growth
    assert sol is not None and torch.equal(sol[0], want_rich)


def test_workspace_growth_inside_a_capture_is_refused():
    """a request that must grow a workspace while the context's stream is capturing returns MIMSEM_ERR_STATE (hipMalloc is illegal
    there) instead of corrupting the capture; Engine.capture's warm-up pass exists to avoid exactly this"""
    import torch
    from mimsem_amd._lib import MimsemError
    dm, eng = _engine()
    nk, n2, nEl = 4, eng.n2e, dm.nEl
    rng = np.random.default_rng(91)
    rho = eng.tensor(np.abs(rng.standard_normal((nEl, nk * n2))) + 1.0); xc = eng.tensor(rng.standard_normal((nEl, nk * n2)))
    area = float(dm.det.mean()) * 4.0 / n2; dz = float(dm.thick.mean())
    lev = lambda nl, lo, hi: eng.tensor(rng.uniform(lo, hi, (nEl, nl * n2)) * area * dz)
    cols = (lev(nk, 280, 320), lev(nk, 0.5, 1.2), lev(nk, 5, 6), lev(nk, 700, 1000))
    F = [eng.tensor(rng.standard_normal((nEl, n * n2)) * 1e8) for n in (nk - 1, nk, nk, nk)]
    s = torch.cuda.Stream(device=eng.device)
    g = torch.cuda.CUDAGraph()
    err = None
    torch.cuda.synchronize()
    from mimsem_amd.device import no_gc
    with no_gc(), torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s), eng.on_current_stream():
            z = xc * 2.0                                                         # (one node, so the capture is not empty)
            try:
                eng.solve_schur_eta(75.0, *cols, *F)                   # first call of this context that needs the column workspace
            except MimsemError as ex:
                err = str(ex)
    torch.cuda.synchronize()
    assert err is not None and "state" in err.lower(), err
    # the context is still usable afterwards
    d = eng.solve_schur_eta(75.0, *cols, *F)
    assert all(bool(torch.isfinite(v).all()) for v in d)


def test_c_abi_graph_replays_hold_no_growing_state():
    """what does the library do PER REPLAY of a graph recorded through mimsem_graph_*?  By reading (csrc/api.hip): one
    hipGraphLaunch on the recording's stream, nothing created, nothing retired; the host's read of the check norms goes through one pinned
    4 KB buffer allocated once.  This test holds it to that: 1 000 replays of a recorded operator sequence with the C++ hosts' read-back
    pattern (mimsem_memcpy_d2h of a few scalars after every replay) -- workspace bytes, open handles and resident memory stay where they were,
    and the result of the last replay equals the first."""
    import ctypes as C
    import os
    import psutil
    import torch
    from mimsem_amd._lib import check
    dm, eng = _engine()
    L = eng.L
    rng = np.random.default_rng(3)
    x = eng.tensor(rng.standard_normal((4, dm.n1))); y = eng.zeros(4, dm.n1); z = eng.zeros(4, dm.n2)
    nrm = eng.zeros(4)

    def seq():
        eng.apply("UMAT", x, lev0=0, scale=SCALE, flags=1, out=y)
        eng.apply("WTQUMAT", y, f=x, lev0=0, scale=SCALE, out=z)
        eng.rowdot(y, y, out=nrm)
    seq(); eng.sync()                                       # (workspaces reach their size outside the recording)
    check(L.mimsem_ctx_use_own_stream(eng.ctx), "use_own_stream")
    seq(); eng.sync()
    g = C.c_void_p()
    check(L.mimsem_graph_begin(eng.ctx), "graph_begin")
    seq()
    check(L.mimsem_graph_end(eng.ctx, C.byref(g)), "graph_end")
    host = (C.c_double * 4)()
    first = None
    proc = psutil.Process()
    ws0 = L.mimsem_ctx_workspace_bytes(eng.ctx)
    for k in range(1100):
        check(L.mimsem_graph_launch(g), "graph_launch")
        check(L.mimsem_memcpy_d2h(eng.ctx, host, nrm.data_ptr(), 32), "d2h")
        if k == 0:
            first = list(host)
        if k == 99:                                         # (the first replays settle the runtime's own pools)
            rss0, fd0 = proc.memory_info().rss, len(os.listdir("/proc/self/fd"))
    rss1, fd1 = proc.memory_info().rss, len(os.listdir("/proc/self/fd"))
    assert list(host) == first and all(v > 0 for v in first)
    assert L.mimsem_ctx_workspace_bytes(eng.ctx) == ws0
    assert fd1 == fd0, (fd0, fd1)
    assert rss1 - rss0 < 8 << 20, (rss0, rss1)
    L.mimsem_graph_destroy(g)
    eng.use_stream(torch.cuda.current_stream(eng.device))


def test_a_recording_outlives_its_context_safely():
    """advisor, round 5: a mimsem_graph kept the stream it was recorded on and replayed there for ever -- after mimsem_ctx_destroy that was a
    destroyed stream.  A recording now belongs to its context: destroyed context => launch returns MIMSEM_ERR_STATE; and a later
    mimsem_ctx_use_own_stream moves the replays with the context (same result)."""
    import ctypes as C
    from mimsem_amd._lib import check
    dm, eng = _engine()
    L = eng.L
    rng = np.random.default_rng(4)
    x = eng.tensor(rng.standard_normal((4, dm.n1))); y = eng.zeros(4, dm.n1)
    seq = lambda: eng.apply("UMAT", x, lev0=0, scale=SCALE, flags=1, out=y)
    check(L.mimsem_ctx_use_own_stream(eng.ctx), "use_own_stream")
    seq(); eng.sync()
    want = y.clone()
    g = C.c_void_p()
    check(L.mimsem_graph_begin(eng.ctx), "graph_begin"); seq(); check(L.mimsem_graph_end(eng.ctx, C.byref(g)), "graph_end")
    y.zero_(); eng.sync()
    check(L.mimsem_graph_launch(g), "graph_launch"); eng.sync()
    import torch
    assert torch.equal(y, want)
    L.mimsem_ctx_destroy(eng.ctx); eng.ctx = C.c_void_p()
    rc = L.mimsem_graph_launch(g)
    assert rc != 0, "a recording whose context is gone must refuse to launch"
    L.mimsem_graph_destroy(g)
