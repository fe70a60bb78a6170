"""The halo exchange behind the C ABI (mimsem_halo_create / _begin / _end): loop-back transport in one process -- slot lists, both
modes, the ordered ADD with repeated targets, begin/end split with work in between, error behaviour.  The several-rank form
(host-callback transport over gloo) runs in tests/test_gpu_multiproc.py."""
import types

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng(oracle):
    from mimsem_amd.device import DeviceMesh, Engine
    from tests.helpers import make_patch
    cs, topo, geom, P, rng = make_patch(oracle, 3, 4, 6, 2, nk=4, seed=5)
    return Engine(DeviceMesh([topo], [geom], nk=4, numbering="local")), P


def _plan(n, ghost, mirror):
    p = types.SimpleNamespace(gids=np.arange(n), ghost_slots={0: np.asarray(ghost, np.int32)}, mirror_slots={0: np.asarray(mirror, np.int32)})
    p.neighbours = lambda: [0]
    return p


def test_loopback_reverse_add_and_forward_insert(eng):
    import torch
    from mimsem_amd.partition import CHalo
    e, P = eng
    n = P.n1
    rng = np.random.default_rng(3)
    ghost = rng.choice(n, 40, replace=False).astype(np.int32)
    mirror = rng.choice(np.setdiff1d(np.arange(n), ghost), 40, replace=False).astype(np.int32)
    h = CHalo(_plan(n, ghost, mirror), e, max_nlev=4, transport="loopback")
    for nlev in (4, 1, 3):
        v0 = rng.standard_normal((nlev, n))
        v = e.tensor(v0)
        h.reverse_add(v)                                           # "owner" slots accumulate the "ghost" partial sums
        want = v0.copy(); want[:, mirror] += v0[:, ghost]
        assert np.array_equal(v.cpu().numpy(), want)
        h.forward_insert(v)                                        # ghosts receive the owners' values
        want[:, ghost] = want[:, mirror]
        assert np.array_equal(v.cpu().numpy(), want)
    # begin / end split: work enqueued in between runs while the exchange is in flight and does not disturb it
    v0 = rng.standard_normal((4, n)); v = e.tensor(v0)
    other = e.tensor(rng.standard_normal((4, n)))
    tok = h.begin("reverse", v, True)
    y = e.apply("UMAT", other, lev0=0, scale=1.0e8, flags=1)
    h.end(tok)
    want = v0.copy(); want[:, mirror] += v0[:, ghost]
    assert np.array_equal(v.cpu().numpy(), want)
    assert torch.isfinite(y).all()
    h.close()


def test_one_sided_transport_with_the_rank_as_its_own_neighbour(eng):
    """the one-sided transport (round 6: the pack kernel writes into the neighbour's receive buffer and publishes a sequence flag, the unpack
    waits for it) in ONE process: a plan whose only neighbour is the rank itself needs no hipIpc handle (its own buffer), so export / connect /
    begin / end run here -- the same results as the loop-back transport over many exchanges (both halves of the double buffer), work in
    between begin and end, a recorded exchange replayed, no time-outs.  Between processes: tests/test_gpu_multiproc.py."""
    import ctypes as C
    import torch
    from mimsem_amd._lib import HALO_PEER_BLOB, check
    from mimsem_amd.partition import CHalo
    e, P = eng
    n = P.n1
    rng = np.random.default_rng(4)
    ghost = rng.choice(n, 40, replace=False).astype(np.int32)
    mirror = rng.choice(np.setdiff1d(np.arange(n), ghost), 40, replace=False).astype(np.int32)
    ref = CHalo(_plan(n, ghost, mirror), e, max_nlev=4, transport="loopback")
    h = CHalo(_plan(n, ghost, mirror), e, max_nlev=4, transport="loopback")
    for name, hd in h.handles.items():                                 # connect every plan of h to itself through the one-sided set-up calls
        blob = C.create_string_buffer(HALO_PEER_BLOB)
        check(e.L.mimsem_halo_peer_export(hd, 0, blob), "halo_peer_export")
        check(e.L.mimsem_halo_set_peer(hd, 0, blob.raw), "halo_set_peer")
    for rep, nlev in enumerate((4, 1, 3, 2, 4)):
        v0 = rng.standard_normal((nlev, n))
        a, b = e.tensor(v0), e.tensor(v0)
        ref.reverse_add(a); h.reverse_add(b)
        assert torch.equal(a, b), rep
        ref.forward_insert(a); h.forward_insert(b)
        assert torch.equal(a, b), rep
    v0 = rng.standard_normal((4, n)); v = e.tensor(v0)
    tok = h.begin("reverse", v, True)
    y = e.apply("UMAT", e.tensor(rng.standard_normal((4, n))), lev0=0, scale=1.0e8, flags=1)
    h.end(tok)
    want = v0.copy(); want[:, mirror] += v0[:, ghost]
    assert np.array_equal(v.cpu().numpy(), want) and torch.isfinite(y).all()
    # recorded: the exchange counter and the buffer parity live in device memory, so every replay is a new, correct exchange
    src = e.tensor(rng.standard_normal((4, n))); buf = torch.zeros_like(src)
    want = src.cpu().numpy().copy(); want[:, mirror] += want[:, ghost]

    def recorded():
        buf.copy_(src)
        h.reverse_add(buf)
    g, _ = e.capture(recorded)
    for _ in range(5):
        buf.zero_()
        g.replay(); torch.cuda.synchronize()
        assert np.array_equal(buf.cpu().numpy(), want)
    assert h.peer_timeouts() == {}
    # errors: connecting before exporting, or with a blob that is not one
    h2 = CHalo(_plan(n, ghost, mirror), e, max_nlev=4, transport="loopback")
    hd = h2.handles["reverse"]
    assert e.L.mimsem_halo_set_peer(hd, 0, bytes(HALO_PEER_BLOB)) != 0
    blob = C.create_string_buffer(HALO_PEER_BLOB)
    check(e.L.mimsem_halo_peer_export(hd, 0, blob), "halo_peer_export")
    assert e.L.mimsem_halo_set_peer(hd, 0, bytes(HALO_PEER_BLOB)) != 0          # (no magic)
    assert e.L.mimsem_halo_set_peer(hd, 1, blob.raw) != 0                        # (exported as rank 0)
    for x in (ref, h, h2):
        x.close()


def test_ordered_add_with_repeated_targets(eng):
    """two neighbours' messages land on the same slots (cube-corner nodes): added in neighbour order, range by range"""
    import ctypes as C
    from mimsem_amd._lib import check
    e, P = eng
    n = P.n0
    ranks = np.array([0, 0], np.int32)                                # the rank is its own neighbour twice
    send_idx = np.array([1, 2, 3, 4, 5, 6], np.int32); send_off = np.array([0, 3, 6], np.int32)
    recv_idx = np.array([10, 11, 12, 12, 11, 13], np.int32); recv_off = np.array([0, 3, 6], np.int32)     # 11 and 12 repeat
    h = C.c_void_p()
    check(e.L.mimsem_halo_create(e.ctx, 2, ranks.ctypes.data, send_idx.ctypes.data, send_off.ctypes.data, recv_idx.ctypes.data,
                                 recv_off.ctypes.data, n, 2, C.byref(h)), "create")
    check(e.L.mimsem_halo_set_loopback(h), "loopback")
    v0 = np.random.default_rng(4).standard_normal((2, n)); v = e.tensor(v0)
    check(e.L.mimsem_halo_begin(h, 1, 2, v.data_ptr(), v.stride(0)), "begin")
    check(e.L.mimsem_halo_end(h), "end")
    want = v0.copy()
    for s, r in zip(send_idx, recv_idx):                               # neighbour 0 first, then neighbour 1
        want[:, r] += v0[:, s]
    assert np.array_equal(v.cpu().numpy(), want)
    e.L.mimsem_halo_destroy(h)


def test_halo_errors(eng):
    import ctypes as C
    from mimsem_amd._lib import MimsemError
    from mimsem_amd.partition import CHalo
    e, P = eng
    n = P.n1
    h = CHalo(_plan(n, [0, 1], [2, 3]), e, max_nlev=2, transport="loopback")
    v = e.zeros(2, n)
    tok = h.begin("reverse", v, True)
    with pytest.raises(MimsemError):
        h.begin("reverse", v, True)                                   # one exchange in flight per plan
    h.end(tok)
    with pytest.raises(MimsemError):
        h.end(tok)                                                    # end without begin
    with pytest.raises(MimsemError):
        h.begin("reverse", e.zeros(3, n), True)                       # more levels than the plan was created for
    h.close()
    out = C.c_void_p()
    bad = np.array([n + 5], np.int32); off = np.array([0, 1], np.int32); rk = np.array([0], np.int32)
    assert e.L.mimsem_halo_create(e.ctx, 1, rk.ctypes.data, bad.ctypes.data, off.ctypes.data, bad.ctypes.data, off.ctypes.data, n, 1, C.byref(out)) == -1
    # a plan without a transport refuses to start
    ok = np.array([1], np.int32)
    assert e.L.mimsem_halo_create(e.ctx, 1, rk.ctypes.data, ok.ctypes.data, off.ctypes.data, ok.ctypes.data, off.ctypes.data, n, 1, C.byref(out)) == 0
    assert e.L.mimsem_halo_begin(out, 1, 1, v.data_ptr(), v.stride(0)) == -4
    e.L.mimsem_halo_destroy(out)


def test_rccl_transport_on_one_rank():
    """mimsem_halo_set_rccl executed on hardware: a size-1 communicator and a plan whose neighbour is the rank itself (grouped
    ncclSend/ncclRecv to self) -- all a one-GPU box can show of the xGMI transport; in its own process (it owns a process group)"""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "exp_rccl_self.py")], env=env, capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "rccl self exchange ok: True" in r.stdout and "forward ok: True" in r.stdout, r.stdout[-2000:]


@pytest.mark.parametrize("op,fl", [("UMAT", 1), ("UHMAT", 1), ("ROTMAT", 0)])
def test_interior_boundary_split_of_an_apply(oracle, op, fl):
    """mimsem_ctx_set_halo_slots + mimsem_op_apply_part: boundary part then interior part = the whole apply bit for bit, and the
    marked (halo) slots are already final after the boundary part"""
    import torch
    from mimsem_amd.device import DeviceMesh, Engine
    import ctypes as C
    from mimsem_amd.geom import Geom
    from mimsem_amd.mesh import CubedSphere, sphere_coords
    from mimsem_amd.topo import Topo
    from mimsem_amd.workloads import SCALE, z_levels
    cs = CubedSphere(3, 4, 6); coords = sphere_coords(3, 4)
    topos = [Topo(cs, p, 9) for p in range(6)]
    geoms = [Geom(t, cs, coords, 9) for t in topos]
    for g in geoms:
        g.set_levels(z_levels(9, g.n0))
    dm = DeviceMesh(topos, geoms, nk=9, numbering="global")
    eng = Engine(dm)
    st = (C.c_int * 5)()
    assert eng.L.mimsem_op_wave_stats(eng.ctx, 9, st) == 1                   # the wave-level form is what runs here
    P = dm
    r = np.random.default_rng(2)
    x = eng.tensor(r.standard_normal((9, P.n1)))
    f = {"UMAT": None, "UHMAT": eng.tensor(r.uniform(0.5, 1.5, (9, P.n2)) * 1e6), "ROTMAT": eng.tensor(r.standard_normal((9, P.n0)) * 1e-4)}[op]
    whole = eng.apply(op, x, f=f, lev0=0, scale=SCALE, flags=fl)
    marked = np.sort(r.choice(P.n1, 60, replace=False)).astype(np.int32)
    eng.set_halo_slots(1, marked)
    again = eng.apply(op, x, f=f, lev0=0, scale=SCALE, flags=fl)            # the re-ordered plan computes the same thing
    assert torch.equal(again, whole)
    out = torch.full_like(whole, float("nan"))
    eng.apply_part(op, "boundary", x, f=f, lev0=0, scale=SCALE, flags=fl, out=out)
    assert torch.equal(out[:, marked.astype(np.int64)], whole[:, marked.astype(np.int64)])
    assert torch.isnan(out).any()                                            # something is left for the interior part
    eng.apply_part(op, "interior", x, f=f, lev0=0, scale=SCALE, flags=fl, out=out)
    assert torch.equal(out, whole)
    # accumulate form
    base = eng.tensor(r.standard_normal((9, P.n1)))
    want = base.clone(); eng.apply(op, x, f=f, lev0=0, scale=SCALE, flags=fl | 2, alpha=0.5, out=want)
    got = base.clone()
    eng.apply_part(op, "boundary", x, f=f, lev0=0, scale=SCALE, flags=fl | 2, alpha=0.5, out=got)
    eng.apply_part(op, "interior", x, f=f, lev0=0, scale=SCALE, flags=fl | 2, alpha=0.5, out=got)
    assert torch.equal(got, want)


def test_split_apply_contract():
    """A pending BOUNDARY part can only be consumed by ITS interior part: anything else that would read or overwrite its partial sums
    is refused (MIMSEM_ERR_STATE), whole applies and other workspace users may run in between without disturbing it"""
    import torch
    from mimsem_amd._lib import MimsemError
    from mimsem_amd.device import DeviceMesh, Engine
    from mimsem_amd.geom import Geom
    from mimsem_amd.mesh import CubedSphere, sphere_coords
    from mimsem_amd.topo import Topo
    from mimsem_amd.workloads import SCALE, z_levels
    cs = CubedSphere(3, 4, 6); coords = sphere_coords(3, 4)
    topos = [Topo(cs, p, 9) for p in range(6)]
    geoms = [Geom(t, cs, coords, 9) for t in topos]
    for g in geoms:
        g.set_levels(z_levels(9, g.n0))
    dm = DeviceMesh(topos, geoms, nk=9, numbering="global")
    eng = Engine(dm)
    r = np.random.default_rng(4)
    x = eng.tensor(r.standard_normal((9, dm.n1))); x2 = eng.tensor(r.standard_normal((9, dm.n1)))
    h = eng.tensor(r.uniform(0.5, 1.5, (9, dm.n2)) * 1e6)
    marked = np.sort(r.choice(dm.n1, 80, replace=False)).astype(np.int32)
    eng.set_halo_slots(1, marked)
    whole = eng.apply("UMAT", x, lev0=0, scale=SCALE, flags=1)
    out = torch.zeros_like(whole); other = torch.zeros_like(whole)
    eng.apply_part("UMAT", "boundary", x, lev0=0, scale=SCALE, flags=1, out=out)
    with pytest.raises(MimsemError):                                         # a second boundary part while one is pending
        eng.apply_part("UMAT", "boundary", x2, lev0=0, scale=SCALE, flags=1, out=other)
    with pytest.raises(MimsemError):                                         # an interior part of another apply (other y)
        eng.apply_part("UMAT", "interior", x, lev0=0, scale=SCALE, flags=1, out=other)
    with pytest.raises(MimsemError):                                         # ... other operator
        eng.apply_part("UHMAT", "interior", x, f=h, lev0=0, scale=SCALE, flags=1, out=out)
    with pytest.raises(MimsemError):                                         # ... other level range
        eng.apply_part("UMAT", "interior", x[:4], lev0=0, scale=SCALE, flags=1, out=out[:4])
    with pytest.raises(MimsemError):                                         # re-planning under a pending part
        eng.set_halo_slots(1, marked[:10])
    # users of the shared workspace in between: whole applies (wave and two-pass kernels), a 2-form-valued operator
    eng.apply("UHMAT", x2, f=h, lev0=0, scale=SCALE, flags=1)
    eng.apply("PMAT", eng.tensor(r.standard_normal((9, dm.n0))), lev0=0, scale=SCALE, flags=0)
    eng.apply("UMAT", x2, lev0=0, scale=SCALE, flags=1)
    eng.apply_part("UMAT", "interior", x, lev0=0, scale=SCALE, flags=1, out=out)
    assert torch.equal(out, whole)
    # error recovery: forget a pending part
    eng.apply_part("UMAT", "boundary", x, lev0=0, scale=SCALE, flags=1, out=out)
    assert eng.L.mimsem_op_apply_part_reset(eng.ctx) == 0
    eng.apply_part("UMAT", "boundary", x2, lev0=0, scale=SCALE, flags=1, out=other)
    eng.apply_part("UMAT", "interior", x2, lev0=0, scale=SCALE, flags=1, out=other)
    assert torch.equal(other, eng.apply("UMAT", x2, lev0=0, scale=SCALE, flags=1))
