"""Device-resident mesh + operator engine: thin Python layer over the C ABI (include/mimsem_hip.h).
PyTorch supplies device memory, streams and torch.distributed -- plumbing, not the product."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import COLOPS, FLAG_ACCUM, FLAG_VERT, OPS, MeshDesc, check


def _ptr(t):
    if t is None:
        return None
    if not (t.dtype in (torch.float64, torch.int32) and t.is_contiguous() and t.is_cuda):
        raise _lib.MimsemError("device float64/int32 contiguous tensor required, got %s %s contiguous=%s" % (t.device, t.dtype, t.is_contiguous()))
    return t.data_ptr()


def _need(cond, what):
    """argument validation that survives python -O: the C ABI takes raw pointers, a wrong-sized tensor would be an out-of-bounds
    device access (a GPU memory fault), so sizes are checked here and reported as MimsemError"""
    if not cond:
        raise _lib.MimsemError("invalid argument: " + what)


class no_gc:
    """Python's cyclic garbage collector must not run inside a stream capture: a collected device tensor is freed by the caching
    allocator with calls that are illegal on a capturing stream, the error is raised inside a destructor and the process aborts
    (seen as 'Fatal Python error: Aborted ... Garbage-collecting' in the middle of a GraphedGMRES capture)."""

    def __enter__(self):
        import gc
        self.was = gc.isenabled()
        gc.disable()                     # (torch.cuda.graph collects once on entry by itself)

    def __exit__(self, *exc):
        import gc
        if self.was:
            gc.enable()
        return False


class DeviceMesh:
    """Flattens any set of patches (Topo+Geom pairs) into the element->slot tables of mimsem_mesh_desc.

    numbering="local": exactly one patch, vectors have the reference's per-rank LOCAL (ghosted) layout
                       (Topo::elInds*_l) -- what VecGetArray on the reference's `*l` vectors gives.
    numbering="global": any number of patches; vector slots are the compacted global ids touched by the
                        patches (all of them => the concatenated PETSc global Vec)."""

    def __init__(self, topos, geoms, nk=1, numbering="global"):
        t0 = topos[0]
        self.n, self.m, self.nk = t0.elOrd, geoms[0].quad_ord, nk
        self.topos, self.geoms = topos, geoms
        n2e = self.n * self.n
        if numbering == "local":
            assert len(topos) == 1
            self.inds0, self.inds1x, self.inds1y = t0.all_inds0_l(), t0.all_inds1x_l(), t0.all_inds1y_l()
            self.n0, self.n1, self.n2 = t0.n0, t0.n1, t0.n2
            self.gid0 = self.gid1 = self.gid2 = self.gidq = None
            self.indsq, self.nq = geoms[0].all_inds0_l(), geoms[0].n0
        else:
            g0 = np.concatenate([t.all_inds0_g() for t in topos])
            g1x = np.concatenate([t.all_inds1x_g() for t in topos])
            g1y = np.concatenate([t.all_inds1y_g() for t in topos])
            self.gid0, inv0 = np.unique(g0, return_inverse=True)
            self.gid1, inv1 = np.unique(np.concatenate([g1x.ravel(), g1y.ravel()]), return_inverse=True)
            self.inds0 = inv0.reshape(g0.shape).astype(np.int32)
            self.inds1x = inv1[:g1x.size].reshape(g1x.shape).astype(np.int32)
            self.inds1y = inv1[g1x.size:].reshape(g1y.shape).astype(np.int32)
            self.gid2 = np.concatenate([t.all_inds2_g().ravel() for t in topos])
            gq = np.concatenate([g.loc0[g.all_inds0_l()] for g in geoms])
            self.gidq, invq = np.unique(gq, return_inverse=True)
            self.indsq, self.nq = invq.reshape(gq.shape).astype(np.int32), self.gidq.size
            self.n0, self.n1 = self.gid0.size, self.gid1.size
            self.n2 = self.gid2.size
        self.nEl = self.inds0.shape[0]
        self.inds2 = np.arange(self.nEl * n2e, dtype=np.int32).reshape(self.nEl, n2e)
        self.det = np.ascontiguousarray(np.concatenate([g.det for g in geoms]))
        self.J = np.ascontiguousarray(np.concatenate([g.J for g in geoms]))
        th = [g.thick_at_elements() for g in geoms]
        self.thick = np.ascontiguousarray(np.concatenate([a for a, _ in th], axis=1))
        self.thickInv = np.ascontiguousarray(np.concatenate([b for _, b in th], axis=1))

    def desc(self):
        d = MeshDesc()
        d.elOrd, d.quadOrd, d.nEl, d.nk = self.n, self.m, self.nEl, self.nk
        d.n0, d.n1, d.n2 = self.n0, self.n1, self.n2
        keep = []
        d.nq = self.nq
        for name in ("inds0", "inds1x", "inds1y", "inds2", "indsq"):
            a = np.ascontiguousarray(getattr(self, name), dtype=np.int32); keep.append(a)
            setattr(d, name, a.ctypes.data)
        for name in ("det", "J", "thick", "thickInv"):
            a = np.ascontiguousarray(getattr(self, name), dtype=np.float64); keep.append(a)
            setattr(d, name, a.ctypes.data)
        d._keep = keep
        return d


class Engine:
    """One mimsem_ctx on one GPU."""

    def __init__(self, dmesh, device=0):
        self.L = _lib.lib()
        if not torch.cuda.is_available():
            raise _lib.MimsemError("no GPU visible: the operator engine has no CPU fallback")
        self.mesh = dmesh
        self.device = torch.device("cuda", device)
        self.ctx = C.c_void_p()
        d = dmesh.desc()
        torch.cuda.set_device(self.device)
        check(self.L.mimsem_ctx_create(C.byref(d), device, C.byref(self.ctx)), "mimsem_ctx_create")
        self.use_stream(torch.cuda.current_stream(self.device))
        n = dmesh.n
        self.n0e, self.n1e, self.n2e, self.mp12 = (n + 1) ** 2, (n + 1) * n, n * n, (n + 1) ** 2
        self.nEl, self.nk = dmesh.nEl, dmesh.nk
        self.sizes = {0: dmesh.n0, 1: dmesh.n1, 2: dmesh.n2, "q": dmesh.nq, "q2": 2 * dmesh.nq}

    def __del__(self):
        try:
            if self.ctx:
                self.L.mimsem_ctx_destroy(self.ctx); self.ctx = C.c_void_p()
        except Exception:
            pass

    def use_stream(self, stream):
        self._stream = stream
        check(self.L.mimsem_ctx_set_stream(self.ctx, C.c_void_p(stream.cuda_stream)), "set_stream")

    def on_current_stream(self):
        """context manager: launch the engine's kernels on torch's CURRENT stream (hipGraph capture, side streams), then
        go back to the stream the engine used before"""
        eng = self

        class _Bind:
            def __enter__(self_b):
                self_b.prev = getattr(eng, "_stream", None)
                eng.use_stream(torch.cuda.current_stream(eng.device))

            def __exit__(self_b, *exc):
                if self_b.prev is not None:
                    eng.use_stream(self_b.prev)
                return False
        return _Bind()

    def capture(self, fn):
        """Capture fn() -- engine calls and torch ops on fixed buffers, no host synchronisation -- in a hipGraph.  Returns
        (graph, outputs): graph.replay() re-runs the whole launch sequence with one submission; outputs are the static
        tensors fn returned (overwritten by every replay)."""
        dev = self.device
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s), self.on_current_stream():
            fn()                                        # warm-up: workspaces reach their final size outside the capture
        torch.cuda.current_stream(dev).wait_stream(s)
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        with no_gc(), torch.cuda.graph(g), self.on_current_stream():
            out = fn()
        torch.cuda.synchronize(dev)
        return g, out

    def sync(self):
        check(self.L.mimsem_ctx_sync(self.ctx), "sync")

    def set_profiling(self, every):
        """0 = off; n > 0 = time every n-th op_apply with hipExtLaunchKernelGGL start/stop events"""
        check(self.L.mimsem_ctx_set_profiling(self.ctx, int(every)), "set_profiling")

    def profile_read(self):
        """(ms in element kernels, ms in gather-sum kernels, launches) since the last read"""
        a, b, n = C.c_double(), C.c_double(), C.c_longlong()
        check(self.L.mimsem_ctx_profile_read(self.ctx, C.byref(a), C.byref(b), C.byref(n)), "profile_read")
        return a.value, b.value, n.value

    def tensor(self, a):
        return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64).to(self.device).contiguous()

    def zeros(self, *shape):
        return torch.zeros(*shape, dtype=torch.float64, device=self.device)

    # ---- horizontal operators ---------------------------------------------------------------
    _SPACES = dict(UMAT=(1, None, 1), UTMAT=(1, None, 1), UHMAT=(1, 2, 1), UTMAT_H=(1, 2, 1), ROTMAT=(1, 0, 1),
                   WMAT=(2, None, 2), WMATINV=(2, None, 2), WHMAT=(2, 2, 2), WHMATINV=(2, 2, 2), PMAT=(0, None, 0),
                   PHMAT=(0, 2, 0), WTQUMAT=(1, 1, 2), WTQDUDZ=(1, 1, 2), UTQWMAT=(2, 1, 1),
                   PHMAT_UP=(0, 2, 0), ROTMAT_UP=(1, 0, 1), UMAT_UP=(1, 1, 1), UHMAT_UP=(1, 2, 1), UVEC_HU_UP=(1, 2, 1), WTQ=("q", None, 2), PTQ=("q", None, 0), UTQ=("q2", None, 1))

    def apply(self, op, x, f=None, lev0=0, scale=1.0, flags=0, alpha=1.0, out=None):
        """y_k = A_op(level lev0+k, f_k) x_k ; x: [nlev, n_in] (or [n_in]) device tensor"""
        sin, sf, sout = self._SPACES[op]
        x2 = x if x.dim() == 2 else x.unsqueeze(0)
        nlev = x2.shape[0]
        assert x2.shape[1] == self.sizes[sin], (x2.shape, self.sizes[sin])
        f2 = None
        if sf is not None:
            f2 = f if f.dim() == 2 else f.unsqueeze(0)
            assert f2.shape == (nlev, self.sizes[sf])
        y = out if out is not None else torch.empty(nlev, self.sizes[sout], dtype=torch.float64, device=self.device)
        y2 = y if y.dim() == 2 else y.unsqueeze(0)
        check(self.L.mimsem_op_apply(self.ctx, OPS[op], lev0, nlev, scale, flags,
                                     _ptr(f2), f2.stride(0) if f2 is not None else 0,
                                     _ptr(x2), x2.stride(0), _ptr(y2), y2.stride(0), alpha), "mimsem_op_apply(%s)" % op)
        return y if x.dim() == 2 else y2[0]

    def set_halo_slots(self, form, slots):
        """mark the 1-form slots that take part in a halo exchange: their element groups move to the front of the plan, so that
        apply_part(..., "boundary") completes exactly those slots (mimsem_ctx_set_halo_slots)"""
        sl = np.ascontiguousarray(slots, dtype=np.int32)
        check(self.L.mimsem_ctx_set_halo_slots(self.ctx, form, sl.ctypes.data, sl.size), "ctx_set_halo_slots")

    def reset_parts(self):
        """forget a BOUNDARY part whose INTERIOR part will not come (error path of a split apply; mimsem_op_apply_part_reset)"""
        check(self.L.mimsem_op_apply_part_reset(self.ctx), "op_apply_part_reset")

    def apply_part(self, op, part, x, f=None, lev0=0, scale=1.0, flags=0, alpha=1.0, out=None):
        """the boundary or the interior part of apply(): part "boundary" first (all marked slots of `out` complete afterwards), then
        "interior" into the SAME out with the same arguments.  The pending boundary part keeps its partial sums in a buffer of its own:
        other calls may run in between, but a second boundary part or a non-matching interior part is refused (MIMSEM_ERR_STATE)"""
        sin, sf, sout = self._SPACES[op]
        x2 = x if x.dim() == 2 else x.unsqueeze(0)
        nlev = x2.shape[0]
        if x2.shape[1] != self.sizes[sin] or out is None or out.shape != (nlev, self.sizes[sout]):
            raise _lib.MimsemError("apply_part: x [nlev, n_in] and out [nlev, n_out] required")
        f2 = None
        if sf is not None:
            f2 = f if f.dim() == 2 else f.unsqueeze(0)
            if f2.shape != (nlev, self.sizes[sf]):
                raise _lib.MimsemError("apply_part: coefficient field of the wrong shape")
        check(self.L.mimsem_op_apply_part(self.ctx, OPS[op], lev0, nlev, scale, flags, _ptr(f2), f2.stride(0) if f2 is not None else 0,
                                          _ptr(x2), x2.stride(0), _ptr(out), out.stride(0), alpha, {"all": 0, "boundary": 1, "interior": 2}[part]),
              "mimsem_op_apply_part(%s)" % op)
        return out

    def apply_up(self, op, x, f, u, fac=None, dt=None, lev0=0, alpha=1.0, flags=0, out=None, scale=1.0, tau=None):
        """upwinded operators: f = the op's field, u = second (velocity) field.  SW ops (PHMAT_UP / ROTMAT_UP) pass fac, dt
        (tau = 1/(1/(fac*dt)), src/Assembly.cpp:541); the eul ops (UMAT_UP / UHMAT_UP / UVEC_HU_UP) pass scale and tau."""
        if tau is None:
            tau = 1.0 / (1.0 / (fac * dt))
        sin, sf, sout = self._SPACES[op]
        x2 = x if x.dim() == 2 else x.unsqueeze(0); f2 = f if f.dim() == 2 else f.unsqueeze(0); u2 = u if u.dim() == 2 else u.unsqueeze(0)
        nlev = x2.shape[0]
        assert x2.shape[1] == self.sizes[sin] and f2.shape == (nlev, self.sizes[sf]) and u2.shape == (nlev, self.sizes[1])
        y = out if out is not None else torch.empty(nlev, self.sizes[sout], dtype=torch.float64, device=self.device)
        y2 = y if y.dim() == 2 else y.unsqueeze(0)
        check(self.L.mimsem_op_apply_up(self.ctx, OPS[op], lev0, nlev, scale, tau, flags, _ptr(f2), f2.stride(0), _ptr(u2), u2.stride(0),
                                        _ptr(x2), x2.stride(0), _ptr(y2), y2.stride(0), alpha), "mimsem_op_apply_up(%s)" % op)
        return y if x.dim() == 2 else y2[0]

    def richardson_sweep(self, op, x, b, dinv, f=None, u=None, tau=0.0, lev0=0, scale=1.0, flags=0, upd=None):
        """x += dinv * (b - Op x) in place (mimsem_op_richardson_sweep: element pass + gather with the update epilogue);
        upd (optional, same shape) receives the update.  [nlev, n] tensors."""
        sin, sf, sout = self._SPACES[op]
        nlev = x.shape[0]
        assert sin == sout and x.dim() == 2 and x.shape == b.shape == dinv.shape and x.shape[1] == self.sizes[sin]
        assert upd is None or upd.shape == x.shape
        check(self.L.mimsem_op_richardson_sweep(self.ctx, OPS[op], lev0, nlev, scale, tau, flags,
                                                _ptr(f), f.stride(0) if f is not None else 0, _ptr(u), u.stride(0) if u is not None else 0,
                                                _ptr(b), b.stride(0), _ptr(dinv), dinv.stride(0), _ptr(x), x.stride(0),
                                                _ptr(upd), upd.stride(0) if upd is not None else 0), "richardson_sweep(%s)" % op)
        return x

    def chebyshev_sweep(self, op, x, b, dinv, p, alpha, beta, f=None, u=None, tau=0.0, lev0=0, scale=1.0, flags=0, upd=None):
        """z = dinv * (b - Op x); p = z + beta p; x += alpha p in place (mimsem_op_chebyshev_sweep: two launches); upd receives z"""
        sin, sf, sout = self._SPACES[op]
        nlev = x.shape[0]
        assert sin == sout and x.dim() == 2 and x.shape == b.shape == dinv.shape == p.shape and x.shape[1] == self.sizes[sin]
        assert upd is None or upd.shape == x.shape
        check(self.L.mimsem_op_chebyshev_sweep(self.ctx, OPS[op], lev0, nlev, scale, tau, flags,
                                               _ptr(f), f.stride(0) if f is not None else 0, _ptr(u), u.stride(0) if u is not None else 0,
                                               _ptr(b), b.stride(0), _ptr(dinv), dinv.stride(0), float(alpha), float(beta), _ptr(p), p.stride(0),
                                               _ptr(x), x.stride(0), _ptr(upd), upd.stride(0) if upd is not None else 0), "chebyshev_sweep(%s)" % op)
        return x

    def block_richardson_sweep(self, op, blocks, x, b, f=None, lev0=0, scale=1.0, flags=0, upd=None):
        """x += sum_e R_e^T B_e R_e (b - Op x) in place on 1-forms; blocks [nEl, 2 n1e, 2 n1e] column-major per element
        (mimsem_block_richardson_sweep: element pass, block pass with on-the-fly gathered residual, gather with update)"""
        nd = 2 * self.n1e
        assert x.dim() == 2 and x.shape == b.shape and x.shape[1] == self.sizes[1] and blocks.shape == (self.nEl, nd, nd)
        assert upd is None or upd.shape == x.shape
        check(self.L.mimsem_block_richardson_sweep(self.ctx, OPS[op], lev0, x.shape[0], scale, flags,
                                                   _ptr(f), f.stride(0) if f is not None else 0, _ptr(blocks),
                                                   _ptr(b), b.stride(0), _ptr(x), x.stride(0),
                                                   _ptr(upd), upd.stride(0) if upd is not None else 0), "block_richardson_sweep(%s)" % op)
        return x

    def block_chebyshev_sweep(self, op, blocks, x, b, p, alpha, beta, f=None, elem_scale=None, lev0=0, scale=1.0, flags=0, upd=None):
        """z = P (b - Op x) with P = sum_e R_e^T (elem_scale[lev, e] B_e) R_e; p = z + beta p; x += alpha p -- in place, three
        launches (mimsem_block_chebyshev_sweep).  blocks [nEl, 2 n1e, 2 n1e] column-major per element."""
        nd = 2 * self.n1e
        assert x.dim() == 2 and x.shape == b.shape == p.shape and x.shape[1] == self.sizes[1] and blocks.shape == (self.nEl, nd, nd)
        assert upd is None or upd.shape == x.shape
        assert elem_scale is None or elem_scale.shape == (x.shape[0], self.nEl)
        check(self.L.mimsem_block_chebyshev_sweep(self.ctx, OPS[op], lev0, x.shape[0], scale, flags,
                                                  _ptr(f), f.stride(0) if f is not None else 0, _ptr(blocks),
                                                  _ptr(elem_scale), elem_scale.stride(0) if elem_scale is not None else 0,
                                                  _ptr(b), b.stride(0), alpha, beta, _ptr(p), p.stride(0), _ptr(x), x.stride(0),
                                                  _ptr(upd), upd.stride(0) if upd is not None else 0), "block_chebyshev_sweep(%s)" % op)
        return x

    def block_chebyshev_solve(self, op, blocks, b, coef, x=None, elem_scale=None, lev0=0, scale=1.0, flags=0, pb=None, upd=None):
        """the whole fixed-length solve from x = 0 as ONE call: len(coef) steps of block_chebyshev_sweep, the first without its operator pass
        and without cleared x / p (mimsem_block_chebyshev_solve; the same bits).  coef: [(alpha, beta)]; pb / upd receive the first / last
        preconditioned residual."""
        import ctypes
        nd = 2 * self.n1e
        assert b.dim() == 2 and b.shape[1] == self.sizes[1] and blocks.shape == (self.nEl, nd, nd)
        x = torch.empty_like(b) if x is None else x
        assert x.shape == b.shape and (pb is None or pb.shape == b.shape) and (upd is None or upd.shape == b.shape)
        assert elem_scale is None or elem_scale.shape == (b.shape[0], self.nEl)
        flat = (ctypes.c_double * (2 * len(coef)))(*[v for ab in coef for v in ab])
        check(self.L.mimsem_block_chebyshev_solve(self.ctx, OPS[op], lev0, b.shape[0], scale, flags, None, 0, _ptr(blocks),
                                                  _ptr(elem_scale), elem_scale.stride(0) if elem_scale is not None else 0,
                                                  _ptr(b), b.stride(0), len(coef), flat, _ptr(x), x.stride(0),
                                                  _ptr(pb), pb.stride(0) if pb is not None else 0,
                                                  _ptr(upd), upd.stride(0) if upd is not None else 0), "block_chebyshev_solve(%s)" % op)
        return x

    def sw_dual_chebyshev(self, coefA, blocks, b1, p1, x1, upd1, coefB, tau, h, u, b0, dinv, p0, x0, upd0, pb1=None, pb0=None):
        """the 1-form mass solve (len(coefA) block-Chebyshev steps on Umat) and the upwinded lumped 0-form mass solve (len(coefB) Chebyshev steps
        on Phmat_up) of one shallow-water Picard iteration, both from x = 0, in SHARED launches (mimsem_sw_dual_chebyshev): the same bits as the
        two sequences of block_chebyshev_sweep / chebyshev_sweep calls on zero iterates.  [1, n] tensors; x1, p1, x0, p0 are outputs / workspaces
        (need not be cleared); upd1 / upd0 receive the last step's preconditioned residual, pb1 / pb0 the first's (P b1, dinv b0)."""
        nd = 2 * self.n1e
        for t, n in ((b1, self.sizes[1]), (p1, self.sizes[1]), (x1, self.sizes[1]), (u, self.sizes[1]), (h, self.sizes[2]), (b0, self.sizes[0]), (dinv, self.sizes[0]),
                     (p0, self.sizes[0]), (x0, self.sizes[0])):
            assert t.dim() == 2 and t.shape == (1, n) and t.is_contiguous(), (tuple(t.shape), n)
        assert blocks.shape == (self.nEl, nd, nd) and blocks.is_contiguous()
        for t, ref in ((upd1, x1), (pb1, x1), (upd0, x0), (pb0, x0)):
            assert t is None or (t.shape == ref.shape and t.stride(1) == 1)
        ca = np.ascontiguousarray(np.asarray(coefA, dtype=np.float64).reshape(-1, 2))
        cb = np.ascontiguousarray(np.asarray(coefB, dtype=np.float64).reshape(-1, 2))
        check(self.L.mimsem_sw_dual_chebyshev(self.ctx, ca.shape[0], ca.ctypes.data, _ptr(blocks), _ptr(b1), _ptr(p1), _ptr(x1), _ptr(upd1), _ptr(pb1),
                                              cb.shape[0], cb.ctypes.data, float(tau), _ptr(h), _ptr(u), _ptr(b0), _ptr(dinv), _ptr(p0), _ptr(x0), _ptr(upd0), _ptr(pb0)),
              "sw_dual_chebyshev")

    def apply_ray(self, x, exner, exner_s, dt, lev0=0, scale=1.0, alpha=1.0, flags=0, out=None):
        """Umat_ray (Held-Suarez friction): x [nlev, n1], exner [nlev, n2] (levels lev0..), exner_s [n2] = level 0."""
        x2 = x if x.dim() == 2 else x.unsqueeze(0); f2 = exner if exner.dim() == 2 else exner.unsqueeze(0)
        nlev = x2.shape[0]
        assert x2.shape[1] == self.sizes[1] and f2.shape == (nlev, self.sizes[2]) and exner_s.shape == (self.sizes[2],)
        y = out if out is not None else torch.empty(nlev, self.sizes[1], dtype=torch.float64, device=self.device)
        y2 = y if y.dim() == 2 else y.unsqueeze(0)
        check(self.L.mimsem_op_apply_up(self.ctx, OPS["UMAT_RAY"], lev0, nlev, scale, dt, flags, _ptr(f2), f2.stride(0),
                                        _ptr(exner_s), 0, _ptr(x2), x2.stride(0), _ptr(y2), y2.stride(0), alpha),
              "mimsem_op_apply_up(UMAT_RAY)")
        return y if x.dim() == 2 else y2[0]

    def prepare_apply(self, op, x, f=None, lev0=0, scale=1.0, flags=0, alpha=1.0, out=None):
        """Validate once, return (call, y): `call()` re-issues the same mimsem_op_apply with pre-marshalled
        arguments (the buffers are fixed) -- the host-side fast path for time-step loops and bench.py."""
        sin, sf, sout = self._SPACES[op]
        x2 = x if x.dim() == 2 else x.unsqueeze(0)
        nlev = x2.shape[0]
        assert x2.shape[1] == self.sizes[sin]
        f2 = None
        if sf is not None:
            f2 = f if f.dim() == 2 else f.unsqueeze(0)
            assert f2.shape == (nlev, self.sizes[sf])
        y = out if out is not None else torch.empty(nlev, self.sizes[sout], dtype=torch.float64, device=self.device)
        y2 = y if y.dim() == 2 else y.unsqueeze(0)
        fn = self.L.mimsem_op_apply
        args = (self.ctx, C.c_int(OPS[op]), C.c_int(lev0), C.c_int(nlev), C.c_double(scale), C.c_uint(flags),
                C.c_void_p(_ptr(f2)), C.c_longlong(f2.stride(0) if f2 is not None else 0),
                C.c_void_p(_ptr(x2)), C.c_longlong(x2.stride(0)), C.c_void_p(_ptr(y2)), C.c_longlong(y2.stride(0)),
                C.c_double(alpha))
        keep = (x2, f2, y2)

        def call(_fn=fn, _args=args, _keep=keep):
            rc = _fn(*_args)
            if rc:
                check(rc, "mimsem_op_apply(%s)" % op)
        return call, y

    def element_matrices(self, op, f=None, lev=0, scale=1.0, flags=0):
        esz = self.L.mimsem_op_elmat_size(self.ctx, OPS[op])
        out = torch.empty(self.nEl, esz, dtype=torch.float64, device=self.device)
        check(self.L.mimsem_op_element_matrices(self.ctx, OPS[op], lev, scale, flags, _ptr(f), _ptr(out)),
              "mimsem_op_element_matrices(%s)" % op)
        return out

    def element_matrices_ray(self, exner, exner_s, dt, lev=0, scale=1.0):
        esz = self.L.mimsem_op_elmat_size(self.ctx, OPS["UMAT_RAY"])
        out = torch.empty(self.nEl, esz, dtype=torch.float64, device=self.device)
        check(self.L.mimsem_op_element_matrices_ex(self.ctx, OPS["UMAT_RAY"], lev, scale, dt, 0, _ptr(exner), _ptr(exner_s), _ptr(out)),
              "mimsem_op_element_matrices_ex(UMAT_RAY)")
        return out

    def blocks_apply(self, form, blocks, x, transpose=False, alpha=1.0, accum=False, out=None, elem_scale=None):
        """y = alpha * sum_e P_e^T B_e P_e x with caller-supplied element blocks [nEl, nd, nd] (same on every level, optionally
        times elem_scale[lev, e]) or [nlev, nEl, nd, nd]; form 0/1/2 (1-forms: nd = 2*n1e, x-edges then y-edges)."""
        x2 = x if x.dim() == 2 else x.unsqueeze(0)
        nlev = x2.shape[0]
        nd = {0: self.n0e, 1: 2 * self.n1e, 2: self.n2e}[form]
        assert blocks.shape[-3:] == (self.nEl, nd, nd) and x2.shape[1] == self.sizes[form]
        bstride = blocks.stride(0) if blocks.dim() == 4 else 0
        assert blocks.dim() == 3 or blocks.shape[0] == nlev
        y = out if out is not None else torch.empty(nlev, self.sizes[form], dtype=torch.float64, device=self.device)
        y2 = y if y.dim() == 2 else y.unsqueeze(0)
        flags = (4 if transpose else 0) | (2 if accum else 0)
        if elem_scale is not None:
            assert blocks.dim() == 3 and elem_scale.shape == (nlev, self.nEl)
        check(self.L.mimsem_elem_blocks_apply(self.ctx, form, nlev, flags, _ptr(blocks), bstride, _ptr(elem_scale),
                                              elem_scale.stride(0) if elem_scale is not None else 0, _ptr(x2), x2.stride(0),
                                              _ptr(y2), y2.stride(0), alpha), "mimsem_elem_blocks_apply")
        return y if x.dim() == 2 else y2[0]

    def wvec(self, rho, lev0=0, scale=1.0, vert_scale=True, out=None):
        """Wvec::assemble(lev, scale, vert_scale, rho) (eul/Assembly.cpp:2457-2495; row B18): the matrix-free 2-form right-hand side
        W^T diag(w s/det [thickInv]) W rho -- Wmat applied to rho.  (The reference leaves its Wt table unfilled and has every call
        commented out; this is the evident intent, see oracle/o_assembly.c.)"""
        return self.apply("WMAT", rho, lev0=lev0, scale=scale, flags=FLAG_VERT if vert_scale else 0, out=out)

    def wvec_K(self, vel1, vel2, lev0=0, scale=1.0, out=None):
        """Wvec::assemble_K(lev, scale, vel1, vel2) (eul/Assembly.cpp:2497-2545): the kinetic-energy 2-form 1/2 <vel2, vel1> as a
        vector -- WtQUmat(vel2) applied to vel1"""
        return self.apply("WTQUMAT", vel1, f=vel2, lev0=lev0, scale=scale, out=out)

    def pvec(self, lev0=0, nlev=1, scale=1.0, h2=None):
        y = torch.empty(nlev, self.sizes[0], dtype=torch.float64, device=self.device)
        check(self.L.mimsem_pvec(self.ctx, lev0, nlev, scale, _ptr(h2), h2.stride(0) if h2 is not None else 0,
                                 _ptr(y), y.stride(0)), "mimsem_pvec")
        return y

    def incidence(self, which, x):
        """which: 'E10','E21','E12','E01'"""
        w = dict(E10=0, E21=1, E12=2, E01=3)[which]
        nout = {0: self.sizes[1], 1: self.sizes[2], 2: self.sizes[1], 3: self.sizes[0]}[w]
        nin = {0: self.sizes[0], 1: self.sizes[1], 2: self.sizes[2], 3: self.sizes[1]}[w]
        x2 = x if x.dim() == 2 else x.unsqueeze(0)
        assert x2.shape[1] == nin, (which, x2.shape, nin)          # the C ABI takes raw pointers: lengths are checked here
        y = torch.empty(x2.shape[0], nout, dtype=torch.float64, device=self.device)      # (every entry is written: faces directly, edges / nodes by the gather pass over all slots)
        check(self.L.mimsem_incidence_apply(self.ctx, w, x2.shape[0], _ptr(x2), x2.stride(0), _ptr(y), y.stride(0)), "incidence")
        return y if x.dim() == 2 else y[0]

    def interp_quad(self, form, x, push_forward=True):
        """Row A7, Geom::interp0 / interp1_l|_g / interp2_l|_g (eul/Geom.cpp:328-417) at every quadrature point:
        x [nlev, n_form] (or [n_form]) -> [nlev, nEl, mp12] (forms 0, 2) or [nlev, nEl, mp12, 2] (1-forms)."""
        x2 = x if x.dim() == 2 else x.unsqueeze(0)
        assert x2.shape[1] == self.sizes[form], (form, x2.shape)
        nc = 2 if form == 1 else 1
        out = torch.empty(x2.shape[0], self.nEl*self.mp12*nc, dtype=torch.float64, device=self.device)
        check(self.L.mimsem_interp_quad(self.ctx, form, 1 if push_forward else 0, x2.shape[0], _ptr(x2), x2.stride(0),
                                        _ptr(out), out.stride(0)), "interp_quad")
        out = out.view(x2.shape[0], self.nEl, self.mp12, 2) if form == 1 else out.view(x2.shape[0], self.nEl, self.mp12)
        return out if x.dim() == 2 else out[0]

    def sw_operator(self, a, grav, H, f0, x, out=None):
        """SWEqn::assemble_operator + MatMult (src/SWEqn_Picard.cpp:622-725) in one element pass: x, y packed rows [u | h]"""
        x2 = x if x.dim() == 2 else x.unsqueeze(0)
        assert x2.shape[1] == self.sizes[1] + self.sizes[2] and x2.stride(1) == 1, x2.shape
        f2 = f0 if f0.dim() == 2 else f0.unsqueeze(0)
        assert f2.shape[1] == self.sizes[0] and f2.shape[0] in (1, x2.shape[0])
        y = out if out is not None else torch.empty(x2.shape[0], x2.shape[1], dtype=torch.float64, device=self.device)
        y2 = y if y.dim() == 2 else y.unsqueeze(0)
        check(self.L.mimsem_sw_operator_apply(self.ctx, x2.shape[0], a, grav, H, f2.data_ptr(), 0 if f2.shape[0] == 1 else f2.stride(0),
                                              x2.data_ptr(), x2.stride(0), y2.data_ptr(), y2.stride(0)), "sw_operator")
        return y if (x.dim() == 2 or out is not None) else y[0]

    def sw_operator_precond(self, a, grav, H, f0, blocks, x, out=None):
        """z = P (A x) in three launches (mimsem_sw_operator_precond_apply): the Krylov body of the shallow-water solve"""
        nd = 2 * self.n1e + self.n2e
        x2 = x if x.dim() == 2 else x.unsqueeze(0)
        assert x2.shape[1] == self.sizes[1] + self.sizes[2] and x2.stride(1) == 1 and blocks.shape == (self.nEl, nd, nd)
        f2 = f0 if f0.dim() == 2 else f0.unsqueeze(0)
        assert f2.shape[1] == self.sizes[0] and f2.shape[0] in (1, x2.shape[0])
        z = out if out is not None else torch.empty(x2.shape[0], x2.shape[1], dtype=torch.float64, device=self.device)
        z2 = z if z.dim() == 2 else z.unsqueeze(0)
        check(self.L.mimsem_sw_operator_precond_apply(self.ctx, x2.shape[0], a, grav, H, f2.data_ptr(), 0 if f2.shape[0] == 1 else f2.stride(0),
                                                      _ptr(blocks), x2.data_ptr(), x2.stride(0), z2.data_ptr(), z2.stride(0)), "sw_operator_precond")
        return z if (x.dim() == 2 or out is not None) else z[0]

    def sw_operator_precond_chebyshev(self, a, grav, H, f0, blocks, ca, cb, x, r, d):
        """one Chebyshev step on B = P A in three launches (mimsem_sw_operator_precond_chebyshev): x += d; r -= P A d; d = ca d + cb r, in place"""
        nd = 2 * self.n1e + self.n2e
        assert x.dim() == 2 and x.shape == r.shape == d.shape and x.shape[1] == self.sizes[1] + self.sizes[2] and blocks.shape == (self.nEl, nd, nd)
        f2 = f0 if f0.dim() == 2 else f0.unsqueeze(0)
        assert f2.shape[1] == self.sizes[0] and f2.shape[0] in (1, x.shape[0])
        check(self.L.mimsem_sw_operator_precond_chebyshev(self.ctx, x.shape[0], a, grav, H, f2.data_ptr(), 0 if f2.shape[0] == 1 else f2.stride(0),
                                                          _ptr(blocks), float(ca), float(cb), _ptr(x), x.stride(0), _ptr(r), r.stride(0),
                                                          _ptr(d), d.stride(0)), "sw_operator_precond_chebyshev")

    def sw_operator_precond_orthogonalize(self, a, grav, H, f0, blocks, x, V, k, h, out, alpha=-1.0):
        """out = P (A x); h[:k] = V[:k] out; out += alpha V[:k]^T h -- the Krylov body and the first Gram-Schmidt pass of the Arnoldi step in four
        launches (mimsem_sw_operator_precond_orthogonalize; bit-identical to sw_operator_precond + orthogonalize, which take five)"""
        nd = 2 * self.n1e + self.n2e
        n = self.sizes[1] + self.sizes[2]
        assert x.numel() == n and out.numel() == n and x.is_contiguous() and out.is_contiguous() and blocks.shape == (self.nEl, nd, nd)
        assert f0.numel() == self.sizes[0] and V.stride(1) == 1 and V.shape[1] == n and 0 <= k <= V.shape[0] and h.numel() >= k
        check(self.L.mimsem_sw_operator_precond_orthogonalize(self.ctx, a, grav, H, f0.data_ptr(), _ptr(blocks), x.data_ptr(), out.data_ptr(),
                                                              k, _ptr(V), V.stride(0), alpha, _ptr(h)), "sw_operator_precond_orthogonalize")
        return out

    def sw_blocks_apply(self, blocks, x, out=None):
        """z = sum_e R_e^T B_e R_e x on packed rows [u | h]; blocks [nEl, ND, ND] stored column-major per element (mimsem_sw_blocks_apply)"""
        nd = 2 * self.n1e + self.n2e
        x2 = x if x.dim() == 2 else x.unsqueeze(0)
        assert x2.shape[1] == self.sizes[1] + self.sizes[2] and x2.stride(1) == 1 and blocks.shape == (self.nEl, nd, nd)
        y = out if out is not None else torch.empty(x2.shape[0], x2.shape[1], dtype=torch.float64, device=self.device)
        y2 = y if y.dim() == 2 else y.unsqueeze(0)
        check(self.L.mimsem_sw_blocks_apply(self.ctx, x2.shape[0], _ptr(blocks), x2.data_ptr(), x2.stride(0), y2.data_ptr(), y2.stride(0)),
              "sw_blocks_apply")
        return y if (x.dim() == 2 or out is not None) else y[0]

    # ---- column operators -------------------------------------------------------------------
    def _col(self, t, slots, name, optional=False):
        """a "vertical" array [nEl, slots*n2e] (L2Vecs::vz of every column)"""
        if t is None:
            _need(optional, name + " is required")
            return
        _need(t.dim() == 2 and t.shape[0] == self.nEl and t.shape[1] == slots * self.n2e,
              "%s must be [nEl=%d, %d*n2e=%d], got %s" % (name, self.nEl, slots, slots * self.n2e, tuple(t.shape)))

    def _colop_slots(self, colop, transpose=False):
        """(input slots, output slots) of a column operator in units of n2e (eul/VertOps.cpp: rows x cols of each Assemble*)"""
        nk = self.nk
        rc = dict(CONST=(nk, nk), CONST_INV=(nk, nk), CONST_RHO=(nk, nk), CONST_RHO_INV=(nk, nk), CONST_THETA=(nk, nk), EOS_BLOCK=(nk, nk),
                  EOS_BLOCK_INV=(nk, nk), LINEAR=(nk - 1, nk - 1), LINEAR_INV=(nk - 1, nk - 1), LINEAR_RT=(nk - 1, nk - 1),
                  LINEAR_THETA=(nk - 1, nk - 1), RAYLEIGH=(nk - 1, nk - 1), LINEAR_RAYLEIGH_INV=(nk - 1, nk - 1),
                  LINEAR_RHO2=(nk + 1, nk + 1), LINEAR_RHO2_UP=(nk + 1, nk + 1), LINCON=(nk - 1, nk), LINCON2=(nk + 1, nk), LINCON2_UP=(nk + 1, nk),
                  CONLIN=(nk, nk - 1), CONLIN_W=(nk, nk - 1), CONLIN_RHODPI=(nk, nk - 1))[colop]
        rows, cols = rc
        return (rows, cols) if transpose else (cols, rows)

    _COLOP_F1 = dict(CONST_RHO="nk", CONST_RHO_INV="nk", CONST_THETA="nk+1", EOS_BLOCK="nk", EOS_BLOCK_INV="nk", LINEAR_RT="nk", LINEAR_THETA="nk+1",
                     LINEAR_RHO2="nk", LINEAR_RHO2_UP="nk", CONLIN_W="nk-1", CONLIN_RHODPI="nk")

    def _check_colop(self, colop, f1, f2, x=None, nout_slots=None, transpose=False, uh=None):
        _need(colop in COLOPS, "unknown column operator %r" % (colop,))
        nk = self.nk
        if colop in self._COLOP_F1:
            self._col(f1, eval(self._COLOP_F1[colop], {"nk": nk}), "f1 of " + colop)
        if colop == "CONLIN_RHODPI":
            self._col(f2, nk - 1, "f2 of CONLIN_RHODPI")
        if colop == "EOS_BLOCK_INV" and f2 is not None:
            self._col(f2, nk + 1, "f2 (theta) of EOS_BLOCK_INV")
        if colop in ("LINEAR_RHO2_UP", "LINCON2_UP"):
            _need(uh is not None and uh.dim() == 2 and uh.shape == (nk, self.sizes[1]), "uh of %s must be [nk, n1]" % colop)
        if x is not None:
            sin, sout = self._colop_slots(colop, transpose)
            self._col(x, sin, "x of " + colop)
            _need(nout_slots == sout, "%s%s maps %d -> %d slots; nout_slots=%r" % (colop, "^T" if transpose else "", sin, sout, nout_slots))

    def l2_horiz_to_vert(self, vh):
        _need(vh.dim() == 2 and vh.shape[1] == self.sizes[2] and vh.shape[0] in (self.nk - 1, self.nk, self.nk + 1), "vh must be [nk-1|nk|nk+1, n2], got %s" % (tuple(vh.shape),))
        nkv = vh.shape[0]
        vz = torch.empty(self.nEl, nkv * self.n2e, dtype=torch.float64, device=self.device)
        check(self.L.mimsem_l2_transpose(self.ctx, 0, nkv, _ptr(vh), vh.stride(0), _ptr(vz)), "l2_transpose")
        return vz

    def l2_vert_to_horiz(self, vz, nkv):
        self._col(vz, nkv, "vz")
        vh = torch.empty(nkv, self.sizes[2], dtype=torch.float64, device=self.device)
        check(self.L.mimsem_l2_transpose(self.ctx, 1, nkv, _ptr(vh), vh.stride(0), _ptr(vz)), "l2_transpose")
        return vh

    def colop_blocks(self, colop, f1=None, f2=None, flags=0):
        self._check_colop(colop, f1, f2)
        nb = self.L.mimsem_colop_nblocks(self.ctx, COLOPS[colop])
        out = torch.empty(self.nEl, nb, self.n2e, self.n2e, dtype=torch.float64, device=self.device)
        check(self.L.mimsem_colop_blocks(self.ctx, COLOPS[colop], flags, _ptr(f1), _ptr(f2), _ptr(out)), "colop_blocks(%s)" % colop)
        return out

    def colop_apply(self, colop, x, f1=None, f2=None, flags=0, transpose=False, nout_slots=None):
        """nout_slots: rows of the operator in units of n2e (required; checked against the operator's shape)"""
        self._check_colop(colop, f1, f2, x, nout_slots, transpose)
        y = torch.empty(self.nEl, nout_slots * self.n2e, dtype=torch.float64, device=self.device)
        check(self.L.mimsem_colop_apply(self.ctx, COLOPS[colop], flags, int(transpose), _ptr(f1), _ptr(f2), _ptr(x), _ptr(y)),
              "colop_apply(%s)" % colop)
        return y

    def colop_apply_blocks(self, colop, blocks, x, nout_slots, transpose=False):
        """MatMult with blocks from colop_blocks (geometry-only operators assembled once)"""
        self._check_colop(colop, None, None, x, nout_slots, transpose)
        _need(blocks.dim() == 4 and blocks.shape[0] == self.nEl and blocks.shape[1] == self.L.mimsem_colop_nblocks(self.ctx, COLOPS[colop])
              and blocks.shape[2:] == (self.n2e, self.n2e), "blocks of %s have shape %s" % (colop, tuple(blocks.shape)))
        y = torch.empty(self.nEl, nout_slots * self.n2e, dtype=torch.float64, device=self.device)
        check(self.L.mimsem_colop_apply_blocks(self.ctx, COLOPS[colop], int(transpose), _ptr(blocks), _ptr(x), _ptr(y)),
              "colop_apply_blocks(%s)" % colop)
        return y

    def colop_blocks_ex(self, colop, param=0.0, f1=None, f2=None, uh=None, flags=0):
        """the Strang / Held-Suarez colops: param = dt_fric or dt, uh = [nk, n1] horizontal velocity (local 1-forms)"""
        self._check_colop(colop, f1, f2, uh=uh)
        nb = self.L.mimsem_colop_nblocks(self.ctx, COLOPS[colop])
        out = torch.empty(self.nEl, nb, self.n2e, self.n2e, dtype=torch.float64, device=self.device)
        check(self.L.mimsem_colop_blocks_ex(self.ctx, COLOPS[colop], flags, param, _ptr(f1), _ptr(f2), _ptr(uh),
                                            uh.stride(0) if uh is not None else 0, _ptr(out)), "colop_blocks_ex(%s)" % colop)
        return out

    def colop_apply_ex(self, colop, x, nout_slots, param=0.0, f1=None, f2=None, uh=None, flags=0, transpose=False):
        self._check_colop(colop, f1, f2, x, nout_slots, transpose, uh=uh)
        y = torch.empty(self.nEl, nout_slots * self.n2e, dtype=torch.float64, device=self.device)
        check(self.L.mimsem_colop_apply_ex(self.ctx, COLOPS[colop], flags, int(transpose), param, _ptr(f1), _ptr(f2), _ptr(uh),
                                           uh.stride(0) if uh is not None else 0, _ptr(x), _ptr(y)), "colop_apply_ex(%s)" % colop)
        return y

    def column_incidence(self, which, x):
        """'V10' | 'V01' | 'V10_full' applied to every column (VertOps::vertOps)"""
        w = dict(V10=0, V01=1, V10_full=2)[which]
        self._col(x, (self.nk - 1, self.nk, self.nk + 1)[w], "x of " + which)
        ny = self.nk - 1 if w == 1 else self.nk
        y = torch.empty(self.nEl, ny * self.n2e, dtype=torch.float64, device=self.device)
        check(self.L.mimsem_column_incidence(self.ctx, w, _ptr(x), _ptr(y)), "column_incidence")
        return y

    def diag_theta_up(self, dt, rho, rt, uh):
        self._col(rho, self.nk, "rho"); self._col(rt, self.nk, "rt")
        _need(uh.dim() == 2 and uh.shape == (self.nk, self.sizes[1]), "uh must be [nk, n1]")
        th = torch.empty(self.nEl, (self.nk + 1) * self.n2e, dtype=torch.float64, device=self.device)
        check(self.L.mimsem_column_diag_theta_up(self.ctx, dt, _ptr(rho), _ptr(rt), _ptr(uh), uh.stride(0), _ptr(th)), "diag_theta_up")
        return th

    def temp_forcing_hs(self, lat, exner, theta, rho):
        _need(lat.dim() == 2 and lat.shape == (self.nEl, self.mp12), "lat must be [nEl, mp12] (latitude of the quadrature points)")
        self._col(exner, self.nk, "exner"); self._col(theta, self.nk + 1, "theta"); self._col(rho, self.nk, "rho")
        out = torch.empty(self.nEl, self.nk * self.n2e, dtype=torch.float64, device=self.device)
        check(self.L.mimsem_column_temp_forcing_hs(self.ctx, _ptr(lat), _ptr(exner), _ptr(theta), _ptr(rho), _ptr(out)), "temp_forcing_hs")
        return out

    def solve_schur_3(self, dt, theta, velz, rho, rt, pi, F_u, F_rho, F_rt, F_pi, want_L=False, flags=0):
        """solve_schur_column_3 for every column; F_* updated in place; returns d_u, d_rho, d_rt, d_pi (, L [nEl,nk,5,n2e,n2e]);
        flags = 3 reproduces the box twin (box/VertSolve.cpp:879-1058)"""
        nk = self.nk
        for a, sl, nm in ((theta, nk + 1, "theta"), (velz, nk - 1, "velz"), (rho, nk, "rho"), (rt, nk, "rt"), (pi, nk, "pi"),
                          (F_u, nk - 1, "F_u"), (F_rho, nk, "F_rho"), (F_rt, nk, "F_rt"), (F_pi, nk, "F_pi")):
            self._col(a, sl, nm)
        N, Nm = self.nk * self.n2e, (self.nk - 1) * self.n2e
        mk = lambda n: torch.empty(self.nEl, n, dtype=torch.float64, device=self.device)
        d_u, d_rho, d_rt, d_pi = mk(Nm), mk(N), mk(N), mk(N)
        L = torch.empty(self.nEl, self.nk, 5, self.n2e, self.n2e, dtype=torch.float64, device=self.device) if want_L else None
        check(self.L.mimsem_column_solve_schur_3(self.ctx, dt, flags, _ptr(theta), _ptr(velz), _ptr(rho), _ptr(rt), _ptr(pi),
                                                 _ptr(F_u), _ptr(F_rho), _ptr(F_rt), _ptr(F_pi),
                                                 _ptr(d_u), _ptr(d_rho), _ptr(d_rt), _ptr(d_pi), _ptr(L)), "solve_schur_3")
        return (d_u, d_rho, d_rt, d_pi, L) if want_L else (d_u, d_rho, d_rt, d_pi)

    def column_eos(self, which, a, b=None, p0=0.0, p1=0.0):
        _need(which in (0, 1, 2, 3), "column_eos which = 0..3")
        self._col(a, self.nk, "a"); self._col(b, self.nk, "b", optional=which in (1, 2))
        out = torch.empty(self.nEl, self.nk * self.n2e, dtype=torch.float64, device=self.device)
        check(self.L.mimsem_column_eos(self.ctx, which, _ptr(a), _ptr(b), p0, p1, _ptr(out)), "column_eos")
        return out

    def diag_theta(self, which, rho, rt):
        _need(which in (0, 1), "diag_theta which = 0 (diagTheta_L2) | 1 (diagTheta2)")
        self._col(rho, self.nk, "rho"); self._col(rt, self.nk, "rt")
        nl = self.nk + (1 if which == 1 else 0)
        th = torch.empty(self.nEl, nl * self.n2e, dtype=torch.float64, device=self.device)
        check(self.L.mimsem_column_diag_theta(self.ctx, which, _ptr(rho), _ptr(rt), _ptr(th)), "diag_theta")
        return th

    def diag_theta_blend(self, rho, rt, blend2=None, blendL=None, wa=1.0, wb=0.0, want2=True, wantL=True):
        """diagTheta2 and diagTheta_L2 in ONE launch, optionally blended with earlier fields: wa * theta(rho, rt) + wb * blend
        (mimsem_column_diag_theta_blend).  Returns (theta2 [nEl, (nk+1) n2e] | None, thetaL [nEl, nk n2e] | None)."""
        self._col(rho, self.nk, "rho"); self._col(rt, self.nk, "rt")
        self._col(blend2, self.nk + 1, "blend2", optional=True); self._col(blendL, self.nk, "blendL", optional=True)
        mk = lambda n: torch.empty(self.nEl, n * self.n2e, dtype=torch.float64, device=self.device)
        th2 = mk(self.nk + 1) if want2 else None
        thL = mk(self.nk) if wantL else None
        check(self.L.mimsem_column_diag_theta_blend(self.ctx, _ptr(rho), _ptr(rt), _ptr(th2), _ptr(blend2), _ptr(thL), _ptr(blendL), wa, wb),
              "diag_theta_blend")
        return th2, thL

    def newton_residual(self, dt, rayleigh, theta, Pi, velz_i, velz_j, rho_i, rho_j, zv, rt_i, rt_j, rho_h, rt_h, exner_j,
                        add_w=None, add_rho=None, add_rt=None):
        """mimsem_column_newton_residual: (F_w, F_rho, F_eta, F_exner, th_w3, eta, k2i) for every column (VertSolve.cpp:1806-1851)"""
        nk = self.nk
        for a, sl, nm in ((theta, nk, "theta"), (Pi, nk, "Pi"), (velz_i, nk - 1, "velz_i"), (velz_j, nk - 1, "velz_j"), (rho_i, nk, "rho_i"),
                          (rho_j, nk, "rho_j"), (zv, nk, "zv"), (rt_i, nk, "rt_i"), (rt_j, nk, "rt_j"), (rho_h, nk, "rho_h"), (rt_h, nk, "rt_h"),
                          (exner_j, nk, "exner_j")):
            self._col(a, sl, nm)
        self._col(add_w, nk - 1, "add_w", optional=True); self._col(add_rho, nk, "add_rho", optional=True); self._col(add_rt, nk, "add_rt", optional=True)
        mk = lambda n: torch.empty(self.nEl, n * self.n2e, dtype=torch.float64, device=self.device)
        F_w, F_rho, F_eta, F_ex, th_w3, eta, k2i = mk(nk - 1), mk(nk), mk(nk), mk(nk), mk(nk), mk(nk), mk(nk - 1)
        check(self.L.mimsem_column_newton_residual(self.ctx, dt, rayleigh, _ptr(theta), _ptr(Pi), _ptr(velz_i), _ptr(velz_j), _ptr(rho_i), _ptr(rho_j),
                                                   _ptr(zv), _ptr(rt_i), _ptr(rt_j), _ptr(rho_h), _ptr(rt_h), _ptr(exner_j),
                                                   _ptr(add_w), _ptr(add_rho), _ptr(add_rt),
                                                   _ptr(F_w), _ptr(F_rho), _ptr(F_eta), _ptr(F_ex), _ptr(th_w3), _ptr(eta), _ptr(k2i)), "newton_residual")
        return F_w, F_rho, F_eta, F_ex, th_w3, eta, k2i

    def newton_update(self, d_w, d_rho, d_eta, d_exner, velz_i, rho_i, rt_i, exner_i, velz_j, rho_j, rt_j, exner_j):
        """mimsem_column_newton_update: velz_j / rho_j / rt_j / exner_j are updated IN PLACE; returns (velz_h, rho_h, rt_h, exner_h, norm_squares
        [8, nEl, nk n2e]) (VertSolve.cpp:1858-1912)"""
        nk = self.nk
        for a, sl, nm in ((d_w, nk - 1, "d_w"), (d_rho, nk, "d_rho"), (d_eta, nk, "d_eta"), (d_exner, nk, "d_exner"), (velz_i, nk - 1, "velz_i"),
                          (rho_i, nk, "rho_i"), (rt_i, nk, "rt_i"), (exner_i, nk, "exner_i"), (velz_j, nk - 1, "velz_j"), (rho_j, nk, "rho_j"),
                          (rt_j, nk, "rt_j"), (exner_j, nk, "exner_j")):
            self._col(a, sl, nm)
        mk = lambda n: torch.empty(self.nEl, n * self.n2e, dtype=torch.float64, device=self.device)
        velz_h, rho_h, rt_h, exner_h = mk(nk - 1), mk(nk), mk(nk), mk(nk)
        nrm = torch.empty(8, self.nEl, nk * self.n2e, dtype=torch.float64, device=self.device)
        check(self.L.mimsem_column_newton_update(self.ctx, _ptr(d_w), _ptr(d_rho), _ptr(d_eta), _ptr(d_exner), _ptr(velz_i), _ptr(rho_i), _ptr(rt_i),
                                                 _ptr(exner_i), _ptr(velz_j), _ptr(rho_j), _ptr(rt_j), _ptr(exner_j),
                                                 _ptr(velz_h), _ptr(rho_h), _ptr(rt_h), _ptr(exner_h), _ptr(nrm)), "newton_update")
        return velz_h, rho_h, rt_h, exner_h, nrm

    def max_norms(self, nrm):
        """mimsem_column_max_norms: VertSolve::MaxNorm for the four pairs of newton_update's norm squares -> device tensor [4] (exner, w, rho, eta)"""
        assert nrm.shape == (8, self.nEl, self.nk * self.n2e) and nrm.is_contiguous()
        ws = torch.empty(4, self.nEl, dtype=torch.float64, device=self.device)
        out = torch.empty(4, dtype=torch.float64, device=self.device)
        check(self.L.mimsem_column_max_norms(self.ctx, _ptr(nrm), _ptr(ws), _ptr(out)), "column_max_norms")
        return out

    def helmholtz_blocks(self, dt, theta, rho, eta, pi):
        for a, nm in ((theta, "theta"), (rho, "rho"), (eta, "eta"), (pi, "pi")):
            self._col(a, self.nk, nm)
        out = torch.empty(self.nEl, self.nk, 3, self.n2e, self.n2e, dtype=torch.float64, device=self.device)
        check(self.L.mimsem_column_helmholtz_blocks(self.ctx, dt, _ptr(theta), _ptr(rho), _ptr(eta), _ptr(pi), _ptr(out)), "helmholtz_blocks")
        return out

    def solve_schur_eta(self, dt, theta, rho, eta, pi, F_u, F_rho, F_eta, F_pi):
        """F_* are updated in place (as the reference does); returns d_u, d_rho, d_eta, d_pi"""
        nk = self.nk
        for a, sl, nm in ((theta, nk, "theta"), (rho, nk, "rho"), (eta, nk, "eta"), (pi, nk, "pi"),
                          (F_u, nk - 1, "F_u"), (F_rho, nk, "F_rho"), (F_eta, nk, "F_eta"), (F_pi, nk, "F_pi")):
            self._col(a, sl, nm)
        N, Nm = self.nk * self.n2e, (self.nk - 1) * self.n2e
        mk = lambda n: torch.empty(self.nEl, n, dtype=torch.float64, device=self.device)
        d_u, d_rho, d_eta, d_pi = mk(Nm), mk(N), mk(N), mk(N)
        check(self.L.mimsem_column_solve_schur_eta(self.ctx, dt, _ptr(theta), _ptr(rho), _ptr(eta), _ptr(pi),
                                                   _ptr(F_u), _ptr(F_rho), _ptr(F_eta), _ptr(F_pi),
                                                   _ptr(d_u), _ptr(d_rho), _ptr(d_eta), _ptr(d_pi)), "solve_schur_eta")
        return d_u, d_rho, d_eta, d_pi

    def solve_status(self):
        """(columns whose refinement did not converge in the last solve_schur_eta, or -1 when that path keeps no status; per-column status
        array: 0 converged, 1 not converged, 2 refinement off, 3 re-solved by the pivoted fallback, 4 flagged for its conditioning only and accepted on its backward error (set_pivot_fallback); per-column |last correction| / |solution|) -- mimsem_column_solve_status"""
        n = C.c_int(-1)
        st = np.zeros(max(self.nEl, 1), dtype=np.int32); ratio = np.zeros(max(self.nEl, 1))
        check(self.L.mimsem_column_solve_status(self.ctx, C.byref(n), st.ctypes.data, ratio.ctypes.data), "column_solve_status")
        return n.value, st[:self.nEl], ratio[:self.nEl]

    def flag_columns_for_test(self, columns):
        """test hook: the next column solve treats these columns as flagged by its block sweep (mimsem_column_flag_for_test)"""
        cols = np.ascontiguousarray(columns, dtype=np.int32)
        check(self.L.mimsem_column_flag_for_test(self.ctx, cols.ctypes.data, int(cols.size)), "column_flag_for_test")

    def set_pivot_fallback(self, on=True):
        """(on by default) solve_schur_eta / solve_schur_3 re-solve the columns their unpivoted sweep flags by an LU with partial pivoting over
        the band (what the reference's PCLU does for every column); verified re-solves report status 3; 0 switches it off --
        mimsem_column_set_pivot_fallback"""
        check(self.L.mimsem_column_set_pivot_fallback(self.ctx, int(on)), "column_set_pivot_fallback")      # (2: every column, validation mode)

    # ---- Krylov building blocks ----------------------------------------------------------------
    def mdot(self, V, w, k=None, out=None):
        """h[i] = <V[i], w> for i < k; V: [m, n] contiguous rows"""
        k = V.shape[0] if k is None else k
        h = out if out is not None else torch.empty(k, dtype=torch.float64, device=self.device)
        check(self.L.mimsem_krylov_mdot(self.ctx, k, w.numel(), _ptr(V), V.stride(0), _ptr(w), _ptr(h)), "krylov_mdot")
        return h

    def maxpy(self, V, h, w, alpha=1.0, k=None):
        """w += alpha * sum_{i<k} h[i] V[i]  (in place)"""
        k = V.shape[0] if k is None else k
        check(self.L.mimsem_krylov_maxpy(self.ctx, k, w.numel(), _ptr(V), V.stride(0), _ptr(h), alpha, _ptr(w)), "krylov_maxpy")
        return w

    def orthogonalize(self, V, w, h, k=None, alpha=-1.0):
        """one classical Gram-Schmidt pass in two launches: h[:k] = V[:k] w, then w += alpha V[:k]^T h (in place)"""
        k = V.shape[0] if k is None else k
        check(self.L.mimsem_krylov_orthogonalize(self.ctx, k, w.numel(), _ptr(V), V.stride(0), alpha, _ptr(w), _ptr(h)), "krylov_orthogonalize")
        return w

    def cgs2(self, V, w, v, k, h1, h2, col, norm_slot, flag=None):
        """both Gram-Schmidt passes of an Arnoldi step + normalisation + Hessenberg column in three launches (mimsem_krylov_cgs2)"""
        assert w.numel() == v.numel() and v.is_contiguous() and w.is_contiguous() and col.dtype == torch.float64 and col.is_contiguous()
        assert col.is_cuda or col.is_pinned(), "col must be device or pinned host memory"
        assert col.numel() > max(k - 1, norm_slot) and h1.numel() >= k and h2.numel() >= k
        check(self.L.mimsem_krylov_cgs2(self.ctx, k, w.numel(), _ptr(V), V.stride(0), _ptr(w), _ptr(v), _ptr(h1), _ptr(h2),
                                        col.data_ptr(), norm_slot, flag.data_ptr() if flag is not None else None), "krylov_cgs2")

    def reorthonormalize(self, V, w, v, k, h1, h2, col, norm_slot, fused=None, flag=None):
        """second Gram-Schmidt pass + normalisation in two launches: h2[:k] = V[:k] w; w -= V[:k]^T h2; v = w/|w|;
        col[:k] = h1 + h2; col[norm_slot] = |w| (col: device or pinned host tensor).  fused / flag given: the explicit form
        (mimsem_krylov_reorthonormalize_ex: the caller's own flag word, nothing shared through the context)"""
        assert w.numel() == v.numel() and v.is_contiguous() and col.dtype == torch.float64 and col.is_contiguous()
        assert col.is_cuda or col.is_pinned(), "col must be device or pinned host memory"
        assert col.numel() > max(k - 1, norm_slot)
        if fused is not None:
            assert flag is None or (flag.dtype == torch.int32 and (flag.is_cuda or flag.is_pinned()))
            check(self.L.mimsem_krylov_reorthonormalize_ex(self.ctx, k, w.numel(), _ptr(V), V.stride(0), _ptr(w), _ptr(v), _ptr(h1), _ptr(h2),
                                                           col.data_ptr(), norm_slot, 1 if fused else 0,
                                                           flag.data_ptr() if flag is not None else None), "krylov_reorthonormalize_ex")
            return
        check(self.L.mimsem_krylov_reorthonormalize(self.ctx, k, w.numel(), _ptr(V), V.stride(0), _ptr(w), _ptr(v), _ptr(h1), _ptr(h2),
                                                    col.data_ptr(), norm_slot), "krylov_reorthonormalize")
        return v

    def normalize(self, w, v, k, h1, h2, col, norm_slot):
        """v = w/|w|; col[:k] = h1 + h2; col[norm_slot] = |w|.  col: device tensor or PINNED host tensor (written by the kernel)"""
        assert w.numel() == v.numel() and v.is_contiguous() and col.dtype == torch.float64 and col.is_contiguous()
        assert col.is_cuda or col.is_pinned(), "col must be device or pinned host memory"
        assert col.numel() > max(k - 1, norm_slot)
        check(self.L.mimsem_krylov_normalize(self.ctx, w.numel(), _ptr(w), _ptr(v), k, _ptr(h1), _ptr(h2) if h2 is not None else None,
                                             col.data_ptr(), norm_slot), "krylov_normalize")
        return v

    def rowdot(self, A, B, out=None):
        """out[i] = <A[i], B[i]> for [nrows, n] tensors (rows contiguous)"""
        out = out if out is not None else torch.empty(A.shape[0], dtype=torch.float64, device=self.device)
        check(self.L.mimsem_krylov_rowdot(self.ctx, A.shape[0], A.shape[1], _ptr(A), A.stride(0), _ptr(B), B.stride(0), _ptr(out)), "rowdot")
        return out

    def rowdot_local(self, A, B, out=None, space=None):
        """the rank's part of rowdot (one rank: all of it); DistEngine weights by ownership and leaves the all-reduce to the caller"""
        return self.rowdot(A, B, out=out)

    def cg_update(self, num, den, p, Ap, x, r):
        """x += (num/den) p ; r -= (num/den) Ap  row-wise, in place"""
        check(self.L.mimsem_krylov_cg_update(self.ctx, p.shape[0], p.shape[1], _ptr(num), _ptr(den), _ptr(p), p.stride(0),
                                             _ptr(Ap), Ap.stride(0), _ptr(x), x.stride(0), _ptr(r), r.stride(0)), "cg_update")

    def chebyshev_start(self, c, s, theta, r, d, x):
        """r = s c ; d = r / theta ; x = 0 row-wise (one launch: mimsem_krylov_chebyshev_start); r may be c"""
        check(self.L.mimsem_krylov_chebyshev_start(self.ctx, x.shape[0], x.shape[1], float(s), float(theta), _ptr(c), c.stride(0), _ptr(r), r.stride(0),
                                                   _ptr(d), d.stride(0), _ptr(x), x.stride(0)), "chebyshev_start")

    def chebyshev_px(self, alpha, beta, y, p, x, b=None, dinv=None, upd=None):
        """z = dinv (b - y) (or y when dinv is None); p = z + beta p; x += alpha p; upd = z -- row-wise, one launch (mimsem_krylov_chebyshev_px: the
        vector algebra of a Chebyshev step on a sharded mesh, where the operator result is completed over the halo between the passes)"""
        st = lambda t: t.stride(0) if t is not None else 0
        check(self.L.mimsem_krylov_chebyshev_px(self.ctx, x.shape[0], x.shape[1], float(alpha), float(beta), _ptr(y), y.stride(0), _ptr(b), st(b), _ptr(dinv), st(dinv),
                                                _ptr(p), p.stride(0), _ptr(x), x.stride(0), _ptr(upd), st(upd)), "chebyshev_px")

    def axpy_dots(self, dx, x, out):
        """x += dx ; out[0] = dx . dx ; out[1] = x . x over ALL entries of the (contiguous) tensors (one launch: mimsem_krylov_axpy_dots)"""
        assert dx.is_contiguous() and x.is_contiguous() and dx.numel() == x.numel() and out.numel() == 2 and out.is_contiguous()
        check(self.L.mimsem_krylov_axpy_dots(self.ctx, x.numel(), _ptr(dx), _ptr(x), _ptr(out)), "axpy_dots")

    def chebyshev_update(self, a, b, Bd, x, r, d):
        """x += d ; r -= Bd ; d = a d + b r  row-wise, in place (one launch: mimsem_krylov_chebyshev_update)"""
        check(self.L.mimsem_krylov_chebyshev_update(self.ctx, x.shape[0], x.shape[1], float(a), float(b), _ptr(Bd), Bd.stride(0),
                                                    _ptr(x), x.stride(0), _ptr(r), r.stride(0), _ptr(d), d.stride(0)), "chebyshev_update")

    def cg_direction(self, num, den, z, p):
        """p = z + (num/den) p  row-wise, in place"""
        check(self.L.mimsem_krylov_cg_direction(self.ctx, p.shape[0], p.shape[1], _ptr(num), _ptr(den), _ptr(z), z.stride(0),
                                                _ptr(p), p.stride(0)), "cg_direction")

    def combine(self, a, alpha=1.0, op=None, b=None, beta=0.0, c=None, out=None):
        """out = alpha * (a, a*b or a/b) + beta * c, row by row ([nrows, n] tensors whose rows are contiguous; slices along the first
        dimension are fine); out may be a or c (mimsem_vec_combine)"""
        a2 = a if a.dim() == 2 else a.unsqueeze(0)
        out = torch.empty_like(a2) if out is None else out
        o2 = out if out.dim() == 2 else out.unsqueeze(0)
        opc = {None: 0, "mul": 1, "div": 2}[op]
        ts = [a2, o2] + ([b] if b is not None else []) + ([c] if c is not None else [])
        for t in ts:
            t2 = t if t.dim() == 2 else t.unsqueeze(0)
            if t2.shape != a2.shape or t2.stride(1) != 1 or t2.dtype != torch.float64 or t2.device != a2.device:
                raise _lib.MimsemError("combine: operands must be float64 [nrows, n] with contiguous rows on one device, got %s vs %s" % (tuple(t2.shape), tuple(a2.shape)))
        b2 = None if b is None else (b if b.dim() == 2 else b.unsqueeze(0))
        c2 = None if c is None else (c if c.dim() == 2 else c.unsqueeze(0))
        if opc and b2 is None:
            raise _lib.MimsemError("combine: op needs b")
        check(self.L.mimsem_vec_combine(self.ctx, a2.shape[0], a2.shape[1], float(alpha), a2.data_ptr(), a2.stride(0), opc,
                                        None if b2 is None else b2.data_ptr(), 0 if b2 is None else b2.stride(0), float(beta),
                                        None if c2 is None else c2.data_ptr(), 0 if c2 is None else c2.stride(0), o2.data_ptr(), o2.stride(0)), "vec_combine")
        return out

    def interface_average(self, a, nk):
        """[nk-1, n] interface field -> [nk, n] level field, 0.5*(k-1) + 0.5*(k) with the missing boundary interfaces left out"""
        out = torch.empty(nk, a.shape[1], dtype=torch.float64, device=a.device)
        check(self.L.mimsem_interface_average(self.ctx, nk, a.shape[1], a.data_ptr(), a.stride(0), out.data_ptr(), out.stride(0)), "interface_average")
        return out

    def block_inverse(self, blocks):
        """inverse of every [n, n] block of a [nblocks, n, n] tensor by the library's batched Gauss-Jordan (mimsem_block_inverse);
        returns a new tensor"""
        if blocks.dim() != 3 or blocks.shape[1] != blocks.shape[2] or blocks.dtype != torch.float64:
            raise _lib.MimsemError("block_inverse: [nblocks, n, n] float64 tensor required")
        out = blocks.contiguous().clone()
        check(self.L.mimsem_block_inverse(self.ctx, out.shape[0], out.shape[1], out.data_ptr()), "block_inverse")
        return out

    def block_inverse_status(self, blocks):
        """block_inverse plus the number of blocks for which the reference's LinAlg::Inv reports a (near-)singular pivot
        (mimsem_block_inverse_status; eul/LinAlg.cpp:243-246)"""
        if blocks.dim() != 3 or blocks.shape[1] != blocks.shape[2] or blocks.dtype != torch.float64:
            raise _lib.MimsemError("block_inverse: [nblocks, n, n] float64 tensor required")
        out = blocks.contiguous().clone()
        ns = C.c_int(0)
        check(self.L.mimsem_block_inverse_status(self.ctx, out.shape[0], out.shape[1], out.data_ptr(), C.byref(ns)), "block_inverse_status")
        return out, ns.value

    def norm(self, x):
        """2-norm of a whole (single-rank) vector by the library's two-stage row-dot; DistEngine overrides with the ownership-weighted,
        all-reduced version"""
        v = x.reshape(1, -1)
        if not v.is_contiguous():
            v = v.contiguous()
        return float(torch.sqrt(self.rowdot(v, v))[0])

    def complete(self, form, y):
        """single rank: results are already complete (DistEngine reduces the halo here)"""
        return y

    def wsum(self, form, t):
        """sum over the GLOBAL vector of one form (every DoF once); DistEngine weights by ownership and all-reduces"""
        return t.sum()

    def allreduce(self, t, op="sum"):
        """sum / max over the ranks of a DistEngine; the identity on one rank"""
        return t

    def space(self, key):
        """context manager naming the vector space of the inner products inside (0, 1, 2 or "uh" = packed [1-form, 2-form]);
        a no-op on one rank, the ownership weights on a DistEngine"""
        import contextlib
        return contextlib.nullcontext()

    # ---- halo pack / unpack ---------------------------------------------------------------------
    def halo_segments(self, idx, seg_off, s_begin, s_end, mode, buf, v):
        """all neighbours in one launch (mimsem_halo_segments): mode 0 pack, 1 insert, 2 add; seg_off: host int32 array"""
        v2 = v if v.dim() == 2 else v.unsqueeze(0)
        check(self.L.mimsem_halo_segments(self.ctx, _ptr(idx), len(seg_off) - 1, seg_off.ctypes.data, s_begin, s_end,
                                          v2.shape[0], mode, _ptr(buf), _ptr(v2), v2.stride(0)), "halo_segments")

    def halo_pack(self, idx, v):
        v2 = v if v.dim() == 2 else v.unsqueeze(0)
        buf = torch.empty(v2.shape[0], idx.numel(), dtype=torch.float64, device=self.device)
        check(self.L.mimsem_halo_pack(self.ctx, _ptr(idx), idx.numel(), v2.shape[0], _ptr(v2), v2.stride(0), _ptr(buf)), "halo_pack")
        return buf

    def halo_unpack(self, idx, buf, v, add):
        v2 = v if v.dim() == 2 else v.unsqueeze(0)
        check(self.L.mimsem_halo_unpack(self.ctx, _ptr(idx), idx.numel(), v2.shape[0], int(add), _ptr(buf), _ptr(v2), v2.stride(0)), "halo_unpack")
