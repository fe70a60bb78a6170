"""One rank's view of a sharded mesh: an Engine over the rank's patches plus the halo exchangers, exposing the same calls as
Engine but with COMPLETE results -- every 0/1-form output is reduced over the halo (owners add the ghosts' partial sums) and
sent back to the ghosts, so all copies of a shared DoF agree on all ranks (what the reference gets from MatMult on MPIAIJ
matrices followed by VecScatter gtol FORWARD).  Inner products count every global DoF once (ownership weights) and are summed
over the ranks.  SURVEY 8(e); used by the sharded SW step (SWEqn over a DistEngine) and testable on one GPU with several
processes (gloo transport staged through the host, tests/test_gpu_multiproc.py).

Cost model: one exchange pair per operator application -- on the BASELINE problem sizes this is latency bound and slower than
one GPU; the class exists for correctness at scale, not for speed on 1e5 unknowns."""
import torch
import torch.distributed as dist

from .device import Engine
from .partition import CHalo, HaloExchanger, build_plans


class RcclComm:
    """An ncclComm_t over the ranks of the default process group, made the way a C++ host would make it (ncclGetUniqueId on rank 0,
    the 128 bytes broadcast, ncclCommInitRank on every rank) through ctypes on the librccl this process has loaded -- PyTorch's,
    when the process group's backend is "nccl".  It is what mimsem_halo_set_rccl takes: the library's grouped ncclSend / ncclRecv
    then run on the plan's communication stream (xGMI), with no Python between pack, transport and unpack.
    One rank per DEVICE: RCCL refuses two ranks on one GPU (the one-GPU rehearsals use the callback transport instead)."""

    @staticmethod
    def find_library():
        """the librccl this process can use, or an exception -- the LOCAL precondition of the constructor, checked by every rank and
        agreed on (all_reduce MIN) before any rank enters the constructor's collectives"""
        import ctypes as C
        import os
        for name in ("librccl.so", "librccl.so.1"):
            try:
                return C.CDLL(name, mode=os.RTLD_NOLOAD | os.RTLD_NOW)       # the copy already in the process (torch's)
            except OSError:
                continue
        return C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))

    def __init__(self, device):
        import ctypes as C
        import os
        torch.cuda.set_device(0 if device is None else device)
        lib = None
        for name in ("librccl.so", "librccl.so.1"):
            try:
                lib = C.CDLL(name, mode=os.RTLD_NOLOAD | os.RTLD_NOW)       # the copy already in the process (torch's)
                break
            except OSError:
                continue
        if lib is None:
            lib = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))
        self.lib = lib

        class UID(C.Structure):
            _fields_ = [("internal", C.c_char * 128)]
        uid = UID()
        world, rank = dist.get_world_size(), dist.get_rank()
        lib.ncclGetUniqueId.restype = C.c_int; lib.ncclGetUniqueId.argtypes = [C.POINTER(UID)]
        lib.ncclCommInitRank.restype = C.c_int
        ok = 1
        if rank == 0 and lib.ncclGetUniqueId(C.byref(uid)) != 0:
            ok = 0
        raw = [bytes(uid) if ok else None] if rank == 0 else [None]
        # (a rank that failed BEFORE this point never reaches the broadcast: `agree` -- called by DistEngine on EVERY rank before the
        #  constructor runs its collectives -- is what keeps the ranks in step; here only rank 0's failure is left to report)
        dist.broadcast_object_list(raw, src=0)
        if raw[0] is None:
            raise RuntimeError("ncclGetUniqueId failed on rank 0")
        C.memmove(C.byref(uid), raw[0], 128)
        comm = C.c_void_p()
        lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UID, C.c_int]
        rc = lib.ncclCommInitRank(C.byref(comm), world, uid, rank)
        if rc != 0 or not comm.value:
            raise RuntimeError("ncclCommInitRank failed (%d)" % rc)
        self.comm = comm.value
        self.handle = lib._handle

    def close(self):
        import ctypes as C
        if self.comm:
            self.lib.ncclCommDestroy.argtypes = [C.c_void_p]
            self.lib.ncclCommDestroy(C.c_void_p(self.comm))
            self.comm = None


class DistEngine:
    def __init__(self, eng, sphere, world, rank, overlap=False, transport="dist", plans=None):
        """overlap=True: 1-form operator results are completed through the C ABI's halo plan (CHalo) with the INTERIOR / BOUNDARY
        split of the apply -- boundary element groups first, exchange started, interior groups while it travels
        (mimsem_op_apply_part + mimsem_halo_begin/_end), and every other completion (complete(), incidence, blocks) goes through the
        C ABI's plans too; transport: "dist" (host callback over the process group), an RcclComm / ncclComm_t, or "auto" = an RcclComm
        of its own when the process group runs on RCCL, else "dist"."""
        self.eng, self.world, self.rank, self.sphere = eng, world, rank, sphere
        dm = eng.mesh
        if plans is None:
            plans = build_plans(sphere, world, rank, dm.gid0, dm.gid1)
        self.halo = {0: HaloExchanger(plans[0], engine=eng), 1: HaloExchanger(plans[1], engine=eng)}
        self.chalo = None
        self.chalo0 = None
        self.rccl = None
        if overlap and world > 1:
            self.transport_note = None
            if transport == "auto":
                transport = "dist"
                if dist.get_backend() == "nccl":
                    # every rank must end up on the same transport: a rank that cannot make its communicator (library not found, ...)
                    # makes all of them fall back to the host-staged callback, and the record says so
                    ok, err = 1, None
                    # step 1: the local preconditions (library found) agreed on BEFORE anybody enters RcclComm's collectives -- a rank
                    # that failed alone would otherwise go straight to the all_reduce below while its peers sit in the broadcast
                    try:
                        RcclComm.find_library()
                    except Exception as ex:                              # noqa: BLE001 -- reported through transport_note
                        ok, err = 0, "%s: %s" % (type(ex).__name__, ex)
                    flag = torch.tensor([ok], dtype=torch.int32, device=eng.device)
                    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                    if int(flag.item()) == 1:
                        # step 2: every rank constructs (the same collectives on every rank), then the outcomes are agreed on
                        try:
                            idx = getattr(eng.device, "index", None)
                            self.rccl = RcclComm(torch.cuda.current_device() if idx is None else idx)
                        except Exception as ex:                          # noqa: BLE001
                            ok, err = 0, "%s: %s" % (type(ex).__name__, ex)
                        flag = torch.tensor([ok], dtype=torch.int32, device=eng.device)
                        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                    if int(flag.item()) == 1:
                        transport = self.rccl
                    else:
                        self.transport_note = "RCCL communicator not available on every rank (%s): host-staged callback transport" % (err or "another rank failed")
                        if self.rccl is not None:
                            self.rccl.close(); self.rccl = None
            if isinstance(transport, RcclComm):
                eng.L.mimsem_halo_use_rccl_library(transport.handle)     # (MIMSEM_ERR_STATE when resolved before: the same library then)
                transport = transport.comm
            self.transport = "rccl" if isinstance(transport, int) else transport
            self.chalo = CHalo(plans[1], eng, max_nlev=eng.nk, transport=transport)
            self.chalo0 = CHalo(plans[0], eng, max_nlev=eng.nk + 1, transport=transport)
            eng.set_halo_slots(1, self.chalo.shared)
        f = lambda m: torch.as_tensor(m, dtype=torch.float64, device=eng.device)
        self.own = {0: f(plans[0].owned), 1: f(plans[1].owned), 2: torch.ones(dm.n2, dtype=torch.float64, device=eng.device)}
        self._spaces = {0: self.own[0], 1: self.own[1], 2: self.own[2], "uh": torch.cat([self.own[1], self.own[2]])}
        self._cur = None
        self._host = None

    def __getattr__(self, name):                 # sizes, nk, nEl, n1e, mesh, tensor, zeros, element_matrices, cg_update, ... are local
        return getattr(self.eng, name)

    # ---- completion ---------------------------------------------------------------------------------------------------------
    def complete(self, form, y):
        if form == 1 and self.chalo is not None and y.shape[0] <= self.eng.nk:
            self.chalo.sum_all(y)                  # edges: one symmetric exchange through the C ABI (pack, transport, unpack in the library)
        elif form == 0 and self.chalo0 is not None and y.shape[0] <= self.eng.nk + 1:
            self.chalo0.reverse_add(y); self.chalo0.forward_insert(y)       # nodes shared by > 2 ranks: owner sums, then scatters
        elif form in (0, 1):
            self.halo[form].sum_all(y)
        return y

    def close(self):
        for h in (self.chalo, self.chalo0):
            if h is not None:
                h.close()
        self.chalo = self.chalo0 = None
        if self.rccl is not None:
            self.rccl.close(); self.rccl = None

    def allreduce(self, t, op="sum"):
        if self.world == 1:
            return t
        rop = dist.ReduceOp.SUM if op == "sum" else dist.ReduceOp.MAX
        if self._host is None:
            self._host = dist.get_backend() == "gloo"
        if self._host and t.is_cuda:
            h = t.cpu(); dist.all_reduce(h, op=rop); t.copy_(h)
        else:
            dist.all_reduce(t, op=rop)
        return t

    def space(self, key):
        """the vector space of the inner products inside the `with` block: 0, 1, 2 or "uh" (packed [1-form, 2-form])"""
        outer = self

        class _Space:
            def __enter__(self_s):
                self_s.prev = outer._cur; outer._cur = outer._spaces[key]

            def __exit__(self_s, *exc):
                outer._cur = self_s.prev
                return False
        return _Space()

    def _weight(self, n):
        if self._cur is None or self._cur.numel() != n:
            raise RuntimeError("DistEngine inner product outside a matching `with eng.space(...)` block (vector length %d)" % n)
        return self._cur

    # ---- operators with complete results ----------------------------------------------------------------------------------------
    def _finish(self, form, tmp, out, accum):
        self.complete(form, tmp)
        if out is None:
            return tmp
        o = out if out.dim() == 2 else out.unsqueeze(0)
        if accum:
            o += tmp
        else:
            o.copy_(tmp)
        return out

    def apply(self, op, x, f=None, lev0=0, scale=1.0, flags=0, alpha=1.0, out=None):
        form = Engine._SPACES[op][2]
        if form == 2:
            return self.eng.apply(op, x, f=f, lev0=lev0, scale=scale, flags=flags, alpha=alpha, out=out)
        x2 = x if x.dim() == 2 else x.unsqueeze(0)
        if self.chalo is not None and form == 1 and Engine._SPACES[op][0] == 1 and op in ("UMAT", "UHMAT", "ROTMAT", "UTMAT", "UTMAT_H"):
            # boundary groups -> exchange in flight -> interior groups -> unpack: the halo travels while the interior is computed
            direct = out is not None and not (flags & 2) and out.dim() == 2 and out.is_contiguous()
            tmp = out if direct else torch.empty(x2.shape[0], self.eng.sizes[1], dtype=torch.float64, device=x2.device)
            try:
                self.eng.apply_part(op, "boundary", x2, f=f, lev0=lev0, scale=scale, flags=flags & ~2, alpha=alpha, out=tmp)
                tok = self.chalo.begin("pair", tmp, True)
                self.eng.apply_part(op, "interior", x2, f=f, lev0=lev0, scale=scale, flags=flags & ~2, alpha=alpha, out=tmp)
            except Exception:
                # a failure between the two parts (transport, halo state) must not leave the BOUNDARY part pending: every later split
                # apply on this context would be refused with MIMSEM_ERR_STATE for good (the contract of mimsem_op_apply_part)
                self.eng.reset_parts()
                raise
            self.chalo.end(tok)
            if direct:
                return out
            if out is None:
                return tmp if x.dim() == 2 else tmp[0]
            o = out if out.dim() == 2 else out.unsqueeze(0)
            if flags & 2:
                o += tmp
            else:
                o.copy_(tmp)
            return out
        tmp = self.eng.apply(op, x2, f=f, lev0=lev0, scale=scale, flags=flags & ~2, alpha=alpha)
        r = self._finish(form, tmp, out, bool(flags & 2))
        return r if (out is not None or x.dim() == 2) else r[0]

    def apply_up(self, op, x, f, u, fac=None, dt=None, lev0=0, alpha=1.0, flags=0, out=None, scale=1.0, tau=None):
        form = Engine._SPACES[op][2]
        tmp = self.eng.apply_up(op, x if x.dim() == 2 else x.unsqueeze(0), f, u, fac=fac, dt=dt, lev0=lev0, alpha=alpha, flags=flags & ~2,
                                scale=scale, tau=tau)
        r = self._finish(form, tmp, out, bool(flags & 2))
        return r if (out is not None or x.dim() == 2) else r[0]

    def incidence(self, which, x):
        y = self.eng.incidence(which, x if x.dim() == 2 else x.unsqueeze(0))
        form = {"E10": 1, "E12": 1, "E01": 0, "E21": 2}[which]
        self.complete(form, y)
        return y if x.dim() == 2 else y[0]

    def pvec(self, lev0=0, nlev=1, scale=1.0, h2=None):
        return self.complete(0, self.eng.pvec(lev0, nlev, scale, h2=h2))

    def blocks_apply(self, form, blocks, x, transpose=False, alpha=1.0, accum=False, out=None, elem_scale=None):
        if form == 2:
            return self.eng.blocks_apply(form, blocks, x, transpose=transpose, alpha=alpha, accum=accum, out=out, elem_scale=elem_scale)
        tmp = self.eng.blocks_apply(form, blocks, x if x.dim() == 2 else x.unsqueeze(0), transpose=transpose, alpha=alpha, elem_scale=elem_scale)
        r = self._finish(form, tmp, out, accum)
        return r if (out is not None or x.dim() == 2) else r[0]

    # ---- inner products over the GLOBAL vector: every DoF counted once, summed over the ranks ----------------------------------
    def rowdot(self, A, B, out=None):
        w = self._weight(A.shape[1])
        r = self.eng.rowdot(A * w, B, out=out)
        return self.allreduce(r)

    def rowdot_local(self, A, B, out=None, space=None):
        """this rank's ownership-weighted part of rowdot, NOT reduced: the fixed-length solves log their check norms with it and the caller
        all-reduces the whole log once (SWEqn: once per Picard iteration)"""
        w = self._spaces[space] if space is not None else self._weight(A.shape[1])
        return self.eng.rowdot(A * w, B, out=out)

    def weights(self, space):
        return self._spaces[space]

    def randn_global(self, space, seed, cpu_generator=False):
        """this rank's entries of the standard-normal vector a single context over the whole mesh draws for `space` (0, 1, 2 or "uh") from
        `seed` -- set-up helper of the spectral estimates: all ranks, and the one-context run, then build the same Krylov space"""
        cs, dm = self.sphere, self.eng.mesh
        ng = {0: cs.nDofs0G, 1: cs.nDofs1G, 2: cs.nDofs2G}
        gid = {0: dm.gid0, 1: dm.gid1, 2: dm.gid2}
        forms = (1, 2) if space == "uh" else (space,)
        n = sum(ng[f] for f in forms)
        dev = self.eng.device
        if cpu_generator:
            g = torch.Generator(device="cpu"); g.manual_seed(seed)
            v = torch.randn((1, n), generator=g, dtype=torch.float64).to(dev)[0]
        else:
            g = torch.Generator(device=dev); g.manual_seed(seed)
            v = torch.randn(n, dtype=torch.float64, device=dev, generator=g)
        idx, off = [], 0
        for f in forms:
            idx.append(torch.as_tensor(gid[f], device=dev).long() + off); off += ng[f]
        return v[torch.cat(idx)].reshape(1, -1)

    # ---- the fused solver steps of Engine with the halo inside (same signatures): what a fixed-length Chebyshev iteration needs per step is the
    #      operator's and the preconditioner's 1-form (0-form) results completed -- exchanges, no inner product, no all-reduce -------------------
    def sw_operator_precond_chebyshev(self, a, grav, H, f0, blocks, ca, cb, x, r, d):
        """x += d; r -= P A d; d = ca d + cb r on the packed [u|h] rows: element pass + gather, EXCHANGE (edges), block pass + gather, EXCHANGE,
        one update launch -- 5 launches and 2 symmetric edge exchanges per step (one context: 3 launches, mimsem_sw_operator_precond_chebyshev)"""
        n1 = self.eng.sizes[1]
        y = self.eng.sw_operator(a, grav, H, f0, d)
        self.complete(1, y[:, :n1])
        z = self.eng.sw_blocks_apply(blocks, y)
        self.complete(1, z[:, :n1])
        self.eng.chebyshev_update(ca, cb, z, x, r, d)

    def block_chebyshev_sweep(self, op, blocks, x, b, p, alpha, beta, f=None, elem_scale=None, lev0=0, scale=1.0, flags=0, upd=None):
        """z = P (b - Op x); p = z + beta p; x += alpha p with both element-local sums completed over the halo (the operator pass split into
        boundary / interior groups around its exchange where the plan allows); blocks column-major per element as for Engine's sweep"""
        y = self.apply(op, x, f=f, lev0=lev0, scale=scale, flags=flags)
        torch.sub(b, y, out=y)
        z = self.blocks_apply(1, blocks, y, transpose=True, elem_scale=elem_scale)       # (column-major storage = the transposed read)
        self.eng.chebyshev_px(alpha, beta, z, p, x, upd=upd)                             # p = z + beta p; x += alpha p; upd = z: one launch
        return x

    def chebyshev_sweep(self, op, x, b, dinv, p, alpha, beta, f=None, u=None, tau=0.0, lev0=0, scale=1.0, flags=0, upd=None):
        """z = dinv (b - Op x); p = z + beta p; x += alpha p; Op's result completed over the halo (nodes: REVERSE/ADD + FORWARD/INSERT)"""
        if u is not None:
            y = self.apply_up(op, x, f, u, lev0=lev0, scale=scale, tau=tau, flags=flags)
        else:
            y = self.apply(op, x, f=f, lev0=lev0, scale=scale, flags=flags)
        self.eng.chebyshev_px(alpha, beta, y, p, x, b=b, dinv=dinv, upd=upd)             # z = dinv (b - y); p = z + beta p; x += alpha p; upd = z: one launch
        return x

    def mdot(self, V, w, k=None, out=None):
        h = self.eng.mdot(V, (w * self._weight(w.numel())).contiguous(), k=k, out=out)
        return self.allreduce(h)

    def wsum(self, form, t):
        return self.allreduce((t * self.own[form]).sum().reshape(1))[0]

    def norm(self, x):
        x2 = x.reshape(1, -1)
        return float(torch.sqrt(self.rowdot(x2, x2))[0])

    def gather_owned(self, form, v, gids, n_global):
        """global vector (on every rank) from the owned entries of each rank's local vector -- tests / output only"""
        g = torch.zeros(v.shape[0], n_global, dtype=v.dtype, device=v.device)
        idx = torch.as_tensor(gids, device=v.device).long()
        g[:, idx] = v * self.own[form]
        return self.allreduce(g)
