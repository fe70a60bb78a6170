"""Device-resident Krylov solves for the mass-matrix systems that follow almost every operator application in
the reference (KSPSolve(ksp1, b, x): GMRES + block-Jacobi on M1, rtol 1e-16; eul/HorizSolve.cpp:77-96, 224, 246).
SURVEY 8(f) row N1 -- the first "next" row after the operator engine.

The mass matrices are symmetric positive definite, so a preconditioned CG converges to the same (unique) solution
the reference's GMRES does; all levels are solved at once (one independent system per level, per-level scalars kept
on the device: no host synchronisation inside the iteration).  Mat-vecs are the matrix-free engine applies."""
import math
import os

import torch

from ._switches import experiment

from .device import no_gc


class KSP:
    """mimsem_ksp_* of the C ABI (round 4, csrc/ksp.hip): the solve loops as library code -- the object a C++ host uses in place of
    PETSc's KSP (mimsem_amd/host/mimsem_shim.hpp: KSP).  Python callers reach the same loops through this wrapper: one code path.
    type "cg" (batched over the rows: one SPD system per level) | "gmres" (restarted, left-preconditioned, the rows as one vector)."""
    REASONS = {2: "rtol", 3: "atol", 4: "its", -3: "diverged_its", -5: "breakdown", -9: "nan_or_inf", 0: "none"}

    def __init__(self, eng, ksp_type="gmres"):
        import ctypes as C
        from .device import check
        self.eng = getattr(eng, "eng", eng)
        self._C, self._check = C, check
        h = C.c_void_p()
        check(self.eng.L.mimsem_ksp_create(self.eng.ctx, {"cg": 0, "gmres": 1}[ksp_type], C.byref(h)), "ksp_create")
        self.h = h
        self._keep = []

    def __del__(self):
        try:
            if getattr(self, "h", None) and self.eng.ctx:
                self.eng.L.mimsem_ksp_destroy(self.h)
        except Exception:
            pass
        self.h = None

    def set_operator(self, op, nlev, lev0=0, scale=1.0, flags=0, f=None):
        from .device import OPS, _ptr
        self._keep.append(f)
        self._check(self.eng.L.mimsem_ksp_set_operator(self.h, OPS[op], lev0, nlev, scale, flags, _ptr(f), f.stride(0) if f is not None else 0),
                    "ksp_set_operator")
        return self

    def set_operator_sw(self, nlev, a, grav, H, f0):
        from .device import _ptr
        self._keep.append(f0)
        self._check(self.eng.L.mimsem_ksp_set_operator_sw(self.h, nlev, a, grav, H, _ptr(f0), f0.stride(0) if f0.dim() > 1 and f0.shape[0] > 1 else 0),
                    "ksp_set_operator_sw")
        return self

    def set_pc(self, kind, blocks=None, elem_scale=None, dinv=None, form=1):
        from .device import _ptr
        L, h = self.eng.L, self.h
        self._keep += [blocks, elem_scale, dinv]
        if kind == "none":
            self._check(L.mimsem_ksp_set_pc_none(h), "ksp_set_pc_none")
        elif kind == "bjacobi":
            self._check(L.mimsem_ksp_set_pc_bjacobi(h), "ksp_set_pc_bjacobi")
        elif kind == "jacobi":
            self._check(L.mimsem_ksp_set_pc_jacobi(h, _ptr(dinv), dinv.stride(0) if dinv.dim() > 1 else 0), "ksp_set_pc_jacobi")
        elif kind == "elem_blocks":
            self._check(L.mimsem_ksp_set_pc_elem_blocks(h, form, _ptr(blocks), _ptr(elem_scale), elem_scale.stride(0) if elem_scale is not None else 0),
                        "ksp_set_pc_elem_blocks")
        elif kind == "sw_blocks":
            self._check(L.mimsem_ksp_set_pc_sw_blocks(h, _ptr(blocks)), "ksp_set_pc_sw_blocks")
        elif kind == "sw_bjacobi":
            self._check(L.mimsem_ksp_set_pc_sw_bjacobi(h), "ksp_set_pc_sw_bjacobi")
        else:
            raise ValueError(kind)
        return self

    def set_tolerances(self, rtol=1e-16, atol=1e-50, maxit=1000, restart=30, check_every=2):
        self._check(self.eng.L.mimsem_ksp_set_tolerances(self.h, rtol, atol, maxit, restart, check_every), "ksp_set_tolerances")
        return self

    def solve(self, b, x=None, guess_nonzero=False):
        from .device import _ptr
        C = self._C
        b2 = b if b.dim() == 2 else b.view(1, -1)
        assert b2.stride(1) == 1
        x2 = torch.zeros_like(b2) if x is None else (x if x.dim() == 2 else x.view(1, -1))
        self._check(self.eng.L.mimsem_ksp_set_initial_guess_nonzero(self.h, 1 if guess_nonzero else 0), "ksp_guess")
        self._check(self.eng.L.mimsem_ksp_solve(self.h, _ptr(b2), b2.stride(0), _ptr(x2), x2.stride(0)), "ksp_solve")
        its, rn, rs = C.c_int(), C.c_double(), C.c_int()
        self._check(self.eng.L.mimsem_ksp_get_info(self.h, C.byref(its), C.byref(rn), C.byref(rs)), "ksp_get_info")
        self.iterations, self.rnorm, self.reason = its.value, rn.value, self.REASONS.get(rs.value, rs.value)
        return x2.view(b.shape)

    def ritz(self, m=40):
        """mimsem_ksp_ritz: (min Re, max Re, max |Im|) of the Ritz values of P A after m Arnoldi steps inside the library"""
        C = self._C
        lo, hi, im = C.c_double(), C.c_double(), C.c_double()
        self._check(self.eng.L.mimsem_ksp_ritz(self.h, m, C.byref(lo), C.byref(hi), C.byref(im)), "ksp_ritz")
        return lo.value, hi.value, im.value

    def pc_blocks(self):
        """mimsem_ksp_get_pc_blocks: (device address of the blocks PCSetUp built, address of the per-(level, element) factors or None, rows of a block)"""
        C = self._C
        b, e, nd = C.c_void_p(), C.c_void_p(), C.c_int()
        self._check(self.eng.L.mimsem_ksp_get_pc_blocks(self.h, C.byref(b), C.byref(e), C.byref(nd)), "ksp_get_pc_blocks")
        return b.value, e.value, nd.value


def pcg(apply_A, b, minv=None, x0=None, rtol=1e-14, maxit=300, check_every=10, allreduce=None, precond=None):
    """Solve A x = b for a batch of systems (rows of b).  apply_A(x)->A x on [nlev, n] tensors.
    minv: elementwise preconditioner (Jacobi) of the same shape, or precond(r)->z (symmetric positive definite).
    allreduce(t): sums per-level scalars over ranks (multi-GPU: each rank holds ghost copies, the caller's dot
    weights handle ownership)."""
    if precond is not None:
        return _pcg_general(apply_A, b, precond, x0, rtol, maxit, check_every, allreduce)
    x = torch.zeros_like(b) if x0 is None else x0.clone()
    r = b - apply_A(x) if x0 is not None else b.clone()
    z = r * minv if minv is not None else r
    p = z.clone()
    dot = (lambda u, v: (u * v).sum(dim=1)) if allreduce is None else (lambda u, v: allreduce((u * v).sum(dim=1)))
    rz = dot(r, z)
    bnorm = torch.sqrt(dot(b, b)).clamp_min(1e-300)
    its = 0
    for it in range(maxit):
        Ap = apply_A(p)
        alpha = rz / dot(p, Ap).clamp_min(1e-300)
        x += alpha[:, None] * p
        r -= alpha[:, None] * Ap
        z = r * minv if minv is not None else r
        rz_new = dot(r, z)
        beta = rz_new / rz.clamp_min(1e-300)
        p = z + beta[:, None] * p
        rz = rz_new
        its = it + 1
        if its % check_every == 0:                    # the only host synchronisation
            if bool((torch.sqrt(dot(r, r)) / bnorm).max() < rtol):
                break
    return x, its


def pcg_engine(eng, apply_A, b, precond, rtol=1e-14, maxit=300, check_every=2, fixed_its=0):
    """PCG for a batch of SPD systems (rows of b) on the engine's fused vector kernels: rowdot (two-stage deterministic reduction),
    cg_update (x += a p, r -= a Ap) and cg_direction (p = z + b p) read their per-row scalars from device memory -- an
    iteration is operator + preconditioner + 6 small launches, no host synchronisation (fixed_its > 0: none at all, so the
    whole solve is hipGraph-capturable; otherwise a convergence test every `check_every` iterations)."""
    x = torch.zeros_like(b)
    r = b.clone()
    z = precond(r)
    p = z.clone()
    rz = eng.rowdot(r, z)
    rz_new = torch.empty_like(rz); pAp = torch.empty_like(rz); rr = torch.empty_like(rz)
    bnorm2 = None if fixed_its else eng.rowdot(b, b).clamp_min(1e-300)
    its = 0
    for it in range(fixed_its if fixed_its else maxit):
        Ap = apply_A(p)
        eng.rowdot(p, Ap, out=pAp)
        eng.cg_update(rz, pAp, p, Ap, x, r)
        its = it + 1
        if not fixed_its and its % check_every == 0:
            eng.rowdot(r, r, out=rr)
            if bool((rr / bnorm2).max() < rtol * rtol):
                break
        z = precond(r)
        eng.rowdot(r, z, out=rz_new)
        eng.cg_direction(rz_new, rz, z, p)
        rz, rz_new = rz_new, rz
    return x, its


def pcg_fixed(apply_A, b, precond, iterations):
    """PCG with a FIXED iteration count and no host synchronisation: capturable in a hipGraph.  For the element-block
    preconditioned mass systems (condition ~1.2, error contraction ~0.05 per iteration) 14 iterations reach round-off."""
    x = torch.zeros_like(b)
    r = b.clone()
    z = precond(r)
    p = z.clone()
    dot = lambda u, v: torch.linalg.vecdot(u, v, dim=1)
    rz = dot(r, z)
    tiny = 1e-300
    for _ in range(iterations):
        Ap = apply_A(p)
        alpha = rz / dot(p, Ap).clamp_min(tiny)
        torch.addcmul(x, alpha[:, None], p, out=x)
        torch.addcmul(r, alpha[:, None], Ap, value=-1.0, out=r)
        z = precond(r)
        rz_new = dot(r, z)
        p = torch.addcmul(z, (rz_new / rz.clamp_min(tiny))[:, None], p)
        rz = rz_new
    return x


def _pcg_general(apply_A, b, precond, x0, rtol, maxit, check_every, allreduce):
    x = torch.zeros_like(b) if x0 is None else x0.clone()
    r = b - apply_A(x) if x0 is not None else b.clone()
    z = precond(r)
    p = z.clone()
    vd = lambda u, v: torch.linalg.vecdot(u, v, dim=1)
    dot = vd if allreduce is None else (lambda u, v: allreduce(vd(u, v)))
    rz = dot(r, z)
    bnorm = torch.sqrt(dot(b, b)).clamp_min(1e-300)
    its = 0
    for it in range(maxit):
        Ap = apply_A(p)
        alpha = rz / dot(p, Ap).clamp_min(1e-300)
        torch.addcmul(x, alpha[:, None], p, out=x)
        torch.addcmul(r, alpha[:, None], Ap, value=-1.0, out=r)
        its = it + 1
        if its % check_every == 0 and bool((torch.sqrt(dot(r, r)) / bnorm).max() < rtol):
            break
        z = precond(r)
        rz_new = dot(r, z)
        p = torch.addcmul(z, (rz_new / rz.clamp_min(1e-300))[:, None], p)
        rz = rz_new
    return x, its


class MassSolver:
    """M1 u = b on all levels (the ksp1 solves, eul/HorizSolve.cpp:77-96: GMRES + PCBJACOBI with one block per element).
    Preconditioner: element blocks P^-1 = sum_e R_e^T D_e (M1_e)^-1 D_e R_e (D_e = 1/multiplicity of the edge), applied by
    mimsem_elem_blocks_apply -- cond(P^-1 M1) ~ 1.2; built once per thickness field, reused over time steps.
    precond="jacobi" keeps the diagonal scaling (cond ~ 3)."""

    def __init__(self, eng, scale=1.0e8, vert_scale=True, precond="blocks"):
        self.eng, self.scale, self.flags = eng, scale, 1 if vert_scale else 0
        n1e = eng.n1e
        dm = eng.mesh
        ix = torch.as_tensor(dm.inds1x, device=eng.device).long()
        iy = torch.as_tensor(dm.inds1y, device=eng.device).long()
        self.kind = precond
        self.fixed_its = 0          # > 0: run exactly that many PCG iterations (hipGraph capture)
        self._cheb = None
        self._cheb_checked = False
        # every fixed-length solve logs {|P r_last|^2, |P b|^2} per level into a slot of this device log (round 6: round 5 verified the first
        # solve only; a later, rougher right-hand side could lose accuracy unseen): verify() reads it once per evaluation / step
        self.MAXLOG = 16
        self._log = None
        self._slot = 0
        self.solves_checked = self.solves_missed = 0
        self.worst_check = 0.0
        # default solver for the block preconditioner on one rank: fixed-length Chebyshev on the fused sweep (no inner products)
        # (round 6: also over a halo -- a DistEngine has the sweep with its two exchanges inside, no inner product: no all-reduce in the solve)
        self.dist = hasattr(eng, "halo")
        self.chebyshev = (precond != "jacobi" and os.environ.get("MIMSEM_MASS_SOLVER", "chebyshev") == "chebyshev" and eng.mesh.n <= 5)
        if precond == "jacobi":
            diag = eng.zeros(eng.nk, dm.n1)
            for k in range(eng.nk):
                em = eng.element_matrices("UMAT", lev=k, scale=scale, flags=self.flags).view(eng.nEl, 4, n1e, n1e)
                diag[k].index_add_(0, ix.reshape(-1), torch.diagonal(em[:, 0], dim1=1, dim2=2).reshape(-1))
                diag[k].index_add_(0, iy.reshape(-1), torch.diagonal(em[:, 3], dim1=1, dim2=2).reshape(-1))
            self.minv = 1.0 / diag
        else:
            idx = torch.cat([ix, iy], dim=1)
            mult = torch.zeros(1, dm.n1, dtype=torch.float64, device=eng.device)
            mult[0].index_add_(0, idx.reshape(-1), torch.ones(idx.numel(), dtype=torch.float64, device=eng.device))
            mult = eng.complete(1, mult)[0]                  # sharded: sharers on other ranks count too
            d = 1.0 / mult[idx]
            # M1_e(k) = U^T diag(c_q thickInv_k(q)) U ~ tau_{k,e} * U^T diag(c_q) U with tau = the element's mean thickInv (exact when
            # the layer thickness is horizontally uniform over the element): ONE thickness-free inverse per element, resident in
            # LDS while the kernel sweeps the levels, times 1/tau per (level, element)
            em = eng.element_matrices("UMAT", lev=0, scale=scale, flags=0).view(eng.nEl, 2, 2, n1e, n1e)
            B = em.permute(0, 1, 3, 2, 4).reshape(eng.nEl, 2 * n1e, 2 * n1e)
            self.blocks = (d[:, :, None] * getattr(eng, "eng", eng).block_inverse(B) * d[:, None, :]).contiguous()
            tau = torch.as_tensor(dm.thickInv, device=eng.device).mean(dim=2) if vert_scale else \
                torch.ones(eng.nk, eng.nEl, dtype=torch.float64, device=eng.device)
            self.escale = (1.0 / tau).contiguous()

    def precond(self, r, lev0=0):
        return self.eng.blocks_apply(1, self.blocks, r, transpose=True, elem_scale=self.escale[lev0:lev0 + r.shape[0]])

    def _blocks_t(self):
        """the blocks as mimsem_ksp_set_pc_elem_blocks applies them (no transpose flag): B^T, contiguous -- the same products as
        precond()'s transposed read of B"""
        if getattr(self, "_bt", None) is None:
            self._bt = self.blocks.transpose(1, 2).contiguous()
        return self._bt

    def apply(self, x, lev0=0):
        return self.eng.apply("UMAT", x, lev0=lev0, scale=self.scale, flags=self.flags)

    def _chebyshev(self):
        """the fixed-length Chebyshev solver on the fused block sweep (single rank, 2 n1e <= 64 rows); spectral bounds over all levels"""
        if self._cheb is None:
            eng = self.eng
            if self.dist:
                # the rank's rows of the global random right-hand side (every level its own draw), ownership-weighted all-reduced inner products
                b = torch.cat([eng.randn_global(1, 1234 + k, cpu_generator=True) for k in range(eng.nk)], dim=0).contiguous()
                w1 = eng.weights(1)
                lmin, lmax, elo, ehi = lanczos_bounds(lambda v: self.apply(v, 0), lambda r: self.precond(r, 0), b, its=40, errors=True,
                                                      dot=lambda u, v: eng.allreduce(torch.linalg.vecdot(u * w1, v, dim=1)))
            else:
                g = torch.Generator(device="cpu"); g.manual_seed(1234)
                b = torch.randn(eng.nk, eng.sizes[1], generator=g, dtype=torch.float64).to(eng.device)
                lmin, lmax, elo, ehi = lanczos_bounds(lambda v: self.apply(v, 0), lambda r: self.precond(r, 0), b, its=40, errors=True)
            self.ritz_errors = (elo / lmin, ehi / lmax)
            # safety margins around the Ritz interval (round 6): as wide as the Ritz values are uncertain -- twice their residual bounds, at
            # least 1 %, at most the 10 % / 5 % of rounds 3-5.  On a smooth thickness field the extreme Ritz values of 40 steps are exact to
            # 1e-4 and the old margins cost 2 of 15 steps (profiles/r06_cheb_margin_probe.txt); every solve is still checked (verify()).
            mg = ritz_margins(lmin, lmax, min(0.10, 2.0 * elo / lmin), min(0.05, 2.0 * ehi / lmax))
            if experiment("MIMSEM_CHEB_MARGIN", ""):
                mg = tuple(float(v) for v in experiment("MIMSEM_CHEB_MARGIN", "").split(","))
            self.margin = mg
            self._cheb = ChebyshevMass(eng, None, lmin, lmax, rtol=1e-15, margin=mg)
            self._blocks_cm = self.blocks.transpose(1, 2).contiguous()
        return self._cheb

    def _logged_solve(self, ch, b):
        """the fixed-length solve with its check norms written to the device log (two extra stores in two sweeps + one two-row-block dot:
        recordable, no host synchronisation)"""
        nlev, n = b.shape
        if self._log is None or self._log.shape[1] != 2 * nlev:
            self._log = torch.zeros(self.MAXLOG, 2 * nlev, dtype=torch.float64, device=b.device)
            self._pair = torch.zeros(2 * nlev, n, dtype=torch.float64, device=b.device)
            self._slot = 0
        ch.upd = self._pair[:nlev]
        x = ch.solve(b, want_residual=True, pb=self._pair[nlev:])
        k = min(self._slot, self.MAXLOG - 1); self._slot += 1
        self.eng.rowdot_local(self._pair, self._pair, out=self._log[k], space=1)       # (sharded: this rank's ownership-weighted part; verify() reduces the log once)
        return x

    def verify(self, rtol=1e-14):
        """the checks of every fixed-length solve since the last call in ONE read: True = every level of every solve had its last
        preconditioned residual below 30 rtol |P b| (one more contraction lies between that residual and the result).  False: the
        Chebyshev mode is switched off (PCG from here on) and the caller redoes its evaluation.  Synchronises."""
        if self._log is None or self._slot == 0:
            return True
        if self.dist:
            self.eng.allreduce(self._log)                       # ONE all-reduce for every solve since the last call
        v = self._log.cpu().numpy()
        self._log.zero_(); self._slot = 0
        nlev = v.shape[1] // 2
        ok = True
        for row in v:
            r2, ref2 = row[:nlev], row[nlev:]
            if not (r2.any() or ref2.any()):
                continue
            import numpy as np
            with np.errstate(divide="ignore", invalid="ignore"):
                rel = np.where(ref2 > 0.0, np.sqrt(r2 / np.where(ref2 > 0.0, ref2, 1.0)), np.where(r2 > 0.0, np.inf, 0.0))
            self.solves_checked += 1
            worst = float(np.nanmax(rel)) if np.isfinite(rel).any() else float("inf")
            self.worst_check = max(self.worst_check, worst)
            if not bool((rel <= 30.0 * rtol).all()):
                ok = False; self.solves_missed += 1
        if not ok:
            self.chebyshev = False
        return ok

    def solve(self, b, lev0=0, rtol=1e-14, maxit=300):
        nlev = b.shape[0]
        if self.kind != "jacobi" and self.chebyshev:
            ch = self._chebyshev()
            es = self.escale[lev0:lev0 + nlev]
            ch.sweep = lambda x, rhs, p, al, be, upd: self.eng.block_chebyshev_sweep(
                "UMAT", self._blocks_cm, x, rhs, p, al, be, elem_scale=es, lev0=lev0, scale=self.scale, flags=self.flags, upd=upd)
            # one context: the whole solve as one call (round 6: no operator pass for the first step, no cleared vectors; MIMSEM_CHEB_WHOLE=0, experiments: a call per sweep)
            ch.whole = None
            if not self.dist and hasattr(self.eng, "block_chebyshev_solve") and self.eng.n1e <= 30 and experiment("MIMSEM_CHEB_WHOLE", "1") != "0":
                ch.whole = lambda rhs, coef, pb, upd: self.eng.block_chebyshev_solve(
                    "UMAT", self._blocks_cm, rhs, coef, elem_scale=es, lev0=lev0, scale=self.scale, flags=self.flags, pb=pb, upd=upd)
            # the calibrated step count holds for the tolerance and the level range it was calibrated on (advisor, round 3): a call that asks
            # for a tighter rtol, or covers levels the calibration did not see, goes back to the bound-based count and re-calibrates
            cal = getattr(self, "_cheb_cal", None)
            if cal is not None and not torch.cuda.is_current_stream_capturing() and \
                    (rtol < cal["rtol"] or lev0 < cal["lev0"] or lev0 + nlev > cal["lev0"] + cal["nlev"]):
                ch.set_steps(cal["bound_steps"])
                self._cheb_checked = False
                self._cheb_cal = None
            x = self._logged_solve(ch, b)
            if self.dist:
                return x, ch.steps           # (sharded: the bound-based count stands -- the one-time calibration below measures true residuals on the host; every solve is checked by verify())
            if not self._cheb_checked and not torch.cuda.is_current_stream_capturing():
                # one-time check of the spectral bounds on a real right-hand side: a step count derived from wrong bounds would
                # silently under-solve; fall back to PCG for good if the true residual is not at round-off
                bn = torch.linalg.vector_norm(b, dim=1).clamp_min(1e-300)
                true_res = lambda y: float((torch.linalg.vector_norm(b - self.apply(y, lev0), dim=1) / bn).max())
                res = true_res(x)
                self._cheb_checked = True
                if not res < 1e-11:
                    self.chebyshev = False
                    return self.solve(b, lev0, rtol, maxit)
                # ... and one-time CALIBRATION of the step count (round 3): the bound-based count (interval widened by 10 % / 5 %, 1e-15)
                # over-solves.  Take the smallest count whose TRUE residual -- on this right-hand side and on a random one, which
                # excites every eigenmode -- meets the tolerance the caller asked for (`rtol`, 1e-14: the stopping rule of the PCG
                # path of this class) or is within a factor 2 of the round-off floor, plus one step; MIMSEM_CHEB_CALIBRATE=0 keeps the bound.
                if os.environ.get("MIMSEM_CHEB_CALIBRATE", "1") != "0":
                    full = ch.steps
                    g = torch.Generator(device="cpu"); g.manual_seed(4321)
                    rnd = torch.randn(b.shape, generator=g, dtype=b.dtype).to(b.device) * bn[:, None] / math.sqrt(b.shape[1])
                    rn = torch.linalg.vector_norm(rnd, dim=1).clamp_min(1e-300)
                    rnd_res = lambda y: float((torch.linalg.vector_norm(rnd - self.apply(y, lev0), dim=1) / rn).max())
                    floor_rnd = rnd_res(ch.solve(rnd))             # a random right-hand side excites every eigenmode: the count must hold for it too
                    need = full
                    for k in range(max(4, full // 2), full):
                        ch.set_steps(k)
                        if true_res(ch.solve(b)) <= max(2.0 * res, rtol) and rnd_res(ch.solve(rnd)) <= max(2.0 * floor_rnd, rtol):
                            need = min(full, k + 1)
                            break
                    ch.set_steps(need)
                    self.cheb_calibration = {"bound_steps": full, "steps": need, "floor": res, "floor_random": floor_rnd}
                    self._cheb_cal = {"rtol": rtol, "lev0": lev0, "nlev": nlev, "bound_steps": full}
                    x = self._logged_solve(ch, b)
            return x, ch.steps
        if self.kind != "jacobi":
            if not hasattr(self.eng, "halo") and not self.fixed_its and os.environ.get("MIMSEM_PCG", "c") == "c":
                # the batched PCG loop of the C ABI (mimsem_ksp_*, csrc/ksp.hip) with this object's blocks: what a C++ host runs
                key = (lev0, nlev)
                if getattr(self, "_ksp_key", None) != key:
                    self._ksp = KSP(self.eng, "cg").set_operator("UMAT", nlev, lev0=lev0, scale=self.scale, flags=self.flags)
                    self._ksp.set_pc("elem_blocks", blocks=self._blocks_t(), elem_scale=self.escale[lev0:lev0 + nlev])
                    self._ksp_key = key
                self._ksp.set_tolerances(rtol=rtol, atol=1e-300, maxit=maxit, check_every=2)
                x = self._ksp.solve(b)
                return x, self._ksp.iterations
            with self.eng.space(1):
                return pcg_engine(self.eng, lambda v: self.apply(v, lev0), b, lambda r: self.precond(r, lev0), rtol=rtol, maxit=maxit,
                                  fixed_its=self.fixed_its)
        if self.kind == "jacobi":
            return pcg(lambda v: self.apply(v, lev0), b, minv=self.minv[lev0:lev0 + nlev], rtol=rtol, maxit=maxit)
        return pcg(lambda v: self.apply(v, lev0), b, precond=lambda r: self.precond(r, lev0), rtol=rtol, maxit=maxit, check_every=2)


def gmres(apply_A, b, precond=None, x0=None, rtol=1e-14, atol=1e-50, restart=30, maxit=1000, dot=None, eng=None):
    """Restarted, left-preconditioned GMRES for ONE (nonsymmetric) system on device tensors of any shape -- the stand-in for
    the reference's KSPGMRES solves on the packed [u,h] operator A and on the upwinded M0h (src/SWEqn_Picard.cpp:600-606, :348-353).
    As in PETSc's default the PRECONDITIONED residual is monitored.  Arnoldi by classical Gram-Schmidt with one
    re-orthogonalisation pass (two GEMVs on the device); the (restart+1) x restart Hessenberg matrix lives on the host:
    one host synchronisation per iteration.  dot(U, w): inner products of the rows of U [k, n] with w [n] -> [k]
    (multi-GPU callers pass an ownership-weighted, all-reduced version).  Returns (x, iterations, relative residual)."""
    shape = b.shape
    bf = b.reshape(-1)
    n = bf.numel()
    A = (lambda v: apply_A(v.view(shape)).reshape(-1))
    M = (lambda v: v) if precond is None else (lambda v: precond(v.view(shape)).reshape(-1))
    if dot is None:                                     # with an engine: its multi-dot (a DistEngine weights and all-reduces it)
        dot = (lambda U, w: U @ w) if eng is None else (lambda U, w: eng.mdot(U.contiguous(), w.contiguous(), k=U.shape[0]))
    nrm = lambda v: float(torch.sqrt(dot(v.view(1, -1), v))[0])
    x = torch.zeros_like(bf) if x0 is None else x0.reshape(-1).clone()
    pb = M(bf)
    bnorm = nrm(pb)
    if bnorm == 0.0:
        return x.view(shape), 0, 0.0
    tol = max(rtol * bnorm, atol)
    V = torch.empty(restart + 1, n, dtype=bf.dtype, device=bf.device)
    its, res = 0, bnorm
    while its < maxit:
        r = pb.clone() if (its == 0 and x0 is None) else M(bf - A(x))
        beta = nrm(r)
        res = beta
        if beta <= tol:
            break
        V[0] = r / beta
        H = [[0.0] * restart for _ in range(restart + 1)]
        cs, sn, g = [0.0] * restart, [0.0] * restart, [0.0] * (restart + 1)
        g[0] = beta
        k = 0
        for j in range(restart):
            w = M(A(V[j]))
            if eng is not None:                         # the engine's one-pass multi-dot / multi-axpy kernels
                w = w.contiguous() if not w.is_contiguous() else w
                h = eng.mdot(V, w, k=j + 1); eng.maxpy(V, h, w, alpha=-1.0, k=j + 1)
                h2 = eng.mdot(V, w, k=j + 1); eng.maxpy(V, h2, w, alpha=-1.0, k=j + 1)
            else:
                h = dot(V[:j + 1], w)
                w = w - h @ V[:j + 1]
                h2 = dot(V[:j + 1], w)                  # re-orthogonalisation
                w = w - h2 @ V[:j + 1]
            hn = torch.sqrt(dot(w.view(1, -1), w))
            col = torch.cat([h + h2, hn]).tolist()      # the host synchronisation of this iteration
            for i in range(j + 2):
                H[i][j] = col[i]
            for i in range(j):                          # previous Givens rotations
                t = cs[i] * H[i][j] + sn[i] * H[i + 1][j]
                H[i + 1][j] = -sn[i] * H[i][j] + cs[i] * H[i + 1][j]
                H[i][j] = t
            d = (H[j][j] ** 2 + H[j + 1][j] ** 2) ** 0.5
            cs[j], sn[j] = (1.0, 0.0) if d == 0.0 else (H[j][j] / d, H[j + 1][j] / d)
            H[j][j] = d; H[j + 1][j] = 0.0
            g[j + 1] = -sn[j] * g[j]; g[j] = cs[j] * g[j]
            its += 1; k = j + 1
            res = abs(g[j + 1])
            if res <= tol or its >= maxit or col[j + 1] == 0.0:
                break
            V[j + 1] = w / col[j + 1]
        y = [0.0] * k                                   # back substitution on the host
        for i in range(k - 1, -1, -1):
            s = g[i] - sum(H[i][l] * y[l] for l in range(i + 1, k))
            y[i] = s / H[i][i]
        x = x + torch.tensor(y, dtype=bf.dtype, device=bf.device) @ V[:k]
        if res <= tol:
            break
    return x.view(shape), its, res / bnorm


class GraphedGMRES:
    """The same restarted, left-preconditioned GMRES with the whole Arnoldi step j -- operator, preconditioner, two
    classical Gram-Schmidt passes against the j+1 basis vectors (the engine's mdot/maxpy kernels), normalisation, the copy
    of the new Hessenberg column to pinned host memory -- captured in a hipGraph (one per j, captured lazily, reused by
    every later solve with the same operator) and replayed: the inner loop is launch-bound (~20 small kernels per
    iteration on 1e5 unknowns), so one graph launch + one stream synchronisation per iteration replaces them.

    body(v) -> M^-1 A v must be capturable: fixed shapes, no host synchronisation, every kernel on the current stream
    (`eng.on_current_stream()` rebinds the engine's HIP stream for the capture)."""

    def __init__(self, eng, n, body, restart=30, dtype=torch.float64, body_orth=None):
        self.eng, self.n, self.m, self.body = eng, n, restart, body
        # body_orth(v, V, k, h, out) -> out = M^-1 A v already orthogonalised once against V[:k] (h[:k] = the projections): an operator that can
        # fold its own last pass into the first Gram-Schmidt pass (SWEqn: the 1-form gather rides in the dots, one launch less per step)
        self.body_orth = body_orth
        dev = eng.device
        self.V = torch.zeros(restart + 1, n, dtype=dtype, device=dev)
        self.h = torch.zeros(restart + 1, dtype=dtype, device=dev)
        self.h2 = torch.zeros(restart + 1, dtype=dtype, device=dev)
        self.col = torch.zeros(restart + 2, dtype=dtype, device=dev)
        self.wbuf = torch.zeros(n, dtype=dtype, device=dev) if body_orth is not None else None
        self.col_host = torch.zeros(restart, restart + 2, dtype=dtype).pin_memory()     # row j: Hessenberg column of step j
        self.graphs = [None] * restart
        self.pool = None
        self.lookahead = int(experiment("MIMSEM_GMRES_LOOKAHEAD", "8"))
        # the two-launch re-orthonormalisation takes its norm from w.w - h2.h2; the kernel raises this (pinned) word should that
        # ever cancel, and the cycle is then repeated on the three-launch form
        # (round 4: this object's OWN word and form, handed to every call -- rounds 2-3 registered the word with the context, where a
        # second GraphedGMRES on the same engine overwrote it)
        self.gs_flag = torch.zeros(1, dtype=torch.int32).pin_memory()
        self.gs_fused = experiment("MIMSEM_GS_FUSED_NORM", "1") != "0"
        # mimsem_krylov_cgs2 (round 4: the whole CGS2 step in three launches) is OPT-IN: measured SLOWER than the four launches it replaces
        # (config 3: 175 against 204 steps/s, config 2: 13.5 against 15.4) -- its middle kernel serialises k block reductions and the last
        # one then reduces n/512 partials per row instead of 32; kept with its parity test as the record (profiles/r04_sw_cgs2_ab.txt)
        self.cgs2 = experiment("MIMSEM_GS_CGS2", "0") == "1" and not hasattr(eng, "halo")      # (a DistEngine reduces its dots over the ranks)

    def _step(self, j):
        eng, V, k = self.eng, self.V, j + 1
        if self.body_orth is not None and not (self.gs_fused and self.cgs2):
            w = self.body_orth(V[j], V, k, self.h, self.wbuf)
            eng.reorthonormalize(V, w, V[j + 1], k, self.h, self.h2, self.col_host[j], self.m + 1, fused=self.gs_fused, flag=self.gs_flag)
            return
        w = self.body(V[j:j + 1]).reshape(-1)
        if not w.is_contiguous():
            w = w.contiguous()
        if self.gs_fused and self.cgs2:
            # round 4: both passes, normalisation and column in THREE launches (mimsem_krylov_cgs2); the column goes straight to pinned memory
            eng.cgs2(V, w, V[j + 1], k, self.h, self.h2, self.col_host[j], self.m + 1, flag=self.gs_flag)
            return
        eng.orthogonalize(V, w, self.h, k=k)
        # re-orthogonalisation + normalisation in two (three) launches
        eng.reorthonormalize(V, w, V[j + 1], k, self.h, self.h2, self.col_host[j], self.m + 1, fused=self.gs_fused, flag=self.gs_flag)

    def _graph(self, j):
        if self.graphs[j] is None:
            dev = self.eng.device
            keep = self.V[j + 1].clone()
            s = torch.cuda.Stream(device=dev)
            s.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(s), self.eng.on_current_stream():
                self._step(j)                           # warm-up: engine / allocator workspaces get their final size here
            torch.cuda.current_stream(dev).wait_stream(s)
            torch.cuda.synchronize(dev)
            g = torch.cuda.CUDAGraph()
            with no_gc(), torch.cuda.graph(g, pool=self.pool), self.eng.on_current_stream():
                self._step(j)
            if self.pool is None:
                self.pool = g.pool()
            torch.cuda.synchronize(dev)
            self.V[j + 1].copy_(keep)
            self.graphs[j] = g
        return self.graphs[j]

    def solve(self, apply_A, b, precond, x0=None, rtol=1e-14, atol=1e-50, maxit=1000):
        """apply_A / precond are used for the (un-graphed) residual at each restart; the graphed steps do the iterations"""
        shape = b.shape
        bf = b.reshape(-1)
        m = self.m
        dev = self.eng.device
        x = torch.zeros_like(bf) if x0 is None else x0.reshape(-1).clone()
        pb = precond(b).reshape(-1)
        bnorm = float(torch.linalg.vector_norm(pb))
        if bnorm == 0.0:
            return x.view(shape), 0, 0.0
        tol = max(rtol * bnorm, atol)
        its, res = 0, bnorm
        while its < maxit:
            r = pb.clone() if (its == 0 and x0 is None) else precond(b - apply_A(x.view(shape))).reshape(-1)
            beta = float(torch.linalg.vector_norm(r))
            res = beta
            if beta <= tol:
                break
            self.V[0] = r / beta
            H = [[0.0] * m for _ in range(m + 1)]
            cs, sn, g = [0.0] * m, [0.0] * m, [0.0] * (m + 1)
            g[0] = beta
            k = 0
            j0, done = 0, False
            hist = [beta]
            while j0 < m and not done:
                # Arnoldi steps j0 .. j0+s-1 are queued back to back and the host looks at their Hessenberg columns afterwards:
                # the device never waits for the host's Givens rotations.  s follows the observed convergence rate so that at
                # most about one step is computed beyond the one that meets the tolerance (its column is then ignored).
                s = 1
                if self.lookahead > 1 and len(hist) >= 3 and 0.0 < hist[-1] < hist[-3]:
                    rate = (hist[-1] / hist[-3]) ** 0.5
                    s = int(math.log(max(tol, 1e-300) / hist[-1]) / math.log(rate)) if hist[-1] > tol else 1
                s = max(1, min(s, self.lookahead, m - j0, maxit - its))
                for j in range(j0, j0 + s):
                    self._graph(j).replay()
                torch.cuda.current_stream(dev).synchronize()
                if int(self.gs_flag[0]) != 0:                    # (never seen on the operators of this repository)
                    self.gs_fused = False
                    self.gs_flag[0] = 0
                    self.graphs = [None] * m                      # re-capture with the three-launch form
                    k = 0; done = False
                    break                                         # restart the cycle from the current x
                for j in range(j0, j0 + s):
                    col = self.col_host[j].tolist()
                    for i in range(j + 1):
                        H[i][j] = col[i]
                    H[j + 1][j] = col[m + 1]
                    for i in range(j):
                        t = cs[i] * H[i][j] + sn[i] * H[i + 1][j]
                        H[i + 1][j] = -sn[i] * H[i][j] + cs[i] * H[i + 1][j]
                        H[i][j] = t
                    d = (H[j][j] ** 2 + H[j + 1][j] ** 2) ** 0.5
                    cs[j], sn[j] = (1.0, 0.0) if d == 0.0 else (H[j][j] / d, H[j + 1][j] / d)
                    H[j][j] = d; H[j + 1][j] = 0.0
                    g[j + 1] = -sn[j] * g[j]; g[j] = cs[j] * g[j]
                    its += 1; k = j + 1
                    res = abs(g[j + 1])
                    hist.append(res)
                    if res <= tol or its >= maxit or col[m + 1] == 0.0:
                        done = True
                        break
                j0 += s
            y = [0.0] * k
            for i in range(k - 1, -1, -1):
                sacc = g[i] - sum(H[i][l] * y[l] for l in range(i + 1, k))
                y[i] = sacc / H[i][i]
            if k > 0:
                self.eng.maxpy(self.V, torch.tensor(y, dtype=bf.dtype, device=bf.device), x, alpha=1.0, k=k)
            if res <= tol:
                break
        # capture ahead: the look-ahead replays run up to `lookahead` steps past the one that converges, and the count moves by a
        # step or two from solve to solve -- a capture (tens of ms) must not land in somebody's time step later
        for j in range(min(m, k + self.lookahead + 2)):
            if self.graphs[j] is None:
                self._graph(j)
        return x.view(shape), its, res / bnorm


class GraphedRichardson:
    """x <- x + P^-1 (b - A x) with `chunk` iterations per hipGraph replay, for systems whose preconditioned operator is a small
    perturbation of the identity: the 1-form mass matrix under its element-block preconditioner (spectrum of P^-1 M1 within
    ~[0.9, 1.1]) and the upwinded lumped 0-form mass under its diagonal (GMRES needs 8 iterations).  No inner products and no
    host synchronisation inside a chunk: 3-6 launches per iteration against 10 for PCG / 9 for a GMRES step, and the norm of
    the last update IS the preconditioned residual |P^-1 (b - A x)| that the reference's KSP monitors, so the stopping rule is
    unchanged (one scalar read per replay).  solve() returns None when the iteration does not contract (the caller then falls
    back to its Krylov solver).

    update(x, b) -> P^-1 (b - A x) must be capturable: fixed shapes, engine calls / torch ops only, coefficients read from
    fixed buffers."""

    def __init__(self, eng, shape, update=None, chunk=8, dtype=torch.float64, sweep=None):
        """update(x, b) -> P^-1 (b - A x)   or   sweep(x, b, upd): x += P^-1 (b - A x) in place, the update stored in upd when
        upd is not None (the engine's fused mimsem_*_richardson_sweep entry points)"""
        assert (update is None) != (sweep is None)
        self.eng, self.update, self.sweep, self.chunk = eng, update, sweep, chunk
        self.upd = torch.zeros(shape, dtype=dtype, device=eng.device) if sweep is not None else None
        self.x = torch.zeros(shape, dtype=dtype, device=eng.device)
        self.b = torch.zeros(shape, dtype=dtype, device=eng.device)
        self.dn = torch.zeros(1, dtype=dtype, device=eng.device)
        self.graphs = {}                 # sweeps per replay -> captured graph

    def _chunk(self, n):
        if self.sweep is not None:
            for i in range(n):
                self.sweep(self.x, self.b, self.upd if i == n - 1 else None)
            d = self.upd
        else:
            for _ in range(n):
                d = self.update(self.x, self.b)
                self.x.add_(d)
        d1 = d.reshape(1, -1)
        self.eng.rowdot(d1, d1, out=self.dn)

    def _graph(self, n):
        if n not in self.graphs:
            keep = self.x.clone()
            self.graphs[n], _ = self.eng.capture(lambda: self._chunk(n))
            self.x.copy_(keep)
        return self.graphs[n]

    def solve(self, b, precond, rtol=1e-14, max_sweeps=200, x0=None):
        """x0: initial guess (e.g. the solution of the previous, nearby system); the tolerance stays relative to |P^-1 b|.
        One graph launch and one scalar read per `chunk` sweeps (an adaptive first-replay length was measured: the extra graph
        captures cost more than the saved sweeps)."""
        self.b.copy_(b)
        self.x.copy_(precond(b))
        bnorm = float(torch.linalg.vector_norm(self.x))
        if bnorm == 0.0:
            return self.x.clone(), 0
        if x0 is not None:
            self.x.copy_(x0)
        done, prev, n = 0, None, self.chunk
        while done < max_sweeps:
            self._graph(n).replay()
            done += n
            dn = float(self.dn.item()) ** 0.5
            if dn <= rtol * bnorm:
                return self.x.clone(), done
            if not (dn == dn) or (prev is not None and dn > 0.9 * prev):     # NaN or not contracting
                return None
            prev = dn
        return None


def arnoldi_ritz(body, n, m, device, seed=1, eng=None, space=None, earlier=None):
    """Ritz values of B = P A from m Arnoldi steps on a random start vector (set-up time: host algebra on the small Hessenberg).
    eng + space (sharded meshes, DistEngine): the start vector is the rank's part of the GLOBAL random vector a single context would draw
    (same seed, same generator -> the same Krylov space), inner products are ownership-weighted and all-reduced -- every rank arrives at
    the same Hessenberg matrix, hence at the same spectral interval and the same fixed step counts.
    earlier = m0 < m: returns (values of m steps, values of the first m0 steps) -- how far the ends still move tells how well they are known."""
    import numpy as np
    dist = eng is not None and hasattr(eng, "halo")
    if dist:
        v = eng.randn_global(space, seed).reshape(-1)
        wgt = eng.weights(space)
        dots = lambda Vj, w: eng.allreduce((Vj * wgt) @ w)
        nrm = lambda w: float(torch.sqrt(eng.allreduce(((w * wgt) @ w).reshape(1)))[0])
        assert v.numel() == n
    else:
        gen = torch.Generator(device=device); gen.manual_seed(seed)
        v = torch.randn(n, dtype=torch.float64, device=device, generator=gen)
        dots = lambda Vj, w: Vj @ w
        nrm = lambda w: float(torch.linalg.vector_norm(w))
    V = torch.zeros(m + 1, n, dtype=torch.float64, device=device); H = np.zeros((m + 1, m))
    V[0] = v / nrm(v)
    k = m
    for j in range(m):
        w = body(V[j:j + 1].contiguous()).reshape(-1)
        for _ in range(2):
            hh = dots(V[:j + 1], w); w = w - hh @ V[:j + 1]; H[:j + 1, j] += hh.cpu().numpy()
        H[j + 1, j] = nrm(w)
        if H[j + 1, j] < 1e-14 * abs(H[0, 0]):
            k = j + 1
            break
        V[j + 1] = w / H[j + 1, j]
    if earlier is not None:
        m0 = min(earlier, k)
        return np.linalg.eigvals(H[:k, :k]), np.linalg.eigvals(H[:m0, :m0])
    return np.linalg.eigvals(H[:k, :k])


def ritz_margins(lo, hi, err_lo, err_hi, widen=1.0):
    """safety margins (factors for the lower / upper end) around a Ritz interval: as wide as the ends are uncertain (err_*: relative; residual
    bounds or the movement between two estimates, already scaled by the caller), at least 1 %; a re-estimate after a missed check (widen > 1)
    opens them by 10 % / 5 % per unit as rounds 5's fixed margins did for widen = 1"""
    return (1.0 - min(0.4, max(0.01, err_lo, 0.1 * (widen - 1.0))), 1.0 + max(0.01, err_hi, 0.05 * (widen - 1.0)))


def chebyshev_ellipse_coefs(d, c2, steps):
    """(alpha_k, beta_k) of the Chebyshev iteration p_k = z_k + beta_k p_{k-1}, x += alpha_k p_k for a spectrum inside an ellipse with centre d
    and foci d +- c (Manteuffel 1977); c2 = c^2 may be NEGATIVE (foci d +- i |c|: a spectrum stretched along the imaginary direction, e.g. an
    advective operator) -- the recurrence stays real.  c2 > 0 with a real interval [d - c, d + c] gives ChebyshevMass' coefficients."""
    al = 1.0 / d
    coef = [(al, 0.0)]
    for k in range(1, steps):
        be = (0.5 if k == 1 else 0.25) * c2 * al * al
        al = 1.0 / (d - be / al)
        coef.append((al, be))
    return coef


def chebyshev_ellipse_rate(d, a_re, a_im):
    """asymptotic convergence factor for an ellipse with centre d > 0 and semi-axes a_re (real direction), a_im (imaginary direction)"""
    import math
    c2 = a_re * a_re - a_im * a_im
    return (a_re + a_im) / (d + math.sqrt(max(d * d - c2, 0.0)))


class GraphedChebyshev:
    """B x = c, B = P A with a spectrum that is (close to) a real interval [lmin, lmax]: the Chebyshev semi-iteration (Saad, Alg. 12.1)
    with a FIXED number of steps -- known in advance from the interval and the tolerance -- captured with the preconditioning of the
    right-hand side and the final norms in ONE hipGraph: a solve is one replay and one read of two scalars.  Per step: body(d) = P A d
    (three launches for the shallow-water operator) and one fused vector update (mimsem_krylov_chebyshev_update); no inner products, no
    host round trip -- against eight launches and a synchronisation per Arnoldi step of the GMRES it replaces.  solve() returns None when
    the recurrence residual misses the tolerance (the caller falls back to its Krylov solver from the iterate reached)."""

    def __init__(self, eng, shape, body, precond, lmin, lmax, rtol=1e-14, margin=(0.9, 1.05), dtype=torch.float64, step=None, space="uh"):
        """step(ca, cb, x, r, d) (optional): the whole step x += d; r -= B d; d = ca d + cb r as ONE fused engine call (the shallow-water
        operator has one: mimsem_sw_operator_precond_chebyshev, three launches) instead of body(d) + the update kernel (four).
        space: the vector space of the two check norms (a DistEngine weights them by ownership; they stay LOCAL partial sums -- _run() holds no
        all-reduce, the caller reduces its whole check vector once)"""
        import math
        self.eng, self.body, self.precond, self.step, self.space = eng, body, precond, step, space
        self.lmin, self.lmax = margin[0] * lmin, margin[1] * lmax
        self.theta, self.delta = 0.5 * (self.lmax + self.lmin), 0.5 * (self.lmax - self.lmin)
        kap = self.lmax / self.lmin
        q = (math.sqrt(kap) - 1.0) / (math.sqrt(kap) + 1.0)
        self.rate = q
        self.steps = max(2, int(math.ceil(math.log(0.5 * rtol) / math.log(q))) + 1)
        self.rtol = rtol
        self.b = torch.zeros(shape, dtype=dtype, device=eng.device)       # the raw right-hand side (un-preconditioned)
        self.x = torch.zeros(shape, dtype=dtype, device=eng.device)
        self.r = torch.zeros(shape, dtype=dtype, device=eng.device)
        self.d = torch.zeros(shape, dtype=dtype, device=eng.device)
        self.nrm = torch.zeros(2, dtype=dtype, device=eng.device)         # |r|^2, |P b|^2
        self.graph = None
        self.graph_steps = 0

    def _run(self, neg_of=None, nrm=None):
        """neg_of = f: solve B x = -P f without forming -f (the Picard iteration's A dx = -f); nrm: where the two check norms go (default self.nrm)"""
        sigma1 = self.theta / self.delta
        rho = 1.0 / sigma1
        nrm = self.nrm if nrm is None else nrm
        loc = getattr(self.eng, "eng", self.eng)              # (element-wise kernels need no halo: a DistEngine's local engine)
        c = self.precond(self.b if neg_of is None else neg_of)
        if hasattr(loc, "chebyshev_start"):
            loc.chebyshev_start(c, 1.0 if neg_of is None else -1.0, self.theta, self.r, self.d, self.x)      # r = +-c; d = r / theta; x = 0: one launch
        else:
            if neg_of is not None:
                c = -c
            self.r.copy_(c)
            self.x.zero_()
            torch.mul(c, 1.0 / self.theta, out=self.d)
        self.eng.rowdot_local(c.reshape(1, -1), c.reshape(1, -1), out=nrm[1:2], space=self.space)
        for _ in range(self.steps):
            rho_new = 1.0 / (2.0 * sigma1 - rho)
            if self.step is not None:
                self.step(rho_new * rho, 2.0 * rho_new / self.delta, self.x, self.r, self.d)
            else:
                Bd = self.body(self.d)
                self.eng.chebyshev_update(rho_new * rho, 2.0 * rho_new / self.delta, Bd, self.x, self.r, self.d)     # x += d; r -= B d; d = rho' rho d + (2 rho'/delta) r
            rho = rho_new
        self.eng.rowdot_local(self.r.reshape(1, -1), self.r.reshape(1, -1), out=nrm[0:1], space=self.space)

    def solve(self, b):
        if hasattr(self.eng, "halo"):                      # sharded: the same launches eagerly (the exchanges are not recorded), ONE all-reduce of the two norms
            self.b.copy_(b)
            self._run()
            self.eng.allreduce(self.nrm)
        else:
            if self.graph is None or self.graph_steps != self.steps:
                self.b.copy_(b)
                self.graph, _ = self.eng.capture(self._run)
                self.graph_steps = self.steps
            self.b.copy_(b)
            self.graph.replay()
        r2, c2 = self.nrm.tolist()
        if c2 == 0.0:
            return self.x.clone(), self.steps, 0.0
        rel = (r2 / c2) ** 0.5
        if not (rel <= self.rtol):                        # (NaN fails too)
            if rel == rel and rel < 1e3 * self.rtol and self.steps < 200:
                import math
                self.steps += max(1, int(math.ceil(math.log(self.rtol / rel) / math.log(self.rate))) + 1)     # the bound was a little short: longer next time
            return None if not (rel == rel) else (self.x.clone(), -self.steps, rel)
        return self.x.clone(), self.steps, rel


def lanczos_bounds(apply_A, precond, b, its=25, dot=None, errors=False):
    """Extreme eigenvalues of P A (A SPD, P SPD) for every row system of b, from the Lanczos tridiagonal that `its` steps of
    preconditioned CG generate (T_kk = 1/a_k + b_{k-1}/a_{k-1}, T_{k,k+1} = sqrt(b_k)/a_k): Ritz values converge to the ends of
    the spectrum first.  Returns (lmin, lmax) over all rows.  Setup-time helper (host synchronisation per step).
    errors: also the residual bounds of the two extreme Ritz values, |beta_k s_ki| (s_ki: last component of the Ritz vector in the Lanczos
    basis) -- an eigenvalue of P A lies within that distance of each; (lmin, lmax, err_min, err_max), the errors the largest over the rows."""
    import numpy as np
    x = torch.zeros_like(b); r = b.clone(); z = precond(r); p = z.clone()
    if dot is None:                                          # (sharded meshes pass the ownership-weighted, all-reduced row dot)
        dot = lambda u, v: torch.linalg.vecdot(u, v, dim=1)
    rz = dot(r, z)
    al, be = [], []
    for _ in range(its):
        Ap = apply_A(p)
        a = rz / dot(p, Ap)
        x = x + a[:, None] * p; r = r - a[:, None] * Ap
        z = precond(r); rz_new = dot(r, z)
        bt = rz_new / rz
        al.append(a.cpu().numpy()); be.append(bt.cpu().numpy())
        if float(rz_new.abs().max()) < 1e-28 * float(rz.abs().max() + 1e-300):
            break
        p = z + bt[:, None] * p; rz = rz_new
    al, be = np.array(al), np.array(be)                     # [k, rows]
    k = al.shape[0]
    lo, hi = np.inf, 0.0
    elo = ehi = 0.0
    for row in range(al.shape[1]):
        T = np.zeros((k, k))
        for j in range(k):
            T[j, j] = 1.0 / al[j, row] + (be[j - 1, row] / al[j - 1, row] if j > 0 else 0.0)
            if j + 1 < k:
                T[j, j + 1] = T[j + 1, j] = np.sqrt(max(be[j, row], 0.0)) / al[j, row]
        ev, S = np.linalg.eigh(T)
        lo, hi = min(lo, ev[0]), max(hi, ev[-1])
        bk = np.sqrt(max(be[k - 1, row], 0.0)) / al[k - 1, row]           # the off-diagonal entry the next step would add
        elo, ehi = max(elo, abs(bk * S[k - 1, 0])), max(ehi, abs(bk * S[k - 1, -1]))
    if errors:
        return float(lo), float(hi), float(elo), float(ehi)
    return float(lo), float(hi)


class ChebyshevMass:
    """M1 x = b (all rows at once) by a FIXED-length Chebyshev semi-iteration on the engine's fused block sweep
    (mimsem_block_chebyshev_sweep: element pass, block pass fed by the on-the-fly gathered residual, gather pass applying
    p = z + beta p, x += alpha p).  The spectrum of P M1 is that of a fixed mesh: its ends are estimated once (Lanczos on 25 CG
    steps, widened by 5 % / 10 %), after which the number of steps for a tolerance and every alpha, beta are known in advance --
    no inner products, no host synchronisation, 3 launches per step against 10 per PCG iteration, and the whole solve can sit
    inside a captured hipGraph.  The norm of the last preconditioned residual is available for a check (`last`)."""

    def __init__(self, eng, sweep, lmin, lmax, rtol=1e-14, margin=(0.90, 1.05)):
        """sweep(x, b, p, alpha, beta, upd): one fused step (closure over op, blocks, elem_scale, lev0, scale, flags)"""
        self.eng, self.sweep = eng, sweep
        self.lmin, self.lmax = margin[0] * lmin, margin[1] * lmax
        kappa = self.lmax / self.lmin
        sg = (math.sqrt(kappa) - 1.0) / (math.sqrt(kappa) + 1.0)
        self.set_steps(max(2, int(math.ceil(math.log(2.0 / rtol) / math.log(1.0 / sg)))))
        self.p = None
        self.upd = None
        self.whole = None          # whole(b, coef, pb, upd) -> x: the solve from x = 0 as ONE engine call (Engine.block_chebyshev_solve: no operator pass in the first step, nothing cleared)

    def set_steps(self, steps):
        """fix the number of steps (the coefficients of step k depend on the spectral interval only, not on how many steps follow)"""
        self.steps = steps
        th, de = 0.5 * (self.lmax + self.lmin), 0.5 * (self.lmax - self.lmin)
        # x_{k+1} = x_k + c1_k z_k + c2_k (x_k - x_{k-1}) written as q_k = z_k + beta_k q_{k-1}, x += alpha_k q_k
        s1 = th / de
        rho_prev = 1.0 / s1
        c1_prev = 1.0 / th
        self.coef = [(c1_prev, 0.0)]
        for _ in range(1, self.steps):
            rho = 1.0 / (2.0 * s1 - rho_prev)
            c1, c2 = 2.0 * rho / de, rho * rho_prev
            self.coef.append((c1, c2 * c1_prev / c1))
            rho_prev, c1_prev = rho, c1

    def solve(self, b, x0=None, want_residual=False, pb=None):
        """returns x; capturable (fixed shapes and step count, no host synchronisation).  want_residual: self.upd receives the preconditioned
        residual the LAST sweep saw; pb (a tensor like b, zero start only): receives the FIRST sweep's update, which is P b -- the two vectors
        of a convergence check without an extra preconditioner application"""
        assert pb is None or x0 is None
        if self.upd is None or self.upd.shape != b.shape:
            self.upd = torch.zeros_like(b)
        if self.whole is not None and x0 is None:
            return self.whole(b, self.coef, pb, self.upd if want_residual else None)
        if self.p is None or self.p.shape != b.shape:
            self.p = torch.zeros_like(b)
        x = torch.zeros_like(b) if x0 is None else x0.clone()           # (p needs no reset: the first step has beta = 0)
        for k, (alpha, beta) in enumerate(self.coef):
            upd = self.upd if (want_residual and k == self.steps - 1) else (pb if k == 0 else None)
            self.sweep(x, b, self.p, alpha, beta, upd)
        return x
