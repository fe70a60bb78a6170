"""Host-side mirror of the per-column residual assembly of the reference's VertSolve (eul/VertSolve.cpp:237-286,
432-502): compositions of VertOps operators and mat-vecs, here issued for ALL columns at once through the C ABI
(mimsem_colop_apply) on "vertical" device arrays [nEl][nslots*n2e] (L2Vecs::vz concatenated)."""
import os

import torch

SCALE = 1.0e8          # eul/VertOps.cpp:21
RAYLEIGH = 4.0 / 120.0  # eul/VertSolve.cpp:32
FLAG_VERT = 1


class VertSolve:
    def __init__(self, eng, dt, rayleigh=RAYLEIGH):
        self.eng, self.dt, self.rayleigh = eng, dt, rayleigh
        self.nk, self.n2e = eng.nk, eng.n2e
        self.k2i_z = 0.0
        self._blocks = {}
        # orders 1..4: residual assembly / update of the Newton loop through the fused entry points (mimsem_column_newton_*: four
        # launches per iteration instead of ~150 single-operator calls); MIMSEM_NEWTON_FUSED=0 keeps the composed form below
        self.fused = os.environ.get("MIMSEM_NEWTON_FUSED", "1") != "0"

    # thin wrappers: vo->AssembleX(ex,ey,...,M); MatMult(M, x, y) for every column
    _GEOMETRY_ONLY = ("CONST", "CONST_INV", "LINEAR", "LINEAR_INV", "RAYLEIGH")

    def _mv(self, colop, x, f1=None, f2=None, flags=0, rows=None, transpose=False):
        if colop in self._GEOMETRY_ONLY and f1 is None and f2 is None and flags == 0:
            # assembled once per VertSolve (the reference re-assembles before every MatMult; the blocks only depend on the mesh)
            if colop not in self._blocks:
                self._blocks[colop] = self.eng.colop_blocks(colop)
            return self.eng.colop_apply_blocks(colop, self._blocks[colop], x, rows, transpose=transpose)
        return self.eng.colop_apply(colop, x, f1=f1, f2=f2, flags=flags, transpose=transpose, nout_slots=rows)

    def V10(self, x):
        """MatMult(vo->V10, x, y): (nk x (nk-1)) -I/+I vertical divergence (eul/VertOps.cpp:134-163)"""
        return self.eng.column_incidence("V10", x.contiguous())

    def V01(self, x):
        """MatMult(vo->V01, x, y): V01 = -V10^T, vertical gradient across interfaces"""
        return self.eng.column_incidence("V01", x.contiguous())

    def diagnose_F_z(self, velz1, velz2, rho1, rho2):
        """eul/VertSolve.cpp:237-260"""
        nm = self.nk - 1
        t1 = self._mv("LINEAR_RT", velz1, f1=rho1, flags=FLAG_VERT, rows=nm)
        t2 = self._mv("LINEAR_RT", velz2, f1=rho1, flags=FLAG_VERT, rows=nm)
        F = (1.0 / 3.0) * self._mv("LINEAR_INV", t1, rows=nm) + (1.0 / 6.0) * self._mv("LINEAR_INV", t2, rows=nm)
        t1 = self._mv("LINEAR_RT", velz1, f1=rho2, flags=FLAG_VERT, rows=nm)
        t2 = self._mv("LINEAR_RT", velz2, f1=rho2, flags=FLAG_VERT, rows=nm)
        F += (1.0 / 6.0) * self._mv("LINEAR_INV", t1, rows=nm) + (1.0 / 3.0) * self._mv("LINEAR_INV", t2, rows=nm)
        return F

    def diagnose_Phi_z(self, velz1, velz2, zv):
        """eul/VertSolve.cpp:262-286"""
        nk = self.nk
        Phi = (1.0 / 6.0) * self._mv("CONLIN_W", velz1, f1=velz1, rows=nk)
        Phi += (1.0 / 6.0) * self._mv("CONLIN_W", velz2, f1=velz1, rows=nk)
        Phi += (1.0 / 6.0) * self._mv("CONLIN_W", velz2, f1=velz2, rows=nk)
        return Phi + zv

    def assemble_residual_ec(self, theta, Pi, velz1, velz2, rho1, rho2, zv):
        """eul/VertSolve.cpp:432-502 -> (fw, F, G, f_theta_corr); theta/Pi/rho on levels, velz on interfaces"""
        nk, nm, dt = self.nk, self.nk - 1, self.dt
        F = self.diagnose_F_z(velz1, velz2, rho1, rho2)
        Phi = self.diagnose_Phi_z(velz1, velz2, zv)
        fw = self._mv("LINEAR", velz2, rows=nm) - self._mv("LINEAR", velz1, rows=nm)
        fw += dt * self.V01(Phi)                                            # bernoulli function term
        tB = self._mv("CONST", Pi, rows=nk)
        tA2 = self._mv("LINEAR_INV", self.V01(tB), rows=nm)                 # pressure gradient
        tA1 = self._mv("LINEAR_RT", tA2, f1=theta, flags=FLAG_VERT, rows=nm)
        fw += 0.5 * dt * tA1
        self.k2i_z += float((F * tA1).sum()) / SCALE                        # kinetic to internal energy power
        G = self._mv("LINEAR_INV", self._mv("LINEAR_RT", F, f1=theta, flags=FLAG_VERT, rows=nm), rows=nm)
        if self.rayleigh:
            fw += 0.5 * dt * self.rayleigh * (self._mv("RAYLEIGH", velz2, rows=nm) + self._mv("RAYLEIGH", velz1, rows=nm))
        # additional terms to ensure conservation of entropy
        tA2 = self._mv("LINEAR_INV", self.V01(self._mv("CONST", theta, rows=nk)), rows=nm)   # theta gradient
        fw += 0.5 * dt * self.V01(self._mv("CONST_RHO", Pi, f1=theta, rows=nk))
        fw -= 0.5 * dt * self._mv("CONLIN_W", Pi, f1=tA2, rows=nm, transpose=True)
        f_theta_corr = 0.5 * dt * self._mv("CONST_RHO", self.V10(F), f1=theta, rows=nk)
        f_theta_corr += 0.5 * dt * self._mv("CONLIN_W", F, f1=tA2, rows=nk)
        return fw, F, G, f_theta_corr

    def init_gz(self, levs, gravity=9.80616):
        """VertSolve::initGZ (eul/VertSolve.cpp:89-175): zv_k = W^T diag(SCALE w_q / 2) (g z_k + g z_{k+1}), the weak-form geopotential
        of every level from the interface heights on the quadrature grid (Geom::levs [nk+1, nq]); returned in the vertical layout"""
        eng = self.eng
        gz = gravity * torch.as_tensor(levs, dtype=torch.float64, device=eng.device)
        zh = (0.5 * SCALE) * eng.apply("WTQ", (gz[:-1] + gz[1:]).contiguous())
        return eng.l2_horiz_to_vert(zh)

    def horiz_forcing_from(self, hs, velx1, velx2):
        """the horizontal transport tendencies the Newton loop adds (eul/VertSolve.cpp:1799): HorizSolve::advection_rhs_ec on the
        horizontal layout, transposed into the vertical one (L2Vecs::HorizToVert)"""
        eng, nk = self.eng, self.nk

        def forcing(rho_i, rho_j, theta_l2_h):
            dF, dG, _, _ = hs.advection_rhs_ec(velx1, velx2, eng.l2_vert_to_horiz(rho_i, nk), eng.l2_vert_to_horiz(rho_j, nk),
                                                eng.l2_vert_to_horiz(theta_l2_h, nk))
            return eng.l2_horiz_to_vert(dF), eng.l2_horiz_to_vert(dG)
        return forcing

    # ---- the vertical implicit solve: the caller of the column path ---------------------------------------
    def solve_schur_eta(self, velz_i, rho_i, rt_i, exner_i, zv, horiz_forcing=None, udwdx=None, hs_lat=None, maxit=20, tol=1.0e-12,
                        verbose=False):
        """VertSolve::solve_schur_eta (eul/VertSolve.cpp:1721-1973) for EVERY column at once: Newton iterations on (w, rho, eta, Pi)
        with solve_schur_column_eta as the linear solve, all state in the "vertical" layout [nEl][slots*n2e] (L2Vecs::vz).

        horiz_forcing(rho_i, rho_j, theta_l2_h) -> (dFx, dGx): the horizontal transport tendencies of advection_rhs_ec in the
        vertical layout ([nEl][nk*n2e]); None = no horizontal wind.  udwdx: optional [nEl][(nk-1)*n2e].  hs_lat: latitude of the
        quadrature points [nEl][mp12] switches the Held-Suarez temperature forcing on.
        Returns (velz, rho, rt, exner) at the new time level and leaves theta_h / theta_l2_h / exner_h (the time-centred fields
        the horizontal corrector reads) and the per-iteration max-norms in self.*"""
        eng, nk, dt = self.eng, self.nk, self.dt
        if self.fused and eng.mesh.n <= 4:
            return self._solve_schur_eta_fused(velz_i, rho_i, rt_i, exner_i, zv, horiz_forcing, udwdx, hs_lat, maxit, tol, verbose)
        mv = self._mv
        velz_j, rho_j, rt_j, exner_j = velz_i.clone(), rho_i.clone(), rt_i.clone(), exner_i.clone()
        theta_i = eng.diag_theta(1, rho_i, rt_i)                          # diagTheta2 :1766
        theta_h = theta_i.clone()
        theta_l2_i = eng.diag_theta(0, rho_i, rt_i)                       # diagTheta_L2 :1773
        theta_l2_h = theta_l2_i.clone()
        exner_h, velz_h, rho_h, rt_h = exner_i.clone(), velz_i.clone(), rho_i.clone(), rt_i.clone()
        self.history = []
        for itt in range(1, maxit + 1):
            self.k2i_z = 0.0
            dFx, dGx = horiz_forcing(rho_i, rho_j, theta_l2_h) if horiz_forcing is not None else (None, None)
            F_w, F_z, G_z, ftc = self.assemble_residual_ec(theta_l2_h, exner_h, velz_i, velz_j, rho_i, rho_j, zv)     # :1806
            if udwdx is not None:
                F_w = F_w + dt * udwdx
            F_exner = eng.column_eos(0, rt_j, exner_j)                    # Assemble_EOS_Residual :1810
            dF_z = rho_j + dt * self.V10(F_z) - rho_i                     # VecAYPX / VecAXPY :1815-1819
            dG_z = rt_j + 0.5 * dt * self.V10(G_z) - rt_i
            F_rho = mv("CONST", dF_z, rows=nk)
            F_rt = mv("CONST", dG_z, rows=nk) + ftc
            if dFx is not None:
                F_rho = F_rho + dt * dFx
                F_rt = F_rt + dt * dGx
            if hs_lat is not None:
                F_rt = F_rt + dt * eng.temp_forcing_hs(hs_lat, exner_h, theta_h, rho_h)                               # :1831-1834
            # entropy residual from the (rho theta) and rho residuals :1836-1842
            t1 = mv("CONST_RHO_INV", F_rt, f1=rt_h, rows=nk) - mv("CONST_RHO_INV", F_rho, f1=rho_h, rows=nk)
            F_eta = mv("CONST", t1, rows=nk)
            # theta_h in W3 and eta_h :1844-1851
            th_w3 = mv("CONST_RHO_INV", mv("CONST", rt_h, rows=nk), f1=rho_h, rows=nk)
            eta = mv("CONST_INV", eng.column_eos(2, th_w3, None), rows=nk)
            d_w, d_rho, d_eta, d_exner = eng.solve_schur_eta(dt, th_w3, rho_h, eta, exner_h, F_w, F_rho, F_eta, F_exner)   # :1855
            # theta_j (before the update) and eta_j :1858-1865
            th_w3 = mv("CONST_RHO_INV", mv("CONST", rt_j, rows=nk), f1=rho_j, rows=nk)
            eta = mv("CONST_INV", eng.column_eos(2, th_w3, d_eta), rows=nk)
            velz_j = velz_j + d_w; rho_j = rho_j + d_rho; exner_j = exner_j + d_exner
            rt_j = mv("CONST_INV", eng.column_eos(3, rho_j, eta), rows=nk)                                            # :1871-1872
            col = lambda dx, x: float((torch.linalg.vector_norm(dx, dim=1) / torch.linalg.vector_norm(x, dim=1)).max())   # MaxNorm :228
            nv = torch.tensor([col(d_exner, exner_j), col(d_w, velz_j), col(d_rho, rho_j), col(d_eta, eta)], dtype=torch.float64, device=eng.device)
            nv = eng.allreduce(nv, op="max").tolist()                    # MPI_Allreduce(MAX) :1915-1918
            norms = dict(exner=nv[0], w=nv[1], rho=nv[2], eta=nv[3])
            self.history.append(norms)
            exner_h = 0.5 * exner_i + 0.5 * exner_j; velz_h = 0.5 * velz_i + 0.5 * velz_j
            rho_h = 0.5 * rho_i + 0.5 * rho_j; rt_h = 0.5 * rt_i + 0.5 * rt_j
            theta_h = 0.5 * eng.diag_theta(1, rho_j, rt_j) + 0.5 * theta_i                                            # :1896-1903
            theta_l2_h = 0.5 * eng.diag_theta(0, rho_j, rt_j) + 0.5 * theta_l2_i                                      # :1905-1912
            if verbose:
                print("\t%d:\t|d_exner|/|exner|: %.6e\t|d_w|/|w|: %.6e\t|d_rho|/|rho|: %.6e\t|d_eta|/|eta|: %.6e"
                      % (itt, norms["exner"], norms["w"], norms["rho"], norms["eta"]))
            if norms["exner"] < tol and norms["rho"] < tol:               # :1922
                break
        self.theta_h, self.theta_l2_h, self.exner_h = theta_h, theta_l2_h, exner_h
        return velz_j, rho_j, rt_j, exner_j

    def _solve_schur_eta_fused(self, velz_i, rho_i, rt_i, exner_i, zv, horiz_forcing, udwdx, hs_lat, maxit, tol, verbose):
        """the same Newton loop on the fused entry points (orders 1..4): per iteration mimsem_column_newton_residual (2 launches),
        mimsem_column_solve_schur_eta (3), mimsem_column_newton_update (1), mimsem_column_diag_theta_blend (1) and one reduction of
        the norm partials; every statement of the composed loop above has its counterpart inside those kernels"""
        eng, dt = self.eng, self.dt
        velz_j, rho_j, rt_j, exner_j = velz_i.clone(), rho_i.clone(), rt_i.clone(), exner_i.clone()
        theta_i, theta_l2_i = eng.diag_theta_blend(rho_i, rt_i)                    # diagTheta2 :1766, diagTheta_L2 :1773
        theta_h, theta_l2_h = theta_i, theta_l2_i
        exner_h, velz_h, rho_h, rt_h = exner_i, velz_i, rho_i, rt_i
        self.history = []
        self._k2i = None
        for itt in range(1, maxit + 1):
            add_rho = add_rt = None
            if horiz_forcing is not None:
                add_rho, add_rt = horiz_forcing(rho_i, rho_j, theta_l2_h)
            if hs_lat is not None:
                hs = eng.temp_forcing_hs(hs_lat, exner_h, theta_h, rho_h)                                            # :1831-1834
                add_rt = hs if add_rt is None else add_rt + hs
            F_w, F_rho, F_eta, F_exner, th_w3, eta, k2i = eng.newton_residual(
                dt, self.rayleigh or 0.0, theta_l2_h, exner_h, velz_i, velz_j, rho_i, rho_j, zv, rt_i, rt_j, rho_h, rt_h, exner_j,
                add_w=udwdx, add_rho=add_rho, add_rt=add_rt)
            self._k2i = k2i
            if getattr(self, "keep_solve_args", False):     # (bench: the linear solve of this iteration again, alone, on the state it was given)
                self.last_solve_args = (dt, th_w3.clone(), rho_h.clone(), eta.clone(), exner_h.clone(), [F_w.clone(), F_rho.clone(), F_eta.clone(), F_exner.clone()])
            d_w, d_rho, d_eta, d_exner = eng.solve_schur_eta(dt, th_w3, rho_h, eta, exner_h, F_w, F_rho, F_eta, F_exner)   # :1855
            velz_h, rho_h, rt_h, exner_h, nrm = eng.newton_update(d_w, d_rho, d_eta, d_exner, velz_i, rho_i, rt_i, exner_i,
                                                                  velz_j, rho_j, rt_j, exner_j)
            if hasattr(eng, "max_norms"):
                nv = eng.max_norms(nrm)                                             # MaxNorm :228 for exner, w, rho, eta (two launches)
            else:
                cs = nrm.sum(dim=2)                                                 # [8, nEl]: column sums of squares
                nv = torch.sqrt(cs[0::2] / cs[1::2]).amax(dim=1)
            nv = eng.allreduce(nv, op="max")                                        # MPI_Allreduce(MAX) :1915-1918
            # (the theta diagnosis does not depend on the norms: launched BEFORE the host waits for them, it runs under the read-back)
            theta_h, theta_l2_h = eng.diag_theta_blend(rho_j, rt_j, blend2=theta_i, blendL=theta_l2_i, wa=0.5, wb=0.5)     # :1896-1912
            nv = nv.tolist()
            norms = dict(exner=nv[0], w=nv[1], rho=nv[2], eta=nv[3])
            self.history.append(norms)
            if verbose:
                print("\t%d:\t|d_exner|/|exner|: %.6e\t|d_w|/|w|: %.6e\t|d_rho|/|rho|: %.6e\t|d_eta|/|eta|: %.6e"
                      % (itt, norms["exner"], norms["w"], norms["rho"], norms["eta"]))
            if norms["exner"] < tol and norms["rho"] < tol:
                break
        self.k2i_z = float(self._k2i.sum()) / SCALE if self._k2i is not None else 0.0
        self.theta_h, self.theta_l2_h, self.exner_h = theta_h, theta_l2_h, exner_h
        return velz_j, rho_j, rt_j, exner_j
