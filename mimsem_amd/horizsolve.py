"""Host-side mirror of the reference's HorizSolve (eul/HorizSolve.cpp): the weak-form differential operators (grad, curl,
laplacian :208-283) and the right-hand sides of the horizontal dynamics (diagnose_fluxes :285-327, advection_rhs_ec :380-417,
diagnose_Phi :419-470, diagnose_q :472-493, momentum_rhs_ec :637-786) composed from engine applies, incidence stencils and
the device mass solves -- EVERY LEVEL IN ONE CALL of each operator (the reference loops `for(kk...)` around per-level
assemble + MatMult + KSPSolve).  SURVEY 8(f) row N2.  Single GPU, global numbering (local == global vectors).

Field layout: horizontal, one row per level: 1-forms [nk, n1], 2-forms [nk, n2], 0-forms [nk, n0]; interface quantities
(velz, dudz, dwdx, Fz) [nk-1, .]."""
import math

import torch

from .krylov import MassSolver

SCALE = 1.0e8
OMEGA = 7.29212e-5             # eul/HorizSolve.cpp:23
RAD_EARTH = 6371220.0
VERT, ACCUM = 1, 2


class HorizSolve:
    def __init__(self, eng, del2=None, quad_coords=None, do_visc=True):
        self.eng, self.nk, self.do_visc = eng, eng.nk, do_visc
        if del2 is None:                                        # viscosity() :112-120, with the GLOBAL node count nDofs0G
            n0g = float(eng.wsum(0, torch.ones(eng.sizes[0], dtype=torch.float64, device=eng.device)))
            dx = math.sqrt(4.0 * math.pi * RAD_EARTH * RAD_EARTH / n0g)
            del2 = -math.sqrt(0.072 * dx ** 3.2)
        self.del2 = del2
        self.m1 = MassSolver(eng, SCALE, True)
        self.m0 = eng.pvec(0, eng.nk, SCALE)                  # M0 is diagonal for the collocated 0-forms (Pvec)
        self.fg = None
        self.k2i_dev = None
        if quad_coords is not None:
            self.coriolis(quad_coords)

    @property
    def k2i(self):
        """horizontal kinetic-to-internal energy exchange of the last momentum_rhs_ec (:697-701)"""
        return 0.0 if self.k2i_dev is None else float(self.k2i_dev)

    def verify(self):
        """every fixed-length 1-form mass solve since the last call met its check (MassSolver.verify: one read of a small device log); False:
        the solver has switched itself to PCG -- redo the evaluation"""
        return self.m1.verify()

    # ---- operators --------------------------------------------------------------------------------------------------
    def _ap(self, op, x, f=None, flags=0, alpha=1.0, out=None):
        return self.eng.apply(op, x, f=f, lev0=0, scale=SCALE, flags=flags, alpha=alpha, out=out)

    def coriolis(self, quad_coords):
        """:124-161: fg[k] = M0(k, scale 1)^-1 PtQ (2 Omega sin(lat))"""
        xq = torch.as_tensor(quad_coords, dtype=torch.float64, device=self.eng.device)
        fq = (2.0 * OMEGA * torch.sin(torch.asin(xq[:, 2] / RAD_EARTH))).unsqueeze(0)
        b = self.eng.apply("PTQ", fq)
        self.fg = b / self.eng.pvec(0, self.nk, 1.0)

    def grad(self, phi):
        """u = M1^-1 E12 M2 phi   (:208-228), phi: [nk, n2]"""
        rhs = self.eng.incidence("E12", self._ap("WMAT", phi, flags=VERT))
        u, self.last_its = self.m1.solve(rhs)
        return u

    def curl(self, u, fg=None):
        """w = M0^-1 E01 M1 u (+ f)   (:233-254), u: [nk, n1]"""
        t = self.eng.incidence("E01", self.m1.apply(u))
        return self.eng.combine(t, 1.0, "div", self.m0, beta=0.0 if fg is None else 1.0, c=fg, out=t)

    def laplacian(self, u):
        """del2 * (grad(E21 u) + E10 curl(u))   (:256-283)"""
        ddu = self.grad(self.eng.incidence("E21", u))
        return self.eng.combine(ddu, self.del2, beta=self.del2, c=self.eng.incidence("E10", self.curl(u)), out=ddu)

    def _uvec_hu4(self, ua, ub, ha, hb):
        """the four m1->assemble_hu(level, SCALE, u, h, false, fac) calls + gtol_1 reverse-add (:300-305, :675-682)"""
        hu = self._ap("UHMAT", ua, f=ha, flags=VERT, alpha=1.0 / 3.0)
        self._ap("UHMAT", ua, f=hb, flags=VERT | ACCUM, alpha=1.0 / 6.0, out=hu)
        self._ap("UHMAT", ub, f=ha, flags=VERT | ACCUM, alpha=1.0 / 6.0, out=hu)
        self._ap("UHMAT", ub, f=hb, flags=VERT | ACCUM, alpha=1.0 / 3.0, out=hu)
        return hu

    # ---- fluxes and the transport right-hand side ---------------------------------------------------------------------
    def diagnose_fluxes(self, u1, u2, h1, h2, theta):
        """:285-327 (theta_in_Wt = false): F = M1^-1 (hu), G = M1^-1 F(theta) F   -- all levels"""
        F, _ = self.m1.solve(self._uvec_hu4(u1, u2, h1, h2))
        G, _ = self.m1.solve(self._ap("UHMAT", F, f=theta, flags=VERT))
        return F, G

    def advection_rhs_ec(self, u1, u2, h1, h2, theta):
        """:380-417 ; returns dF, dG in the horizontal layout (the caller's HorizToVert is mimsem_l2_transpose) and Fk, Gk"""
        eng = self.eng
        Fk, Gk = self.diagnose_fluxes(u1, u2, h1, h2, theta)
        dFk = eng.incidence("E21", Fk)
        dF = self._ap("WMAT", dFk, flags=VERT)
        # (sums of operator results are accumulated by the operator kernels themselves: MIMSEM_FLAG_ACCUM + alpha, no elementwise passes)
        dG = self._ap("WMAT", eng.incidence("E21", Gk), flags=VERT, alpha=0.5)
        self._ap("WHMAT", dFk, f=theta, flags=VERT | ACCUM, alpha=0.5, out=dG)
        self.dTheta = self.grad(theta)                                           # (kept: momentum_rhs_ec of the same stage needs the same gradient)
        self._ap("WTQUMAT", Fk, f=self.dTheta, flags=ACCUM, out=dG)             # K incl. its 0.5 factor
        self.Fk, self.Gk = Fk, Gk
        return dF, dG, Fk, Gk

    # ---- momentum right-hand side ---------------------------------------------------------------------------------------
    def _to_levels(self, a):
        """0.5*(interface k-1) + 0.5*(interface k) with the missing boundary interfaces left out (:451-459)"""
        return self.eng.interface_average(a, self.nk)

    def diagnose_Phi(self, u1, u2, velz1, velz2):
        """:419-470"""
        Phi = self._ap("WTQUMAT", u1, f=u1, alpha=1.0 / 3.0)
        self._ap("WTQUMAT", u2, f=u1, flags=ACCUM, alpha=1.0 / 3.0, out=Phi)
        self._ap("WTQUMAT", u2, f=u2, flags=ACCUM, alpha=1.0 / 3.0, out=Phi)
        z1, z2 = self._to_levels(velz1), self._to_levels(velz2)
        self._ap("WHMAT", z1, f=z1, flags=ACCUM, alpha=1.0 / 6.0, out=Phi)
        self._ap("WHMAT", z2, f=z1, flags=ACCUM, alpha=1.0 / 6.0, out=Phi)
        self._ap("WHMAT", z2, f=z2, flags=ACCUM, alpha=1.0 / 6.0, out=Phi)
        return Phi

    def diagnose_q(self, rho, u):
        """:472-493: (M0h(rho)) q = E01 M1 u + M0 f ; M0h is diagonal"""
        eng = self.eng
        rhs = eng.incidence("E01", self.m1.apply(u))
        eng.combine(self.m0, 1.0, "mul", self.fg.expand_as(self.m0) if self.fg.shape != self.m0.shape else self.fg, beta=1.0, c=rhs, out=rhs)
        return eng.combine(rhs, 1.0, "div", eng.pvec(0, self.nk, SCALE, h2=rho), out=rhs)

    def momentum_rhs_ec(self, theta, dudz1, dudz2, velz1, velz2, Pi, velx1, velx2, rho1, rho2, Fx=None, Fz=None,
                        dwdx1=None, dwdx2=None, Fk=None, dTheta=None):
        """:637-786 for every level at once; returns fu [nk, n1]; self.k2i = the kinetic-to-internal exchange (needs Fk).
        dTheta (optional, like Fx): grad(theta) when the caller has it already -- advection_rhs_ec of the same stage solved the same system
        for the same theta (:403 and :659 in the reference, two KSPSolves with one answer); passing self.dTheta saves one of the seven
        1-form mass solves of a right-hand-side evaluation"""
        eng = self.eng
        Phi = self.diagnose_Phi(velx1, velx2, velz1, velz2)
        dPi = self.grad(Pi)
        if dTheta is None:
            dTheta = self.grad(theta)
        fu = eng.incidence("E12", Phi)
        uh = eng.combine(velx1, 0.5, beta=0.5, c=velx2)
        q = self.diagnose_q(eng.combine(rho1, 0.5, beta=0.5, c=rho2), uh)
        if Fx is None:
            Fx, _ = self.m1.solve(self._uvec_hu4(velx1, velx2, rho1, rho2))
        self._ap("ROTMAT", Fx, f=q, flags=ACCUM, out=fu)
        self._ap("UHMAT", dPi, f=theta, flags=VERT | ACCUM, alpha=0.5, out=fu)  # pressure gradient force
        self._ap("UHMAT", dTheta, f=Pi, flags=VERT | ACCUM, alpha=-0.5, out=fu)
        dp = eng.incidence("E12", self._ap("WHMAT", theta, f=Pi, flags=VERT))
        eng.combine(dp, 0.5, beta=1.0, c=fu, out=fu)
        if Fk is not None:
            self.k2i_dev = self.eng.wsum(1, eng.combine(Fk, 1.0, "mul", dp)) / SCALE   # stays on the device (no host sync: hipGraph-capturable)
        # second vorticity term: interface i feeds levels i and i+1 (:704-746)
        dz = eng.combine(dudz1, 0.5, beta=0.5, c=dudz2)
        if dwdx1 is not None:
            eng.combine(dwdx1, -0.5, beta=1.0, c=dz, out=dz)
            eng.combine(dwdx2, -0.5, beta=1.0, c=dz, out=dz)
        v = Fz if Fz is not None else eng.combine(velz1, 0.5, beta=0.5, c=velz2)
        t = eng.apply("UTQWMAT", v, f=dz, lev0=0, scale=SCALE)                  # UtQWmat::assemble(u1, scale): no thickness
        eng.combine(t, 0.5, beta=1.0, c=fu[1:], out=fu[1:])
        eng.combine(t, 0.5, beta=1.0, c=fu[:-1], out=fu[:-1])
        if self.do_visc:
            self._ap("UMAT", self.laplacian(self.laplacian(uh)), flags=VERT | ACCUM, out=fu)
        return fu
