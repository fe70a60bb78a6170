"""Host-side mirror of the weak-form differential operators of the reference's HorizSolve
(eul/HorizSolve.cpp:208-283): grad, curl, laplacian composed from engine applies, incidence stencils and the
device mass solves -- every level in one call.  SURVEY 8(f) row N2 (first pieces).  Single-GPU global numbering."""
import torch

from .krylov import MassSolver

SCALE = 1.0e8


class HorizSolve:
    def __init__(self, eng, del2=1.0):
        self.eng, self.del2 = eng, del2
        self.m1 = MassSolver(eng, SCALE, True)
        self.m0 = eng.pvec(0, eng.nk, SCALE)                  # M0 is diagonal for the collocated 0-forms (Pvec)

    def grad(self, phi):
        """u = M1^-1 E12 M2 phi   (HorizSolve::grad :208-228), phi: [nk, n2]"""
        Mphi = self.eng.apply("WMAT", phi, lev0=0, scale=SCALE, flags=1)
        rhs = self.eng.incidence("E12", Mphi)
        u, self.last_its = self.m1.solve(rhs)
        return u

    def curl(self, u, fg=None):
        """w = M0^-1 E01 M1 u (+ f)   (HorizSolve::curl :233-254), u: [nk, n1]"""
        Mu = self.m1.apply(u)
        w = self.eng.incidence("E01", Mu) / self.m0
        return w if fg is None else w + fg

    def laplacian(self, u):
        """del2 * (grad(E21 u) + E10 curl(u))   (HorizSolve::laplacian :256-283)"""
        ddu = self.grad(self.eng.incidence("E21", u))
        ddu = ddu + self.eng.incidence("E10", self.curl(u))
        return self.del2 * ddu
