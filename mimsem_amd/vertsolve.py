"""Host-side mirror of the per-column residual assembly of the reference's VertSolve (eul/VertSolve.cpp:237-286,
432-502): compositions of VertOps operators and mat-vecs, here issued for ALL columns at once through the C ABI
(mimsem_colop_apply) on "vertical" device arrays [nEl][nslots*n2e] (L2Vecs::vz concatenated)."""
import torch

SCALE = 1.0e8          # eul/VertOps.cpp:21
RAYLEIGH = 4.0 / 120.0  # eul/VertSolve.cpp:32
FLAG_VERT = 1


class VertSolve:
    def __init__(self, eng, dt, rayleigh=RAYLEIGH):
        self.eng, self.dt, self.rayleigh = eng, dt, rayleigh
        self.nk, self.n2e = eng.nk, eng.n2e
        self.k2i_z = 0.0

    # thin wrappers: vo->AssembleX(ex,ey,...,M); MatMult(M, x, y) for every column
    def _mv(self, colop, x, f1=None, f2=None, flags=0, rows=None, transpose=False):
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
