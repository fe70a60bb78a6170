"""CPU restatement of the reference's HorizSolve right-hand sides (eul/HorizSolve.cpp) -- TEST INFRASTRUCTURE ONLY
(see oracle/oracle.h): importable from tests/ only, never from mimsem_amd/.

Every PETSc Mat of the reference becomes a DENSE global numpy matrix filled from the C oracle's element matrices with the
reference's MatSetValues(ADD_VALUES) pattern; every KSPSolve becomes a dense LU solve ("direct solve to round-off",
SURVEY 8(c)).  Sized for small spheres.  eul/ flavour: SCALE = 1e8, layer thickness, |det|."""
import numpy as np

from . import pyoracle

SCALE = 1.0e+8
OMEGA = 7.29212e-5            # eul/HorizSolve.cpp:23
RAD_EARTH = 6371220.0


class GlobalDense:
    """dense global matrices of the Assembly.cpp operator classes on a whole (small) cubed sphere"""

    def __init__(self, sphere, topos, geoms, coords, levs):
        self.cs, self.topos, self.geoms = sphere, topos, geoms
        self.nk = levs.shape[0] - 1
        pn = topos[0].elOrd
        self.N0, self.N1, self.N2 = sphere.nDofs0G, sphere.nDofs1G, sphere.nDofs2G
        self.P = []
        for t, g in zip(topos, geoms):
            P = pyoracle.Patch(pn, pn, sphere.nel, self.nk)
            P.set_sphere_geometry(coords[g.loc0])
            P.set_levels(levs)
            self.P.append(P)
        self.NQ = int(max(g.loc0.max() for g in geoms)) + 1
        self.xq = np.zeros((self.NQ, 3))
        for g in geoms:
            self.xq[g.loc0] = coords[g.loc0]
        self.E21 = self._e21(); self.E10 = self._e10()
        self.E12 = -self.E21.T; self.E01 = -self.E10.T

    # local views of global fields
    def l1(self, t, u): return np.ascontiguousarray(u[t.loc1])
    def l0(self, t, q): return np.ascontiguousarray(q[t.loc0])
    def l2(self, t, h): return np.ascontiguousarray(h[t.pi * t.n2 + np.arange(t.n2)])

    def mat(self, op, lev, flag=0, field=None, scale=SCALE):
        """global dense matrix of one operator class at one level; field is a GLOBAL vector of the op's coefficient space"""
        sp = dict(UMAT=(1, 1, None), UTMAT=(1, 1, None), UHMAT=(1, 1, 2), UTMAT_H=(1, 1, 2), ROTMAT=(1, 1, 0), WMAT=(2, 2, None),
                  WHMAT=(2, 2, 2), PMAT=(0, 0, None), PHMAT=(0, 0, 2), WTQUMAT=(2, 1, 1), WTQDUDZ=(2, 1, 1), UTQWMAT=(1, 2, 1))[op]
        rows, cols, fs = sp
        N = {0: self.N0, 1: self.N1, 2: self.N2}
        M = np.zeros((N[rows], N[cols]))
        for t, P in zip(self.topos, self.P):
            f = None if fs is None else {0: self.l0, 1: self.l1, 2: self.l2}[fs](t, field)
            em = P.op_elmats(op, lev, scale, flag, f)
            gx, gy, g2, g0 = t.all_inds1x_g(), t.all_inds1y_g(), t.all_inds2_g(), t.all_inds0_g()
            n1e, n2e, n0e = P.n1e, P.n2e, P.n0e
            for e in range(P.nEl):
                if op == "ROTMAT":
                    b = em[e].reshape(2, n1e, n1e)
                    M[np.ix_(gx[e], gy[e])] += b[0]; M[np.ix_(gy[e], gx[e])] += b[1]
                elif rows == 1 and cols == 1:
                    b = em[e].reshape(4, n1e, n1e)
                    M[np.ix_(gx[e], gx[e])] += b[0]; M[np.ix_(gx[e], gy[e])] += b[1]
                    M[np.ix_(gy[e], gx[e])] += b[2]; M[np.ix_(gy[e], gy[e])] += b[3]
                elif rows == 2 and cols == 2:
                    M[np.ix_(g2[e], g2[e])] += em[e].reshape(n2e, n2e)
                elif rows == 0:
                    M[np.ix_(g0[e], g0[e])] += em[e].reshape(n0e, n0e)
                elif rows == 2 and cols == 1:
                    b = em[e].reshape(2, n2e, n1e)
                    M[np.ix_(g2[e], gx[e])] += b[0]; M[np.ix_(g2[e], gy[e])] += b[1]
                else:                                   # 1 x 2
                    b = em[e].reshape(2, n1e, n2e)
                    M[np.ix_(gx[e], g2[e])] += b[0]; M[np.ix_(gy[e], g2[e])] += b[1]
        return M

    def _e21(self):
        E = np.zeros((self.N2, self.N1))
        for t, P in zip(self.topos, self.P):
            g2 = t.pi * t.n2 + np.arange(t.n2)
            for j in range(P.n1):
                x = np.zeros(P.n1); x[j] = 1.0
                col = P.e21(x); nz = np.nonzero(col)[0]
                E[g2[nz], t.loc1[j]] = col[nz]
        return E

    def _e10(self):
        E = np.zeros((self.N1, self.N0))
        for t, P in zip(self.topos, self.P):
            for j in range(P.n0):
                x = np.zeros(P.n0); x[j] = 1.0
                col = P.e10(x); nz = np.nonzero(col)[0]
                E[t.loc1[nz], t.loc0[j]] = col[nz]
        return E

    def uvec_hu(self, lev, u, rho, fac):
        """Uvec::assemble_hu(lev, SCALE, ul, rho, ., fac) on every patch + the gtol_1 REVERSE/ADD  (eul/Assembly.cpp:2198-2279)"""
        out = np.zeros(self.N1)
        for t, P in zip(self.topos, self.P):
            np.add.at(out, t.loc1, P.uvec_hu(lev, SCALE, self.l1(t, u), self.l2(t, rho), fac))
        return out

    def project0(self, fq):
        """PtQmat applied to a global quad-grid field"""
        out = np.zeros(self.N0)
        for t, g, P in zip(self.topos, self.geoms, self.P):
            np.add.at(out, t.loc0, P.project_from_quad(1, np.ascontiguousarray(fq[g.loc0])))
        return out


class HorizOracle:
    """eul/HorizSolve.cpp on dense matrices; vectors are global, one row per level"""

    def __init__(self, gd, do_visc=True):
        self.g, self.nk, self.do_visc = gd, gd.nk, do_visc
        ae = 4.0 * np.pi * RAD_EARTH * RAD_EARTH                      # viscosity() :112-120
        dx = np.sqrt(ae / gd.N0)
        self.del2 = -np.sqrt(0.072 * dx ** 3.2)
        self.M1 = [gd.mat("UMAT", k, 1) for k in range(self.nk)]      # M1->assemble(lev, SCALE, true)
        self.M2 = [gd.mat("WMAT", k, 1) for k in range(self.nk)]
        self.M0 = [gd.mat("PMAT", k, 0) for k in range(self.nk)]
        self.coriolis()

    def coriolis(self):
        """:124-161  fg[k] = M0(k, scale 1)^-1 PtQ f"""
        lat = np.arcsin(self.g.xq[:, 2] / RAD_EARTH)
        b = self.g.project0(2.0 * OMEGA * np.sin(lat))
        self.fg = np.stack([np.linalg.solve(self.g.mat("PMAT", k, 0, scale=1.0), b) for k in range(self.nk)])

    def grad(self, phi, lev):
        return np.linalg.solve(self.M1[lev], self.g.E12 @ (self.M2[lev] @ phi))                   # :208-228

    def curl(self, u, lev, add_f=False):
        w = np.linalg.solve(self.M0[lev], self.g.E01 @ (self.M1[lev] @ u))                        # :233-254
        return w + self.fg[lev] if add_f else w

    def laplacian(self, u, lev):
        ddu = self.grad(self.g.E21 @ u, lev)                                                       # :256-283
        ddu = ddu + self.g.E10 @ self.curl(u, lev)
        return self.del2 * ddu

    def diagnose_fluxes(self, lev, u1, u2, h1, h2, theta):
        """:285-327 with theta_in_Wt = false (the _ec callers): F = M1^-1 sum Uvec_hu ; G = M1^-1 F(theta) F"""
        g = self.g
        hu = g.uvec_hu(lev, u1, h1, 1.0 / 3.0) + g.uvec_hu(lev, u1, h2, 1.0 / 6.0) \
            + g.uvec_hu(lev, u2, h1, 1.0 / 6.0) + g.uvec_hu(lev, u2, h2, 1.0 / 3.0)                   # m1->assemble_hu x4, :300-305
        F = np.linalg.solve(self.M1[lev], hu)
        G = np.linalg.solve(self.M1[lev], g.mat("UHMAT", lev, 1, theta[lev]) @ F)
        return F, G

    def advection_rhs_ec(self, u1, u2, h1, h2, theta):
        """:380-417 ; returns dF, dG [nk, N2] (horizontal layout, before HorizToVert) and the fluxes Fk, Gk"""
        g = self.g
        dF, dG, Fk, Gk = (np.zeros((self.nk, n)) for n in (g.N2, g.N2, g.N1, g.N1))
        for kk in range(self.nk):
            Fk[kk], Gk[kk] = self.diagnose_fluxes(kk, u1[kk], u2[kk], h1[kk], h2[kk], theta)
            dFk = g.E21 @ Fk[kk]
            dF[kk] = self.M2[kk] @ dFk
            dGk = g.E21 @ Gk[kk]
            dG[kk] = 0.5 * (self.M2[kk] @ dGk)
            dG[kk] += 0.5 * (g.mat("WHMAT", kk, 1, theta[kk]) @ dFk)
            dTheta = self.grad(theta[kk], kk)
            dG[kk] += g.mat("WTQUMAT", kk, 0, dTheta) @ Fk[kk]                                      # K incl. the 0.5 factor
        return dF, dG, Fk, Gk

    def diagnose_Phi(self, lev, u1, u2, velz1, velz2):
        """:419-470"""
        g = self.g
        K1 = g.mat("WTQUMAT", lev, 0, u1)
        Phi = (1.0 / 3.0) * (K1 @ u1) + (1.0 / 3.0) * (K1 @ u2)
        Phi += (1.0 / 3.0) * (g.mat("WTQUMAT", lev, 0, u2) @ u2)
        z1 = np.zeros(g.N2); z2 = np.zeros(g.N2)
        if lev > 0:
            z1 += 0.5 * velz1[lev - 1]; z2 += 0.5 * velz2[lev - 1]
        if lev < self.nk - 1:
            z1 += 0.5 * velz1[lev]; z2 += 0.5 * velz2[lev]
        W1 = g.mat("WHMAT", lev, 0, z1)
        Phi += (1.0 / 6.0) * (W1 @ z1) + (1.0 / 6.0) * (W1 @ z2)
        Phi += (1.0 / 6.0) * (g.mat("WHMAT", lev, 0, z2) @ z2)
        return Phi

    def diagnose_q(self, lev, rho, u):
        """:472-493"""
        g = self.g
        rhs = g.E01 @ (self.M1[lev] @ u)                 # m1->assemble(level, SCALE, true, ul): Uvec = M1 u
        rhs = rhs + self.M0[lev] @ self.fg[lev]
        return np.linalg.solve(g.mat("PHMAT", lev, 0, rho), rhs)

    def momentum_rhs_ec(self, lev, theta, dudz1, dudz2, velz1, velz2, Pi, velx1, velx2, rho1, rho2, Fx=None, Fz=None,
                        dwdx1=None, dwdx2=None, Fk=None):
        """:637-786 ; theta/Pi: this level's 2-forms; dudz*, velz*, Fz, dwdx*: [nk-1, .] interface arrays.
        returns fu (and the level's kinetic-to-internal exchange term when Fk is given)"""
        g = self.g
        Phi = self.diagnose_Phi(lev, velx1, velx2, velz1, velz2)
        dPi = self.grad(Pi, lev)
        dTheta = self.grad(theta, lev)
        fu = g.E12 @ Phi
        uh = 0.5 * velx1 + 0.5 * velx2
        rh = 0.5 * rho1 + 0.5 * rho2
        q = self.diagnose_q(lev, rh, uh)
        R = g.mat("ROTMAT", lev, 0, q)
        if Fx is None:
            dp = g.uvec_hu(lev, velx1, rho1, 1.0 / 3.0) + g.uvec_hu(lev, velx2, rho1, 1.0 / 6.0) \
                + g.uvec_hu(lev, velx1, rho2, 1.0 / 6.0) + g.uvec_hu(lev, velx2, rho2, 1.0 / 3.0)       # :675-682
            dp = R @ np.linalg.solve(self.M1[lev], dp)
        else:
            dp = R @ Fx
        fu = fu + dp
        dp = g.mat("UHMAT", lev, 1, theta) @ dPi
        fu = fu + 0.5 * dp
        dp = g.mat("UHMAT", lev, 1, Pi) @ dTheta
        fu = fu - 0.5 * dp
        dp = g.E12 @ (g.mat("WHMAT", lev, 1, Pi) @ theta)
        fu = fu + 0.5 * dp
        k2i = None if Fk is None else float(Fk @ dp) / SCALE
        for il in ((lev - 1,) if lev > 0 else ()) + ((lev,) if lev < self.nk - 1 else ()):
            dz = 0.5 * dudz1[il] + 0.5 * dudz2[il]
            if dwdx1 is not None:
                dz = dz - 0.5 * dwdx1[il] - 0.5 * dwdx2[il]
            Rh = g.mat("UTQWMAT", 0, 0, dz)                       # UtQWmat::assemble(u1, scale): no level, no thickness
            v = Fz[il] if Fz is not None else 0.5 * velz1[il] + 0.5 * velz2[il]
            fu = fu + 0.5 * (Rh @ v)
        if self.do_visc:
            d2u = self.laplacian(uh, lev)
            d4u = self.laplacian(d2u, lev)
            fu = fu + self.M1[lev] @ d4u
        return (fu, k2i) if Fk is not None else fu
