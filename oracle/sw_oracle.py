"""CPU restatement of the reference's shallow-water Picard step, SWEqn::solve (src/SWEqn_Picard.cpp:727-791) and everything
it calls -- TEST INFRASTRUCTURE ONLY (see oracle/oracle.h): importable from tests/ and bench.py's cpu_baseline leg, never from
mimsem_amd/.

Every PETSc Mat of the reference becomes a DENSE global numpy matrix, filled from the C oracle's element matrices with the
reference's MatSetValues(ADD_VALUES) pattern (global indices Topo::elInds*_g); every KSPSolve (GMRES, rtol 1e-16) becomes a
dense LU solve -- "direct solve to round-off", SURVEY 8(c).  Sized for small spheres (a few hundred DoFs).
src/ flavour: scale 1, unit thickness (nk = 1, levels 0 and 1), signed Jacobian determinant (src/Geom.cpp:248-252)."""
import numpy as np

from . import pyoracle

RAD_EARTH = 6371220.0
RAD_SPHERE = 6371220.0
H_MEAN = 1.0e+4          # src/SWEqn_Picard.cpp:27
ROS_ALPHA = 0.5          # :29
UP_TAU = 0.5             # :30


class SWOracle:
    def __init__(self, sphere, topos, geoms, coords):
        """sphere: mimsem_amd.mesh.CubedSphere (sizes only); topos/geoms: one per patch (index tables only); coords: the global
        quadrature-grid coordinate table (geom_*.txt content)."""
        self.cs, self.topos, self.geoms = sphere, topos, geoms
        pn = sphere.pn if hasattr(sphere, "pn") else topos[0].elOrd
        self.N0, self.N1, self.N2 = sphere.nDofs0G, sphere.nDofs1G, sphere.nDofs2G
        self.P = []
        for t, g in zip(topos, geoms):
            P = pyoracle.Patch(pn, pn, sphere.nel, 1)
            P.set_sphere_geometry(coords[g.loc0], abs_det=False)
            P.set_levels(np.stack([np.zeros(P.n0q), np.ones(P.n0q)]))
            self.P.append(P)
        self.NQ = int(max(g.loc0.max() for g in geoms)) + 1
        self.xq = np.zeros((self.NQ, 3))
        for g in geoms:
            self.xq[g.loc0] = coords[g.loc0]
        self.grav = 9.80616 * (RAD_SPHERE / RAD_EARTH)
        self.omega = 7.292e-5
        # fixed matrices
        self.M1 = self._assemble1("UMAT")
        self.M2 = self._assemble2("WMAT")
        self.M0 = self._assemble0("PMAT")
        self.E21 = self._e21()
        self.E10 = self._e10()
        self.E12 = -self.E21.T                       # E21mat: E12 = -E21^T (eul/Assembly.cpp:1209-1213 / src twin)
        self.E01 = -self.E10.T
        self.E01M1 = self.E01 @ self.M1              # :73
        self.E12M2 = self.E12 @ self.M2              # :74
        self.coriolis()

    # ---- local <-> global helpers ------------------------------------------------------------------------------------------
    def _local1(self, t, ug): return np.ascontiguousarray(ug[t.loc1])
    def _local0(self, t, qg): return np.ascontiguousarray(qg[t.loc0])
    def _local2(self, t, hg): return np.ascontiguousarray(hg[t.pi * t.n2 + np.arange(t.n2)])

    def _add11(self, M, t, P, em):
        gx, gy = t.all_inds1x_g(), t.all_inds1y_g()
        b = em.reshape(P.nEl, 4, P.n1e, P.n1e)
        for e in range(P.nEl):
            M[np.ix_(gx[e], gx[e])] += b[e, 0]; M[np.ix_(gx[e], gy[e])] += b[e, 1]
            M[np.ix_(gy[e], gx[e])] += b[e, 2]; M[np.ix_(gy[e], gy[e])] += b[e, 3]

    def _assemble1(self, op, field=None, space=None):
        M = np.zeros((self.N1, self.N1))
        for t, P in zip(self.topos, self.P):
            f = None if field is None else (self._local2(t, field) if space == 2 else self._local0(t, field))
            em = P.op_elmats(op, 0, 1.0, 0, f)
            if op == "ROTMAT":                       # blocks UtQV (x rows, y cols), VtQU (y rows, x cols)
                gx, gy = t.all_inds1x_g(), t.all_inds1y_g()
                b = em.reshape(P.nEl, 2, P.n1e, P.n1e)
                for e in range(P.nEl):
                    M[np.ix_(gx[e], gy[e])] += b[e, 0]; M[np.ix_(gy[e], gx[e])] += b[e, 1]
            else:
                self._add11(M, t, P, em)
        return M

    def _assemble2(self, op):
        M = np.zeros((self.N2, self.N2))
        for t, P in zip(self.topos, self.P):
            g2 = t.all_inds2_g(); em = P.op_elmats(op, 0, 1.0, 0).reshape(P.nEl, P.n2e, P.n2e)
            for e in range(P.nEl):
                M[np.ix_(g2[e], g2[e])] += em[e]
        return M

    def _assemble0(self, op, h=None):
        M = np.zeros((self.N0, self.N0))
        for t, P in zip(self.topos, self.P):
            g0 = t.all_inds0_g()
            em = P.op_elmats(op, 0, 1.0, 0, None if h is None else self._local2(t, h)).reshape(P.nEl, P.n0e, P.n0e)
            for e in range(P.nEl):
                M[np.ix_(g0[e], g0[e])] += em[e]
        return M

    def _e21(self):
        E = np.zeros((self.N2, self.N1))
        for t, P in zip(self.topos, self.P):
            g2 = t.pi * t.n2 + np.arange(t.n2)
            for j in range(P.n1):
                x = np.zeros(P.n1); x[j] = 1.0
                col = P.e21(x)
                nz = np.nonzero(col)[0]
                E[g2[nz], t.loc1[j]] = col[nz]       # every face row is owned by exactly one patch: plain insert
        return E

    def _e10(self):
        E = np.zeros((self.N1, self.N0))
        for t, P in zip(self.topos, self.P):
            for j in range(P.n0):
                x = np.zeros(P.n0); x[j] = 1.0
                col = P.e10(x)
                nz = np.nonzero(col)[0]
                E[t.loc1[nz], t.loc0[j]] = col[nz]   # rows of the patch's own edges only (Assembly.cpp:1102-1162)
        return E

    def K(self, ug):
        """WtQUmat::assemble(ul)  src/Assembly.cpp:1172-1299"""
        M = np.zeros((self.N2, self.N1))
        for t, P in zip(self.topos, self.P):
            gx, gy, g2 = t.all_inds1x_g(), t.all_inds1y_g(), t.all_inds2_g()
            em = P.op_elmats("WTQUMAT", 0, 1.0, 0, self._local1(t, ug)).reshape(P.nEl, 2, P.n2e, P.n1e)
            for e in range(P.nEl):
                M[np.ix_(g2[e], gx[e])] += em[e, 0]; M[np.ix_(g2[e], gy[e])] += em[e, 1]
        return M

    def M1h(self, hg): return self._assemble1("UHMAT", hg, 2)        # Uhmat::assemble(h)  src/Assembly.cpp:675-750
    def R(self, qg): return self._assemble1("ROTMAT", qg, 0)         # RotMat::assemble(q) src/Assembly.cpp:1346-1396
    def M0h(self, hg): return self._assemble0("PHMAT", hg)           # Phmat::assemble(h)  src/Assembly.cpp:396-497

    def R_up(self, qg, ug, dt):
        """RotMat_up::assemble(q0, ul, fac, dt)  src/Assembly.cpp:1784-1853"""
        M = np.zeros((self.N1, self.N1))
        for t, P in zip(self.topos, self.P):
            _, em = P.apply_up(1, np.zeros(P.n1), UP_TAU, dt, self._local0(t, qg), self._local1(t, ug))
            gx, gy = t.all_inds1x_g(), t.all_inds1y_g()
            b = em.reshape(P.nEl, 2, P.n1e, P.n1e)
            for e in range(P.nEl):
                M[np.ix_(gx[e], gy[e])] += b[e, 0]; M[np.ix_(gy[e], gx[e])] += b[e, 1]
        return M

    def M0h_up(self, ug, hg, dt):
        """Phmat::assemble_up(ul, hl, fac, dt)  src/Assembly.cpp:499-567"""
        M = np.zeros((self.N0, self.N0))
        for t, P in zip(self.topos, self.P):
            _, em = P.apply_up(0, np.zeros(P.n0), UP_TAU, dt, self._local2(t, hg), self._local1(t, ug))
            g0 = t.all_inds0_g(); b = em.reshape(P.nEl, P.n0e, P.n0e)
            for e in range(P.nEl):
                M[np.ix_(g0[e], g0[e])] += b[e]
        return M

    def project(self, which, fq):
        """WtQmat / PtQmat / UtQmat applied to a global quad-grid field (which 0/1/2)"""
        out = np.zeros([self.N2, self.N0, self.N1][which])
        for t, g, P in zip(self.topos, self.geoms, self.P):
            loc = np.ascontiguousarray(fq[g.loc0])
            y = P.project_from_quad(which, loc.reshape(-1) if which == 2 else loc)
            l2g = [t.pi * t.n2 + np.arange(t.n2), t.loc0, t.loc1][which]
            np.add.at(out, l2g, y)
        return out

    # ---- SWEqn methods -------------------------------------------------------------------------------------------------------
    def coriolis(self):
        """:186-233"""
        lat = np.arcsin(self.xq[:, 2] / RAD_SPHERE)
        fq = 2.0 * self.omega * np.sin(lat)
        self.fg = np.linalg.solve(self.M0, self.project(1, fq))

    def curl(self, u):
        return np.linalg.solve(self.M0, self.E01M1 @ u)                  # :236-250

    def diagnose_F(self, ui, uj, hi, hj):
        """:253-284"""
        hu = np.zeros(self.N1)
        M = self.M1h(hi)
        hu += (1.0 / 3.0) * (M @ ui); hu += (1.0 / 6.0) * (M @ uj)
        M = self.M1h(hj)
        hu += (1.0 / 6.0) * (M @ ui); hu += (1.0 / 3.0) * (M @ uj)
        return np.linalg.solve(self.M1, hu)

    def diagnose_Phi(self, ui, uj, hi, hj):
        """:289-320"""
        Phi = np.zeros(self.N2)
        Kx = self.K(ui)
        Phi += (1.0 / 3.0) * (Kx @ ui); Phi += (1.0 / 3.0) * (Kx @ uj)
        Kx = self.K(uj)
        Phi += (1.0 / 3.0) * (Kx @ uj)
        Phi += (self.grav / 2.0) * (self.M2 @ hi); Phi += (self.grav / 2.0) * (self.M2 @ hj)
        return Phi

    def diagnose_q(self, dt, u, h):
        """:322-341"""
        rhs = self.M0 @ self.fg + self.E01M1 @ u
        M = self.M0h_up(u, h, dt) if dt > 1.0e-6 else self.M0h(h)
        return np.linalg.solve(M, rhs)

    def assemble_residual(self, ui, hi, uj, hj, dt, q_exact=False, bot=None):
        """:402-607; returns (f_u, f_h)"""
        F = self.diagnose_F(ui, uj, hi, hj)
        Phi = self.diagnose_Phi(ui, uj, hi, hj)
        if bot is not None:
            Phi = Phi + self.grav * (self.M2 @ bot)
        fu = self.E12 @ Phi
        if q_exact:
            q = self.diagnose_q(0.0, 0.5 * ui + 0.5 * uj, 0.5 * hi + 0.5 * hj)
            fu = fu + 1.0 * (self.R(q) @ F)
        else:
            qi = self.diagnose_q(dt, ui, hi); qj = self.diagnose_q(dt, uj, hj)
            fu = fu + 0.5 * (self.R_up(qi, ui, dt) @ F)
            fu = fu + 0.5 * (self.R_up(qj, uj, dt) @ F)
        fh = self.M2 @ (self.E21 @ F)
        return (self.M1 @ uj - self.M1 @ ui) + dt * fu, (self.M2 @ hj - self.M2 @ hi) + dt * fh

    def assemble_operator(self, dt):
        """:609-725"""
        a = ROS_ALPHA * dt
        A = np.zeros((self.N1 + self.N2, self.N1 + self.N2))
        A[:self.N1, :self.N1] = self.M1 + a * self.R(self.fg)
        A[:self.N1, self.N1:] = (a * self.grav) * (self.E12 @ self.M2)
        A[self.N1:, :self.N1] = (a * H_MEAN) * (self.M2 @ self.E21)
        A[self.N1:, self.N1:] = self.M2
        return A

    def solve(self, un, hn, dt, nits=99, q_exact=False, bot=None):
        """:727-791"""
        A = self.assemble_operator(dt)
        ui, hi = un.copy(), hn.copy()
        x = np.concatenate([un, hn])
        uj, hj = un.copy(), hn.copy()
        it, hist = 0, []
        while True:
            fu, fh = self.assemble_residual(ui, hi, uj, hj, dt, q_exact, bot)
            dx = np.linalg.solve(A, -np.concatenate([fu, fh]))
            x = x + dx
            uj, hj = x[:self.N1].copy(), x[self.N1:].copy()
            norm = np.linalg.norm(dx) / np.linalg.norm(x)
            hist.append(norm)
            it += 1
            if not (norm > 1.0e-14 and it < nits):
                break
        self.history = hist
        return uj, hj

    def conservation(self, u, h, bot=None):
        """int2 / int0 / intE / enstrophy of writeConservation (:1202-1359), element by element and point by point as the reference
        does (interp at every quadrature point, sum w det f)"""
        mass = vort = ener = 0.0
        w0 = self.curl(u)
        for t, P in zip(self.topos, self.P):
            ul, hl, wl = self._local1(t, u), self._local2(t, h), self._local0(t, w0)
            bl = None if bot is None else self._local2(t, bot)
            Q = P.arr("Q", (P.mp12,))
            for e in range(P.nEl):
                ex, ey = e % P.nElsX, e // P.nElsX
                for ii in range(P.mp12):
                    det = P.det[e, ii]
                    px, py = ii % P.mp1, ii // P.mp1
                    hq = P.interp("2g", ex, ey, px, py, hl)[0]
                    bq = 0.0 if bl is None else P.interp("2g", ex, ey, px, py, bl)[0]
                    uq = P.interp("1g", ex, ey, px, py, ul)
                    wq = P.interp("0", ex, ey, px, py, wl)[0]
                    mass += det * Q[ii] * hq
                    vort += det * Q[ii] * wq
                    ener += det * Q[ii] * 0.5 * (self.grav * (hq + bq) * (hq + bq) + hq * (uq[0] * uq[0] + uq[1] * uq[1]))
        q = self.diagnose_q(0.0, u, h)
        enst = float(q @ (self.M0h(h) @ q))
        return dict(mass=mass, vorticity=vort, energy=ener, enstrophy=enst)

    def err_norms(self, form, vg, exact, lat_cut=False):
        """SWEqn::err0 / err1 / err2 (:981-1200) with fu = NULL gradients: [L1, L2, Linf], point by point.
        exact: values at the quadrature-grid points ([nq] or [nq, 2]), indexed like geom->x."""
        l1 = [0.0, 0.0]; l2 = [0.0, 0.0]; li = [0.0, 0.0]
        for t, g, P in zip(self.topos, self.geoms, self.P):
            vl = (self._local0, self._local1, self._local2)[form](t, vg)
            Q = P.arr("Q", (P.mp12,))
            inds0 = g.all_inds0_l()
            for e in range(P.nEl):
                ex, ey = e % P.nElsX, e // P.nElsX
                for ii in range(P.mp12):
                    if lat_cut and abs(g.s[inds0[e, ii], 1]) > 0.45 * np.pi:
                        continue
                    px, py = ii % P.mp1, ii // P.mp1
                    wd = P.det[e, ii] * Q[ii]
                    ua = np.atleast_1d(exact[g.loc0[inds0[e, ii]]])
                    un = P.interp(("0", "1g", "2g")[form], ex, ey, px, py, vl)[:ua.size]
                    a1, r1 = np.abs(un - ua).sum(), np.abs(ua).sum()
                    l1[0] += wd * a1; l1[1] += wd * r1
                    l2[0] += wd * ((un - ua) ** 2).sum(); l2[1] += wd * (ua * ua).sum()
                    if abs(wd * a1) > li[0]:
                        li = [abs(wd * a1), abs(wd * r1)]
        return [l1[0] / l1[1], np.sqrt(l2[0] / l2[1]), li[0] / li[1]]

    def init1(self, uq):
        return np.linalg.solve(self.M1, self.project(2, uq))             # :880-932

    def init2(self, hq):
        return np.linalg.solve(self.M2, self.project(0, hq))             # :934-975
