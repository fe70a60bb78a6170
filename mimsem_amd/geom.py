"""Host-side mirror of the reference's Geom (eul/Geom.h:8-58, eul/Geom.cpp): quadrature-point
coordinates of a patch, the Guba et al. (2014) sphere Jacobian at every element quadrature point,
level heights / thickness.  Init-time only (scope row A6/A9); the results are uploaded once into the
device context.  Vectorised numpy, not a transcription of the reference's per-point loops."""
import numpy as np

from .mesh import RAD_SPHERE

_GLL_X = {
    1: lambda: np.array([-1.0, 1.0]),
    2: lambda: np.array([-1.0, 0.0, 1.0]),
    3: lambda: np.array([-1.0, -np.sqrt(0.2), np.sqrt(0.2), 1.0]),
    4: lambda: np.array([-1.0, -np.sqrt(3.0 / 7.0), 0.0, np.sqrt(3.0 / 7.0), 1.0]),
}


def gll_points(n):
    """GLL abscissae (eul/Basis.cpp:31-89); orders > 4 via the same closed forms as the device library."""
    if n in _GLL_X:
        return _GLL_X[n]()
    if n == 5:
        a = 2.0 * np.sqrt(7.0) / 21.0
        p, q = np.sqrt(1.0 / 3.0 + a), np.sqrt(1.0 / 3.0 - a)
        return np.array([-1.0, -p, -q, q, p, 1.0])
    if n == 6:
        a = 2.0 * np.sqrt(5.0 / 3.0) / 11.0
        p, q = np.sqrt(5.0 / 11.0 + a), np.sqrt(5.0 / 11.0 - a)
        return np.array([-1.0, -p, -q, 0.0, q, p, 1.0])
    if n == 7:
        return np.array([-1.0, -0.871740148509607, -0.591700181433142, -0.209299217902479,
                         0.209299217902479, 0.591700181433142, 0.871740148509607, 1.0])
    raise ValueError("invalid gauss-lobatto quadrature order: %d" % n)


def gll_weights(n):
    """GLL weights w_i = 2 / (n (n+1) P_n(x_i)^2) (the values tabulated in eul/Basis.cpp:31-89)"""
    x = gll_points(n)
    pn = np.polynomial.legendre.legval(x, [0.0] * n + [1.0])
    return 2.0 / (n * (n + 1) * pn * pn)


class Geom:
    def __init__(self, topo, quad_sphere, coords, nk=1, radius=RAD_SPHERE, signed_det=False):
        """topo: Topo of this patch; quad_sphere: CubedSphere built with the QUADRATURE order (the
        reference's second ParaCube pass, scr/Setup.py:55-58); coords: global coordinate table
        (mesh.sphere_coords(qn, ne)).  signed_det=True reproduces the src/ flavour."""
        self.topo, self.pi, self.nk, self.radius = topo, topo.pi, nk, radius
        self.quad_ord = quad_sphere.pn
        self.nElsX = topo.nElsX
        self.nDofsX = quad_sphere.D                      # quad-point grid size per side
        qp = quad_sphere.patches[topo.pi]
        self.loc0 = qp.loc0
        self.n0, self.n0l = qp.loc0.size, qp.n0l
        self.nDofs0G = quad_sphere.nDofs0G
        self.x = np.array(coords[self.loc0], dtype=np.float64)           # geom_%04u.txt content
        self.s = np.stack([np.arctan2(self.x[:, 1], self.x[:, 0]), np.arcsin(self.x[:, 2] / radius)], axis=1)
        self.qx = gll_points(self.quad_ord)
        self.updateGlobalCoords()
        self.initJacobians(signed_det)
        self.topog = np.zeros(self.n0)
        self.levs = np.zeros((nk + 1, self.n0))
        self.thick = np.ones((nk, self.n0))
        self.thickInv = np.ones((nk, self.n0))

    def all_inds0_l(self):
        """Geom::elInds0_l for every element (eul/Geom.cpp:799-811): [nEl][(m+1)^2]"""
        m, E = self.quad_ord, self.nElsX
        ey, ex, iy, ix = np.meshgrid(np.arange(E), np.arange(E), np.arange(m + 1), np.arange(m + 1), indexing="ij")
        return ((ey * m + iy) * (self.nDofsX + 1) + ex * m + ix).reshape(E * E, (m + 1) ** 2).astype(np.int32)

    def _corner_blend(self):
        """bilinear map of the 4 element corners evaluated at all quad points: [nEl][mp12][3]"""
        mp1 = self.quad_ord + 1
        inds = self.all_inds0_l()
        c = self.x[inds[:, [0, mp1 - 1, mp1 * mp1 - 1, (mp1 - 1) * mp1]]]          # [nEl][4][3]
        x1 = np.tile(self.qx, mp1); x2 = np.repeat(self.qx, mp1)                   # q = qy*mp1+qx
        wts = 0.25 * np.stack([(1 - x1) * (1 - x2), (1 + x1) * (1 - x2), (1 + x1) * (1 + x2), (1 - x1) * (1 + x2)], axis=1)
        return np.einsum("qc,ecd->eqd", wts, c), c, x1, x2

    def updateGlobalCoords(self):
        """eul/Geom.cpp:682-724: element-interior points re-projected through the corner map"""
        mp1 = self.quad_ord + 1
        inds = self.all_inds0_l()
        rt, _, _, _ = self._corner_blend()
        new = self.radius * rt / np.linalg.norm(rt, axis=2, keepdims=True)
        keep = np.ones(mp1 * mp1, dtype=bool)
        keep[[0, mp1 - 1, mp1 * mp1 - 1, mp1 * (mp1 - 1)]] = False
        # corners keep their file values; shared interior points get identical values from either side
        self.x[inds[:, keep].ravel()] = new[:, keep].reshape(-1, 3)
        self.s = np.stack([np.arctan2(self.x[:, 1], self.x[:, 0]), np.arcsin(self.x[:, 2] / self.radius)], axis=1)

    def initJacobians(self, signed_det=False):
        """eul/Geom.cpp:245-326, 726-741: J = A.B.C.D R/(4|r~|) at every element quad point"""
        inds = self.all_inds0_l()
        rt, c, x1, x2 = self._corner_blend()
        rinv = 1.0 / np.linalg.norm(rt, axis=2)                                   # [nEl][mp12]
        lam, phi = self.s[inds, 0], self.s[inds, 1]
        sl, cl, sp, cp = np.sin(lam), np.cos(lam), np.sin(phi), np.cos(phi)
        z = np.zeros_like(sl); o = np.ones_like(sl)
        A = np.stack([np.stack([-sl, cl, z], -1), np.stack([z, z, o], -1)], -2)                    # [..][2][3]
        B = np.stack([np.stack([sl * sl * cp * cp + sp * sp, -0.5 * np.sin(2 * lam) * cp * cp, -0.5 * cl * np.sin(2 * phi)], -1),
                      np.stack([-0.5 * np.sin(2 * lam) * cp * cp, cl * cl * cp * cp + sp * sp, -0.5 * sl * np.sin(2 * phi)], -1),
                      np.stack([-cl * sp, -sl * sp, cp], -1)], -2)                                  # [..][3][3]
        Cm = np.swapaxes(c, 1, 2)                                                                   # [nEl][3][4]
        D = np.stack([np.stack([-1 + x2, -1 + x1], -1), np.stack([1 - x2, -1 - x1], -1),
                      np.stack([1 + x2, 1 + x1], -1), np.stack([-1 - x2, 1 - x1], -1)], -2)        # [mp12][4][2]
        AB = A @ B
        ABC = np.einsum("eqik,ekj->eqij", AB, Cm)
        J = np.einsum("eqik,qkj->eqij", ABC, D) * (0.25 * self.radius * rinv)[..., None, None]
        self.J = np.ascontiguousarray(J.reshape(J.shape[0], J.shape[1], 4))        # J00 J01 J10 J11
        d = self.J[..., 0] * self.J[..., 3] - self.J[..., 1] * self.J[..., 2]
        self.det = d if signed_det else np.abs(d)

    def initTopog(self, ft, fl):
        """eul/Geom.cpp:743-764; ft(x)->topography, fl(x,k)->normalised level height"""
        max_height = fl(self.x[0], self.nk) if fl else 1.0
        self.topog = np.array([ft(x) for x in self.x])
        for k in range(self.nk + 1):
            zo = np.array([fl(x, k) for x in self.x])
            self.levs[k] = (max_height - self.topog) * zo / max_height + self.topog
        self.set_levels(self.levs)

    def set_levels(self, levs):
        self.levs = np.asarray(levs, dtype=np.float64)
        self.thick = self.levs[1:] - self.levs[:-1]
        self.thickInv = 1.0 / self.thick

    def thick_at_elements(self):
        """[nk][nEl][mp12] thickness / inverse at each element's own quad points (device layout)"""
        inds = self.all_inds0_l()
        return np.ascontiguousarray(self.thick[:, inds]), np.ascontiguousarray(self.thickInv[:, inds])


class BoxGeom(Geom):
    """box/ flavour (BASELINE config 5): doubly periodic planar patch with the constant diagonal Jacobian
    J = diag(LX / (2 * elements per side)) of box/Geom.cpp:132-143; same members as the sphere Geom."""

    def __init__(self, topo, quad_box, coords, nk=1, lx=1000.0):
        self.topo, self.pi, self.nk, self.radius = topo, topo.pi, nk, None
        self.quad_ord = quad_box.pn
        self.nElsX = topo.nElsX
        self.nDofsX = quad_box.D
        qp = quad_box.patches[topo.pi]
        self.loc0 = qp.loc0
        self.n0, self.n0l = qp.loc0.size, qp.n0l
        self.nDofs0G = quad_box.nDofs0G
        self.x = np.array(coords[self.loc0], dtype=np.float64)
        self.s = self.x[:, :2].copy()
        self.qx = gll_points(self.quad_ord)
        nEl, mp12 = self.nElsX ** 2, (self.quad_ord + 1) ** 2
        j = 0.5 * lx / (topo.nElsX * quad_box.npx)
        self.J = np.zeros((nEl, mp12, 4)); self.J[..., 0] = j; self.J[..., 3] = j
        self.det = np.abs(self.J[..., 0] * self.J[..., 3] - self.J[..., 1] * self.J[..., 2])
        self.topog = np.zeros(self.n0)
        self.levs = np.zeros((nk + 1, self.n0))
        self.thick = np.ones((nk, self.n0))
        self.thickInv = np.ones((nk, self.n0))
