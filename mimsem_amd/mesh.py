"""Cubed-sphere patch topology and coordinates (host side, offline -- the reference's is Python too).

What the reference produces with scr/Setup.py -> scr/Proc2.py (ParaCube: 6.n^2 square patches, global
node/edge/face numbering with two hanging cube-corner nodes, ghost layer on the east/north side of every
patch) and scr/Geom2.py (equi-angular gnomonic GLL coordinates), re-derived here in closed form instead
of by stitching neighbour objects: every local (ghosted) slot of a patch is mapped straight to its
global id from the face adjacency table.  Integers must be bit-identical to Proc2.py's (pinned by
tests/golden/topo_*.npz); coordinates to the last ulp (tests/golden/geom_*.npz).

Conventions (scr/Proc2.py:54-130, eul/Topo.cpp:200-305):
  D = pn*ne/npx dofs per patch side, F = pn*ne per face side, patch id = face*npx^2 + py*npx + px
  loc0  [(D+1)*(D+1)]  node (ix,iy)          -> face-row-major node id  face*F^2 + gy*F + gx
  loc1x [D*(D+1)]      x-normal edge (ix,iy) -> 2*(patch*D^2 + elem*pn^2 + py*pn+px)       (iy<D, ix<=D)
  loc1y [(D+1)*D]      y-normal edge (ix,iy) -> the same + 1                                (iy<=D, ix<D)
  loc2  [D*D]          face                  -> patch*D^2 + elem*pn^2 + py*pn+px
  hanging nodes: 6F^2 (south-east corner shared by faces 0,2,4) and 6F^2+1 (north-west corner of 1,3,5).
"""
import numpy as np

# face adjacency, scr/Proc2.py:419-479.  Even faces see their EAST neighbour rotated (its south side,
# reversed) and their NORTH neighbour aligned; odd faces the other way round.
EAST_FACE = (2, 2, 4, 4, 0, 0)
NORTH_FACE = (1, 3, 3, 5, 5, 1)
RAD_SPHERE = 6371220.0   # eul/Geom.cpp:20, scr/Geom2.py:242


def _isqrt(n):
    r = int(np.sqrt(n))
    while r * r > n:
        r -= 1
    while (r + 1) * (r + 1) <= n:
        r += 1
    return r


class Patch:
    """The index data one reference rank reads from input/*_%04u.txt (eul/Topo.cpp:28-139)."""

    def __init__(self, pid, face, px, py):
        self.pid, self.face, self.px, self.py = pid, face, px, py
        self.loc0 = self.loc1x = self.loc1y = self.loc2 = None
        self.n0l = self.n1xl = self.n1yl = self.n2l = 0

    @property
    def loc1(self):
        """eul/Topo.cpp:82-86: interleave x/y edges."""
        out = np.empty(self.loc1x.size + self.loc1y.size, dtype=np.int32)
        out[0::2] = self.loc1x
        out[1::2] = self.loc1y
        return out


class CubedSphere:
    """pn: polynomial order, ne: elements per face side, n_procs = 6*npx^2 patches."""

    def __init__(self, pn, ne, n_procs=6):
        if n_procs % 6 or _isqrt(n_procs // 6) ** 2 != n_procs // 6:
            raise ValueError("patch count must be 6*n^2 (README.md:32, scr/Setup.py:25-29)")
        self.pn, self.ne, self.n_procs = pn, ne, n_procs
        self.npx = _isqrt(n_procs // 6)
        if ne % self.npx:
            raise ValueError("elements per face side must divide evenly over patches per side")
        self.nel = ne // self.npx            # elements per patch side (Topo::nElsX)
        self.D = pn * self.nel               # Topo::nDofsX
        self.F = pn * ne
        self.hang0 = n_procs * self.D * self.D
        self.hang1 = self.hang0 + 1
        self.nDofs0G = self.hang0 + 2        # eul/Topo.cpp:113-115
        self.nDofs1G = 2 * self.hang0
        self.nDofs2G = self.hang0
        self.patches = [self._build(f, px, py) for f in range(6)
                        for py in range(self.npx) for px in range(self.npx)]

    # ---- global ids from face coordinates ------------------------------------------------
    def node_id(self, f, gx, gy):
        """gx,gy in [0,F] (F = one past the face: owned by a neighbouring face or hanging)."""
        F = self.F
        gx = np.asarray(gx, dtype=np.int64); gy = np.asarray(gy, dtype=np.int64)
        E, N = EAST_FACE[f], NORTH_FACE[f]
        own = f * F * F + gy * F + gx
        if f % 2 == 0:
            east = np.where(gy == 0, self.hang0, E * F * F + (F - gy))
            north = N * F * F + gx
            corner = E * F * F
        else:
            east = E * F * F + gy * F
            north = np.where(gx == 0, self.hang1, N * F * F + (F - gx) * F)
            corner = N * F * F
        out = np.where((gx < F) & (gy < F), own,
              np.where((gx == F) & (gy < F), east,
              np.where((gx < F) & (gy == F), north, corner)))
        return out.astype(np.int32)

    def _elem_slot(self, gx, gy):
        """(patch-on-face index, slot inside patch) of the element-contiguous numbering, Proc2.py:105-130."""
        D, pn, nel, npx = self.D, self.pn, self.nel, self.npx
        pj = (gy // D) * npx + gx // D
        lx, ly = gx % D, gy % D
        slot = ((ly // pn) * nel + lx // pn) * pn * pn + (ly % pn) * pn + lx % pn
        return pj, slot

    def _edge_own(self, f, gx, gy):
        pj, slot = self._elem_slot(gx, gy)
        return 2 * ((f * self.npx * self.npx + pj) * self.D * self.D + slot)

    def edge_x_id(self, f, gx, gy):
        """x-normal edge with gx in [0,F], gy in [0,F)."""
        F = self.F
        gx = np.asarray(gx, dtype=np.int64); gy = np.asarray(gy, dtype=np.int64)
        E = EAST_FACE[f]
        inside = gx < F
        gxi = np.where(inside, gx, 0)
        own = self._edge_own(f, gxi, gy)
        if f % 2 == 0:   # east face rotated: its south-side y-edges, reversed
            ghost = self._edge_own(E, F - 1 - gy, np.zeros_like(gy)) + 1
        else:            # aligned: its west-side x-edges
            ghost = self._edge_own(E, np.zeros_like(gy), gy)
        return np.where(inside, own, ghost).astype(np.int32)

    def edge_y_id(self, f, gx, gy):
        """y-normal edge with gx in [0,F), gy in [0,F]."""
        F = self.F
        gx = np.asarray(gx, dtype=np.int64); gy = np.asarray(gy, dtype=np.int64)
        N = NORTH_FACE[f]
        inside = gy < F
        gyi = np.where(inside, gy, 0)
        own = self._edge_own(f, gx, gyi) + 1
        if f % 2 == 0:   # aligned: north face's south-side y-edges
            ghost = self._edge_own(N, gx, np.zeros_like(gx)) + 1
        else:            # rotated: north face's west-side x-edges, reversed
            ghost = self._edge_own(N, np.zeros_like(gx), F - 1 - gx)
        return np.where(inside, own, ghost).astype(np.int32)

    def face_id(self, f, gx, gy):
        pj, slot = self._elem_slot(np.asarray(gx, dtype=np.int64), np.asarray(gy, dtype=np.int64))
        return ((f * self.npx * self.npx + pj) * self.D * self.D + slot).astype(np.int32)

    # ---- one patch ------------------------------------------------------------------------
    def _build(self, f, px, py):
        D, npx = self.D, self.npx
        p = Patch(f * npx * npx + py * npx + px, f, px, py)
        x0, y0 = px * D, py * D
        iy, ix = np.meshgrid(np.arange(D + 1), np.arange(D + 1), indexing="ij")
        p.loc0 = self.node_id(f, x0 + ix, y0 + iy).ravel()
        iy, ix = np.meshgrid(np.arange(D), np.arange(D + 1), indexing="ij")
        p.loc1x = self.edge_x_id(f, x0 + ix, y0 + iy).ravel()
        iy, ix = np.meshgrid(np.arange(D + 1), np.arange(D), indexing="ij")
        p.loc1y = self.edge_y_id(f, x0 + ix, y0 + iy).ravel()
        iy, ix = np.meshgrid(np.arange(D), np.arange(D), indexing="ij")
        p.loc2 = self.face_id(f, x0 + ix, y0 + iy).ravel()
        # owned counts, Proc2.py:54-66: the SE-corner patch of face 0 / NW-corner patch of face 1
        # additionally own one hanging node each
        p.n0l = D * D + int((f == 0 and px == npx - 1 and py == 0) or (f == 1 and px == 0 and py == npx - 1))
        p.n1xl = p.n1yl = p.n2l = D * D
        return p


# ---- coordinates ---------------------------------------------------------------------------
_GLL = {
    2: lambda: np.array([-1.0, 0.0, +1.0]),
    3: lambda: np.array([-1.0, -np.sqrt(0.2), +np.sqrt(0.2), +1.0]),
    4: lambda: np.array([-1, -np.sqrt(3.0 / 7.0), 0.0, +np.sqrt(3.0 / 7.0), +1]),
    5: lambda: (lambda a: np.array([-1.0, -np.sqrt((7.0 + 2.0 * a) / 21.0), -np.sqrt((7.0 - 2.0 * a) / 21.0),
                                    +np.sqrt((7.0 - 2.0 * a) / 21.0), +np.sqrt((7.0 + 2.0 * a) / 21.0), +1.0]))(np.sqrt(7.0)),
    6: lambda: (lambda d: np.array([-1.0, -np.sqrt(5.0 / 11.0 + d), -np.sqrt(5.0 / 11.0 - d), 0.0,
                                    +np.sqrt(5.0 / 11.0 - d), +np.sqrt(5.0 / 11.0 + d), +1.0]))((2.0 / 11.0) * np.sqrt(5.0 / 3.0)),
    7: lambda: np.array([-1.0, -0.871740148509607, -0.591700181433142, -0.209299217902479,
                         +0.209299217902479, +0.591700181433142, +0.871740148509607, +1.0]),
}


def sphere_coords(pn, ne, radius=RAD_SPHERE):
    """Cartesian coordinates of every global node id (6F^2+2 of them), scr/Geom2.py:10-277.

    Face 0 is the equi-angular gnomonic panel centred on (1,0,0); the other five follow by the
    reference's chain of quarter turns (signed axis permutations, exact), then everything is pushed
    through (lon,lat) back to radius*unit-vector exactly as Geom2.py:241-250 does.
    """
    if pn not in _GLL:
        raise ValueError("coordinates exist for orders 2..7 (scr/Geom2.py:22-35)")
    q = _GLL[pn]()
    F = pn * ne
    dx = 0.5 * np.pi / ne
    X = np.zeros(F + 1)
    for el in range(ne):
        X[el * pn:(el + 1) * pn] = dx * 0.5 * (q[:pn] + 1.0) + el * dx - 0.25 * np.pi
    X[F] = +0.25 * np.pi

    def gnomonic(ax, ay, at):
        tx, ty = np.tan(ax), np.tan(ay)
        phi = np.arcsin(ty / np.sqrt(1.0 + tx * tx + ty * ty))
        return np.cos(phi) * np.cos(at), np.cos(phi) * np.sin(at), np.sin(phi)

    gy, gx = np.meshgrid(np.arange(F), np.arange(F), indexing="ij")
    x0, y0, z0 = gnomonic(X[gx].ravel(), X[gy].ravel(), X[gx].ravel())
    # hanging node 0 sits at angle (X[F], X[0]) of panel 0 (Geom2.py:62-69: tan taken at X[0] twice)
    hx0, hy0, hz0 = gnomonic(X[0], X[0], X[F])
    # panel 1 = panel 0 turned north; hanging node 1 from (X[0], X[F]) (Geom2.py:93-100)
    x1, y1, z1 = -z0, y0, x0
    tx = np.tan(X[F]); ty = np.tan(X[F]); phi = np.arcsin(ty / np.sqrt(1.0 + tx * tx + ty * ty))
    hx1, hy1, hz1 = -np.sin(phi), np.cos(phi) * np.sin(X[0]), np.cos(phi) * np.cos(X[0])
    x2, y2, z2 = x1, z1, -y1            # east
    x3, y3, z3 = -y2, x2, z2            # north
    x4, y4, z4 = -z3, y3, x3            # east
    x5, y5, z5 = x4, z4, -y4            # north
    xg = np.concatenate([x0, x1, x2, x3, x4, x5, [hx0, hx1]])
    yg = np.concatenate([y0, y1, y2, y3, y4, y5, [hy0, hy1]])
    zg = np.concatenate([z0, z1, z2, z3, z4, z5, [hz0, hz1]])
    theta = np.arctan2(yg, xg)
    phi = np.arcsin(zg)
    return np.stack([radius * np.cos(phi) * np.cos(theta),
                     radius * np.cos(phi) * np.sin(theta),
                     radius * np.sin(phi)], axis=1)


def patch_coords(mesh_q, coords, pid):
    """geom_%04u.txt content: coordinates of a patch's quad-grid nodes (scr/Setup.py:61-68)."""
    return coords[mesh_q.patches[pid].loc0]


# ---- doubly periodic planar box (box/ flavour, BASELINE config 5) ------------------------------------------------
class PeriodicBox:
    """npx^2 square patches of one doubly periodic face (scr/ProcBox.py:6-245): same local layout and
    element-contiguous edge/face numbering as the sphere, neighbours wrap around, no hanging nodes
    (nDofs0G has no +2, box/Topo.cpp:112)."""

    def __init__(self, pn, ne, n_procs=1):
        npx = _isqrt(n_procs)
        if npx * npx != n_procs or ne % npx:
            raise ValueError("box patch count must be a square that divides the elements per side")
        self.pn, self.ne, self.n_procs, self.npx = pn, ne, n_procs, npx
        self.nel = ne // npx
        self.D = pn * self.nel
        self.F = pn * ne
        self.nDofs0G = self.F * self.F
        self.nDofs1G = 2 * self.F * self.F
        self.nDofs2G = self.F * self.F
        self.patches = [self._build(px, py) for py in range(npx) for px in range(npx)]

    def _slot(self, gx, gy):
        D, pn, nel, npx = self.D, self.pn, self.nel, self.npx
        pj = (gy // D) * npx + gx // D
        lx, ly = gx % D, gy % D
        return pj * D * D + ((ly // pn) * nel + lx // pn) * pn * pn + (ly % pn) * pn + lx % pn

    def _build(self, px, py):
        D, F = self.D, self.F
        p = Patch(py * self.npx + px, 0, px, py)
        x0, y0 = px * D, py * D
        iy, ix = np.meshgrid(np.arange(D + 1), np.arange(D + 1), indexing="ij")
        p.loc0 = (((y0 + iy) % F) * F + (x0 + ix) % F).astype(np.int32).ravel()
        iy, ix = np.meshgrid(np.arange(D), np.arange(D + 1), indexing="ij")
        p.loc1x = (2 * self._slot((x0 + ix) % F, (y0 + iy) % F)).astype(np.int32).ravel()
        iy, ix = np.meshgrid(np.arange(D + 1), np.arange(D), indexing="ij")
        p.loc1y = (2 * self._slot((x0 + ix) % F, (y0 + iy) % F) + 1).astype(np.int32).ravel()
        iy, ix = np.meshgrid(np.arange(D), np.arange(D), indexing="ij")
        p.loc2 = self._slot(x0 + ix, y0 + iy).astype(np.int32).ravel()
        p.n0l = p.n1xl = p.n1yl = p.n2l = D * D
        return p


def box_coords(pn, ne, lx=1000.0):
    """planar GLL coordinates of every global node id (scr/GeomBox.py:9-74)"""
    q = {1: np.array([-1.0, +1.0])}.get(pn)
    if q is None:
        q = _GLL[pn]()
    F = pn * ne
    dx = lx / ne
    i = np.arange(F)
    c = (i // pn) * dx + 0.5 * dx * (1.0 + q[i % pn])
    gy, gx = np.meshgrid(i, i, indexing="ij")
    return np.stack([c[gx].ravel(), c[gy].ravel(), np.zeros(F * F)], axis=1)
