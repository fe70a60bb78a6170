"""Host-side mirror of the reference's Topo (eul/Topo.h:1-51, eul/Topo.cpp): the per-patch sizes,
global-index arrays and element->index maps that every operator class is constructed from.
Built from the in-memory mesh generator (mimsem_amd.mesh) instead of input/*.txt files; the integer
content is identical (tests/test_mesh.py)."""
import numpy as np


class Topo:
    def __init__(self, sphere, pi, nk=1):
        p = sphere.patches[pi]
        self.pi = pi
        self.nk = nk
        self.elOrd = sphere.pn
        self.nElsX = sphere.nel
        self.nDofsX = sphere.D
        self.loc0, self.loc1x, self.loc1y, self.loc2 = p.loc0, p.loc1x, p.loc1y, p.loc2
        self.loc1 = p.loc1
        self.n0, self.n1x, self.n1y, self.n2 = p.loc0.size, p.loc1x.size, p.loc1y.size, p.loc2.size
        self.n1 = self.n1x + self.n1y
        self.n0l, self.n1xl, self.n1yl, self.n2l = p.n0l, p.n1xl, p.n1yl, p.n2l
        self.n1l = self.n1xl + self.n1yl
        self.nDofs0G, self.nDofs1G, self.nDofs2G = sphere.nDofs0G, sphere.nDofs1G, sphere.nDofs2G

    # element -> local index maps (eul/Topo.cpp:200-251), vectorised over the patch: [nEl][dofs]
    def _grid(self, ny, nx):
        n, E = self.elOrd, self.nElsX
        ey, ex, iy, ix = np.meshgrid(np.arange(E), np.arange(E), np.arange(ny), np.arange(nx), indexing="ij")
        return (ey * n + iy).reshape(E * E, ny * nx), (ex * n + ix).reshape(E * E, ny * nx)

    def all_inds0_l(self):
        r, c = self._grid(self.elOrd + 1, self.elOrd + 1)
        return (r * (self.nDofsX + 1) + c).astype(np.int32)

    def all_inds1x_l(self):
        r, c = self._grid(self.elOrd, self.elOrd + 1)
        return (2 * (r * (self.nDofsX + 1) + c)).astype(np.int32)

    def all_inds1y_l(self):
        r, c = self._grid(self.elOrd + 1, self.elOrd)
        return (2 * (r * self.nDofsX + c) + 1).astype(np.int32)

    def all_inds2_l(self):
        n2 = self.elOrd * self.elOrd
        return np.arange(self.nElsX ** 2 * n2, dtype=np.int32).reshape(-1, n2)

    def elInds0_l(self, ex, ey): return self.all_inds0_l()[ey * self.nElsX + ex]
    def elInds1x_l(self, ex, ey): return self.all_inds1x_l()[ey * self.nElsX + ex]
    def elInds1y_l(self, ex, ey): return self.all_inds1y_l()[ey * self.nElsX + ex]
    def elInds2_l(self, ex, ey): return self.all_inds2_l()[ey * self.nElsX + ex]

    # global variants (eul/Topo.cpp:253-305)
    def all_inds0_g(self): return self.loc0[self.all_inds0_l()]
    def all_inds1x_g(self): return self.loc1x[self.all_inds1x_l() // 2]
    def all_inds1y_g(self): return self.loc1y[(self.all_inds1y_l() - 1) // 2]
    def all_inds2_g(self): return (self.all_inds2_l() + self.pi * self.n2).astype(np.int32)
    def elInds0_g(self, ex, ey): return self.all_inds0_g()[ey * self.nElsX + ex]
    def elInds1x_g(self, ex, ey): return self.all_inds1x_g()[ey * self.nElsX + ex]
    def elInds1y_g(self, ex, ey): return self.all_inds1y_g()[ey * self.nElsX + ex]
    def elInds2_g(self, ex, ey): return self.all_inds2_g()[ey * self.nElsX + ex]
