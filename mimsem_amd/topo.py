"""Host-side mirror of the reference's Topo (eul/Topo.h:1-51, eul/Topo.cpp): the per-patch sizes,
global-index arrays and element->index maps that every operator class is constructed from.
Built from the in-memory mesh generator (mimsem_amd.mesh) instead of input/*.txt files; the integer
content is identical (tests/test_mesh.py)."""
import numpy as np


class Topo:
    """paired=False: the reference's local 1-form layout (eul/Topo.cpp:82-86, 215-240): x-edge (r, c) at slot 2(r(D+1) + c), y-edge
    (r, c) at slot 2(rD + c) + 1 -- the two halves of an aligned slot pair {2k, 2k+1} drift apart by one column per row, so the
    wave-level plan of the engine (one 16-byte access per slot pair) does not exist for it and the two-pass kernels run.
    paired=True: the same slots handed out so that a pair is CO-LOCATED -- y-edge (r, c) at 2(r(D+1) + c) + 1 next to x-edge (r, c) for
    r < D; the D y-edges of the patch's top row take the odd slots left over beside the right-hand column of x-edges,
    2(c(D+1) + D) + 1.  It is a permutation of the local vector (loc1 and both element maps change together, nothing else in the
    reference looks inside a local 1-form vector), shown as a patch to Topo.cpp in INTEGRATION.md."""

    def __init__(self, sphere, pi, nk=1, paired=False):
        p = sphere.patches[pi]
        self.paired = paired
        self.pi = pi
        self.nk = nk
        self.elOrd = sphere.pn
        self.nElsX = sphere.nel
        self.nDofsX = sphere.D
        self.loc0, self.loc1x, self.loc1y, self.loc2 = p.loc0, p.loc1x, p.loc1y, p.loc2
        self.loc1 = p.loc1
        if paired:
            D = sphere.D
            r, c = np.meshgrid(np.arange(D + 1), np.arange(D), indexing="ij")
            self._yslot = np.where(r < D, 2 * (r * (D + 1) + c) + 1, 2 * (c * (D + 1) + D) + 1).astype(np.int32)     # [D+1][D]
            loc1 = np.empty_like(self.loc1)
            loc1[0::2] = p.loc1x
            loc1[self._yslot.ravel()] = p.loc1y
            self.loc1 = loc1
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
        if self.paired:
            return self._yslot[r, c]
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
    def all_inds1y_g(self): return self.loc1[self.all_inds1y_l()]
    def all_inds2_g(self): return (self.all_inds2_l() + self.pi * self.n2).astype(np.int32)
    def elInds0_g(self, ex, ey): return self.all_inds0_g()[ey * self.nElsX + ex]
    def elInds1x_g(self, ex, ey): return self.all_inds1x_g()[ey * self.nElsX + ex]
    def elInds1y_g(self, ex, ey): return self.all_inds1y_g()[ey * self.nElsX + ex]
    def elInds2_g(self, ex, ey): return self.all_inds2_g()[ey * self.nElsX + ex]
