"""Row A6 directly: the host-side Jacobian (mimsem_amd/geom.py, numpy, vectorised) against the oracle's restatement of
Geom::jacobian / jacDet / initJacobians / updateGlobalCoords (eul/Geom.cpp:245-326, 682-741; oracle/o_patch.c) on the same
coordinate tables, plus the known answer sum_e sum_q w_q det = 4 pi R^2."""
import numpy as np
import pytest

from mimsem_amd.geom import Geom, gll_weights
from mimsem_amd.mesh import CubedSphere, RAD_SPHERE, sphere_coords
from mimsem_amd.topo import Topo


@pytest.mark.parametrize("pn,ne,npatch,pids", [(3, 4, 6, (0, 3, 5)), (3, 4, 24, (1, 13, 22)), (4, 2, 6, (2,)), (2, 3, 54, (0, 31, 53)),
                                                 (3, 8, 6, (4,)), (5, 1, 6, (1,))])
@pytest.mark.parametrize("signed", [False, True])
def test_jacobian_and_determinant_match_oracle(oracle, pn, ne, npatch, pids, signed):
    cs = CubedSphere(pn, ne, npatch); coords = sphere_coords(pn, ne)
    for pi in pids:
        g = Geom(Topo(cs, pi), cs, coords, 1, signed_det=signed)
        P = oracle.Patch(pn, pn, cs.nel, 1)
        P.set_sphere_geometry(coords[cs.patches[pi].loc0], abs_det=not signed)
        # re-projected quadrature-point coordinates (Geom::updateGlobalCoords)
        assert np.abs(g.x - P.xq).max() <= 1e-15 * RAD_SPHERE * 8
        scale = np.abs(P.J).max()
        assert np.abs(g.J - P.J).max() <= 1e-13 * scale, (pi, np.abs(g.J - P.J).max() / scale)
        assert np.abs(g.det - P.det).max() <= 1e-13 * np.abs(P.det).max()
        if signed:
            assert np.array_equal(np.sign(g.det), np.sign(P.det))
        else:
            assert (g.det > 0).all()


def test_sphere_area_from_determinants():
    """sum over all elements and quadrature points of w_q |det J| = 4 pi R^2 up to the quadrature error of the GLL rule"""
    errs = []
    for ne in (4, 8):
        cs = CubedSphere(3, ne, 6); coords = sphere_coords(3, ne)
        w = gll_weights(3); wq = np.outer(w, w).ravel()
        area = sum((Geom(Topo(cs, p), cs, coords, 1).det @ wq).sum() for p in range(6))
        errs.append(abs(area / (4.0 * np.pi * RAD_SPHERE ** 2) - 1.0))
    assert errs[0] < 1e-5 and errs[1] < errs[0] / 32.0, errs       # converges (measured: order 6) to the analytic area
