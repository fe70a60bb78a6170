"""Shared builders for the parity tests: the same patch handed to the oracle (CPU checker) and to the
HIP engine, on real cubed-sphere geometry with a stretched level set."""
import numpy as np

from mimsem_amd.geom import Geom
from mimsem_amd.mesh import CubedSphere, sphere_coords
from mimsem_amd.topo import Topo

SCALE = 1.0e8      # eul/Assembly.cpp:20


def z_levels(nk, n0q, rng=None, ztop=30000.0, mu=15.0):
    """UMJS14-like stretched levels (eul/UMJS14.cpp:124-129 shape), with a small per-point perturbation
    so that thickness really varies over the quad-point grid."""
    k = np.arange(nk + 1) / nk
    z = ztop * (np.sqrt(mu * k * k + 1.0) - 1.0) / (np.sqrt(mu + 1.0) - 1.0)
    levs = np.repeat(z[:, None], n0q, axis=1)
    if rng is not None:
        levs[1:-1] *= 1.0 + 0.01 * rng.uniform(-1, 1, (nk - 1, n0q))
    return levs


def make_patch(oracle, pn=3, ne=4, nprocs=6, pi=0, nk=3, seed=0):
    rng = np.random.default_rng(seed)
    cs = CubedSphere(pn, ne, nprocs)
    coords = sphere_coords(pn, ne)
    topo = Topo(cs, pi, nk)
    geom = Geom(topo, cs, coords, nk)
    geom.set_levels(z_levels(nk, geom.n0, rng))
    P = oracle.Patch(pn, pn, cs.nel, nk)
    P.set_sphere_geometry(coords[cs.patches[pi].loc0])
    P.set_levels(geom.levs)
    return cs, topo, geom, P, rng


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64).ravel(); b = np.asarray(b, dtype=np.float64).ravel()
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)
