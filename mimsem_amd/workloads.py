"""Synthetic inputs shared by bench.py, scripts/ and the parity tests (SURVEY 8(d) "synthetic inputs"): part of the
host-side mirror, no checker code in here."""
import numpy as np

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
