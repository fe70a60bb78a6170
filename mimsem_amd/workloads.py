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


def write_sw_case(path, dm, fg, u, h, dt, nsteps, nits, q_exact, bot=None):
    """A shallow-water case for the C++ hosts (mimsem_amd/host/sw_io.hpp: tests/cpp/test_sw.cpp, mimsem_amd/host/sw_call.cpp): the tables of
    a DeviceMesh (nk = 1) as mimsem_mesh_desc takes them, the Coriolis 0-form, a start state (host arrays), the step parameters and
    (optional) the bottom topography 2-form."""
    import numpy as np
    with open(path, "wb") as f:
        np.array([dm.n, dm.m, dm.nEl, 1, dm.n0, dm.n1, dm.n2, dm.nq, nsteps, nits, int(q_exact), 0 if bot is None else 1], dtype=np.int32).tofile(f)
        for a in (dm.inds0, dm.inds1x, dm.inds1y, dm.inds2, dm.indsq):
            np.ascontiguousarray(a, dtype=np.int32).tofile(f)
        for a in (dm.det, dm.J, dm.thick[:1], dm.thickInv[:1], fg, u, h, np.array([dt])) + (() if bot is None else (bot,)):
            np.ascontiguousarray(a, dtype=np.float64).tofile(f)


def write_arrays(path, arrays):
    """named int32 / float64 arrays for the C++ hosts (mimsem_amd/host/sw_io.hpp::read_arrays)"""
    import struct

    import numpy as np
    with open(path, "wb") as f:
        f.write(b"MSEMARR1"); f.write(struct.pack("<i", len(arrays)))
        for name, a in arrays.items():
            a = np.asarray(a)
            kind = 0 if np.issubdtype(a.dtype, np.integer) else 1
            a = np.ascontiguousarray(a, dtype=np.int32 if kind == 0 else np.float64)
            nb = name.encode()
            f.write(struct.pack("<i", len(nb))); f.write(nb); f.write(struct.pack("<iq", kind, a.size))
            a.tofile(f)


def mesh_arrays(dm):
    """the tables of a DeviceMesh under their mimsem_mesh_desc names (sw_io.hpp::desc_of)"""
    import numpy as np
    d = {"sizes": np.array([dm.n, dm.m, dm.nEl, dm.nk, dm.n0, dm.n1, dm.n2, dm.nq], dtype=np.int32)}
    for k in ("inds0", "inds1x", "inds1y", "inds2", "indsq", "det", "J", "thick", "thickInv"):
        d[k] = getattr(dm, k)
    return d
