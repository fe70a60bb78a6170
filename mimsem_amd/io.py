"""The reference's on-disk mesh format: the per-rank text files scr/Setup.py writes and
eul/Topo.cpp:28-139 / eul/Geom.cpp:38-105 read.  One integer per line (`%u`), coordinates as three `%.18e` columns,
`grid_res*.txt` as two lines without a trailing newline.  write_input() emits byte-identical files
(tests/test_io_format.py against files produced by the reference's own Setup path); read_input() loads them back."""
import os

import numpy as np

from .mesh import CubedSphere, sphere_coords


def _ints(path, a):
    np.savetxt(path, np.asarray(a), fmt="%u")


def write_input(dirname, pn, ne, n_procs, qn=None):
    """equivalent of `scr/Setup.py pn ne n_procs qn proj` (scr/Setup.py:39-78)"""
    qn = pn if qn is None else qn
    d = os.path.join(dirname, "input")
    os.makedirs(d, exist_ok=True)
    cs = CubedSphere(pn, ne, n_procs)
    for p in cs.patches:
        _ints(os.path.join(d, "nodes_%.4u.txt" % p.pid), p.loc0)
        _ints(os.path.join(d, "edges_x_%.4u.txt" % p.pid), p.loc1x)
        _ints(os.path.join(d, "edges_y_%.4u.txt" % p.pid), p.loc1y)
        _ints(os.path.join(d, "faces_%.4u.txt" % p.pid), p.loc2)
        _ints(os.path.join(d, "local_sizes_%.4u.txt" % p.pid), np.array([p.n0l, p.n1xl, p.n1yl, p.n2l], dtype=np.int32))
    with open(os.path.join(d, "grid_res.txt"), "w") as f:
        f.write(str(pn) + "\n"); f.write(str(ne // cs.npx))
    cq = CubedSphere(qn, ne, n_procs)
    coords = sphere_coords(qn, ne)
    for p in cq.patches:
        _ints(os.path.join(d, "quads_%.4u.txt" % p.pid), p.loc0)
        np.savetxt(os.path.join(d, "geom_%.4u.txt" % p.pid), coords[p.loc0], fmt="%.18e")
        _ints(os.path.join(d, "local_sizes_quad_%.4u.txt" % p.pid), np.array([p.n0l], dtype=np.int32))
    with open(os.path.join(d, "grid_res_quad.txt"), "w") as f:
        f.write(str(qn) + "\n"); f.write(str(ne // cq.npx))
    return cs, cq, coords


def read_input(dirname, pi):
    """what Topo::Topo and Geom::Geom read for rank `pi`: dict of arrays"""
    d = os.path.join(dirname, "input")
    li = lambda name: np.loadtxt(os.path.join(d, name), dtype=np.int64, ndmin=1).astype(np.int32)
    out = {}
    with open(os.path.join(d, "grid_res.txt")) as f:
        out["elOrd"], out["nElsX"] = (int(v) for v in f.read().split())
    with open(os.path.join(d, "grid_res_quad.txt")) as f:
        out["quadOrd"], out["nElsX_quad"] = (int(v) for v in f.read().split())
    out["loc0"] = li("nodes_%.4u.txt" % pi); out["loc1x"] = li("edges_x_%.4u.txt" % pi)
    out["loc1y"] = li("edges_y_%.4u.txt" % pi); out["loc2"] = li("faces_%.4u.txt" % pi)
    out["local_sizes"] = li("local_sizes_%.4u.txt" % pi)
    out["quads"] = li("quads_%.4u.txt" % pi)
    out["geom"] = np.loadtxt(os.path.join(d, "geom_%.4u.txt" % pi), dtype=np.float64, ndmin=2)
    out["n0l_quad"] = int(li("local_sizes_quad_%.4u.txt" % pi)[0])
    return out


# ---- PETSc binary Vec files (the reference's checkpoints: VecView / VecLoad through PetscViewerBinaryOpen) ------------------
# Third-party format (PETSc is not part of /root/reference; version unpinned there, src/Makefile:14-15 mentions 3.12.2 / 3.17.4).
# Published layout of a Vec in a PETSc binary file, unchanged across those versions, default (32-bit index) build:
#     int32  VEC_FILE_CLASSID = 1211214      (big-endian)
#     int32  n                               (big-endian, global length)
#     n x float64                            (big-endian), entries in global PETSc ordering
# Call sites mirrored: eul/UMJS14.cpp:238-267 (LoadVecs / LoadVecsVert: "output/<field>_<lev %.3u>_<step %.4u>.vec"),
# src/Williamson2.cpp:104-113 ("output/pressure_%.4u.vec", "output/velocity_%.4u.vec"), Geom::write0/1/2.
VEC_FILE_CLASSID = 1211214


def vec_filename(fieldname, step, lev=None, outdir="output"):
    """the reference's naming: eul flavour has a level field (%.3u), src flavour has none"""
    if lev is None:
        return os.path.join(outdir, "%s_%.4u.vec" % (fieldname, step))
    return os.path.join(outdir, "%s_%.3u_%.4u.vec" % (fieldname, lev, step))


def write_vec(path, a):
    a = np.ascontiguousarray(a, dtype=np.float64).reshape(-1)
    with open(path, "wb") as f:
        f.write(np.array([VEC_FILE_CLASSID, a.size], dtype=">i4").tobytes())
        f.write(a.astype(">f8").tobytes())


def read_vec(path):
    with open(path, "rb") as f:
        head = np.frombuffer(f.read(8), dtype=">i4")
        if head.size != 2 or int(head[0]) != VEC_FILE_CLASSID:
            raise ValueError("%s: not a PETSc binary Vec (classid %s)" % (path, head[:1]))
        n = int(head[1])
        data = np.frombuffer(f.read(8 * n), dtype=">f8")
        if data.size != n:
            raise ValueError("%s: truncated (%d of %d entries)" % (path, data.size, n))
    return data.astype(np.float64)


def save_levels(fieldname, step, fields, outdir="output"):
    """one file per level, like the reference's dump loops (eul/Euler_2.cpp: geom->write2(vec, fieldname, step, lev))"""
    os.makedirs(outdir, exist_ok=True)
    for k, v in enumerate(fields):
        write_vec(vec_filename(fieldname, step, k, outdir), v)


def load_levels(fieldname, step, nk, outdir="output"):
    """LoadVecs (eul/UMJS14.cpp:238-249)"""
    return np.stack([read_vec(vec_filename(fieldname, step, k, outdir)) for k in range(nk)])
