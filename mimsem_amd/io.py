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
