#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ FROM THE REFERENCE (run in the authoring
container only; /root/reference does not travel to the GPU box, the .npz files do).

  topo_*.npz / geom_*.npz : scr/Proc2.py (ParaCube) and scr/Geom2.py (init_geom), imported as-is.
  basis_linalg.npz        : eul/Basis.cpp + eul/LinAlg.cpp compiled in place (oracle/_ref, `make -C oracle ref`).

Fixtures are data (inputs + expected outputs) only.  Usage: python tests/golden/make_fixtures.py
"""
import contextlib
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/scr")


def topo_fixture(pn, ne, n_procs):
    from Proc2 import ParaCube
    with contextlib.redirect_stdout(io.StringIO()):
        pc = ParaCube(n_procs, pn, ne, "/tmp/")
    out = {}
    for pi, pr in enumerate(pc.procs):
        out[f"loc0_{pi}"] = np.asarray(pr.loc0, dtype=np.int32)
        out[f"loc1x_{pi}"] = np.asarray(pr.loc1x, dtype=np.int32)
        out[f"loc1y_{pi}"] = np.asarray(pr.loc1y, dtype=np.int32)
        out[f"loc2_{pi}"] = np.asarray(pr.loc2, dtype=np.int32)
        out[f"sizes_{pi}"] = np.array([pr.n0l, pr.n1xl, pr.n1yl, pr.n2l], dtype=np.int32)
    out["meta"] = np.array([pn, ne, n_procs], dtype=np.int32)
    np.savez_compressed(os.path.join(HERE, f"topo_p{pn}_ne{ne}_np{n_procs}.npz"), **out)


def geom_fixture(pn, ne):
    from Geom2 import init_geom
    with contextlib.redirect_stdout(io.StringIO()):
        xg, yg, zg = init_geom(pn, ne, False, True)
    np.savez_compressed(os.path.join(HERE, f"geom_p{pn}_ne{ne}.npz"),
                        coords=np.stack([xg, yg, zg], axis=1), meta=np.array([pn, ne], dtype=np.int32))


def basis_linalg_fixture():
    from oracle import pyoracle
    import ctypes as C
    pyoracle.build(ref=True)
    R = pyoracle.ref_lib()
    assert R is not None, "oracle/_ref not built"
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    out = {}
    for n in range(1, 8):
        x = np.zeros(n + 1); w = np.zeros(n + 1)
        R.ref_gll(n, dp(x), dp(w))
        out[f"gll_x_{n}"] = x; out[f"gll_w_{n}"] = w
    for n, m in [(1, 1), (2, 2), (3, 3), (4, 4), (5, 5), (6, 6), (7, 7), (2, 3), (3, 4), (3, 5), (4, 6)]:
        l = np.zeros((m + 1, n + 1)); e = np.zeros((m + 1, n))
        R.ref_tables(n, m, dp(l), dp(e))
        out[f"ljxi_{n}_{m}"] = l; out[f"ejxi_{n}_{m}"] = e
    rng = np.random.default_rng(20241024)
    pts = rng.uniform(-1.3, 1.3, size=12)
    out["eval_pts"] = pts
    for n in (3, 4):
        out[f"node_eval_{n}"] = np.array([[R.ref_node_eval_q(n, n, C.c_double(x), i) for i in range(n + 1)] for x in pts])
        out[f"node_deriv_{n}"] = np.array([[R.ref_node_deriv(n, n, C.c_double(x), i) for i in range(n + 1)] for x in pts])
        out[f"edge_eval_{n}"] = np.array([[R.ref_edge_eval(n, n, C.c_double(x), i) for i in range(n)] for x in pts])
    # dense kernels on seeded inputs at the hot-path shapes
    for tag, (ni, nj, nk) in dict(u=(12, 12, 16), w=(9, 9, 16), p=(16, 16, 16), wu=(9, 12, 16), p4=(20, 20, 25)).items():
        A = rng.standard_normal((ni, nk)); B = rng.standard_normal((nk, nj)); d = rng.standard_normal(nk)
        Cm = np.zeros((ni, nj)); FD = np.zeros((ni, nk)); T = np.zeros((nk, ni)); y = np.zeros(ni)
        R.ref_Mult_IP(ni, nj, nk, dp(A), dp(B), dp(Cm))
        R.ref_Mult_FD_IP(ni, nk, nk, dp(A), dp(d), dp(FD))
        R.ref_Tran_IP(ni, nk, dp(A), dp(T))
        R.ref_Ax_b(ni, nk, dp(A), dp(d), dp(y))
        out.update({f"mm_A_{tag}": A, f"mm_B_{tag}": B, f"mm_d_{tag}": d, f"mm_C_{tag}": Cm,
                    f"mm_FD_{tag}": FD, f"mm_T_{tag}": T, f"mm_y_{tag}": y})
    for n in (4, 9, 16):
        A = rng.standard_normal((n, n)) + n * np.eye(n)
        Ai = np.zeros((n, n))
        err = R.ref_Inv(dp(A), dp(Ai), n)
        out[f"inv_A_{n}"] = A; out[f"inv_Ai_{n}"] = Ai; out[f"inv_err_{n}"] = np.array([err])
    # a permuted (pivot-heavy) and a singular case for Inv's error path
    A = np.eye(9)[rng.permutation(9)] * rng.uniform(1, 2, 9)
    Ai = np.zeros((9, 9)); err = R.ref_Inv(dp(A), dp(Ai), 9)
    out["inv_A_perm"] = A; out["inv_Ai_perm"] = Ai; out["inv_err_perm"] = np.array([err])
    A = np.ones((4, 4)); Ai = np.zeros((4, 4)); err = R.ref_Inv(dp(A), dp(Ai), 4)
    out["inv_A_sing"] = A; out["inv_err_sing"] = np.array([err])
    np.savez_compressed(os.path.join(HERE, "basis_linalg.npz"), **out)


def box_fixture(pn, ne, n_procs, lx=1000.0):
    from ProcBox import ParaBox
    from GeomBox import init_geom as box_geom
    with contextlib.redirect_stdout(io.StringIO()):
        pc = ParaBox(n_procs, pn, ne, "/tmp/")
        xg, yg, zg = box_geom(pn, ne, False, lx)
    out = {}
    for pi, pr in enumerate(pc.procs):
        for k in ("loc0", "loc1x", "loc1y", "loc2"):
            out[f"{k}_{pi}"] = np.asarray(getattr(pr, k), dtype=np.int32)
        out[f"sizes_{pi}"] = np.array([pr.n0l, pr.n1xl, pr.n1yl, pr.n2l], dtype=np.int32)
    out["coords"] = np.stack([xg, yg, zg], axis=1)
    out["meta"] = np.array([pn, ne, n_procs], dtype=np.int32)
    np.savez_compressed(os.path.join(HERE, f"box_p{pn}_ne{ne}_np{n_procs}.npz"), **out)


def input_files_fixture(pn, ne, n_procs):
    """the input/ directory exactly as scr/Setup.py produces it (same calls, same order), into tests/golden/"""
    import shutil
    from Proc2 import ParaCube
    from Geom2 import init_geom
    root = os.path.join(HERE, f"input_p{pn}_ne{ne}_np{n_procs}")
    shutil.rmtree(root, ignore_errors=True)
    os.makedirs(os.path.join(root, "input"))
    with contextlib.redirect_stdout(io.StringIO()):
        pc = ParaCube(n_procs, pn, ne, root + "/")
        for pi in np.arange(n_procs):
            pc.print_nodes(pi, "nodes"); pc.print_edges(pi, 0); pc.print_edges(pi, 1); pc.print_faces(pi)
            pc.procs[pi].writeLocalSizes()
        with open(root + "/input/grid_res.txt", "w") as f:
            f.write(str(pn) + "\n"); f.write(str(ne // pc.npx))
        pc = ParaCube(n_procs, pn, ne, root + "/")
        for pi in np.arange(n_procs):
            pc.print_nodes(pi, "quads")
        xg, yg, zg = init_geom(pn, ne, False, True)
        for pi in np.arange(n_procs):
            proc = pc.procs[pi]
            coords = np.stack([xg[proc.loc0], yg[proc.loc0], zg[proc.loc0]], axis=1)
            np.savetxt(root + "/input/geom_%.4u" % pi + ".txt", coords, fmt="%.18e")
            np.savetxt(root + "/input/local_sizes_quad_%.4u" % pi + ".txt", np.array([proc.n0l], dtype=np.int32), fmt="%u")
        with open(root + "/input/grid_res_quad.txt", "w") as f:
            f.write(str(pn) + "\n"); f.write(str(ne // pc.npx))


if __name__ == "__main__":
    input_files_fixture(2, 2, 6)
    for pn, ne, npr in [(4, 4, 4), (2, 3, 9), (3, 2, 1)]:
        box_fixture(pn, ne, npr)
    for pn, ne, npr in [(3, 2, 6), (3, 4, 24), (2, 3, 54), (4, 2, 6), (1, 2, 24)]:
        topo_fixture(pn, ne, npr)
    for pn, ne in [(3, 2), (3, 4), (4, 2), (2, 3)]:
        geom_fixture(pn, ne)
    basis_linalg_fixture()
    print("fixtures written to", HERE)
