"""The C++ host shim (mimsem_amd/host/mimsem_shim.hpp) compiled with g++ against the C ABI and run as a
reference-style call site; parity against the oracle inside the executable."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp, name="test_shim"):
    exe = os.path.join(tmp, name)
    subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "cpp", name + ".cpp"), "-o", exe,
                           "-L" + os.path.join(ROOT, "mimsem_amd"), "-lmimsem_hip", "-L" + os.path.join(ROOT, "oracle"), "-loracle",
                           "-Wl,-rpath," + os.path.join(ROOT, "mimsem_amd"), "-Wl,-rpath," + os.path.join(ROOT, "oracle"),
                           "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_shim_compiles_and_links(tmp_path, oracle):
    """CPU: the header-only shim compiles with plain g++ (no HIP headers) and links against the C ABI"""
    from mimsem_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    assert os.path.exists(_build(str(tmp_path)))
    assert os.path.exists(_build(str(tmp_path), "test_ksp"))
    assert os.path.exists(_build(str(tmp_path), "test_sw"))


@pytest.mark.gpu
def test_shim_call_sites_match_oracle(tmp_path, oracle):
    out = subprocess.run([_build(str(tmp_path))], capture_output=True, text=True, timeout=300)
    print(out.stdout, out.stderr)
    assert out.returncode == 0 and "OK" in out.stdout


@pytest.mark.gpu
def test_ksp_call_sites_match_dense_solves(tmp_path, oracle):
    """HorizSolve::grad, HorizSolve::diagnose_fluxes and one kspA solve of SWEqn::solve written in C++ over the shim's KSP class
    (mimsem_ksp_*: the solve loops inside the library) against dense solves of the oracle's matrices"""
    out = subprocess.run([_build(str(tmp_path), "test_ksp")], capture_output=True, text=True, timeout=300)
    print(out.stdout, out.stderr)
    assert out.returncode == 0 and "OK" in out.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("q_exact,nits,dt", [(False, 2, 360.0), (True, 4, 600.0)], ids=["galewsky_style", "williamson2_style"])
def test_sw_step_driven_from_cpp(tmp_path, oracle, q_exact, nits, dt):
    """SWEqn::solve (src/SWEqn_Picard.cpp:727-791) orchestrated in C++ (mimsem_amd/host/mimsem_sweqn.hpp: KSP objects, fixed-length
    Chebyshev solves, the same as one hipGraph per Picard iteration) on the cubed sphere of tests/test_gpu_sweqn.py, against the numpy
    oracle's step (oracle/sw_oracle.py: dense matrices, LU for every KSPSolve).  Tolerance 1e-9 as for the Python host."""
    import numpy as np
    from mimsem_amd.device import DeviceMesh
    from mimsem_amd.geom import Geom
    from mimsem_amd.mesh import CubedSphere, sphere_coords
    from mimsem_amd.topo import Topo
    from mimsem_amd.workloads import write_sw_case
    from oracle import sw_oracle
    from tests.helpers import rel_l2
    pn, ne, nsteps = 3, 2, 2
    cs = CubedSphere(pn, ne, 6); coords = sphere_coords(pn, ne)
    topos = [Topo(cs, p, 1) for p in range(6)]
    geoms = [Geom(t, cs, coords, 1, signed_det=True) for t in topos]
    for g in geoms:
        g.set_levels(np.stack([np.zeros(g.n0), np.ones(g.n0)]))
    dm = DeviceMesh(topos, geoms, nk=1, numbering="global")
    O = sw_oracle.SWOracle(cs, topos, geoms, coords)
    th = np.arcsin(O.xq[:, 2] / 6371220.0); lam = np.arctan2(O.xq[:, 1], O.xq[:, 0])
    U0, H0 = 38.61068276698372, 2998.1154702758267
    uq = np.stack([U0 * np.cos(th) + 3.0 * np.sin(2 * lam) * np.cos(th), 2.0 * np.cos(lam) * np.cos(th) ** 2], axis=1)
    hq = H0 - (6371220.0 * 7.292e-5 * U0 + 0.5 * U0 * U0) * np.sin(th) ** 2 / 9.80616 + 40.0 * np.cos(th) * np.sin(lam)
    u0, h0 = O.init1(uq), O.init2(hq)
    fin, fout = str(tmp_path / "sw_in.bin"), str(tmp_path / "sw_out.bin")
    write_sw_case(fin, dm, O.fg, u0, h0, dt, nsteps, nits, q_exact)
    out = subprocess.run([_build(str(tmp_path), "test_sw"), fin, fout], capture_output=True, text=True, timeout=600)
    print(out.stdout, out.stderr)
    assert out.returncode == 0 and "ALL OK" in out.stdout
    ur, hr = u0, h0
    for _ in range(nsteps):
        ur, hr = O.solve(ur, hr, dt, nits=nits, q_exact=q_exact)
    res = np.fromfile(fout, dtype=np.float64).reshape(3, dm.n1 + dm.n2)
    for mode in range(3):
        assert rel_l2(res[mode, :dm.n1], ur) < 1e-9 and rel_l2(res[mode, dm.n1:], hr) < 1e-9, mode
