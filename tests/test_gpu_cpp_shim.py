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
