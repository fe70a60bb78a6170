"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports every
symbol include/mimsem_hip.h declares.  No compute calls (there is no GPU here)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    from mimsem_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.lib()


def test_header_symbols_all_exported(L):
    from mimsem_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "mimsem_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(mimsem_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/mimsem_hip.h but not exported"
    assert sorted(_lib.exported_symbols()) == declared       # the binding covers exactly the header


def test_abi_version_and_errors(L):
    assert L.mimsem_abi_version() == 1
    assert L.mimsem_strerror(0) == b"ok"
    assert b"unsupported" in L.mimsem_strerror(-2)
    # argument validation happens before any HIP call
    out = C.c_void_p()
    assert L.mimsem_ctx_create(None, 0, C.byref(out)) == -1
    from mimsem_amd._lib import MeshDesc
    d = MeshDesc(); d.elOrd = 3; d.quadOrd = 4; d.nEl = 1; d.nk = 1
    assert L.mimsem_ctx_create(C.byref(d), 0, C.byref(out)) == -2      # quadrature order != element order
    d.quadOrd = 3; d.elOrd = d.quadOrd = 9
    assert L.mimsem_ctx_create(C.byref(d), 0, C.byref(out)) == -2      # order outside 1..7


def test_engine_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from mimsem_amd import _lib
    from mimsem_amd.device import Engine
    with pytest.raises(_lib.MimsemError):
        Engine(object())
