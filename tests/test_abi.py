"""The C-ABI library loads on a CPU-only box and exports every symbol
include/dynamite_amd.h declares (no compute calls here)."""
import ctypes as C
import os
import re

from dynamite_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "dynamite_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dnm_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound():
    L = _lib.lib()
    names = declared_symbols()
    assert len(names) >= 35
    for n in names:
        assert hasattr(L, n), n
        assert n in _lib.SIGNATURES, "no ctypes signature for " + n
    for n in _lib.SIGNATURES:
        assert n in names, n + " bound but not declared in the header"


def test_version_and_error_string():
    L = _lib.lib()
    assert L.dnm_version() >= 100
    # a failing call sets the error string
    d = C.c_int64()
    bad = _lib.Subspace()
    bad.type, bad.L = 99, 4
    assert L.dnm_subspace_dim(C.byref(bad), C.byref(d)) != 0
    assert b"unknown subspace type" in L.dnm_last_error()


def test_struct_sizes_match_native():
    """DevPass export checks sizeof on the native side; exercise it."""
    from plan_emulator import HostMat
    import numpy as np
    sub = _lib.Subspace()
    sub.type, sub.L = 0, 12
    hm = HostMat(np.array([0, 3]), np.array([0, 1, 3]), np.array([1, 0, 3]),
                 np.array([0.5, 0.25, -0.25], dtype=complex), sub, sub)
    assert hm.tiled == 1 and hm.n_local_passes >= 1
