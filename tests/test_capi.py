"""The C-ABI shared library: loads, exports every symbol include/cirkit_hip.h declares, and its
argument validation answers with the documented status codes -- all without touching a GPU
(validation precedes every launch)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT
from cirkit_amd import _capi as capi

HEADER = os.path.join(ROOT, "include", "cirkit_hip.h")


INTERNAL = os.path.join(ROOT, "include", "cirkit_hip_internal.h")  # test hooks: exported, not part of the boundary


def _declared_symbols(headers=(HEADER, INTERNAL)):
    names = set()
    for h in headers:
        text = open(h, encoding="utf-8").read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(ck_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_library_loads_and_exports_every_declared_symbol():
    lib = capi.load()
    names = _declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in cirkit_hip.h but not exported"
    # and the ctypes table binds exactly the declared entry points
    assert set(capi.SIGNATURES) | {"ck_last_error"} == set(names)
    assert lib.ck_abi_version() == capi.ABI_VERSION
    # the boundary header holds no test hooks, the internal one nothing else
    assert not [n for n in _declared_symbols((HEADER,)) if n.startswith("ck_debug_")]
    assert all(n.startswith("ck_debug_") for n in _declared_symbols((INTERNAL,)))


def test_invalid_arguments_return_status_and_message():
    lib = capi.load()
    st = lib.ck_sum_lse_fwd(None, None, None, None, 1, 1, 1, 32, 32, 0, 0, None)
    assert st == -1 and b"null pointer" in lib.ck_last_error()
    buf = (C.c_float * 64)()
    p = C.cast(buf, C.c_void_p)
    st = lib.ck_sum_lse_fwd(p, p, p, p, 1, 1, 0, 32, 32, 0, 0, None)
    assert st == -1 and b"non-positive" in lib.ck_last_error()
    st = lib.ck_sum_lse_fwd(p, p, p, p, 1, 1, 4, 32, 32, 7, 0, None)
    assert st == -1 and b"unknown mode" in lib.ck_last_error()
    st = lib.ck_hadamard_fwd(p, p, p, 1, 2, 4, 8, 3, None)
    assert st == -1 and b"esize" in lib.ck_last_error()
    with pytest.raises(ValueError):  # python shim maps CK_ERR_INVALID to ValueError like the reference's shape errors
        capi.call("ck_param_unary", 99, p, p, 4, 0.0, 1.0, None)
    with pytest.raises(NotImplementedError):  # CK_ERR_UNSUPPORTED
        wl = (C.c_void_p * 1)(p)
        no = (C.c_int32 * 2)(0, 0)
        capi.call("ck_subtree_cat_cpt_fwd", p, None, p, p, p, wl, p, no, 0, p, 1, 1, 32, 64, 4, 0, None)
    st = lib.ck_sum_lse_fwd(p, p, p, p, 1, 2, 4, 16, 16, 0, 1, None)  # tiled layout needs K = 32
    assert st == -1 and b"tiled" in lib.ck_last_error()


def test_program_recording_needs_no_device():
    """While a program is being recorded the ck_* calls only append closures."""
    lib = capi.load()
    prog = C.c_void_p()
    assert lib.ck_program_begin(C.byref(prog)) == 0
    other = C.c_void_p()
    assert lib.ck_program_begin(C.byref(other)) == -4  # CK_ERR_STATE: already recording
    buf = (C.c_float * 64)()
    p = C.cast(buf, C.c_void_p)
    assert lib.ck_param_unary(capi.CK_UNARY_EXP, p, p, 16, 0.0, 1.0, None) == 0
    assert lib.ck_ll_sum(p, 8, 1, p, None) == 0
    assert lib.ck_program_end(prog) == 0
    assert lib.ck_program_num_ops(prog) == 2
    assert lib.ck_program_destroy(prog) == 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(capi, "_lib", None)
    monkeypatch.setattr(capi, "_LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(capi.HipExtensionError):
        capi.load()


def test_hip_circuit_refuses_cpu_device():
    from conftest import load_case
    from cirkit_amd.circuit import HipCircuit

    plan, tensors, _ = load_case("cfg1_rbt8")
    with pytest.raises(capi.HipExtensionError):
        HipCircuit(plan, tensors, device="cpu")
