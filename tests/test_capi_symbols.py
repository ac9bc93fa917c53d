"""The C-ABI library loads on a CPU-only box and exports every symbol the header declares."""
import ctypes
import os
import re

from opencorr_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "opencorr_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(oc_hip_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_list_the_same_symbols():
    assert header_symbols() == sorted(capi.SYMBOLS)


def test_library_exports_every_declared_symbol():
    assert os.path.exists(capi.LIB_PATH), "build with `python -m opencorr_amd.build`"
    L = ctypes.CDLL(capi.LIB_PATH)
    for name in header_symbols():
        assert hasattr(L, name), "missing export " + name


def test_abi_version_and_error_string():
    L = capi.lib()
    assert L.oc_hip_abi_version() == 3
    # a null handle is rejected with a readable message (no GPU needed for this path)
    assert L.oc_hip_prepare(None) == capi.ERR_INVALID
    assert b"null engine" in L.oc_hip_last_error()
