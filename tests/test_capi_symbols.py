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


def test_one_hip_runtime_per_process():
    """opencorr_amd loaded BEFORE torch must still end up on the HIP runtime torch uses (two copies of libamdhip64 in one
    process crash as soon as a stream handle crosses, see capi._one_hip_runtime).  Runs in a fresh interpreter; no GPU
    needed: only what gets mapped is looked at."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from opencorr_amd import capi\n"
            "capi.lib()\n"
            "before = capi._mapped('libamdhip64')\n"
            "import torch\n"
            "after = capi._mapped('libamdhip64')\n"
            "print(len(before), len(after), after == before)\n" % ROOT)
    out = subprocess.run([sys.executable, "-W", "error::RuntimeWarning", "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-1500:]
    assert out.stdout.split() == ["1", "1", "True"], out.stdout
