"""Host-side Deformation2D1 / Deformation2D2 / Deformation3D1 (include/opencorr_compat/oc_deformation.h; the reference's
public API src/oc_deformation.h:26-100: setDeformation(...), setWarp(), warp(point), warp_matrix(i, j)) against the
reference's OWN classes compiled unmodified into oracle/_ref (oracle/ref_driver.cpp oc_ref_deformation), bit for bit: the
parameter -> warp-matrix map (2D2: the 18 polynomial entries), the warped point, and the parameters read back from an
arbitrary warp matrix.  The header's small matrix type (product, inverse) is held against the oracle's restatement of the
solvers' warp update W * (dW)^-1.  CPU only."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import oracle
from oracle import ref as oref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

KINDS = {1: (6, 3, 2), 2: (12, 6, 2), 3: (12, 4, 3)}   # parameters, matrix side, point dimension


@pytest.fixture(scope="module")
def mine(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("deformation") / "libdeformation_check.so")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Werror", "-ffp-contract=off", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "deformation_check.cpp"), "-o", so])
    lib = ctypes.CDLL(so)
    fp = ctypes.POINTER(ctypes.c_float)
    lib.oc_test_deformation.argtypes = [ctypes.c_int, fp, fp, fp, fp, fp, fp]
    lib.oc_test_warp_compose.argtypes = [ctypes.c_int, fp, fp, fp]
    return lib


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _run(fn, kind, p, pt, mat_in):
    npar, side, dim = KINDS[kind]
    mat_out = np.zeros(side * side, np.float32)
    warped = np.zeros(dim, np.float32)
    back = np.zeros(npar, np.float32)
    rc = fn(kind, _fp(p), _fp(pt), _fp(mat_in), _fp(mat_out), _fp(warped), _fp(back))
    assert rc == 0, rc
    return mat_out, warped, back


@pytest.mark.skipif(not (oref.available() and hasattr(oref.lib(), "oc_ref_deformation")),
                    reason="needs oracle/_ref/liboc_ref.so built from /root/reference (make -C oracle ref)")
@pytest.mark.parametrize("kind", [1, 2, 3])
def test_deformation_classes_equal_the_reference_bit_for_bit(mine, kind):
    L = oref.lib()
    fp = ctypes.POINTER(ctypes.c_float)
    L.oc_ref_deformation.argtypes = [ctypes.c_int, fp, fp, fp, fp, fp, fp]
    npar, side, dim = KINDS[kind]
    rng = np.random.default_rng(100 + kind)
    for trial in range(200):
        scale = [1e-3, 1e-1, 3.0][trial % 3]
        p = (rng.standard_normal(npar) * scale).astype(np.float32)
        p[0] = np.float32(rng.uniform(-9, 9))                      # translations of pixel size
        p[npar // (2 if kind != 3 else 3)] = np.float32(rng.uniform(-9, 9))
        pt = rng.uniform(-40, 40, dim).astype(np.float32)
        mat_in = rng.standard_normal(side * side).astype(np.float32)
        want = _run(L.oc_ref_deformation, kind, p, pt, mat_in)
        got = _run(mine.oc_test_deformation, kind, p, pt, mat_in)
        for a, b, what in zip(got, want, ("warp_matrix", "warp(point)", "setDeformation()")):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (kind, trial, what, a, b)


@pytest.mark.parametrize("n", [3, 4, 6])
def test_small_matrix_compose_matches_the_oracle_algebra(mine, n):
    """W * (dW)^-1 with the header's matrices: within float rounding of the oracle's own restatement of the solvers' warp
    update (cofactor inverse for 3 x 3 / 4 x 4, LU for 6 x 6 there; LU with partial pivoting here for every size)."""
    rng = np.random.default_rng(n)
    for _ in range(50):
        w = (np.eye(n) + 0.05 * rng.standard_normal((n, n))).astype(np.float32)
        dw = (np.eye(n) + 0.02 * rng.standard_normal((n, n))).astype(np.float32)
        out = np.zeros(n * n, np.float32)
        mine.oc_test_warp_compose(n, _fp(np.ascontiguousarray(w.ravel())), _fp(np.ascontiguousarray(dw.ravel())), _fp(out))
        want = w.astype(np.float64) @ np.linalg.inv(dw.astype(np.float64))
        assert np.abs(out.reshape(n, n) - want).max() <= 2e-6
        inv = np.zeros((n, n), np.float32)
        rc = oracle.inverse(dw) if hasattr(oracle, "inverse") else None
        if rc is not None:
            assert np.abs(np.asarray(rc, np.float64) - np.linalg.inv(dw.astype(np.float64))).max() <= 2e-6
