"""Shared by tests/test_oracle_vs_ref_epipolar.py (CPU) and tests/test_gpu_epipolar.py (GPU): a synthetic stereo-like case for
the EpipolarSearch consumer (src/oc_epipolar_search.cpp:133-195) and the C wrapper of the product's host-side
candidate generation (include/opencorr_compat/oc_epipolar.h)."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RX = RY = 12
CONV, STOP = 0.001, 10.0
SEARCH_RADIUS, SEARCH_STEP = 11, 3          # trials at 0, +-3, +-6, +-9 px along the line: up to 7 per POI


def cameras(width, height):
    """Two cameras with the same intrinsics; the second one translated mostly along x, slightly down, with a small rotation:
    epipolar lines of slope ~ -0.5 that pass within a pixel of the synthetic pair's true displacement (2.3, -1.7)."""
    intr = np.zeros(13, dtype=np.float32)
    intr[0], intr[1], intr[2], intr[3], intr[4] = 2000.0, 2000.0, 0.0, width / 2.0, height / 2.0   # fx fy fs cx cy
    cam1 = (intr.copy(), np.zeros(6, dtype=np.float32))
    extr2 = np.array([-100.0, 50.0, 0.0, 1.0e-4, -2.0e-4, 3.0e-4], dtype=np.float32)               # tx ty tz rx ry rz
    cam2 = (intr.copy(), extr2)
    return cam1, cam2


PARALLAX_X = np.array([0.0, 0.0, 2.3], dtype=np.float32)
PARALLAX_Y = np.array([0.0, 0.0, -1.7], dtype=np.float32)


def candidate_lib(tmp_dir):
    so = os.path.join(str(tmp_dir), "libepipolar_candidates.so")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "epipolar_candidates.cpp"), "-o", so])
    L = ctypes.CDLL(so)
    fp = ctypes.POINTER(ctypes.c_float)
    L.oc_test_epipolar_candidates.restype = ctypes.c_long
    L.oc_test_epipolar_candidates.argtypes = [fp, ctypes.c_long, fp, fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                              ctypes.c_int, ctypes.c_int, fp, ctypes.c_long, ctypes.POINTER(ctypes.c_uint)]
    return L


def candidates(L, pois, F, width, height):
    """(candidate queue, segment starts) of include/opencorr_compat/oc_epipolar.h epipolarCandidates for the whole queue."""
    fp = ctypes.POINTER(ctypes.c_float)
    n = len(pois)
    cap = n * (2 * (SEARCH_RADIUS // SEARCH_STEP) + 2)
    cand = np.zeros((cap, 25), dtype=np.float32)
    starts = np.zeros(n + 1, dtype=np.uint32)
    F = np.ascontiguousarray(F, dtype=np.float32).reshape(9)
    got = L.oc_test_epipolar_candidates(pois.ctypes.data_as(fp), n, F.ctypes.data_as(fp), PARALLAX_X.ctypes.data_as(fp),
                                        PARALLAX_Y.ctypes.data_as(fp), SEARCH_RADIUS, SEARCH_STEP, RX, RY, width, height,
                                        cand.ctypes.data_as(fp), cap, starts.ctypes.data_as(ctypes.POINTER(ctypes.c_uint)))
    assert got >= 0, got
    return cand[:got].copy(), starts


def select_like_the_reference(cand, starts, pois):
    """poi->deformation = best.deformation; poi->result = best.result with best = the candidate of highest ZNCC
    (std::sort(sortByZNCC)[0], src/oc_epipolar_search.cpp:190-194).  Several trials of a POI usually converge to the SAME
    minimum and then share their ZNCC to the last bit, so the choice among equals matters: the standard leaves it open
    (std::sort is not stable), but for the <= 16 candidates of a search fan libstdc++'s std::sort IS its insertion sort, which
    keeps equal elements in their order -- the EARLIEST candidate of highest ZNCC wins, which is oc_hip_select_best's rule.
    Returns (result, unique): `unique[k]` is False where the maximum is shared."""
    out = pois.copy()
    unique = np.ones(len(pois), dtype=bool)
    for k in range(len(pois)):
        seg = cand[starts[k]:starts[k + 1]]
        z = seg[:, 16]
        j = int(np.argmax(z))
        unique[k] = (z == z[j]).sum() == 1
        out[k, 2:20] = seg[j, 2:20]
    return out, unique
