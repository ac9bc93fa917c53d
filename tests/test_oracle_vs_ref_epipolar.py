"""The EpipolarSearch consumer pinned on the reference's own loop (CPU only).

oracle/_ref/liboc_ref.so now also holds src/oc_epipolar_search.cpp, src/oc_calibration.cpp and src/oc_stereovision.cpp,
compiled unmodified.  `EpipolarSearch::compute(poi_queue)` (src/oc_epipolar_search.cpp:133-204) is run on a synthetic
stereo-like pair; beside it the batched form this repository offers: the host-side candidate generation of
include/opencorr_compat/oc_epipolar.h (compiled from the header) -> ONE queue of all trials through the solver -> the
selection "highest ZNCC wins".  With the oracle in the reference's loop order as the solver the two must agree in EVERY bit
of deformation and result of every POI -- which pins the candidate generation (the line, the truncations, the bounds tests,
the order of the trials) and the selection semantics, ties included (the earliest candidate of highest ZNCC: what libstdc++'s
std::sort leaves first among equals for fans of <= 16 trials, and oc_hip_select_best's documented rule).  The GPU twin
(tests/test_gpu_epipolar.py) swaps the solver for the HIP engine + oc_hip_select_best."""
import numpy as np
import pytest

import oracle
from oracle import ref as oref
from opencorr_amd import synth

import epipolar_case as ec

pytestmark = pytest.mark.skipif(not (oref.available() and hasattr(oref.lib(), "oc_ref_epipolar_search")),
                                reason="needs oracle/_ref/liboc_ref.so built from /root/reference (make -C oracle ref)")


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def test_batched_epipolar_search_equals_the_reference_loop(tmp_path):
    ref, tar = synth.speckle_pair_2d(300, 320, seed=20260925)
    h, w = ref.shape
    xs, ys = synth.poi_grid_2d(h, w, 17, 15, 30)
    # POIs near the right and the lower border lose trials to the bounds tests of :166-179
    xs = np.concatenate([xs, [w - 22.0, w - 19.0, 40.0]]).astype(np.float32)
    ys = np.concatenate([ys, [150.0, 60.0, h - 18.0]]).astype(np.float32)
    pois = oracle.make_pois2d(xs, ys)
    pois[:, 20:23] = 7.5    # strain: nobody touches it
    cam1, cam2 = ec.cameras(w, h)
    want = pois.copy()
    F = oref.epipolar_search(ref, tar, cam1, cam2, ec.SEARCH_RADIUS, ec.SEARCH_STEP, ec.PARALLAX_X, ec.PARALLAX_Y, ec.RX, ec.RY, ec.CONV,
                             ec.STOP, want)
    assert F is not None and np.isfinite(F).all()
    cand, starts = ec.candidates(ec.candidate_lib(tmp_path), pois, F, w, h)
    counts = np.diff(starts.astype(np.int64))
    assert counts.max() == 7 and counts.min() >= 1 and (counts < 7).any()      # full fans and fans cut by the bounds tests
    oracle.icgn2d1(oracle.Prepared2D(ref, tar), ec.RX, ec.RY, ec.CONV, ec.STOP, cand, order=oracle.ORDER_SEQ)
    got, unique = ec.select_like_the_reference(cand, starts, pois)
    # several trials of a POI converge to the same minimum and share their ZNCC to the last bit: the earliest of them wins in
    # the reference (libstdc++'s std::sort of <= 16 elements is a stable insertion sort) as in oc_hip_select_best
    assert (~unique).sum() > 10 and counts.max() <= 16
    bad = np.argwhere(_bits(got) != _bits(want))
    assert bad.size == 0, bad[:10].tolist()
    ok = want[:, 16] > 0.9
    assert ok.mean() > 0.9
    assert np.abs(want[ok, 2] - 2.3).max() < 1.0 and np.abs(want[ok, 8] + 1.7).max() < 1.0   # the pair's true displacement +- its gradients
