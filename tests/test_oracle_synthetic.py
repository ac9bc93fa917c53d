"""Oracle sanity on synthetic speckle with analytically known warps (CPU only).

ICGN2D2 / FFTCC3D / ICGN3D1 have no usable golden vectors in the reference tree (SURVEY 8c:
"parity unpinned"); these known-answer tests are what pins their oracle functions.
"""
import numpy as np
import pytest

import oracle
from oracle import P2, P3
from opencorr_amd import synth


def test_icgn2d1_recovers_affine_field():
    ref, tar = synth.speckle_pair_2d(260, 280, seed=3)
    xs, ys = synth.poi_grid_2d(260, 280, 9, 8, 30)
    pois = oracle.make_pois2d(xs, ys)
    oracle.fftcc2d(ref, tar, 16, 16, pois)
    eu, ev = synth.expected_deformation_2d(xs, ys, 260, 280)
    assert np.abs(pois[:, P2["u"]] - eu).max() <= 1.0 and np.abs(pois[:, P2["v"]] - ev).max() <= 1.0
    prep = oracle.Prepared2D(ref, tar)
    for order in (oracle.ORDER_SEQ, oracle.ORDER_LANES):
        p = pois.copy()
        oracle.icgn2d1(prep, 16, 16, 0.001, 10, p, order=order)
        assert (p[:, P2["zncc"]] > 0.99).all()
        assert np.abs(p[:, P2["u"]] - eu).max() < 0.03 and np.abs(p[:, P2["v"]] - ev).max() < 0.03
        assert abs(np.median(p[:, P2["ux"]]) - synth.DEFAULT_WARP_2D["ux"]) < 5e-4
        assert abs(np.median(p[:, P2["vy"]]) - synth.DEFAULT_WARP_2D["vy"]) < 5e-4


def test_icgn2d2_recovers_second_order_field():
    so = dict(uxx=4e-5, uxy=-2e-5, uyy=3e-5, vxx=-3e-5, vxy=2e-5, vyy=-4e-5)
    ref, tar = synth.speckle_pair_2d(300, 320, seed=11, second_order=so)
    xs, ys = synth.poi_grid_2d(300, 320, 9, 8, 34)
    pois = oracle.make_pois2d(xs, ys)
    oracle.fftcc2d(ref, tar, 20, 20, pois)
    prep = oracle.Prepared2D(ref, tar)
    eu, ev = synth.expected_deformation_2d(xs, ys, 300, 320, second_order=so)
    for order in (oracle.ORDER_SEQ, oracle.ORDER_LANES):
        p = pois.copy()
        oracle.icgn2d2(prep, 20, 20, 0.001, 10, p, order=order)
        ok = p[:, P2["zncc"]] > 0.99
        assert ok.mean() > 0.95
        assert np.abs(p[ok, P2["u"]] - eu[ok]).max() < 0.03 and np.abs(p[ok, P2["v"]] - ev[ok]).max() < 0.03
        for key in ("uxx", "uxy", "uyy", "vxx", "vxy", "vyy"):
            assert abs(np.median(p[ok, P2[key]]) - so[key]) < 2e-5, key


def test_bicubic_lut_reproduces_pixels_and_rejects_out_of_range():
    rng = np.random.default_rng(0)
    img = rng.uniform(0, 255, (40, 50)).astype(np.float32)
    lut = oracle.bspline2d_lut(img)
    # interior integer positions: the local 4x4 B-spline fit interpolates the pixel itself
    for (x, y) in [(1, 1), (7, 9), (46, 36), (20.0, 5.0)]:
        assert abs(oracle.bspline2d_eval(lut, x, y) - img[int(y), int(x)]) < 1e-3
    for (x, y) in [(0.99, 5), (5, 0.5), (48.0, 5), (5, 38.0), (float("nan"), 5)]:
        assert oracle.bspline2d_eval(lut, x, y) == -1.0  # src/oc_cubic_bspline.cpp:137-142


def test_gradient_matches_central_difference_formula():
    rng = np.random.default_rng(1)
    img = rng.uniform(0, 255, (30, 33)).astype(np.float32)
    gx, gy = oracle.gradient2d(img)
    want = (-img[:, 4:] + 8 * img[:, 3:-1] - 8 * img[:, 1:-3] + img[:, :-4]) / 12.0
    assert np.abs(gx[:, 2:-2] - want).max() < 1e-3
    assert (gx[:, :2] == 0).all() and (gx[:, -2:] == 0).all() and (gy[:2] == 0).all() and (gy[-2:] == 0).all()


def test_fftcc2d_surface_peak_and_tie_rule():
    """arg-max uses strict '>' from index 0: on a constant (all-zero) pair index 0 wins -> u = v = 0."""
    flat = np.full((80, 80), 100, np.float32)
    pois = oracle.make_pois2d([40], [40])
    surf = oracle.fftcc2d(flat, flat, 8, 8, pois, want_surface=True)
    assert pois[0, P2["u"]] == 0 and pois[0, P2["v"]] == 0
    assert np.all(surf == 0)
    # a pure integer shift is found exactly
    rng = np.random.default_rng(2)
    a = rng.uniform(0, 255, (120, 120)).astype(np.float32)
    b = np.roll(a, (3, -5), axis=(0, 1))
    pois = oracle.make_pois2d([60], [60])
    oracle.fftcc2d(a, b, 16, 16, pois)
    assert pois[0, P2["u"]] == -5 and pois[0, P2["v"]] == 3 and pois[0, P2["zncc"]] > 0.5


@pytest.mark.parametrize("order", [oracle.ORDER_SEQ, oracle.ORDER_LANES])
def test_dvc_recovers_affine_field(order):
    vol_shape = (72, 76, 80)
    ref, tar = synth.speckle_pair_3d(*vol_shape, seed=21)
    xs, ys, zs = synth.poi_grid_3d(*vol_shape, 3, 3, 3, 26)
    pois = oracle.make_pois3d(xs, ys, zs)
    oracle.fftcc3d(ref, tar, 8, 8, 8, pois)
    w = synth.DEFAULT_WARP_3D
    xp, yp, zp = xs - (vol_shape[2] - 1) * 0.5, ys - (vol_shape[1] - 1) * 0.5, zs - (vol_shape[0] - 1) * 0.5
    eu = w["u"] + w["ux"] * xp + w["uy"] * yp + w["uz"] * zp
    ev = w["v"] + w["vx"] * xp + w["vy"] * yp + w["vz"] * zp
    ew = w["w"] + w["wx"] * xp + w["wy"] * yp + w["wz"] * zp
    assert np.abs(pois[:, P3["u"]] - eu).max() <= 1.0 and np.abs(pois[:, P3["w"]] - ew).max() <= 1.0
    prep = oracle.Prepared3D(ref, tar)
    oracle.icgn3d1(prep, 8, 8, 8, 0.001, 20, pois, order=order, lanes=256)
    assert (pois[:, P3["zncc"]] > 0.97).all()
    assert np.abs(pois[:, P3["u"]] - eu).max() < 0.05
    assert np.abs(pois[:, P3["v"]] - ev).max() < 0.05
    assert np.abs(pois[:, P3["w"]] - ew).max() < 0.05
