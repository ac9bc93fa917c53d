"""Strain (2D, 3D) and RegionFit2D / RegionFit3D of the oracle against the REFERENCE's own sources
(/root/reference/src/oc_strain.cpp, oc_region_fit.cpp, oc_nearest_neighbor.cpp compiled unmodified into oracle/_ref by
`make -C oracle ref`).

What the reference leaves to third-party code is restated by stand-ins written for this purpose (oracle/ref_stubs):
nanoflann -> a brute-force radius / knn search with nanoflann's documented semantics (strict `<` on the squared radius,
knn by ascending distance); Eigen's float `colPivHouseholderQr().solve()` -> the same algorithm in scalar float loops.
The oracle solves the plane fit through double-precision normal equations instead, so the bars are
  * WHICH POIs are computed (ZNCC filter of the POI itself, radius search, KNN fallback, ZNCC filter of the neighbours,
    minimum count): identical -- every float the reference leaves untouched is untouched, bit for bit;
  * the values written: within 1e-6 absolute (strains and fitted gradients, magnitudes <= 1e-2) and 1e-5 for fitted
    displacements of a few pixels (a few float ulps of the QR, amplified where the query extrapolates).
Queues use irregular positions: on a regular grid the K nearest neighbours are not unique (equal distances) and the
reference's own choice depends on nanoflann's tree traversal -- nothing to pin there.  Skipped where the reference tree
is not mounted (the GPU box).
"""
import numpy as np
import pytest

import oracle
from oracle import P2, P3
from oracle import ref as oref

pytestmark = pytest.mark.skipif(not oref.available(), reason="reference tree not mounted: oracle/_ref cannot be built")

S2 = [P2["exx"], P2["eyy"], P2["exy"]]
S3 = [P3[k] for k in ("exx", "eyy", "ezz", "exy", "eyz", "ezx")]
D2 = [P2[k] for k in ("u", "ux", "uy", "v", "vx", "vy")]
D3 = [P3[k] for k in ("u", "ux", "uy", "uz", "v", "vx", "vy", "vz", "w", "wx", "wy", "wz")]


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def queue2d(n, seed, extent=300.0, garbage=True):
    rng = np.random.default_rng(seed)
    x = (rng.random(n) * extent).astype(np.float32)
    y = (rng.random(n) * extent * 0.8).astype(np.float32)
    p = oracle.make_pois2d(x, y)
    xd, yd = x.astype(np.float64), y.astype(np.float64)
    p[:, P2["u"]] = 1.5 + 2e-3 * xd - 7e-4 * yd + 3e-6 * xd * yd + rng.normal(0, 2e-3, n)
    p[:, P2["v"]] = -0.5 + 4e-4 * xd - 3e-3 * yd + rng.normal(0, 2e-3, n)
    p[:, P2["zncc"]] = rng.uniform(0.85, 1.0, n)          # ~a third below the 0.9 threshold
    if garbage:
        p[::17, P2["zncc"]] = [-3.0, -4.0, -5.0, 0.0][seed % 4]  # failed POIs with garbage displacements
        p[::17, P2["u"]] = 1e3
    p[:, S2] = 77.0                                        # sentinel: untouched fields must stay exactly this
    return p


def queue3d(n, seed, extent=120.0, garbage=True):
    rng = np.random.default_rng(seed)
    x, y, z = [(rng.random(n) * extent * s).astype(np.float32) for s in (1.0, 0.9, 0.8)]
    p = oracle.make_pois3d(x, y, z)
    xd, yd, zd = x.astype(np.float64), y.astype(np.float64), z.astype(np.float64)
    p[:, P3["u"]] = 1.0 + 2e-3 * xd - 7e-4 * yd + 5e-4 * zd + rng.normal(0, 1e-3, n)
    p[:, P3["v"]] = -0.5 + 4e-4 * xd - 3e-3 * yd - 2e-4 * zd + rng.normal(0, 1e-3, n)
    p[:, P3["w"]] = 0.25 - 6e-4 * xd + 1e-3 * yd + 1.5e-3 * zd + rng.normal(0, 1e-3, n)
    p[:, P3["zncc"]] = rng.uniform(0.85, 1.0, n)
    if garbage:
        p[::13, P3["zncc"]] = -4.0
        p[::13, P3["w"]] = -1e3
    p[:, S3] = 77.0
    return p


def compare(got, want, cols, tol):
    """`cols` may differ by `tol`; the same POIs must have been written; everything else must be bit-identical."""
    rest = np.setdiff1d(np.arange(got.shape[1]), cols)
    assert np.array_equal(_bits(got[:, rest]), _bits(want[:, rest]))
    d = np.abs(got[:, cols].astype(np.float64) - want[:, cols].astype(np.float64))
    assert d.max() <= tol, d.max()
    return d.max()


@pytest.mark.parametrize("radius,nmin,approx,seed", [(22.0, 6, 1, 1), (22.0, 6, 2, 2), (9.0, 5, 1, 3), (40.0, 12, 2, 4)])
def test_strain2d_matches_the_reference_sources(radius, nmin, approx, seed):
    p = queue2d(2500, seed)
    want, got = p.copy(), p.copy()
    oref.strain(want, radius, nmin, 0.9, approx)
    oracle.strain2d(got, radius, nmin, 0.9, approx)
    written_ref = (want[:, S2] != 77.0).any(1)
    written = (got[:, S2] != 77.0).any(1)
    assert np.array_equal(written, written_ref)             # same POIs computed (filters, KNN fallback, minimum count)
    assert 0.3 * len(p) < written.sum() < len(p)            # both branches of every filter occur
    assert not written[p[:, P2["zncc"]] < 0.9].any()
    compare(got, want, S2, 1e-6)
    if radius == 9.0:
        # sparse neighbourhoods: the radius search falls short for many POIs and the K-nearest fallback decides
        assert (~written & (p[:, P2["zncc"]] >= 0.9)).sum() > 50


@pytest.mark.parametrize("radius,nmin,approx,seed", [(16.0, 8, 1, 5), (16.0, 8, 2, 6), (8.0, 6, 1, 7)])
def test_strain3d_matches_the_reference_sources(radius, nmin, approx, seed):
    p = queue3d(3000, seed)
    want, got = p.copy(), p.copy()
    oref.strain(want, radius, nmin, 0.9, approx)
    oracle.strain3d(got, radius, nmin, 0.9, approx)
    written_ref = (want[:, S3] != 77.0).any(1)
    written = (got[:, S3] != 77.0).any(1)
    assert np.array_equal(written, written_ref)
    assert 0.1 * len(p) < written.sum() < len(p)
    compare(got, want, S3, 1e-6)


def test_strain_zncc_threshold_and_small_queues():
    """setZnccThreshold reaches both filters; a queue shorter than neighbor_number_min computes nothing (knn returns fewer
    points than asked for, src/oc_nearest_neighbor.cpp:165-167)."""
    p = queue2d(600, 9)
    for thr in (0.0, 0.95):
        want, got = p.copy(), p.copy()
        oref.strain(want, 30.0, 5, thr, 1)
        oracle.strain2d(got, 30.0, 5, thr, 1)
        assert np.array_equal((got[:, S2] != 77.0).any(1), (want[:, S2] != 77.0).any(1))
        compare(got, want, S2, 1e-6)
    tiny = queue2d(4, 10)
    tiny[:, P2["zncc"]] = 0.99
    want, got = tiny.copy(), tiny.copy()
    oref.strain(want, 500.0, 6, 0.9, 1)
    oracle.strain2d(got, 500.0, 6, 0.9, 1)
    assert np.array_equal(_bits(got), _bits(want)) and (got[:, S2] == 77.0).all()


@pytest.mark.parametrize("radius,nmin,seed", [(25.0, 7, 11), (6.0, 5, 12), (60.0, 20, 13)])
def test_region_fit2d_matches_the_reference_sources(radius, nmin, seed):
    reliable = queue2d(1500, seed, garbage=False)
    reliable[:, P2["zncc"]] = 0.99                      # RegionFit does not filter the reliable queue
    rng = np.random.default_rng(seed + 100)
    q = oracle.make_pois2d((rng.random(900) * 340 - 20).astype(np.float32), (rng.random(900) * 280 - 20).astype(np.float32))
    q[:, D2] = 55.0
    q[:, P2["zncc"]] = -4.0
    q[:, P2["u0"]], q[:, P2["iteration"]] = 3.0, 10.0   # must be left alone
    want, got = q.copy(), q.copy()
    oref.region_fit(reliable, want, radius, nmin)
    oracle.region_fit(reliable, got, radius, nmin)
    fitted_ref, fitted = want[:, P2["zncc"]] == 0.0, got[:, P2["zncc"]] == 0.0
    assert np.array_equal(fitted, fitted_ref) and fitted.all()   # the K-nearest fallback always finds nmin reliable POIs
    # gradients within 1e-6; the fitted displacement itself (2 - 3 px, one float ulp = 2.4e-7) within 1e-5: a third of the
    # queries lie outside the reliable cloud (extrapolation) or find 5 neighbours in a 6 px disc -- ill-conditioned fits
    # in which the reference's float QR loses a few more ulps than the oracle's double normal equations
    compare(got, want, D2, 1e-5 if radius > 6 else 5e-5)
    g = [P2[k] for k in ("ux", "uy", "vx", "vy")]
    if radius > 6:
        assert np.abs(got[:, g].astype(np.float64) - want[:, g]).max() <= 1e-6
    assert (got[:, P2["u0"]] == 3.0).all() and (got[:, P2["iteration"]] == 10.0).all()
    # fewer reliable POIs than neighbor_number_min: nothing can be fitted, everything stays as it came
    few = reliable[:nmin - 1].copy()
    want, got = q.copy(), q.copy()
    oref.region_fit(few, want, radius, nmin)
    oracle.region_fit(few, got, radius, nmin)
    assert np.array_equal(_bits(got), _bits(q)) and np.array_equal(_bits(want), _bits(q))


@pytest.mark.parametrize("radius,nmin,seed", [(18.0, 9, 21), (5.0, 6, 22)])
def test_region_fit3d_matches_the_reference_sources(radius, nmin, seed):
    reliable = queue3d(2500, seed, garbage=False)
    reliable[:, P3["zncc"]] = 0.99
    rng = np.random.default_rng(seed + 100)
    q = oracle.make_pois3d(*[(rng.random(700) * e).astype(np.float32) for e in (120.0, 108.0, 96.0)])
    q[:, D3] = 55.0
    q[:, P3["zncc"]] = -3.0
    q[:, P3["w0"]] = 4.0
    want, got = q.copy(), q.copy()
    oref.region_fit(reliable, want, radius, nmin)
    oracle.region_fit(reliable, got, radius, nmin)
    assert np.array_equal(got[:, P3["zncc"]] == 0.0, want[:, P3["zncc"]] == 0.0) and (got[:, P3["zncc"]] == 0.0).all()
    compare(got, want, D3, 1e-5 if radius > 5 else 1e-4)
    g = [P3[k] for k in ("ux", "uy", "uz", "vx", "vy", "vz", "wx", "wy", "wz")]
    if radius > 5:
        assert np.abs(got[:, g].astype(np.float64) - want[:, g]).max() <= 1e-6
    assert (got[:, P3["w0"]] == 4.0).all()
