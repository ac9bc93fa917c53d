"""The oracle against the REFERENCE ITSELF: oracle/_ref/liboc_ref.so is /root/reference/src/*.cpp compiled unmodified
(oracle/Makefile `ref`, stand-in Eigen / FFTW / OpenCV headers under oracle/ref_stubs).

Bars
  * every solver -- ICGN2D1, ICGN2D2, the centre-offset overloads, self-adaptive subsets, ICLM2D1 / ICLM2D2, NR2D1,
    ICGN3D1: oracle(OC_ORDER_SEQ) == reference, BIT FOR BIT, every float of every POI record, including rejected,
    out-of-range, non-converged and NaN POIs;
  * gradients and interpolators: bit for bit;
  * FFTCC2D / FFTCC3D: integer displacements identical, ZNCC within 1e-5 (2D) / 1e-4 (3D) -- the correlation surface
    passes through the FFT, whose internal arithmetic differs (reference: FFTW -> here a double DFT rounded to a
    float spectrum; oracle: double throughout), see oracle/ref.py.
Skipped where the reference tree is not mounted (the GPU box).
"""
import numpy as np
import pytest

import oracle
from oracle import ref as oref

pytestmark = pytest.mark.skipif(not oref.available(), reason="reference tree not mounted: oracle/_ref cannot be built")


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _same(got, want):
    mism = np.argwhere(_bits(got) != _bits(want))
    assert mism.size == 0, "first mismatches (poi, field): %s" % mism[:10].tolist()


@pytest.fixture(scope="module")
def pair():
    from opencorr_amd import synth
    ref, tar = synth.speckle_pair_2d(160, 176, seed=5, second_order=dict(uxx=2e-5, vyy=-1e-5))
    return ref, tar, oracle.Prepared2D(ref, tar)


def _queue(ref, tar, r, extra=True):
    from opencorr_amd import synth
    h, w = ref.shape
    xs, ys = synth.poi_grid_2d(h, w, 6, 5, r + 10)
    pois = oracle.make_pois2d(xs, ys)
    oracle.fftcc2d(ref, tar, r, r, pois)
    if extra:
        P = oracle.P2
        e = oracle.make_pois2d([3.0, 80.0, 80.0, 80.0, 80.5, w - 14.0], [70.0, 70.0, 70.0, 70.0, 70.25, 90.0])
        e[1, P["u"]] = 120.0       # warped subset leaves the image -> -3 inside the loop
        e[2, P["zncc"]] = -2.0     # rejected on entry, flag preserved
        e[3, P["v"]] = np.nan
        e[4, P["u"]], e[4, P["v"]] = 2.0, -2.0   # non-integer POI position
        e[5, P["u"]], e[5, P["ux"]] = 2.0, 0.3   # large gradient guess next to the border: partly out of range
        pois = np.concatenate([pois, e]).astype(np.float32)
    return pois


def test_gradient_and_bicubic_interpolation(pair):
    ref, tar, prep = pair
    gx, gy = oref.gradient2d(ref)
    _same(prep.gx, gx)
    _same(prep.gy, gy)
    rng = np.random.default_rng(1)
    h, w = tar.shape
    xy = np.stack([rng.uniform(-2, w + 2, 4000), rng.uniform(-2, h + 2, 4000)], 1).astype(np.float32)
    xy[:8] = [[1, 1], [w - 2, 5], [w - 2.0001, 5], [5, h - 2], [0.9999, 5], [np.nan, 5], [5, np.inf], [1, h - 2.001]]
    want = oref.bspline2d_eval(tar, xy)
    got = np.array([oracle.bspline2d_eval(prep.lut, x, y) for x, y in xy], dtype=np.float32)
    _same(got, want)
    assert (want == -1).sum() > 8 and (want > 0).sum() > 3000


@pytest.mark.parametrize("r", [(16, 16), (9, 12)])
def test_fftcc2d_against_reference(pair, r):
    from opencorr_amd import synth
    ref, tar, _ = pair
    rx, ry = r
    h, w = ref.shape
    xs, ys = synth.poi_grid_2d(h, w, 7, 6, 20)
    want = oracle.make_pois2d(np.concatenate([xs, [5.0, w - 3.0]]), np.concatenate([ys, [40.0, 40.0]]))  # + guarded POIs
    want[3, oracle.P2["u"]] = 3.0   # an integer initial guess displaces the target window
    got = want.copy()
    oref.fftcc2d(ref, tar, rx, ry, want)
    oracle.fftcc2d(ref, tar, rx, ry, got)
    P = oracle.P2
    for key in ("u", "v", "u0", "v0"):
        assert np.array_equal(got[:, P[key]], want[:, P[key]]), key
    assert np.abs(got[:, P["zncc"]] - want[:, P["zncc"]]).max() <= 1e-5
    rest = [c for c in range(25) if c not in (P["u"], P["v"], P["u0"], P["v0"], P["zncc"])]
    _same(got[:, rest], want[:, rest])
    assert (want[:-2, P["zncc"]] > 0.3).mean() > 0.9 and (want[-2:, P["zncc"]] == 0).all()


@pytest.mark.parametrize("engine,r", [("icgn2d1", 16), ("icgn2d1", 7), ("icgn2d2", 12)])
def test_icgn2d_bit_exact_against_reference(pair, engine, r):
    ref, tar, prep = pair
    pois = _queue(ref, tar, r)
    want = pois.copy()
    got = pois.copy()
    oref.solve2d(oref.ICGN2D1 if engine == "icgn2d1" else oref.ICGN2D2, ref, tar, r, r, 0.001, 10, want)
    getattr(oracle, engine)(prep, r, r, 0.001, 10, got, order=oracle.ORDER_SEQ)
    _same(got, want)
    P = oracle.P2
    assert (want[:30, P["zncc"]] > 0.9).mean() > 0.7  # r = 7: a 14 x 14 FFTCC window misses some guesses
    assert want[-5, P["zncc"]] == -3 and want[-4, P["zncc"]] == -2 and want[-3, P["zncc"]] == -3


def test_icgn2d_stop_condition_and_minus4(pair):
    """stop = 2 leaves most POIs unconverged: zncc = -4 with the last iterate kept (src/oc_icgn.cpp:329-332)."""
    ref, tar, prep = pair
    pois = _queue(ref, tar, 12, extra=False)
    want, got = pois.copy(), pois.copy()
    oref.solve2d(oref.ICGN2D1, ref, tar, 12, 12, 1e-6, 2, want)
    oracle.icgn2d1(prep, 12, 12, 1e-6, 2, got, order=oracle.ORDER_SEQ)
    _same(got, want)
    assert (want[:, oracle.P2["zncc"]] == -4).sum() > 10


@pytest.mark.parametrize("engine", ["icgn2d1", "icgn2d2"])
def test_center_offset_overloads_against_reference(pair, engine):
    """compute(poi_queue, center_offset_queue), src/oc_icgn.cpp:353-557 / 910-1136."""
    ref, tar, prep = pair
    r = 10
    pois = _queue(ref, tar, r)
    rng = np.random.default_rng(3)
    off = rng.uniform(-2.5, 2.5, (len(pois), 2)).astype(np.float32)
    off[::4] = np.round(off[::4])
    want, got = pois.copy(), pois.copy()
    oref.solve2d(oref.ICGN2D1 if engine == "icgn2d1" else oref.ICGN2D2, ref, tar, r, r, 0.001, 10, want, center_offsets=off)
    getattr(oracle, engine)(prep, r, r, 0.001, 10, got, order=oracle.ORDER_SEQ, center_offsets=off)
    _same(got, want)


@pytest.mark.parametrize("engine", ["icgn2d1", "icgn2d2"])
def test_self_adaptive_subsets_against_reference(pair, engine):
    """DIC::setSelfAdaptive(true): per-POI radius from poi.subset_radius (src/oc_icgn.cpp:152-158)."""
    ref, tar, prep = pair
    pois = _queue(ref, tar, 14)
    P = oracle.P2
    rng = np.random.default_rng(4)
    pois[:, P["srx"]] = rng.integers(6, 15, len(pois))
    pois[:, P["sry"]] = rng.integers(6, 15, len(pois))
    want, got = pois.copy(), pois.copy()
    oref.solve2d(oref.ICGN2D1 if engine == "icgn2d1" else oref.ICGN2D2, ref, tar, 14, 14, 0.001, 10, want, self_adaptive=True)
    getattr(oracle, engine)(prep, 14, 14, 0.001, 10, got, order=oracle.ORDER_SEQ, self_adaptive=True)
    _same(got, want)


@pytest.mark.parametrize("dof,damping", [(6, (100.0, 0.1, 10.0)), (12, (100.0, 0.1, 10.0)), (6, (1.0, 0.5, 2.0))])
def test_iclm2d_against_reference(pair, dof, damping):
    """ICLM2D1 / ICLM2D2 (src/oc_iclm.cpp).  The first damping value is powf(lambda, znssd / 4) - 1: glibc's powf here,
    a fixed double sequence in the oracle -- they agree except for one ulp in ~0.05 % of arguments, so POIs whose
    trajectory forks on that ulp are allowed to differ (none does on this queue; the bar is: at most one)."""
    ref, tar, prep = pair
    r = 11
    pois = _queue(ref, tar, r)
    want, got = pois.copy(), pois.copy()
    oref.solve2d(oref.ICLM2D1 if dof == 6 else oref.ICLM2D2, ref, tar, r, r, 0.001, 10, want, damping=damping)
    (oracle.iclm2d1 if dof == 6 else oracle.iclm2d2)(prep, r, r, 0.001, 10, got, damping=damping, order=oracle.ORDER_SEQ)
    bad = np.unique(np.argwhere(_bits(got) != _bits(want))[:, 0])
    assert len(bad) <= 1, "POIs that differ: %s" % bad.tolist()


def test_nr2d1_against_reference(pair):
    ref, tar, _ = pair
    r = 12
    pois = _queue(ref, tar, r)
    want, got = pois.copy(), pois.copy()
    oref.solve2d(oref.NR2D1, ref, tar, r, r, 0.001, 10, want)
    oracle.nr2d1(oracle.PreparedNR2D(ref, tar), r, r, 0.001, 10, got, order=oracle.ORDER_SEQ)
    _same(got, want)


# ---- DVC ------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def volumes():
    from opencorr_amd import synth
    ref, tar = synth.speckle_pair_3d(56, 60, 64, seed=9)
    return ref, tar, oracle.Prepared3D(ref, tar)


def test_gradient3d_and_tricubic_interpolation(volumes):
    ref, tar, prep = volumes
    rng = np.random.default_rng(2)
    dz, dy, dx = tar.shape
    xyz = np.stack([rng.uniform(-1, dx + 1, 3000), rng.uniform(-1, dy + 1, 3000), rng.uniform(-1, dz + 1, 3000)], 1).astype(np.float32)
    xyz[:4] = [[1, 1, 1], [dx - 2, 5, 5], [5, 5, dz - 2.001], [np.nan, 3, 3]]
    gx, gy, gz, want = oref.prepare3d(ref, tar, xyz)
    _same(prep.gx, gx)
    _same(prep.gy, gy)
    _same(prep.gz, gz)
    got = np.array([oracle.bspline3d_eval(prep.coef, *p) for p in xyz], dtype=np.float32)
    _same(got, want)
    assert (want == -1).sum() > 3 and (want > 0).sum() > 2000


@pytest.mark.parametrize("r", [(8, 8, 8), (5, 7, 4), (6, 4, 8), (4, 6, 7)])
def test_fftcc3d_against_reference(volumes, r):
    """Unequal radii too: the reference plans fftwf_plan_dft_r2c_3d(2rx, 2ry, 2rz) over a buffer it fills x-fastest
    (src/oc_fftcc.cpp:68-70, 349-360), i.e. it correlates the window's floats read as an array [2rx][2ry][2rz] and decodes the
    peak's buffer position as a window position (:401-403).  Its own code runs here (on the stand-in FFTW, which implements the
    documented row-major meaning of n0, n1, n2); the oracle, the rocFFT pipeline and fftcc3d_box.hip reproduce that result."""
    from opencorr_amd import synth
    ref, tar, _ = volumes
    xs, ys, zs = synth.poi_grid_3d(*ref.shape, 3, 3, 2, 20)
    want = oracle.make_pois3d(xs, ys, zs)
    want[2, oracle.P3["w"]] = 2.0
    got = want.copy()
    oref.fftcc3d(ref, tar, r[0], r[1], r[2], want)
    oracle.fftcc3d(ref, tar, r[0], r[1], r[2], got)
    P = oracle.P3
    for key in ("u", "v", "w", "u0", "v0", "w0"):
        assert np.array_equal(got[:, P[key]], want[:, P[key]]), key
    assert np.abs(got[:, P["zncc"]] - want[:, P["zncc"]]).max() <= 1e-4
    if r[0] == r[2]:   # (a reshaped window correlates worse: no statement about its peak height)
        assert (want[:, P["zncc"]] > 0.5).mean() > 0.9


@pytest.mark.parametrize("r", [(6, 6, 6), (5, 7, 4)])
def test_icgn3d1_bit_exact_against_reference(volumes, r):
    from opencorr_amd import synth
    ref, tar, prep = volumes
    rx, ry, rz = r
    xs, ys, zs = synth.poi_grid_3d(*ref.shape, 3, 2, 2, 18)
    pois = oracle.make_pois3d(xs, ys, zs)
    oracle.fftcc3d(ref, tar, 8, 8, 8, pois)
    P = oracle.P3
    extra = oracle.make_pois3d([2, 30, 30, 30, 30.5], [28, 28, 28, 28, 28.25], [26, 26, 26, 26, 26.75])
    extra[1, P["u"]] = 50.0
    extra[2, P["zncc"]] = -1.0
    extra[3, P["w"]] = np.nan
    extra[4, P["u"]], extra[4, P["v"]], extra[4, P["w"]] = 2.0, -2.0, 1.0
    pois = np.concatenate([pois, extra]).astype(np.float32)
    want, got = pois.copy(), pois.copy()
    oref.icgn3d1(ref, tar, rx, ry, rz, 0.001, 20, want)
    oracle.icgn3d1(prep, rx, ry, rz, 0.001, 20, got, order=oracle.ORDER_SEQ)
    _same(got, want)
    assert (want[:12, P["zncc"]] > 0.9).all()


@pytest.mark.parametrize("seed", range(4))
def test_fuzz_2d_solvers_against_reference(seed):
    """Randomised: image size, rx != ry, limits, POIs at NON-INTEGER positions, on the guard's borders, with first-order
    guesses, NaN / rejected / far-off records -- the reference's own sources and the oracle (sequential order) must agree
    on every bit of every record, for all five 2D solvers."""
    from opencorr_amd import synth
    rng = np.random.default_rng(7000 + seed)
    h, w = int(rng.integers(120, 180)), int(rng.integers(120, 180))
    warp = dict(u=float(rng.uniform(-2, 2)), ux=float(rng.uniform(-3e-3, 3e-3)), uy=float(rng.uniform(-3e-3, 3e-3)),
                v=float(rng.uniform(-2, 2)), vx=float(rng.uniform(-3e-3, 3e-3)), vy=float(rng.uniform(-3e-3, 3e-3)))
    ref, tar = synth.speckle_pair_2d(h, w, seed=800 + seed, warp=warp)
    rx, ry = int(rng.integers(4, 14)), int(rng.integers(4, 14))
    conv, stop = float(rng.choice([1e-3, 1e-4])), float(rng.choice([10, 5]))
    P = oracle.P2
    n = 60
    m = max(rx, ry) + 3
    xs = rng.uniform(m, w - 1 - m, n).astype(np.float32)
    ys = rng.uniform(m, h - 1 - m, n).astype(np.float32)
    xs[:15], ys[:15] = np.round(xs[:15]), np.round(ys[:15])
    xs[-4:] = np.array([rx - 0.5, rx, w - 1 - rx, w - 1 - rx + 0.5], np.float32)
    ys[-4:] = np.array([ry + 2, ry - 0.25, h - 1 - ry, h - 2 - ry], np.float32)
    pois = oracle.make_pois2d(xs, ys)
    pois[:, P["u"]] = warp["u"] + rng.normal(0, 0.4, n)
    pois[:, P["v"]] = warp["v"] + rng.normal(0, 0.4, n)
    for k in ("ux", "uy", "vx", "vy"):
        pois[:, P[k]] = rng.normal(0, 0.01, n)
    pois[:, P["zncc"]] = rng.uniform(0, 1, n)
    pois[20, P["u"]] = np.nan
    pois[21, P["zncc"]] = -1.0
    pois[22, P["u"]] += 8.0
    pois[23, P["v"]] = 1e6
    pois = pois.astype(np.float32)
    prep = oracle.Prepared2D(ref, tar)
    for eng, fn in ((oref.ICGN2D1, oracle.icgn2d1), (oref.ICGN2D2, oracle.icgn2d2)):
        want, got = pois.copy(), pois.copy()
        oref.solve2d(eng, ref, tar, rx, ry, conv, stop, want)
        fn(prep, rx, ry, conv, stop, got, order=oracle.ORDER_SEQ)
        _same(got, want)
    dmp = (100.0, 0.1, 10.0)
    for eng, fn in ((oref.ICLM2D1, oracle.iclm2d1), (oref.ICLM2D2, oracle.iclm2d2)):
        want, got = pois.copy(), pois.copy()
        oref.solve2d(eng, ref, tar, rx, ry, conv, stop, want, damping=dmp)
        fn(prep, rx, ry, conv, stop, got, damping=dmp, order=oracle.ORDER_SEQ)
        _same(got, want)
    want, got = pois.copy(), pois.copy()
    oref.solve2d(oref.NR2D1, ref, tar, rx, ry, conv, stop, want)
    oracle.nr2d1(oracle.PreparedNR2D(ref, tar), rx, ry, conv, stop, got, order=oracle.ORDER_SEQ)
    _same(got, want)


@pytest.mark.parametrize("seed", range(2))
def test_fuzz_icgn3d1_against_reference(seed):
    """Randomised DVC case: volume shape, rx != ry != rz, float POI positions, noisy guesses, NaN / rejected records."""
    from opencorr_amd import synth
    rng = np.random.default_rng(7100 + seed)
    dz, dy, dx = (int(rng.integers(36, 46)) for _ in range(3))
    ref, tar = synth.speckle_pair_3d(dz, dy, dx, seed=850 + seed)
    rx, ry, rz = (int(rng.integers(3, 7)) for _ in range(3))
    P = oracle.P3
    n = 24
    m = max(rx, ry, rz) + 4
    pois = oracle.make_pois3d(rng.uniform(m, dx - 1 - m, n).astype(np.float32), rng.uniform(m, dy - 1 - m, n).astype(np.float32),
                              rng.uniform(m, dz - 1 - m, n).astype(np.float32))
    w3 = synth.DEFAULT_WARP_3D
    for k in ("u", "v", "w"):
        pois[:, P[k]] = w3[k] + rng.normal(0, 0.3, n)
    pois[2, P["u"]] = np.nan
    pois[3, P["zncc"]] = -2.0
    pois[4, P["w"]] += 6.0
    pois[5, P["x"]] = rx - 0.5
    pois = pois.astype(np.float32)
    stop = float(rng.choice([20, 6]))
    want, got = pois.copy(), pois.copy()
    oref.icgn3d1(ref, tar, rx, ry, rz, 0.001, stop, want)
    oracle.icgn3d1(oracle.Prepared3D(ref, tar), rx, ry, rz, 0.001, stop, got, order=oracle.ORDER_SEQ)
    _same(got, want)
    assert (want[:, P["zncc"]] > 0.5).sum() > 8
