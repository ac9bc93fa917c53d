"""The fused arithmetic contract on the GPU (`oc_hip_set_tuning("arith_fma", 1)`, round 5).

Every per-sample multiply-add of ICGN2D1 / ICGN2D2 / ICLM2D1 / ICLM2D2 / ICGN3D1 becomes ONE fused multiply-add -- the
contraction a compiler with FMA hardware makes of the reference's source expressions (src/oc_cubic_bspline.cpp:159-177,
390-401; src/oc_icgn.cpp:198-205, 266-276, 1314-1445; the reference's build files fix no contraction mode).  The oracle
restates the same fused sites with explicit fmaf (oracle/oc_oracle.h "Arithmetic contract", OC_ORDER_LANES_FMA); IEEE
fusedMultiplyAdd is defined bit for bit, so the kernels must equal it in EVERY bit -- for every kernel variant, with
centre offsets, self-adaptive radii, early leavers, the golden OHT pair, and across a device group.  What the fused
results look like next to the reference's separately rounded loop order is asserted in tests/test_gpu_fullsize.py
(all BASELINE configs) and on CPU in tests/test_order_tolerance.py.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.fixture(scope="module")
def eng():
    import opencorr_amd
    return opencorr_amd


@pytest.fixture(scope="module")
def case2d(speckle_small):
    import oracle
    from opencorr_amd import synth
    ref, tar = speckle_small
    xs, ys = synth.poi_grid_2d(ref.shape[0], ref.shape[1], 19, 23, 26)
    pois = oracle.make_pois2d(xs, ys)
    oracle.fftcc2d(ref, tar, 16, 16, pois)
    P = oracle.P2
    extra = oracle.make_pois2d([3.0, 90.0, 90.0, 90.0], [80.0, 80.0, 80.0, 80.0])
    extra[1, P["u"]] = 200.0      # leaves the image inside the loop: -3
    extra[2, P["zncc"]] = -1.0    # rejected on entry
    extra[3, P["v"]] = np.nan
    pois = np.concatenate([extra[:2], pois, extra[2:]]).astype(np.float32)
    return ref, tar, pois, oracle.Prepared2D(ref, tar)


@pytest.mark.parametrize("variant", [-1, 1, 2, 3, 4, 5, 7])
@pytest.mark.parametrize("dof", [6, 12])
def test_icgn2d_fma_every_variant_equals_oracle_lanes_fma(eng, case2d, variant, dof):
    import oracle
    ref, tar, pois, prep = case2d
    r = 16 if dof == 6 else 12
    fn = oracle.icgn2d1 if dof == 6 else oracle.icgn2d2
    want = pois.copy()
    fn(prep, r, r, 0.001, 10, want, order=oracle.ORDER_LANES_FMA, lanes=64)
    sep = pois.copy()
    fn(prep, r, r, 0.001, 10, sep, order=oracle.ORDER_LANES, lanes=64)
    assert not np.array_equal(_bits(want), _bits(sep))          # the contract does change bits ...
    ok = (want[:, 16] >= 0) & (sep[:, 16] >= 0) & (want[:, 17] == sep[:, 17])
    assert ok.sum() > 0.95 * len(pois)
    assert np.abs(want[ok][:, [2, 8]] - sep[ok][:, [2, 8]]).max() <= 1e-4   # ... by rounding only
    icgn = (eng.ICGN2D1 if dof == 6 else eng.ICGN2D2)(r, r, 0.001, 10)
    icgn.set_images(ref, tar)
    icgn.prepare()
    icgn.set_tuning("icgn2d_variant", variant)
    icgn.set_tuning("arith_fma", 1)
    assert np.array_equal(_bits(icgn.compute(pois.copy())), _bits(want))
    icgn.set_tuning("arith_fma", 0)                              # and back: the default build, the default oracle order
    assert np.array_equal(_bits(icgn.compute(pois.copy())), _bits(sep))


@pytest.mark.parametrize("dof", [6, 12])
def test_icgn2d_fma_offsets_self_adaptive_rectangular(eng, case2d, dof):
    import oracle
    ref, tar, pois, prep = case2d
    rx, ry = (13, 9) if dof == 6 else (10, 12)
    fn = oracle.icgn2d1 if dof == 6 else oracle.icgn2d2
    icgn = (eng.ICGN2D1 if dof == 6 else eng.ICGN2D2)(rx, ry, 0.001, 10)
    icgn.set_images(ref, tar)
    icgn.prepare()
    icgn.set_tuning("arith_fma", 1)
    off = np.random.default_rng(5).uniform(-2, 2, (len(pois), 2)).astype(np.float32)
    want = pois.copy()
    fn(prep, rx, ry, 0.001, 10, want, order=oracle.ORDER_LANES_FMA, lanes=64, center_offsets=off)
    assert np.array_equal(_bits(icgn.compute_with_offsets(pois.copy(), off)), _bits(want))
    sa = pois.copy()
    P = oracle.P2
    sa[:, P["srx"]] = np.random.default_rng(1).integers(6, rx + 1, len(sa))
    sa[:, P["sry"]] = np.random.default_rng(2).integers(6, ry + 1, len(sa))
    want = sa.copy()
    fn(prep, rx, ry, 0.001, 10, want, order=oracle.ORDER_LANES_FMA, lanes=64, self_adaptive=True)
    icgn.set_self_adaptive(True)
    assert np.array_equal(_bits(icgn.compute(sa.copy())), _bits(want))


def test_icgn2d1_fma_large_queue_tile_schedule_and_lockstep(eng, speckle_small):
    """A queue long enough for the default launch shape of the bench line (variant 5: coordinate table, lockstep sweeps,
    cooperative inverse, tile-ordered visiting) under the fused contract."""
    import oracle
    from opencorr_amd import synth
    ref, tar = speckle_small
    h, w = ref.shape
    xs, ys = synth.poi_grid_2d(h, w, 210, 160, 24)   # 33 600 POIs >= 32 768
    pois = oracle.make_pois2d(xs, ys)
    oracle.fftcc2d(ref, tar, 16, 16, pois)
    icgn = eng.ICGN2D1(16, 16, 0.001, 10)
    icgn.set_images(ref, tar)
    icgn.prepare()
    icgn.set_tuning("arith_fma", 1)
    got = icgn.compute(pois.copy())
    want = pois.copy()
    oracle.icgn2d1(oracle.Prepared2D(ref, tar), 16, 16, 0.001, 10, want, order=oracle.ORDER_LANES_FMA, lanes=64)
    assert np.array_equal(_bits(got), _bits(want))


@pytest.mark.parametrize("dof", [6, 12])
def test_iclm2d_fma(eng, case2d, dof):
    import oracle
    ref, tar, pois, prep = case2d
    r = 12
    fn = oracle.iclm2d1 if dof == 6 else oracle.iclm2d2
    want = pois.copy()
    fn(prep, r, r, 0.001, 10, want, order=oracle.ORDER_LANES_FMA, lanes=64)
    lm = (eng.ICLM2D1 if dof == 6 else eng.ICLM2D2)(r, r, 0.001, 10)
    lm.set_images(ref, tar)
    lm.prepare()
    lm.set_tuning("arith_fma", 1)
    assert np.array_equal(_bits(lm.compute(pois.copy())), _bits(want))


def test_golden_oht_on_gpu_fma(eng, golden):
    """The reference's own OHT example under the fused contract: same bars against its golden table as the default
    build (|d u|, |d v| <= 2e-4 px and |d ZNCC| <= 1e-5 on the POIs the CSV shows converged, >= 99 % equal
    iteration counts), and bit-identical to the oracle in OC_ORDER_LANES_FMA."""
    import oracle
    P = oracle.P2
    tab = golden["table"]   # x y u v u0 v0 zncc iteration convergence
    pois = oracle.make_pois2d(tab[:, 0], tab[:, 1])
    f = eng.FFTCC2D(golden["rx"], golden["ry"])
    f.set_images(golden["ref"], golden["tar"])
    f.compute(pois)
    same = (pois[:, P["u"]] == tab[:, 4]) & (pois[:, P["v"]] == tab[:, 5])
    g = eng.ICGN2D1(golden["rx"], golden["ry"], golden["conv"], golden["stop"])
    g.share_images(f)
    g.prepare()
    g.set_tuning("arith_fma", 1)
    want = pois.copy()
    g.compute(pois)
    oracle.icgn2d1(oracle.Prepared2D(golden["ref"], golden["tar"]), golden["rx"], golden["ry"], golden["conv"], golden["stop"], want,
                   order=oracle.ORDER_LANES_FMA, lanes=64)
    assert np.array_equal(_bits(pois), _bits(want))
    m = (tab[:, 7] < golden["stop"]) & same
    assert m.sum() > 28000
    assert np.abs(pois[m, P["u"]] - tab[m, 2]).max() <= 2e-4 and np.abs(pois[m, P["v"]] - tab[m, 3]).max() <= 2e-4
    assert np.abs(pois[m, P["zncc"]] - tab[m, 6]).max() <= 1e-5
    assert (pois[m, P["iteration"]] == tab[m, 7]).mean() >= 0.99


@pytest.mark.parametrize("r", [5, 16])
def test_icgn3d1_fma(eng, r):
    import oracle
    from opencorr_amd import synth
    dim = 2 * (r + 8) + 30
    ref, tar = synth.speckle_pair_3d(dim, dim, dim, seed=20260927)
    xs, ys, zs = synth.poi_grid_3d(dim, dim, dim, 3, 3, 3, r + 8)
    pois = oracle.make_pois3d(xs, ys, zs)
    P = oracle.P3
    w = synth.DEFAULT_WARP_3D
    pois[:, P["u"]], pois[:, P["v"]], pois[:, P["w"]] = round(w["u"]), round(w["v"]), round(w["w"])
    extra = oracle.make_pois3d([dim // 2, dim // 2], [dim // 2, dim // 2], [dim // 2, dim // 2])
    extra[0, P["u"]] = 90.0
    extra[1, P["zncc"]] = -2.0
    pois = np.concatenate([pois, extra]).astype(np.float32)
    prep = oracle.Prepared3D(ref, tar)
    want = pois.copy()
    oracle.icgn3d1(prep, r, r, r, 0.001, 20.0, want, order=oracle.ORDER_LANES_FMA, lanes=512)
    sep = pois.copy()
    oracle.icgn3d1(prep, r, r, r, 0.001, 20.0, sep, order=oracle.ORDER_LANES, lanes=512)
    assert not np.array_equal(_bits(want), _bits(sep))
    icgn = eng.ICGN3D1(r, r, r, 0.001, 20.0)
    icgn.set_images(ref, tar)
    icgn.prepare()
    icgn.set_tuning("arith_fma", 1)
    got = icgn.compute(pois.copy())
    mism = np.argwhere(_bits(got) != _bits(want))
    assert mism.size == 0, mism[:10].tolist()
    assert (got[:27, P["zncc"]] > 0.9).all()
    icgn.set_tuning("arith_fma", 0)
    assert np.array_equal(_bits(icgn.compute(pois.copy())), _bits(sep))


def test_arith_fma_travels_through_a_device_group_and_is_refused_elsewhere(eng, case2d):
    import oracle
    ref, tar, pois, prep = case2d
    want = pois.copy()
    oracle.icgn2d1(prep, 16, 16, 0.001, 10, want, order=oracle.ORDER_LANES_FMA, lanes=64)
    icgn = eng.ICGN2D1(16, 16, 0.001, 10)
    icgn.set_devices([0, 0, 0])
    icgn.set_tuning("arith_fma", 1)       # fans out over the members
    icgn.set_images(ref, tar)
    icgn.prepare()
    assert np.array_equal(_bits(icgn.compute(pois.copy())), _bits(want))
    icgn2 = eng.ICGN2D1(16, 16, 0.001, 10)
    icgn2.set_tuning("arith_fma", 1)      # set BEFORE the group is formed: the clones inherit it
    icgn2.set_devices([0, 0])
    icgn2.set_images(ref, tar)
    icgn2.prepare()
    assert np.array_equal(_bits(icgn2.compute(pois.copy())), _bits(want))
    for make in (lambda: eng.FFTCC2D(16, 16), lambda: eng.NR2D1(16, 16, 0.001, 10)):
        e = make()
        with pytest.raises(Exception):
            e.set_tuning("arith_fma", 1)
        e.set_tuning("arith_fma", 0)
