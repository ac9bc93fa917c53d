"""GPU parity of ICLM2D1 / ICLM2D2 (inverse-compositional Levenberg-Marquardt) against the oracle.
Bars: bit-exact vs the oracle in OC_ORDER_LANES (same flags, iteration counts and float bits), including the
damping schedule; on the OHT example the same acceptance as the oracle's own test."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _assert_same(got, want):
    both_nan = np.isnan(got) & np.isnan(want)  # NaN payload bits are not part of the contract
    mism = np.argwhere((_bits(got) != _bits(want)) & ~both_nan)
    assert mism.size == 0, "first mismatches (poi, field): %s" % mism[:10].tolist()


def _queue(ref, tar, rx, ry):
    import oracle
    from opencorr_amd import synth
    h, w = ref.shape
    xs, ys = synth.poi_grid_2d(h, w, 21, 17, 28)
    pois = oracle.make_pois2d(xs, ys)
    oracle.fftcc2d(ref, tar, rx, ry, pois)
    # edge cases: outside the guard (-3); a wild guess whose warped subset leaves the image (IC-LM has no abort: the
    # samples enter as -1.f); negative ZNCC on entry (kept); NaN guess (-3 by the guard); a guess 1.5 px off (steps get
    # rejected and lambda grows); a subset touching the interpolation border
    extra = oracle.make_pois2d([5, 160, 160, 160, 160, 30], [150, 150, 150, 150, 150, 30])
    P = oracle.P2
    extra[1, P["u"]] = 140.0
    extra[2, P["zncc"]] = -2.5
    extra[3, P["u"]] = np.nan
    extra[4, P["u"]], extra[4, P["v"]] = 3.8, -0.2
    extra[5, P["u"]], extra[5, P["v"]] = -14.5, -13.0
    return np.concatenate([pois, extra]).astype(np.float32)


@pytest.mark.parametrize("dof,rx,ry,damping", [(6, 16, 16, None), (6, 9, 12, (10.0, 0.5, 4.0)), (6, 16, 16, (1.0, 0.1, 10.0)),
                                               (12, 16, 16, None), (12, 12, 10, (1000.0, 0.2, 5.0))])
def test_iclm2d_bit_exact_vs_oracle(speckle_small, dof, rx, ry, damping):
    import opencorr_amd as eng
    import oracle
    ref, tar = speckle_small
    pois = _queue(ref, tar, rx, ry)
    want = pois.copy()
    prep = oracle.Prepared2D(ref, tar)
    ofn = oracle.iclm2d1 if dof == 6 else oracle.iclm2d2
    ofn(prep, rx, ry, 0.001, 10, want, damping=damping or oracle.DEFAULT_DAMPING, order=oracle.ORDER_LANES, lanes=64)
    lm = (eng.ICLM2D1 if dof == 6 else eng.ICLM2D2)(rx, ry, 0.001, 10)
    lm.set_images(ref, tar)
    lm.prepare()
    if damping:
        lm.set_damping(*damping)
    got = lm.compute(pois.copy())
    _assert_same(got, want)
    P = oracle.P2
    assert want[-6, P["zncc"]] == -3.0 and want[-4, P["zncc"]] == -2.5 and want[-3, P["zncc"]] == -3.0
    # with the default damping the regular grid converges; heavy damping legitimately runs into the iteration
    # limit (-4), which is reference behaviour and covered by the bit comparison above
    if damping is None:
        assert (want[:-6, P["zncc"]] > 0.9).mean() > 0.9


@pytest.mark.parametrize("dof", [6, 12])
def test_iclm2d_self_adaptive_radius_bit_exact(speckle_small, dof):
    import opencorr_amd as eng
    import oracle
    ref, tar = speckle_small
    pois = _queue(ref, tar, 16, 16)[:-6]
    rng = np.random.default_rng(11)
    P = oracle.P2
    pois[:, P["srx"]] = rng.integers(6, 19, len(pois))
    pois[:, P["sry"]] = rng.integers(6, 19, len(pois))
    want = pois.copy()
    ofn = oracle.iclm2d1 if dof == 6 else oracle.iclm2d2
    ofn(oracle.Prepared2D(ref, tar), 16, 16, 0.001, 10, want, order=oracle.ORDER_LANES, lanes=64, self_adaptive=True)
    lm = (eng.ICLM2D1 if dof == 6 else eng.ICLM2D2)(16, 16, 0.001, 10)
    lm.set_images(ref, tar)
    lm.prepare()
    lm.set_self_adaptive(True)
    got = lm.compute(pois.copy())
    _assert_same(got, want)


def test_iclm2d1_on_the_oht_example(golden):
    import opencorr_amd as eng
    from test_oracle_golden import iclm1_golden_check
    tab = golden["table"]
    fftcc = eng.FFTCC2D(golden["rx"], golden["ry"])
    fftcc.set_images(golden["ref"], golden["tar"])
    pois = eng.make_pois2d(tab[:, 0], tab[:, 1])
    fftcc.compute(pois)
    lm = eng.ICLM2D1(golden["rx"], golden["ry"], golden["conv"], golden["stop"])
    lm.share_images(fftcc)
    lm.prepare()
    lm.compute(pois)
    iclm1_golden_check(pois, tab, golden["stop"])


def test_set_damping_rejects_what_the_engine_cannot_honour(speckle_small):
    import opencorr_amd as eng
    lm = eng.ICLM2D1(16, 16, 0.001, 10)
    with pytest.raises(eng.capi.OpenCorrHipError):
        lm.set_damping(-1.0, 0.1, 10.0)
    icgn = eng.ICGN2D1(16, 16, 0.001, 10)
    with pytest.raises(eng.capi.OpenCorrHipError):
        eng.capi.check(eng.capi.lib().oc_hip_set_damping(icgn._h, 100.0, 0.1, 10.0))
