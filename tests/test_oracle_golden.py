"""Pins the CPU oracle against the reference's own golden vectors.

examples/2d_dic/oht_cfrp_4_fftcc_icgn1_r16.csv + _deformation.csv were written by the
reference (examples/test_2d_dic_fftcc_icgn1.cpp: FFTCC2D -> ICGN2D1, r = 16, conv 1e-3,
stop 10, 30 000 POIs) and are stored, with the two input images, in tests/golden/.
Acceptance: FFTCC u0,v0 identical on >= 99.9 % of POIs; on POIs the CSV shows
converged (< 10 iterations) |du|,|dv| <= 2e-4 px, |dZNCC| <= 1e-5, iteration counts
agree on >= 99 %.
"""
import numpy as np
import pytest

import oracle
from oracle import P2


@pytest.fixture(scope="module")
def oracle_run(golden):
    tab = golden["table"]
    pois = oracle.make_pois2d(tab[:, 0], tab[:, 1])
    oracle.fftcc2d(golden["ref"], golden["tar"], golden["rx"], golden["ry"], pois)
    after_fftcc = pois.copy()
    prep = oracle.Prepared2D(golden["ref"], golden["tar"])
    out = {}
    for order in (oracle.ORDER_SEQ, oracle.ORDER_LANES, oracle.ORDER_SEQ_FMA, oracle.ORDER_LANES_FMA):
        p = after_fftcc.copy()
        oracle.icgn2d1(prep, golden["rx"], golden["ry"], golden["conv"], golden["stop"], p, order=order)
        out[order] = p
    return after_fftcc, out


def test_fftcc_initial_guess_matches_golden(golden, oracle_run):
    after_fftcc, _ = oracle_run
    tab = golden["table"]
    same = (after_fftcc[:, P2["u"]] == tab[:, 4]) & (after_fftcc[:, P2["v"]] == tab[:, 5])
    assert same.mean() >= 0.999, "FFTCC guess differs on %d POIs" % (~same).sum()


@pytest.mark.parametrize("order", [oracle.ORDER_SEQ, oracle.ORDER_LANES, oracle.ORDER_SEQ_FMA, oracle.ORDER_LANES_FMA])
def test_icgn2d1_matches_golden(golden, oracle_run, order):
    """(round 5: the fused arithmetic contract -- OC_ARITH_FMA, oracle/oc_oracle.h -- meets the same bars against the
    reference authors' own run as the separately rounded orders)"""
    after_fftcc, out = oracle_run
    p = out[order]
    tab, de = golden["table"], golden["deformation"]
    same_init = (after_fftcc[:, P2["u"]] == tab[:, 4]) & (after_fftcc[:, P2["v"]] == tab[:, 5])
    m = (tab[:, 7] < golden["stop"]) & same_init  # converged in the reference's run
    assert m.sum() > 28000
    assert np.abs(p[m, P2["u"]] - tab[m, 2]).max() <= 2e-4
    assert np.abs(p[m, P2["v"]] - tab[m, 3]).max() <= 2e-4
    assert np.median(np.abs(p[m, P2["u"]] - tab[m, 2])) <= 1e-6
    assert np.abs(p[m, P2["zncc"]] - tab[m, 6]).max() <= 1e-5
    # full deformation vector: ux uy vx vy
    got = p[m][:, [P2["ux"], P2["uy"], P2["vx"], P2["vy"]]]
    want = de[m][:, [3, 4, 6, 7]]
    assert np.abs(got - want).max() <= 5e-5
    agree = (p[m, P2["iteration"]] == tab[m, 7]).mean()
    assert agree >= 0.99, "iteration counts agree on only %.4f" % agree
    # u0, v0 written by ICGN are FFTCC's output (src/oc_icgn.cpp:318-319)
    assert np.array_equal(p[m, P2["u0"]], tab[m, 4]) and np.array_equal(p[m, P2["v0"]], tab[m, 5])


def test_lanes_order_close_to_sequential(oracle_run):
    """The GPU's summation order is a re-association only: same flags, ~1e-6 px apart."""
    _, out = oracle_run
    a, b = out[oracle.ORDER_SEQ], out[oracle.ORDER_LANES]
    conv = (a[:, P2["zncc"]] >= 0) & (b[:, P2["zncc"]] >= 0)
    assert ((a[:, P2["zncc"]] < 0) != (b[:, P2["zncc"]] < 0)).mean() < 1e-3
    assert (a[conv, P2["iteration"]] == b[conv, P2["iteration"]]).mean() >= 0.999
    same_it = conv & (a[:, P2["iteration"]] == b[:, P2["iteration"]])
    assert np.abs(a[same_it, P2["u"]] - b[same_it, P2["u"]]).max() <= 1e-4
    assert np.abs(a[same_it, P2["v"]] - b[same_it, P2["v"]]).max() <= 1e-4


def test_fused_arithmetic_close_to_the_reference_order(oracle_run):
    """OC_ORDER_SEQ (the reference's compiled sources, bit for bit) against the fused contract in the GPU's association
    on the 30 000 golden POIs: north_star's bars -- same flags and codes, >= 99.5 % equal iteration counts, |d u, v| <=
    1e-4 and |d ZNCC| <= 1e-5 on equal counts."""
    _, out = oracle_run
    a, b = out[oracle.ORDER_SEQ], out[oracle.ORDER_LANES_FMA]
    fa, fb = a[:, P2["zncc"]] < 0, b[:, P2["zncc"]] < 0
    assert (fa != fb).sum() == 0 and np.array_equal(a[fa, P2["zncc"]], b[fa, P2["zncc"]])
    conv = ~fa
    same_it = conv & (a[:, P2["iteration"]] == b[:, P2["iteration"]])
    assert same_it.sum() / conv.sum() >= 0.995
    assert np.abs(a[same_it, P2["u"]] - b[same_it, P2["u"]]).max() <= 1e-4
    assert np.abs(a[same_it, P2["v"]] - b[same_it, P2["v"]]).max() <= 1e-4
    assert np.abs(a[same_it, P2["zncc"]] - b[same_it, P2["zncc"]]).max() <= 1e-5


def icgn2_soft_anchor_check(p, tab):
    """Shared by the oracle and the GPU test: converged ICGN2D2 results vs the reference's CUDA ICGN2D2 CSV,
    both started from the CSV's (u0, v0) -- the CSV does not keep the gradient part of the SIFT/affine guess,
    so trajectories differ and only the converged values are compared."""
    m = (tab[:, 6] > 0.9) & (p[:, P2["zncc"]] > 0.9) & (tab[:, 7] < 10)
    assert m.sum() > 28000
    du, dv = np.abs(p[m, P2["u"]] - tab[m, 2]), np.abs(p[m, P2["v"]] - tab[m, 3])
    assert np.median(du) <= 2e-5 and np.median(dv) <= 2e-5
    assert np.percentile(du, 99) <= 3e-4 and np.percentile(dv, 99) <= 3e-4
    assert du.max() <= 2e-3 and dv.max() <= 2e-3
    dz = np.abs(p[m, P2["zncc"]] - tab[m, 6])
    assert np.median(dz) <= 1e-6 and np.percentile(dz, 99) <= 1e-5


def test_icgn2d2_converges_to_the_reference_cuda_results(golden, golden_icgn2):
    """Soft external anchor for the 12-DoF engine."""
    tab = golden_icgn2
    p = oracle.make_pois2d(tab[:, 0], tab[:, 1])
    p[:, P2["u"]], p[:, P2["v"]] = tab[:, 4], tab[:, 5]
    prep = oracle.Prepared2D(golden["ref"], golden["tar"])
    oracle.icgn2d2(prep, 16, 16, golden["conv"], golden["stop"], p)
    icgn2_soft_anchor_check(p, tab)


def nr1_golden_check(p, after_fftcc, tab, stop):
    """Shared by the oracle and the GPU test: NR2D1 results vs the reference's own CSV
    (examples/2d_dic/oht_cfrp_4_fftcc_nr1_r16.csv)."""
    same_init = (after_fftcc[:, P2["u"]] == tab[:, 4]) & (after_fftcc[:, P2["v"]] == tab[:, 5])
    assert same_init.mean() >= 0.999
    m = (tab[:, 7] < stop) & same_init & (tab[:, 6] > 0)  # converged in the reference's run
    assert m.sum() > 28000
    assert np.abs(p[m, P2["u"]] - tab[m, 2]).max() <= 2e-4 and np.abs(p[m, P2["v"]] - tab[m, 3]).max() <= 2e-4
    assert np.median(np.abs(p[m, P2["u"]] - tab[m, 2])) <= 1e-6
    assert np.abs(p[m, P2["zncc"]] - tab[m, 6]).max() <= 1e-5
    assert (p[m, P2["iteration"]] == tab[m, 7]).mean() >= 0.99
    assert np.array_equal(p[m, P2["u0"]], tab[m, 4]) and np.array_equal(p[m, P2["v0"]], tab[m, 5])


@pytest.mark.parametrize("order", [oracle.ORDER_SEQ, oracle.ORDER_LANES])
def test_nr2d1_matches_golden(golden, golden_nr1, order):
    """The Newton-Raphson engine is pinned on the reference's golden CSV like ICGN2D1 is."""
    tab = golden_nr1
    pois = oracle.make_pois2d(tab[:, 0], tab[:, 1])
    oracle.fftcc2d(golden["ref"], golden["tar"], golden["rx"], golden["ry"], pois)
    after = pois.copy()
    prep = oracle.PreparedNR2D(golden["ref"], golden["tar"])
    oracle.nr2d1(prep, golden["rx"], golden["ry"], golden["conv"], golden["stop"], pois, order=order)
    nr1_golden_check(pois, after, tab, golden["stop"])


# ---- IC-LM (src/oc_iclm.cpp).  The reference ships no result table for it (only a timing CSV), so the oracle is
# ---- anchored through what the algorithm implies: with lambda = 1 the damping vanishes and the accepted steps are
# ---- Gauss-Newton steps; with the default damping it converges to the minimum the ICGN2D1 golden CSV records.
def test_pow_lambda_is_the_correctly_rounded_power():
    """First damping value powf(lambda, q) (src/oc_iclm.cpp:253): the fixed-arithmetic restatement equals libm's
    powf except for rare 1-ulp cases (glibc's powf is not correctly rounded; the restatement is)."""
    import ctypes
    libm = ctypes.CDLL("libm.so.6")
    libm.powf.argtypes = [ctypes.c_float, ctypes.c_float]
    libm.powf.restype = ctypes.c_float
    rng = np.random.default_rng(5)
    qs = np.concatenate([rng.random(20000), rng.random(2000) * 1e-3, [0.0, 1.0, 0.25, 0.5]]).astype(np.float32)
    off = 0
    for lam in (100.0, 10.0, 2.5):
        for q in qs:
            a, b = np.float32(oracle.pow_lambda(lam, q)), np.float32(libm.powf(lam, float(q)))
            if a != b:
                off += 1
                assert abs(int(a.view(np.uint32)) - int(b.view(np.uint32))) == 1
                exact = np.float32(np.exp(np.float64(q) * np.log(np.float64(lam))))
                assert a == exact  # the restatement is the correctly rounded one
    assert off <= 0.002 * 3 * len(qs)
    assert oracle.pow_lambda(1.0, 0.37) == 1.0 and oracle.pow_lambda(100.0, 0.0) == 1.0


@pytest.fixture(scope="module")
def iclm_run(golden, oracle_run):
    after_fftcc, _ = oracle_run
    prep = oracle.Prepared2D(golden["ref"], golden["tar"])
    out = {}
    for name, fn, damping in (("lm1", oracle.iclm2d1, oracle.DEFAULT_DAMPING), ("lm1_unit", oracle.iclm2d1, (1.0, 0.1, 10.0)),
                              ("lm2", oracle.iclm2d2, oracle.DEFAULT_DAMPING)):
        p = after_fftcc.copy()
        fn(prep, golden["rx"], golden["ry"], golden["conv"], golden["stop"], p, damping=damping)
        out[name] = p
    return out


def iclm1_golden_check(p, tab, stop):
    """Shared with the GPU test: ICLM2D1 against the reference's ICGN2D1 table (same ZNSSD minimum)."""
    m = (tab[:, 7] < stop) & (p[:, P2["zncc"]] > 0)
    assert m.sum() > 27000
    du, dv = np.abs(p[m, P2["u"]] - tab[m, 2]), np.abs(p[m, P2["v"]] - tab[m, 3])
    assert np.median(du) <= 2e-6 and np.median(dv) <= 2e-6
    assert du.max() <= 2e-3 and dv.max() <= 2e-3
    assert np.abs(p[m, P2["zncc"]] - tab[m, 6]).max() <= 5e-5
    assert np.array_equal(p[m, P2["u0"]], tab[m, 4]) and np.array_equal(p[m, P2["v0"]], tab[m, 5])


def test_iclm2d1_reaches_the_minimum_of_the_golden_icgn1_table(golden, iclm_run):
    iclm1_golden_check(iclm_run["lm1"], golden["table"], golden["stop"])


def test_iclm2d1_with_unit_lambda_takes_gauss_newton_steps(oracle_run, iclm_run):
    """lambda = 1: powf(1, q) - 1 = 0, so (H + 0 I)^-1 = H^-1 bit for bit and every accepted step is ICGN2D1's.
    The two differ only where a step was rejected (ZNSSD not strictly lower, typically the last one)."""
    _, out = oracle_run
    gn, lm = out[oracle.ORDER_SEQ], iclm_run["lm1_unit"]
    same = (gn.view(np.uint32) == lm.view(np.uint32)).all(axis=1)
    assert same.mean() >= 0.5
    both = (gn[:, P2["zncc"]] > 0) & (lm[:, P2["zncc"]] > 0)
    assert np.abs(gn[both, P2["u"]] - lm[both, P2["u"]]).max() <= 2e-3
    # a rejected step can only cost iterations, never save them
    assert (lm[both, P2["iteration"]] >= gn[both, P2["iteration"]]).all()


def test_iclm2d2_agrees_with_icgn2d2(golden, oracle_run, iclm_run):
    after_fftcc, _ = oracle_run
    prep = oracle.Prepared2D(golden["ref"], golden["tar"])
    gn2 = after_fftcc.copy()
    oracle.icgn2d2(prep, golden["rx"], golden["ry"], golden["conv"], golden["stop"], gn2)
    lm2 = iclm_run["lm2"]
    m = (gn2[:, P2["zncc"]] > 0.9) & (lm2[:, P2["zncc"]] > 0.9)
    assert m.sum() > 22000
    for k in ("u", "v"):
        d = np.abs(gn2[m, P2[k]] - lm2[m, P2[k]])
        assert np.median(d) <= 5e-6 and d.max() <= 5e-3
