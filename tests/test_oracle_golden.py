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
    for order in (oracle.ORDER_SEQ, oracle.ORDER_LANES):
        p = after_fftcc.copy()
        oracle.icgn2d1(prep, golden["rx"], golden["ry"], golden["conv"], golden["stop"], p, order=order)
        out[order] = p
    return after_fftcc, out


def test_fftcc_initial_guess_matches_golden(golden, oracle_run):
    after_fftcc, _ = oracle_run
    tab = golden["table"]
    same = (after_fftcc[:, P2["u"]] == tab[:, 4]) & (after_fftcc[:, P2["v"]] == tab[:, 5])
    assert same.mean() >= 0.999, "FFTCC guess differs on %d POIs" % (~same).sum()


@pytest.mark.parametrize("order", [oracle.ORDER_SEQ, oracle.ORDER_LANES])
def test_icgn2d1_matches_golden(golden, oracle_run, order):
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
