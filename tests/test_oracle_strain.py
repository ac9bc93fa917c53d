"""Pins the Strain restatement (oracle/oc_oracle.cpp, Strain section) on the reference's golden table and on
analytic displacement fields (src/oc_strain.cpp:149-247, 372-488)."""
import numpy as np
import pytest

import oracle
from oracle import P2, P3


def strain_queue_from_golden(g):
    t = g["table"]
    p = oracle.make_pois2d(t[:, 0], t[:, 1])
    p[:, P2["u"]], p[:, P2["v"]], p[:, P2["zncc"]] = t[:, 2], t[:, 3], t[:, 4]
    return p


def strain_golden_check(p, g):
    """Shared with the GPU test.  The table was printed with 8 decimals (5e-9) by Eigen's float QR (~1e-6 relative
    on strains of up to 7e-2): 3e-7 absolute, median at the print resolution."""
    t = g["table"]
    m = t[:, 4] >= g["zncc_threshold"]
    assert m.sum() > 29000
    for k, c in (("exx", 5), ("eyy", 6), ("exy", 7)):
        d = np.abs(p[m, P2[k]].astype(np.float64) - t[m, c])
        assert d.max() <= 3e-7, (k, d.max())
        assert np.median(d) <= 6e-9, (k, np.median(d))
    # POIs below the threshold are not computed (src/oc_strain.cpp:241): fields stay as they came
    assert not p[~m][:, [P2["exx"], P2["eyy"], P2["exy"]]].any()


def test_strain2d_matches_the_reference_table(golden_strain):
    p = strain_queue_from_golden(golden_strain)
    oracle.strain2d(p, golden_strain["radius"], golden_strain["neighbors"], golden_strain["zncc_threshold"],
                    golden_strain["approximation"])
    strain_golden_check(p, golden_strain)


def affine_queue_2d(n=4000, seed=7, extent=400.0):
    rng = np.random.default_rng(seed)
    xs, ys = rng.random(n) * extent, rng.random(n) * extent * 0.7
    p = oracle.make_pois2d(xs.astype(np.float32), ys.astype(np.float32))
    x, y = p[:, 0].astype(np.float64), p[:, 1].astype(np.float64)
    g = dict(ux=2e-3, uy=-7e-4, vx=4e-4, vy=-3e-3)
    p[:, P2["u"]] = 1.5 + g["ux"] * x + g["uy"] * y
    p[:, P2["v"]] = -0.5 + g["vx"] * x + g["vy"] * y
    p[:, P2["zncc"]] = 0.99
    return p, g


@pytest.mark.parametrize("approximation", [1, 2])
def test_strain2d_recovers_an_affine_field(approximation):
    p, g = affine_queue_2d()
    # a fifth of the POIs failed: they must be neither computed nor used (their displacements are garbage)
    bad = np.arange(len(p)) % 5 == 0
    p[bad, P2["zncc"]] = -4.0
    p[bad, P2["u"]] = 1e3
    p[bad, P2["exx"]] = 123.0
    oracle.strain2d(p, 25.0, 6, 0.9, approximation)
    ux, uy, vx, vy = g["ux"], g["uy"], g["vx"], g["vy"]
    if approximation == 1:
        want = (ux, vy, 0.5 * (uy + vx))
    else:
        want = (ux + 0.5 * (ux * ux + vx * vx), vy + 0.5 * (uy * uy + vy * vy), 0.5 * (uy + vx + uy * ux + vy * vx))
    ok = ~bad
    for k, w in zip(("exx", "eyy", "exy"), want):
        assert np.abs(p[ok, P2[k]] - w).max() <= 2e-6  # float32 u, v over a ~50 px baseline
    assert (p[bad, P2["exx"]] == 123.0).all() and not p[bad, P2["eyy"]].any()


def test_strain2d_knn_fallback_and_too_few_neighbours():
    """Fewer than neighbor_number_min POIs inside the radius: the K nearest are used instead
    (src/oc_strain.cpp:177-186); if the ZNCC filter then leaves fewer than K, nothing is written (:190)."""
    xs, ys = np.meshgrid(np.arange(8, dtype=np.float32) * 10, np.arange(6, dtype=np.float32) * 10)
    p = oracle.make_pois2d(xs.ravel(), ys.ravel())
    p[:, P2["u"]] = 0.01 * p[:, 0]
    p[:, P2["v"]] = -0.02 * p[:, 1]
    p[:, P2["zncc"]] = 1.0
    q = p.copy()
    oracle.strain2d(q, 5.0, 5)  # radius 5 < spacing 10: only the POI itself is inside -> KNN with K = 5
    assert np.abs(q[:, P2["exx"]] - 0.01).max() < 1e-6 and np.abs(q[:, P2["eyy"]] + 0.02).max() < 1e-6
    q = p.copy()
    q[9, P2["zncc"]] = 0.1  # one of the 5 nearest of its neighbours fails the filter -> those stay untouched
    oracle.strain2d(q, 5.0, 5)
    untouched = ~q[:, [P2["exx"], P2["eyy"], P2["exy"]]].any(axis=1)
    assert untouched[9] and 3 <= untouched.sum() <= 9
    # more neighbours requested than POIs exist: nothing can be fitted
    q = p[:4].copy()
    oracle.strain2d(q, 5.0, 5)
    assert not q[:, P2["exx"]].any()


def test_strain2d_does_not_depend_on_the_queue_order_beyond_rounding():
    p, _ = affine_queue_2d(n=1500, seed=3)
    rng = np.random.default_rng(1)
    p[:, P2["u"]] += rng.normal(0, 0.01, len(p)).astype(np.float32)
    a = p.copy()
    oracle.strain2d(a, 30.0, 5)
    perm = rng.permutation(len(p))
    b = p[perm].copy()
    oracle.strain2d(b, 30.0, 5)
    for k in ("exx", "eyy", "exy"):
        assert np.abs(a[perm, P2[k]] - b[:, P2[k]]).max() <= 1e-9


def affine_queue_3d(n=3000, seed=5, extent=120.0):
    rng = np.random.default_rng(seed)
    xyz = (rng.random((n, 3)) * extent).astype(np.float32)
    p = oracle.make_pois3d(xyz[:, 0], xyz[:, 1], xyz[:, 2])
    G = np.array([[1e-3, -2e-3, 5e-4], [3e-4, 2e-3, -1e-3], [-7e-4, 6e-4, -1.5e-3]])  # rows: grad u, grad v, grad w
    x = p[:, :3].astype(np.float64)
    for r, k in enumerate(("u", "v", "w")):
        p[:, P3[k]] = 0.3 * (r + 1) + x @ G[r]
    p[:, P3["zncc"]] = 0.95
    return p, G


@pytest.mark.parametrize("approximation", [1, 2])
def test_strain3d_recovers_an_affine_field(approximation):
    p, G = affine_queue_3d()
    bad = np.arange(len(p)) % 7 == 0
    p[bad, P3["zncc"]] = -3.0
    p[bad, P3["w"]] = -500.0
    oracle.strain3d(p, 22.0, 8, 0.9, approximation)
    (ux, uy, uz), (vx, vy, vz), (wx, wy, wz) = G
    if approximation == 1:
        want = dict(exx=ux, eyy=vy, ezz=wz, exy=0.5 * (uy + vx), eyz=0.5 * (vz + wy), ezx=0.5 * (wx + uz))
    else:
        want = dict(exx=ux + 0.5 * (ux * ux + vx * vx + wx * wx), eyy=vy + 0.5 * (uy * uy + vy * vy + wy * wy),
                    ezz=wz + 0.5 * (uz * uz + vz * vz + wz * wz), exy=0.5 * (uy + vx + uy * ux + vy * vx + wy * wx),
                    eyz=0.5 * (vz + wy + uz * uy + vz * vy + wz * wy), ezx=0.5 * (wx + uz + ux * uz + vx * vz + wx * wz))
    ok = ~bad
    fitted = p[ok][:, P3["exx"]] != 0
    assert fitted.mean() > 0.99
    for k, w in want.items():
        assert np.abs(p[ok][fitted, P3[k]] - w).max() <= 3e-6, k
    assert not p[bad][:, P3["exx"]:P3["ezx"] + 1].any()


# ---- RegionFit2D / RegionFit3D (src/oc_region_fit.cpp): the same plane fit, evaluated for POIs of a second queue ----
def test_region_fit2d_transfers_an_affine_field_to_new_pois():
    cloud, g = affine_queue_2d(n=3000, seed=21)
    rng = np.random.default_rng(8)
    qx, qy = (rng.random(500) * 380 + 10).astype(np.float32), (rng.random(500) * 260 + 10).astype(np.float32)
    q = oracle.make_pois2d(qx, qy)
    q[:, P2["zncc"]] = -4.0  # the unreliable POIs being re-initialised (examples/test_3d_reconstruction_sift_icgn2_regfit.cpp:233-239)
    q[:, P2["uxx"]] = 9.0    # not part of the fit: must survive
    oracle.region_fit(cloud, q, 20.0, 7)
    x, y = qx.astype(np.float64), qy.astype(np.float64)
    assert np.abs(q[:, P2["u"]] - (1.5 + g["ux"] * x + g["uy"] * y)).max() <= 2e-5
    assert np.abs(q[:, P2["v"]] - (-0.5 + g["vx"] * x + g["vy"] * y)).max() <= 2e-5
    for k in ("ux", "uy", "vx", "vy"):
        assert np.abs(q[:, P2[k]] - g[k]).max() <= 2e-6
    assert (q[:, P2["zncc"]] == 0).all() and (q[:, P2["uxx"]] == 9.0).all()


def test_region_fit2d_knn_fallback_and_small_clouds():
    cloud, g = affine_queue_2d(n=40, seed=2, extent=1000.0)  # ~1 POI per 150 px
    q = oracle.make_pois2d(np.array([100.0, 5000.0], dtype=np.float32), np.array([100.0, 300.0], dtype=np.float32))
    q[:, P2["zncc"]] = 0.5
    a = q.copy()
    oracle.region_fit(cloud, a, 10.0, 6)  # nobody inside 10 px: the 6 nearest are used, also far outside the cloud
    assert (a[:, P2["zncc"]] == 0).all()
    assert np.abs(a[:, P2["ux"]] - g["ux"]).max() <= 1e-5 and np.abs(a[:, P2["vy"]] - g["vy"]).max() <= 1e-5
    b = q.copy()
    oracle.region_fit(cloud[:4], b, 10.0, 6)  # fewer reliable POIs than required: untouched
    assert np.array_equal(b, q)


def test_region_fit3d_transfers_an_affine_field():
    cloud, G = affine_queue_3d(n=4000, seed=13)
    rng = np.random.default_rng(3)
    xyz = (rng.random((300, 3)) * 100 + 10).astype(np.float32)
    q = oracle.make_pois3d(xyz[:, 0], xyz[:, 1], xyz[:, 2])
    oracle.region_fit(cloud, q, 18.0, 10)
    x = xyz.astype(np.float64)
    for r, k in enumerate(("u", "v", "w")):
        assert np.abs(q[:, P3[k]] - (0.3 * (r + 1) + x @ G[r])).max() <= 3e-5
        for c, ax in enumerate("xyz"):
            assert np.abs(q[:, P3[k + ax]] - G[r, c]).max() <= 3e-6
    assert (q[:, P3["zncc"]] == 0).all()
