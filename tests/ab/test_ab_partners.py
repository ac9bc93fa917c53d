"""The measured losers that stay around as A/B partners -- icgn2d variants 0 and 6, the LDS-band kernel (variant 9,
icgn2d_band.hip, round 6) and the ICGN3D1 row mapping (icgn3d_rows.hip) -- are compiled only into the A/B build of the library (opencorr_amd/build.py --ab ->
lib/ab/libopencorr_hip_ab.so, -DOC_BUILD_AB=1).  These tests keep them bit-exact against their oracle orders.  They run in
a process of their own whose OPENCORR_HIP_LIB points at that build: tests/test_gpu_ab_build.py starts it on the GPU box;
collected anywhere else they skip.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.skipif(os.environ.get("OC_AB_RUN") != "1", reason="runs inside tests/test_gpu_ab_build.py (needs the A/B build + a GPU)")


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def eng():
    import opencorr_amd
    assert opencorr_amd.capi.LIB_PATH.endswith("libopencorr_hip_ab.so"), opencorr_amd.capi.LIB_PATH
    return opencorr_amd


@pytest.mark.parametrize("variant,xcd", [(0, 0), (6, 1), (6, 0), (8, 1), (8, 0), (9, 1), (9, 0), (5, 1), (7, 1)])
def test_icgn2d1_ab_variants_identical_bits(eng, speckle_small, variant, xcd):
    import oracle
    from opencorr_amd import synth
    ref, tar = speckle_small
    xs, ys = synth.poi_grid_2d(ref.shape[0], ref.shape[1], 19, 23, 26)
    pois = oracle.make_pois2d(xs, ys)
    oracle.fftcc2d(ref, tar, 16, 16, pois)
    want = pois.copy()
    oracle.icgn2d1(oracle.Prepared2D(ref, tar), 16, 16, 0.001, 10, want, order=oracle.ORDER_LANES, lanes=64)
    icgn = eng.ICGN2D1(16, 16, 0.001, 10)
    icgn.set_images(ref, tar)
    icgn.prepare()
    icgn.set_tuning("icgn2d_variant", variant)
    icgn.set_tuning("icgn2d_xcd", xcd)
    assert np.array_equal(_bits(icgn.compute(pois.copy())), _bits(want))
    icgn.set_tuning("arith_fma", 1)     # the A/B variants exist in both arithmetic modes
    want = pois.copy()
    oracle.icgn2d1(oracle.Prepared2D(ref, tar), 16, 16, 0.001, 10, want, order=oracle.ORDER_LANES_FMA, lanes=64)
    assert np.array_equal(_bits(icgn.compute(pois.copy())), _bits(want))


@pytest.mark.parametrize("dof", [6, 12])
def test_icgn2d_variant6_lockstep_barriers_with_mixed_wave_lifetimes(eng, speckle_small, dof):
    """Variant 6 (4-wave lockstep workgroups): every workgroup mixes guard rejects, first-sweep aborts, NaN guesses,
    one-iteration POIs and stop-limited POIs (the product build runs the same case for variants 4 and 5)."""
    import oracle
    from opencorr_amd import synth
    ref, tar = speckle_small
    h, w = ref.shape
    r = 16
    P = oracle.P2
    xs, ys = synth.poi_grid_2d(h, w, 32, 30, 26)
    pois = oracle.make_pois2d(xs, ys)
    oracle.fftcc2d(ref, tar, r, r, pois)
    prep = oracle.Prepared2D(ref, tar)
    fn = oracle.icgn2d1 if dof == 6 else oracle.icgn2d2
    solved = pois.copy()
    fn(prep, r, r, 0.001, 10, solved, order=oracle.ORDER_LANES, lanes=64)
    q = pois.copy()
    slot = np.arange(len(q)) % 4
    q[slot == 0, P["zncc"]] = -2.0
    q[(np.arange(len(q)) % 8) == 1, P["u"]] = w - 30.0
    q[(np.arange(len(q)) % 8) == 2, P["v"]] = np.nan
    one = (np.arange(len(q)) % 8) == 3
    q[one, 2:14] = solved[one, 2:14]
    far = (np.arange(len(q)) % 8) == 5
    q[far, P["u"]] += 6.5
    q[far, P["v"]] -= 5.5
    want = q.copy()
    fn(prep, r, r, 0.001, 10, want, order=oracle.ORDER_LANES, lanes=64)
    icgn = (eng.ICGN2D1 if dof == 6 else eng.ICGN2D2)(r, r, 0.001, 10)
    icgn.set_images(ref, tar)
    icgn.prepare()
    icgn.set_tuning("icgn2d_variant", 6)
    for _ in range(3):
        assert np.array_equal(_bits(icgn.compute(q.copy())), _bits(want))


BIG = (96, 100, 104)  # dz, dy, dx


@pytest.fixture(scope="module")
def big_volumes():
    import oracle
    from opencorr_amd import synth
    ref, tar = synth.speckle_pair_3d(*BIG, seed=23)
    return ref, tar, oracle.Prepared3D(ref, tar)


@pytest.mark.parametrize("r", [13, 14, 15, 16, 17, 20, 29, 30, 31, 32])
def test_icgn3d1_both_mappings_against_their_oracle_orders(big_volumes, r):
    """The two ICGN3D1 kernels on the radii where the row mapping changes shape: r = 13 (27 samples per row: no body, the row
    kernel IS the old mapping), 14 / 15 (one partial chunk of 29 / 31 lanes, no tail), 16 / 17 (one chunk + a tail of 1 / 3
    columns), 20 / 29 (tail of 9 / 27 columns), 30 / 31 (two chunks, the second partial), 32 (two chunks + a tail column).
    "icgn3d_mapping" = 1 (icgn3d_rows.hip, the default) must equal the oracle in OC_ORDER_ROWS bit for bit, = 0 (icgn3d.hip) the
    oracle in OC_ORDER_LANES; the two orders differ by a re-association only (same iteration counts here, |d u| <= 1e-4)."""
    import opencorr_amd
    import oracle
    from opencorr_amd import synth
    ref, tar, prep = big_volumes
    P = oracle.P3
    c = [BIG[2] // 2, BIG[1] // 2, BIG[0] // 2]
    span = [BIG[2] - 2 * (r + 7), BIG[1] - 2 * (r + 7), BIG[0] - 2 * (r + 7)]   # the warped subvolume + its taps stay inside
    rng = np.random.default_rng(1000 + r)
    n = 5 if r < 24 else 3
    xs = [c[0] + int(rng.integers(-(span[0] // 2), span[0] // 2 + 1)) for _ in range(n)]
    ys = [c[1] + int(rng.integers(-(span[1] // 2), span[1] // 2 + 1)) for _ in range(n)]
    zs = [c[2] + int(rng.integers(-(span[2] // 2), span[2] // 2 + 1)) for _ in range(n)]
    pois = oracle.make_pois3d(xs, ys, zs)
    w = synth.DEFAULT_WARP_3D
    pois[:, P["u"]], pois[:, P["v"]], pois[:, P["w"]] = round(w["u"]), round(w["v"]), round(w["w"])
    extra = oracle.make_pois3d([c[0], c[0]], [c[1], c[1]], [c[2], c[2]])
    extra[0, P["u"]] = 90.0       # leaves the volume inside the loop: -3
    extra[1, P["zncc"]] = -2.0    # rejected on entry
    pois = np.concatenate([pois, extra]).astype(np.float32)
    icgn = opencorr_amd.ICGN3D1(r, r, r, 0.001, 20.0)
    icgn.set_images(ref, tar)
    icgn.prepare()
    results = {}
    for mapping, order in ((1, oracle.ORDER_ROWS), (0, oracle.ORDER_LANES)):
        want = pois.copy()
        oracle.icgn3d1(prep, r, r, r, 0.001, 20.0, want, order=order, lanes=512)
        icgn.set_tuning("icgn3d_mapping", mapping)
        got = icgn.compute(pois.copy())
        mism = np.argwhere(_bits(got) != _bits(want))
        assert mism.size == 0, "mapping %d: first mismatches (poi, field): %s" % (mapping, mism[:10].tolist())
        results[mapping] = got
    a, b = results[1][:n], results[0][:n]
    assert (a[:, P["zncc"]] > 0.9).all()
    assert np.array_equal(a[:, P["iteration"]], b[:, P["iteration"]])
    assert np.abs(a[:, [P["u"], P["v"], P["w"]]] - b[:, [P["u"], P["v"], P["w"]]]).max() <= 1e-4
    if r == 13:
        assert np.array_equal(_bits(results[1]), _bits(results[0]))   # no body: the same mapping, the same bits


def test_icgn3d1_row_mapping_block_schedule_and_reduced_slots(eng):
    """The row mapping under the block-ordered queue, and with fewer persistent workgroups than the default 512 (the A/B
    build's OC_ICGN3D_BLOCKS knob is what sizes the scratch: the launcher must use the SAME count -- ADVICE r4)."""
    import oracle
    from opencorr_amd import synth
    shape = (72, 76, 80)
    ref, tar = synth.speckle_pair_3d(*shape, seed=21)
    P = oracle.P3
    xs, ys, zs = synth.poi_grid_3d(*shape, 13, 13, 13, 14)
    pois = oracle.make_pois3d(xs, ys, zs)[:2191]
    w = synth.DEFAULT_WARP_3D
    pois[:, P["u"]], pois[:, P["v"]], pois[:, P["w"]] = round(w["u"]), round(w["v"]), round(w["w"])
    pois[5::97, P["zncc"]] = -1.0
    icgn = eng.ICGN3D1(5, 5, 5, 0.001, 20)
    icgn.set_images(ref, tar)
    icgn.prepare()
    icgn.set_tuning("icgn3d_mapping", 1)
    icgn.set_tuning("icgn3d_tile_vox", 0)
    want = icgn.compute(pois.copy())
    for tile in (8, 48):
        icgn.set_tuning("icgn3d_tile_vox", tile)
        assert np.array_equal(_bits(icgn.compute(pois.copy())), _bits(want)), tile


@pytest.mark.parametrize("dof", [6, 12])
def test_icgn2d_split_launch_shape_pipeline_same_bits(eng, speckle_small, dof):
    """Variant 8 -- the set-up kernel files mean, norm and H^-1 per POI, the iteration kernel reads them -- back to back and
    as the two-stream pipeline over 3 and 7 chunks of the visiting order (`icgn2d_split_chunks`; set-up kernels one or two
    chunks ahead of the iteration kernels): every bit equal to the single-kernel default, with centre offsets too, twice in
    a row (the set-up records of the previous call are overwritten)."""
    import oracle
    from opencorr_amd import synth
    ref, tar = speckle_small
    h, w = ref.shape
    r = 16 if dof == 6 else 12
    xs, ys = synth.poi_grid_2d(h, w, 150, 120, 24)  # 18000 POIs >= the 16384 of the tile schedule and of the pipeline
    pois = oracle.make_pois2d(xs, ys)
    oracle.fftcc2d(ref, tar, 12, 12, pois)
    pois[7::101, oracle.P2["zncc"]] = -1.0
    pois[11::103, oracle.P2["u"]] = 250.0
    icgn = (eng.ICGN2D1 if dof == 6 else eng.ICGN2D2)(r, r, 0.001, 10)
    icgn.set_images(ref, tar)
    icgn.prepare()
    want = icgn.compute(pois.copy())
    off = np.random.default_rng(3).uniform(-2, 2, (len(pois), 2)).astype(np.float32)
    want_off = icgn.compute_with_offsets(pois.copy(), off)
    sample = pois[::37].copy()
    (oracle.icgn2d1 if dof == 6 else oracle.icgn2d2)(oracle.Prepared2D(ref, tar), r, r, 0.001, 10, sample, order=oracle.ORDER_LANES, lanes=64)
    assert np.array_equal(_bits(sample), _bits(want[::37]))
    icgn.set_tuning("icgn2d_variant", 8)
    for chunks in (0, 3, 7):
        icgn.set_tuning("icgn2d_split_chunks", chunks)
        for _ in range(2):
            assert np.array_equal(_bits(icgn.compute(pois.copy())), _bits(want)), chunks
        assert np.array_equal(_bits(icgn.compute_with_offsets(pois.copy(), off)), _bits(want_off)), chunks


@pytest.mark.parametrize("dof", [6, 12])
def test_icgn2d_band_kernel_trippers_offsets_rectangular(eng, speckle_small, dof):
    """Variant 9 (icgn2d_band.hip: the workgroup's band of the bicubic table staged in LDS, the warped subset in registers, all
    eight waves resident until the workgroup's last POI is done): guard trippers, rejected and NaN POIs, a far-off guess whose
    samples leave the staged box (per-lane global gathers), a queue that does not fill its last workgroup, a non-square
    subset, centre offsets -- bit-identical to the oracle in both arithmetic modes."""
    import oracle
    from opencorr_amd import synth
    ref, tar = speckle_small
    h, w = ref.shape
    rx, ry = (13, 9) if dof == 6 else (10, 12)
    xs, ys = synth.poi_grid_2d(h, w, 11, 9, 24)
    pois = oracle.make_pois2d(xs, ys)
    oracle.fftcc2d(ref, tar, 12, 12, pois)
    P = oracle.P2
    extra = oracle.make_pois2d([3.0, 90.0, 90.0, 90.0, w - 12.0], [80.0, 80.0, 80.0, 80.0, 100.0])
    extra[1, P["u"]] = 200.0
    extra[2, P["zncc"]] = -1.0
    extra[3, P["v"]] = np.nan
    extra[4, P["u"]], extra[4, P["ux"]] = 1.0, 0.4
    pois = np.concatenate([extra[:2], pois, extra[2:]]).astype(np.float32)
    prep = oracle.Prepared2D(ref, tar)
    fn = oracle.icgn2d1 if dof == 6 else oracle.icgn2d2
    icgn = (eng.ICGN2D1 if dof == 6 else eng.ICGN2D2)(rx, ry, 0.001, 10)
    icgn.set_images(ref, tar)
    icgn.prepare()
    icgn.set_tuning("icgn2d_variant", 9)
    off = np.random.default_rng(9).uniform(-2, 2, (len(pois), 2)).astype(np.float32)
    for fma, order in ((0, oracle.ORDER_LANES), (1, oracle.ORDER_LANES_FMA)):
        icgn.set_tuning("arith_fma", fma)
        want = pois.copy()
        fn(prep, rx, ry, 0.001, 10, want, order=order, lanes=64)
        assert np.array_equal(_bits(icgn.compute(pois.copy())), _bits(want)), fma
        want = pois.copy()
        fn(prep, rx, ry, 0.001, 10, want, order=order, lanes=64, center_offsets=off)
        assert np.array_equal(_bits(icgn.compute_with_offsets(pois.copy(), off)), _bits(want)), fma


@pytest.mark.parametrize("dof", [6, 12])
def test_icgn2d_band_kernel_mixed_wave_lifetimes_and_tile_schedule(eng, speckle_small, dof):
    """Variant 9's workgroups keep all eight waves until the last POI is done (finished waves go on staging the band): every
    workgroup mixes guard rejects, first-sweep aborts, NaN guesses, one-iteration POIs and stop-limited POIs; then a queue
    long enough for the tile schedule (workgroups whose POIs straddle two grid rows: part of the waves gather globally)."""
    import oracle
    from opencorr_amd import synth
    ref, tar = speckle_small
    h, w = ref.shape
    r = 16
    P = oracle.P2
    xs, ys = synth.poi_grid_2d(h, w, 32, 30, 26)
    pois = oracle.make_pois2d(xs, ys)
    oracle.fftcc2d(ref, tar, r, r, pois)
    prep = oracle.Prepared2D(ref, tar)
    fn = oracle.icgn2d1 if dof == 6 else oracle.icgn2d2
    solved = pois.copy()
    fn(prep, r, r, 0.001, 10, solved, order=oracle.ORDER_LANES, lanes=64)
    q = pois.copy()
    slot = np.arange(len(q)) % 8
    q[slot == 0, P["zncc"]] = -2.0
    q[slot == 1, P["u"]] = w - 30.0
    q[slot == 2, P["v"]] = np.nan
    q[slot == 3, 2:14] = solved[slot == 3, 2:14]
    q[slot == 4, P["u"]] += 6.5
    q[slot == 4, P["v"]] -= 5.5
    want = q.copy()
    fn(prep, r, r, 0.001, 10, want, order=oracle.ORDER_LANES, lanes=64)
    icgn = (eng.ICGN2D1 if dof == 6 else eng.ICGN2D2)(r, r, 0.001, 10)
    icgn.set_images(ref, tar)
    icgn.prepare()
    icgn.set_tuning("icgn2d_variant", 9)
    for _ in range(3):
        assert np.array_equal(_bits(icgn.compute(q.copy())), _bits(want))
    xs, ys = synth.poi_grid_2d(h, w, 150, 120, 24)  # 18000 POIs >= the 16384 of the tile schedule
    big = oracle.make_pois2d(xs, ys)
    oracle.fftcc2d(ref, tar, 12, 12, big)
    got = icgn.compute(big.copy())
    icgn.set_tuning("icgn2d_variant", -1)
    assert np.array_equal(_bits(got), _bits(icgn.compute(big.copy())))
    sample = big[::41].copy()
    fn(prep, r, r, 0.001, 10, sample, order=oracle.ORDER_LANES, lanes=64)
    assert np.array_equal(_bits(sample), _bits(got[::41]))


def test_fftcc3d_fused32_against_the_kernel_of_rounds_1_to_5(big_volumes):
    """fftcc3d_fused.hip (round 6: y-lines gathered side by side in x, mirror-closed waves, the inverse on the Hermitian half, one
    launch) against fftcc3d_fused_r5.hip (tuning "fftcc3d_fused" = 2): the same integers on every POI -- inner ones with integer
    guesses that displace the target window, and windows clamped at each of the six volume faces, which the old kernel handed to a
    second launch -- and ZNCC within 1e-6 (the means are summed in another order; everything behind them is the same arithmetic)."""
    import opencorr_amd
    import oracle
    from opencorr_amd import synth
    ref, tar, _ = big_volumes
    dz, dy, dx = BIG
    xs, ys, zs = synth.poi_grid_3d(dz, dy, dx, 6, 5, 4, 18)
    pois = oracle.make_pois3d(xs, ys, zs)
    P = oracle.P3
    rng = np.random.default_rng(5)
    pois[::3, P["u"]] = rng.integers(-3, 4, len(pois[::3]))
    pois[1::3, P["v"]] = rng.integers(-3, 4, len(pois[1::3]))
    pois[2::3, P["w"]] = rng.integers(-3, 4, len(pois[2::3]))
    border = oracle.make_pois3d([3.0, dx - 4.0, 50.0, 50.0, 50.0, 50.0, 2.0], [50.0, 50.0, 5.0, dy - 2.0, 50.0, 50.0, 3.0],
                                [48.0, 48.0, 48.0, 48.0, 1.0, dz - 6.0, dz - 2.0])
    pois = np.concatenate([pois, border]).astype(np.float32)
    f = opencorr_amd.FFTCC3D(16, 16, 16)
    f.set_images(ref, tar)
    new = f.compute(pois.copy())
    f.set_tuning("fftcc3d_fused", 2)
    old = f.compute(pois.copy())
    zc = P["zncc"]
    other = [c for c in range(31) if c != zc]
    assert np.array_equal(_bits(new[:, other]), _bits(old[:, other]))
    assert np.abs(new[:, zc] - old[:, zc]).max() <= 1e-6
    assert (new[:len(xs), zc] > 0.5).mean() > 0.9
    assert not np.array_equal(new[:, [P["u"], P["v"], P["w"]]], pois[:, [P["u"], P["v"], P["w"]]])
