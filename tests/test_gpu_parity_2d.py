"""GPU parity: HIP engines (through the C-ABI) vs the CPU oracle and the golden vectors.

Bars:
  * prepare() fields (gradients, bicubic LUT): bit-exact vs the oracle.
  * FFTCC2D: integer u, v identical; ZNCC within 1e-5 (rocFFT float32 vs the oracle's
    double-precision DFT).
  * ICGN2D1: bit-exact vs the oracle in OC_ORDER_LANES (same reduction association),
    i.e. identical iteration counts, flags and float bits for every POI.
  * Golden OHT-CFRP example of the reference: same acceptance as the oracle's own test.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import opencorr_amd
    assert opencorr_amd.capi.device_count() >= 1
    return opencorr_amd


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def test_prepare_fields_bit_exact(eng, speckle_small):
    import oracle
    ref, tar = speckle_small
    icgn = eng.ICGN2D1(16, 16, 0.001, 10)
    icgn.set_images(ref, tar)
    icgn.prepare()
    gx, gy = oracle.gradient2d(ref)
    lut = oracle.bspline2d_lut(tar)
    assert np.array_equal(_bits(icgn.read_field("gx")), _bits(gx))
    assert np.array_equal(_bits(icgn.read_field("gy")), _bits(gy))
    assert np.array_equal(_bits(icgn.read_field("lut")), _bits(lut))
    assert np.array_equal(icgn.read_field("ref"), ref)


def test_column_major_images_match_row_major(eng, speckle_small):
    """Image2D::eg_mat is column-major (src/oc_image.h:37); the engine transposes on upload."""
    ref, tar = speckle_small
    a = eng.ICGN2D1(16, 16, 0.001, 10)
    a.set_images(np.asfortranarray(ref), np.asfortranarray(tar), layout=eng.capi.COL_MAJOR)
    assert np.array_equal(a.read_field("ref"), ref)
    assert np.array_equal(a.read_field("tar"), tar)


@pytest.mark.parametrize("rx,ry", [(16, 16), (15, 15), (8, 12)])
def test_fftcc2d_matches_oracle(eng, speckle_small, rx, ry):
    import oracle
    from opencorr_amd import synth
    ref, tar = speckle_small
    h, w = ref.shape
    xs, ys = synth.poi_grid_2d(h, w, 23, 19, 24)
    # a few POIs that trip the bounds guard (left untouched by the reference, src/oc_fftcc.cpp:190-196)
    xs = np.concatenate([xs, [3, w - 2, 150]]).astype(np.float32)
    ys = np.concatenate([ys, [100, 100, 2]]).astype(np.float32)
    want = oracle.make_pois2d(xs, ys)
    got = want.copy()
    oracle.fftcc2d(ref, tar, rx, ry, want)
    f = eng.FFTCC2D(rx, ry)
    f.set_images(ref, tar)
    f.compute(got)
    P = oracle.P2
    assert np.array_equal(got[:, P["u"]], want[:, P["u"]])
    assert np.array_equal(got[:, P["v"]], want[:, P["v"]])
    assert np.array_equal(got[:, [P["u0"], P["v0"]]], want[:, [P["u0"], P["v0"]]])
    assert np.abs(got[:, P["zncc"]] - want[:, P["zncc"]]).max() <= 1e-5
    # guarded POIs are bit-for-bit untouched
    assert np.array_equal(_bits(got[-3:]), _bits(want[-3:]))
    # fields FFTCC never writes stay as they were
    untouched = [c for c in range(25) if c not in (P["u"], P["v"], P["u0"], P["v0"], P["zncc"])]
    assert np.array_equal(_bits(got[:, untouched]), _bits(want[:, untouched]))


@pytest.mark.parametrize("r", [16, 8, 9, 10, 12, 15, 18, 20, 24, 25, 30, 32, 7, 11, 13, 14, 21, 23, 27, 31])
def test_fftcc2d_fused_kernel_matches_rocfft_pipeline(eng, speckle_small, r):
    """Square windows of side 16, 18, 20, 24, 30, 32, 36, 40, 48, 50, 60, 64 run a single-kernel FFT (fftcc2d_fused.hip for 32, the mixed-radix
    fftcc2d_fusedn.hip otherwise) by default; the rocFFT pipeline and the oracle must agree: identical integer
    results, ZNCC to float rounding, guarded POIs untouched."""
    import oracle
    from opencorr_amd import synth
    ref, tar = speckle_small
    h, w = ref.shape
    xs, ys = synth.poi_grid_2d(h, w, 31, 29, r + 8)
    xs = np.concatenate([xs, [3, w - 2, 150]]).astype(np.float32)
    ys = np.concatenate([ys, [100, 100, 2]]).astype(np.float32)
    base = eng.make_pois2d(xs, ys)
    base[::7, 2] = 1.0   # non-zero initial guesses shift the target window (src/oc_fftcc.cpp:215-216)
    base[::5, 8] = -2.0
    f = eng.FFTCC2D(r, r)
    f.set_images(ref, tar)
    fused = f.compute(base.copy())
    f.set_tuning("fftcc2d_fused", 0)
    piped = f.compute(base.copy())
    want = base.copy()
    oracle.fftcc2d(ref, tar, r, r, want)
    for col in (2, 8, 14, 15):
        assert np.array_equal(fused[:, col], piped[:, col]), col
        assert np.array_equal(fused[:, col], want[:, col]), col
    assert np.abs(fused[:, 16] - piped[:, 16]).max() <= 2e-6
    # the oracle restates the reference's sequential float sums of means and norms (src/oc_fftcc.cpp:198-231); over
    # 1600+ samples they differ from the GPU's tree sums by ~1e-5 (the rocFFT pipeline deviates by the same amount)
    assert np.abs(fused[:, 16] - want[:, 16]).max() <= (1.5e-5 if r <= 16 else 3e-5)
    other = [c for c in range(25) if c not in (2, 8, 14, 15, 16)]
    assert np.array_equal(_bits(fused[:, other]), _bits(base[:, other]))
    assert np.array_equal(_bits(fused[-3:]), _bits(base[-3:]))
    assert (fused[:-3, 16] > 0.5).mean() > 0.9


def test_fftcc2d_every_fused_shape(eng, speckle_small):
    """All 71 window shapes with a single-kernel FFTCC2D -- EVERY square radius from 4 to 32 (29 sides from 8 to 64: the
    5-smooth ones, and since round 4 the sides with a prime factor of 7 ... 31: 14, 22, 26, 28, 34, 38, 42, 44, 46, 52, 56,
    58, 62) and 42 rectangular pairs; one template instance each, fftcc2d_fusedn_impl.h -- on an odd-length queue with
    fractional positions, integer initial guesses and three guard trippers: integers as the oracle and the rocFFT
    pipeline, ZNCC within 3e-5, tripped POIs untouched."""
    import oracle
    ref, tar = speckle_small
    h, w = ref.shape
    P = oracle.P2
    sides = [16, 20, 24, 32, 40, 48, 64]
    shapes = [(r, r) for r in range(4, 33)] + [(a // 2, b // 2) for a in sides for b in sides if a != b]
    assert len(shapes) == 71
    for rx, ry in shapes:
        rng = np.random.default_rng(rx * 100 + ry)
        n = 75
        m = max(rx, ry) + 6
        xs = rng.uniform(m, w - m, n).astype(np.float32)
        ys = rng.uniform(m, h - m, n).astype(np.float32)
        xs[::2] = np.floor(xs[::2])
        ys[::2] = np.floor(ys[::2])
        base = eng.make_pois2d(xs, ys)
        base[:, P["u"]] = rng.integers(-3, 4, n).astype(np.float32)
        base[:, P["v"]] = rng.integers(-3, 4, n).astype(np.float32)
        trip = [7, 40, n - 1]
        base[7, P["x"]] = 2.0
        base[40, P["y"]] = h - 1.0
        base[n - 1, P["u"]] = 5000.0
        f = eng.FFTCC2D(rx, ry)
        f.set_images(ref, tar)
        fused = f.compute(base.copy())
        f.set_tuning("fftcc2d_fused", 0)
        piped = f.compute(base.copy())
        want = base.copy()
        oracle.fftcc2d(ref, tar, rx, ry, want)
        for c in ("u", "v", "u0", "v0"):
            assert np.array_equal(fused[:, P[c]], want[:, P[c]]), (rx, ry, c)
            assert np.array_equal(fused[:, P[c]], piped[:, P[c]]), (rx, ry, c)
        assert np.abs(fused[:, P["zncc"]] - want[:, P["zncc"]]).max() <= 3e-5, (rx, ry)
        assert np.array_equal(_bits(fused[trip]), _bits(base[trip])), (rx, ry)


@pytest.mark.parametrize("count", [1, 2, 101, 128, 777])
def test_fftcc2d_two_pois_per_wave_edge_cases(eng, speckle_small, count):
    """The 32 x 32 kernel serves two POIs per wave (fftcc2d_fused32x2_kernel): odd queue lengths (a lone half-wave at the
    end), guard trippers in the first or the second half of a wave next to a live partner, fractional POI coordinates and
    fractional initial guesses (the window indices are truncations of float sums, src/oc_fftcc.cpp:192-216) -- integers as
    the oracle and the rocFFT pipeline, ZNCC within 1e-5, tripped POIs untouched."""
    import oracle
    ref, tar = speckle_small
    h, w = ref.shape
    rng = np.random.default_rng(count)
    xs = rng.uniform(30, w - 30, count).astype(np.float32)
    ys = rng.uniform(30, h - 30, count).astype(np.float32)
    xs[::3] = np.floor(xs[::3])          # a third on integer positions, the rest fractional
    ys[::3] = np.floor(ys[::3])
    base = eng.make_pois2d(xs, ys)
    P = oracle.P2
    base[:, P["u"]] = rng.uniform(-3, 3, count).astype(np.float32)
    base[:, P["v"]] = rng.uniform(-3, 3, count).astype(np.float32)
    base[::4, P["u"]] = 0.0
    trip = [i for i in (0, 5, 50, 51, count - 1) if 0 <= i < count and count > 2]
    for n, i in enumerate(trip):   # out at the left / right / top / by the displaced target window
        if n % 4 == 0: base[i, P["x"]] = 3.0
        elif n % 4 == 1: base[i, P["x"]] = w - 2.0
        elif n % 4 == 2: base[i, P["y"]] = 2.0
        else: base[i, P["u"]] = 4000.0
    f = eng.FFTCC2D(16, 16)
    f.set_images(ref, tar)
    fused = f.compute(base.copy())
    f.set_tuning("fftcc2d_fused", 0)
    piped = f.compute(base.copy())
    want = base.copy()
    oracle.fftcc2d(ref, tar, 16, 16, want)
    for col in (P["u"], P["v"], P["u0"], P["v0"]):
        assert np.array_equal(fused[:, col], want[:, col]), col
        assert np.array_equal(fused[:, col], piped[:, col]), col
    assert np.abs(fused[:, P["zncc"]] - want[:, P["zncc"]]).max() <= 1e-5
    assert np.abs(fused[:, P["zncc"]] - piped[:, P["zncc"]]).max() <= 2e-6
    other = [c for c in range(25) if c not in (P["u"], P["v"], P["u0"], P["v0"], P["zncc"])]
    assert np.array_equal(_bits(fused[:, other]), _bits(base[:, other]))
    if trip:
        assert np.array_equal(_bits(fused[trip]), _bits(base[trip]))
        assert np.array_equal(_bits(want[trip]), _bits(base[trip]))


@pytest.mark.parametrize("rx,ry", [(8, 16), (16, 8), (10, 24), (32, 12), (20, 16), (24, 32), (16, 20), (12, 8)])
def test_fftcc2d_fused_rectangular_windows(eng, speckle_small, rx, ry):
    """rx != ry with both window sides out of {16, 20, 24, 32, 40, 48, 64}: the register-FFT kernel (fftcc2d_fusedr.hip) instead
    of the five-kernel rocFFT pipeline.  The reference transforms the window's linear buffer re-cut into 2 * rx lines of
    2 * ry elements there (fftwf_plan_dft_r2c_2d(width, height) over data filled [row * width + col], src/oc_fftcc.cpp:40-42,
    204-221) and decodes the peak with the window's width; the oracle restates exactly that, the pipeline and the fused
    kernel must both reproduce it: identical integers, ZNCC to float rounding, guarded POIs untouched."""
    import oracle
    from opencorr_amd import synth
    ref, tar = speckle_small
    h, w = ref.shape
    xs, ys = synth.poi_grid_2d(h, w, 27, 23, max(rx, ry) + 8)
    xs = np.concatenate([xs, [3, w - 2, 150]]).astype(np.float32)
    ys = np.concatenate([ys, [100, 100, 2]]).astype(np.float32)
    base = eng.make_pois2d(xs, ys)
    base[::7, 2] = 1.0
    base[::5, 8] = -2.0
    f = eng.FFTCC2D(rx, ry)
    f.set_images(ref, tar)
    fused = f.compute(base.copy())
    f.set_tuning("fftcc2d_fused", 0)
    piped = f.compute(base.copy())
    want = base.copy()
    oracle.fftcc2d(ref, tar, rx, ry, want)
    for col in (2, 8, 14, 15):
        assert np.array_equal(fused[:, col], piped[:, col]), col
        assert np.array_equal(fused[:, col], want[:, col]), col
    assert np.abs(fused[:, 16] - piped[:, 16]).max() <= 3e-6
    assert np.abs(fused[:, 16] - want[:, 16]).max() <= 3e-5
    other = [c for c in range(25) if c not in (2, 8, 14, 15, 16)]
    assert np.array_equal(_bits(fused[:, other]), _bits(base[:, other]))
    assert np.array_equal(_bits(fused[-3:]), _bits(base[-3:]))


# every radius 4 ... 32 once as rx and once as ry, paired with a different one (a rotation of the 29 radii)
RECT_RADII = [(r, 4 + ((r - 4) + 11) % 29) for r in range(4, 33)]
assert all(rx != ry for rx, ry in RECT_RADII) and sorted(ry for _, ry in RECT_RADII) == list(range(4, 33))


@pytest.mark.parametrize("rx,ry", RECT_RADII + [(4, 32), (32, 4), (31, 32), (5, 4)])
def test_fftcc2d_rect_kernel_every_line_length(eng, speckle_small, rx, ry):
    """Rectangular windows outside the 42 instantiated pairs: ONE kernel with run-time sides (fftcc2d_rect.hip; the pairs that DO
    have an instantiation take it, as before).  Same bars as the instantiated shapes: the reference's re-cut transform
    reproduced -- identical integers against the oracle and the rocFFT pipeline, ZNCC to float rounding, guarded POIs
    untouched -- on an odd-length queue (the last wave of the two-POIs-per-wave instantiation is half empty)."""
    import oracle
    from opencorr_amd import synth
    ref, tar = speckle_small
    h, w = ref.shape
    xs, ys = synth.poi_grid_2d(h, w, 11, 9, max(rx, ry) + 8)
    xs = np.concatenate([xs, [3, w - 2, 150]]).astype(np.float32)
    ys = np.concatenate([ys, [100, 100, 2]]).astype(np.float32)
    base = eng.make_pois2d(xs, ys)
    assert len(base) % 2 == 0
    base = base[1:]
    base[::7, 2] = 1.0
    base[::5, 8] = -2.0
    f = eng.FFTCC2D(rx, ry)
    f.set_images(ref, tar)
    fused = f.compute(base.copy())
    assert np.array_equal(_bits(fused), _bits(f.compute(base.copy())))
    f.set_tuning("fftcc2d_fused", 0)
    piped = f.compute(base.copy())
    want = base.copy()
    oracle.fftcc2d(ref, tar, rx, ry, want)
    for col in (2, 8, 14, 15):
        assert np.array_equal(fused[:, col], piped[:, col]), col
        assert np.array_equal(fused[:, col], want[:, col]), col
    assert np.abs(fused[:, 16] - piped[:, 16]).max() <= 3e-6
    assert np.abs(fused[:, 16] - want[:, 16]).max() <= 3e-5
    other = [c for c in range(25) if c not in (2, 8, 14, 15, 16)]
    assert np.array_equal(_bits(fused[:, other]), _bits(base[:, other]))
    assert np.array_equal(_bits(fused[-3:]), _bits(base[-3:]))
    assert not np.array_equal(_bits(fused[:, 16]), _bits(piped[:, 16]))   # two implementations, not the pipeline twice


def test_fftcc2d_setsubset_replans(eng, speckle_small):
    """FFTCC2D::setSubset between computes (examples/test_3d_dic_epipolar_sift.cpp:188-190): the engine drops its FFT
    plans / picks another kernel for the new window -- fused 32 -> the run-time-sides kernel (rx != ry) -> fused 40 -> fused 32
    again, each pass equal to a fresh engine of that radius and to the oracle."""
    import oracle
    from opencorr_amd import synth
    ref, tar = speckle_small
    h, w = ref.shape
    xs, ys = synth.poi_grid_2d(h, w, 17, 13, 30)
    base = eng.make_pois2d(xs, ys)
    f = eng.FFTCC2D(16, 16)
    f.set_images(ref, tar)
    for rx, ry in [(16, 16), (9, 13), (20, 20), (16, 16)]:
        f.set_subset(rx, ry)
        got = f.compute(base.copy())
        fresh = eng.FFTCC2D(rx, ry)
        fresh.set_images(ref, tar)
        assert np.array_equal(_bits(got), _bits(fresh.compute(base.copy()))), (rx, ry)
        want = base.copy()
        oracle.fftcc2d(ref, tar, rx, ry, want)
        for col in (2, 8, 14, 15):
            assert np.array_equal(got[:, col], want[:, col]), (rx, ry, col)
        assert np.abs(got[:, 16] - want[:, 16]).max() <= 3e-5


@pytest.mark.parametrize("variant,xcd", [(1, 1), (2, 1), (2, 0), (3, 0), (4, 1), (4, 0), (5, 1), (5, 0), (7, 1), (7, 0)])
def test_icgn2d1_variants_identical_bits(eng, speckle_small, variant, xcd):
    """Every kernel variant / workgroup mapping of oc_hip_set_tuning computes the same bits (variants 0, 6 and 8, the measured
    losers, live in the A/B build of the library only: tests/ab/, run by tests/test_gpu_ab_build.py)."""
    import oracle
    from opencorr_amd import synth
    ref, tar = speckle_small
    xs, ys = synth.poi_grid_2d(ref.shape[0], ref.shape[1], 19, 23, 26)
    pois = oracle.make_pois2d(xs, ys)
    oracle.fftcc2d(ref, tar, 16, 16, pois)
    want = pois.copy()
    prep = oracle.Prepared2D(ref, tar)
    oracle.icgn2d1(prep, 16, 16, 0.001, 10, want, order=oracle.ORDER_LANES, lanes=64)
    icgn = eng.ICGN2D1(16, 16, 0.001, 10)
    icgn.set_images(ref, tar)
    icgn.prepare()
    icgn.set_tuning("icgn2d_variant", variant)
    icgn.set_tuning("icgn2d_xcd", xcd)
    got = icgn.compute(pois.copy())
    assert np.array_equal(_bits(got), _bits(want))


@pytest.mark.parametrize("variant", [4, 5])
@pytest.mark.parametrize("dof", [6, 12])
def test_icgn2d_coordinate_table_variants(eng, speckle_small, variant, dof):
    """The variants with a per-workgroup coordinate table (one barrier, then waves may leave early): guard trippers,
    rejected and NaN POIs, a queue length that fills neither the last workgroup nor its last wave's table share, a
    non-square subset, centre offsets, and the fallback to a table-free variant under self-adaptive radii -- all
    bit-identical to the oracle."""
    import oracle
    from opencorr_amd import synth
    ref, tar = speckle_small
    h, w = ref.shape
    rx, ry = (13, 9) if dof == 6 else (10, 12)
    xs, ys = synth.poi_grid_2d(h, w, 11, 9, 24)   # 99 POIs
    pois = oracle.make_pois2d(xs, ys)
    oracle.fftcc2d(ref, tar, 12, 12, pois)
    P = oracle.P2
    extra = oracle.make_pois2d([3.0, 90.0, 90.0, 90.0, w - 12.0], [80.0, 80.0, 80.0, 80.0, 100.0])
    extra[1, P["u"]] = 200.0
    extra[2, P["zncc"]] = -1.0
    extra[3, P["v"]] = np.nan
    extra[4, P["u"]], extra[4, P["ux"]] = 1.0, 0.4
    pois = np.concatenate([extra[:2], pois, extra[2:]]).astype(np.float32)   # 104 POIs, the trippers spread over workgroups
    prep = oracle.Prepared2D(ref, tar)
    fn = oracle.icgn2d1 if dof == 6 else oracle.icgn2d2
    icgn = (eng.ICGN2D1 if dof == 6 else eng.ICGN2D2)(rx, ry, 0.001, 10)
    icgn.set_images(ref, tar)
    icgn.prepare()
    icgn.set_tuning("icgn2d_variant", variant)
    want = pois.copy()
    fn(prep, rx, ry, 0.001, 10, want, order=oracle.ORDER_LANES, lanes=64)
    assert np.array_equal(_bits(icgn.compute(pois.copy())), _bits(want))
    off = np.random.default_rng(variant).uniform(-2, 2, (len(pois), 2)).astype(np.float32)
    want = pois.copy()
    fn(prep, rx, ry, 0.001, 10, want, order=oracle.ORDER_LANES, lanes=64, center_offsets=off)
    assert np.array_equal(_bits(icgn.compute_with_offsets(pois.copy(), off)), _bits(want))
    # self-adaptive radii cannot share a table: the engine falls back to a table-free variant by itself
    sa = pois.copy()
    sa[:, P["srx"]] = np.random.default_rng(1).integers(6, rx + 1, len(sa))
    sa[:, P["sry"]] = np.random.default_rng(2).integers(6, ry + 1, len(sa))
    want = sa.copy()
    fn(prep, rx, ry, 0.001, 10, want, order=oracle.ORDER_LANES, lanes=64, self_adaptive=True)
    icgn.set_self_adaptive(True)
    assert np.array_equal(_bits(icgn.compute(sa.copy())), _bits(want))


@pytest.mark.parametrize("variant", [4, 5])
@pytest.mark.parametrize("dof", [6, 12])
def test_icgn2d_lockstep_barriers_with_mixed_wave_lifetimes(eng, speckle_small, variant, dof):
    """The lockstep sweep barriers (icgn2d.hip, SWEEP_SYNC) sit inside the per-iteration sweep of 8-wave workgroups whose
    waves live for different numbers of iterations.  EVERY workgroup of this queue (8 consecutive POIs, queue order: the
    queue is below the tile-schedule threshold) holds: a guard reject (zncc < 0 on entry), a POI whose first sweep leaves
    the image (abort with -3 after ONE sweep), a NaN guess, a POI that converges in its first iteration (it starts from
    its own converged parameters), a far-off guess that runs to the stop limit (-4), and normal 2-4 iteration POIs.  Must
    terminate and equal the oracle bit for bit, three times in a row (the barriers carry no data, but a wrong pairing
    would hang or desynchronise the COOP inverse)."""
    import oracle
    from opencorr_amd import synth
    ref, tar = speckle_small
    h, w = ref.shape
    r = 16
    P = oracle.P2
    xs, ys = synth.poi_grid_2d(h, w, 32, 30, 26)     # 960 POIs = 120 workgroups of 8
    pois = oracle.make_pois2d(xs, ys)
    oracle.fftcc2d(ref, tar, r, r, pois)
    prep = oracle.Prepared2D(ref, tar)
    fn = oracle.icgn2d1 if dof == 6 else oracle.icgn2d2
    solved = pois.copy()
    fn(prep, r, r, 0.001, 10, solved, order=oracle.ORDER_LANES, lanes=64)
    q = pois.copy()
    slot = np.arange(len(q)) % 8
    q[slot == 0, P["zncc"]] = -2.0                                   # guard: leaves before any sweep
    q[slot == 1, P["u"]] = w - 30.0                                   # in range for the guard (|u| < width), outside for the LUT
    q[slot == 2, P["v"]] = np.nan                                     # guard
    one = slot == 3                                                   # starts converged: exactly one iteration
    q[one, 2:14] = solved[one, 2:14]
    q[slot == 4, P["u"]] += 6.5                                       # far-off guess: wanders to the stop limit or aborts
    q[slot == 4, P["v"]] -= 5.5
    want = q.copy()
    fn(prep, r, r, 0.001, 10, want, order=oracle.ORDER_LANES, lanes=64)
    it = want[:, P["iteration"]]
    assert (want[slot == 1, P["zncc"]] == -3.0).all() and (want[slot == 0, P["zncc"]] == -2.0).all()
    # (ICGN2D2 promotes a first-order guess, src/oc_icgn.cpp:765-770: its second-order terms start at zero -> two iterations)
    assert (it[one & (want[:, P["zncc"]] > 0)] <= (1 if dof == 6 else 3)).mean() > 0.9
    assert ((want[slot == 4, P["zncc"]] == -4.0) | (it[slot == 4] >= 6)).mean() > 0.5    # long-lived waves in (almost) every workgroup
    icgn = (eng.ICGN2D1 if dof == 6 else eng.ICGN2D2)(r, r, 0.001, 10)
    icgn.set_images(ref, tar)
    icgn.prepare()
    icgn.set_tuning("icgn2d_variant", variant)
    for _ in range(3):
        assert np.array_equal(_bits(icgn.compute(q.copy())), _bits(want))


def test_icgn2d_tile_schedule_changes_no_bits(eng, speckle_small):
    """The locality schedule (poi_order.hip) only reorders the independent per-POI solves: a queue long
    enough to engage it gives the same bits as queue order, and the same bits as the oracle."""
    import oracle
    from opencorr_amd import synth
    ref, tar = speckle_small
    h, w = ref.shape
    xs, ys = synth.poi_grid_2d(h, w, 150, 120, 24)  # 18000 POIs >= the 16384 threshold
    rng = np.random.default_rng(3)
    shuffle = rng.permutation(len(xs))  # an arbitrary (not row-major) caller order
    xs, ys = xs[shuffle], ys[shuffle]
    xs = np.concatenate([xs, [2.0, np.float32(w + 50), 100.0]]).astype(np.float32)  # guard trippers
    ys = np.concatenate([ys, [100.0, 60.0, np.float32(-7)]]).astype(np.float32)
    fftcc = eng.FFTCC2D(16, 16)
    fftcc.set_images(ref, tar)
    start = eng.make_pois2d(xs, ys)
    fftcc.compute(start)
    icgn = eng.ICGN2D1(16, 16, 0.001, 10)
    icgn.share_images(fftcc)
    icgn.prepare()
    outs = []
    for px in (0, 64, 160):
        icgn.set_tuning("icgn2d_tile_px", px)
        pois = start.copy()
        icgn.compute(pois)
        outs.append(pois)
    assert np.array_equal(_bits(outs[0]), _bits(outs[1]))
    assert np.array_equal(_bits(outs[0]), _bits(outs[2]))
    sample = start[::9].copy()
    oracle.icgn2d1(oracle.Prepared2D(ref, tar), 16, 16, 0.001, 10, sample, order=oracle.ORDER_LANES, lanes=64)
    assert np.array_equal(_bits(sample), _bits(outs[1][::9]))


@pytest.mark.parametrize("rx,ry", [(16, 16), (15, 15), (7, 9), (20, 20)])
def test_icgn2d1_bit_exact_vs_oracle(eng, speckle_small, rx, ry):
    import oracle
    from opencorr_amd import synth
    ref, tar = speckle_small
    h, w = ref.shape
    xs, ys = synth.poi_grid_2d(h, w, 21, 17, 26)
    pois = oracle.make_pois2d(xs, ys)
    oracle.fftcc2d(ref, tar, rx, ry, pois)
    # edge cases: outside the guard, a wild initial guess (warped subset leaves the image ->
    # -3 from inside the loop), a negative ZNCC on entry, a NaN guess
    extra = oracle.make_pois2d([5, 160, 160, 160], [150, 150, 150, 150])
    extra[1, oracle.P2["u"]] = 200.0
    extra[2, oracle.P2["zncc"]] = -1.0
    extra[3, oracle.P2["u"]] = np.nan
    pois = np.concatenate([pois, extra]).astype(np.float32)
    want = pois.copy()
    prep = oracle.Prepared2D(ref, tar)
    oracle.icgn2d1(prep, rx, ry, 0.001, 10, want, order=oracle.ORDER_LANES, lanes=64)
    got = pois.copy()
    icgn = eng.ICGN2D1(rx, ry, 0.001, 10)
    icgn.set_images(ref, tar)
    icgn.prepare()
    icgn.compute(got)
    P = oracle.P2
    assert np.array_equal(got[:, P["iteration"]], want[:, P["iteration"]])
    assert np.array_equal(got[:, P["zncc"]] < 0, want[:, P["zncc"]] < 0)
    mism = np.argwhere(_bits(got) != _bits(want))
    assert mism.size == 0, "first mismatches (poi, field): %s" % mism[:10].tolist()
    assert want[-4, P["zncc"]] == -3.0 and want[-3, P["zncc"]] == -3.0 and want[-2, P["zncc"]] == -1.0
    ok = want[:-4, P["zncc"]] > 0.9  # the regular grid converges (tiny subsets may hit the -4 path)
    if min(rx, ry) >= 15:
        assert ok.all()


def test_icgn2d1_stop_condition_and_not_converged_flag(eng, speckle_small):
    """stop = 2 forces the -4 path (src/oc_icgn.cpp:329-332); set_iteration mirrors setIteration."""
    import oracle
    from opencorr_amd import synth
    ref, tar = speckle_small
    xs, ys = synth.poi_grid_2d(ref.shape[0], ref.shape[1], 9, 9, 30)
    pois = oracle.make_pois2d(xs, ys)
    oracle.fftcc2d(ref, tar, 16, 16, pois)
    want = pois.copy()
    prep = oracle.Prepared2D(ref, tar)
    oracle.icgn2d1(prep, 16, 16, 1e-7, 2, want, order=oracle.ORDER_LANES)
    icgn = eng.ICGN2D1(16, 16, 0.001, 10)
    icgn.set_images(ref, tar)
    icgn.prepare()
    icgn.set_iteration(1e-7, 2)
    got = icgn.compute(pois.copy())
    assert np.array_equal(_bits(got), _bits(want))
    assert (want[:, oracle.P2["zncc"]] == -4.0).all()


def test_device_resident_pois_and_images(eng, speckle_small):
    """OC_HIP_DEVICE buffers (torch tensors) are used in place on the caller's stream."""
    import torch
    import oracle
    from opencorr_amd import synth
    ref, tar = speckle_small
    xs, ys = synth.poi_grid_2d(ref.shape[0], ref.shape[1], 16, 16, 30)
    host = oracle.make_pois2d(xs, ys)
    f = eng.FFTCC2D(16, 16)
    f.set_images(ref, tar)
    icgn = eng.ICGN2D1(16, 16, 0.001, 10)
    icgn.set_images(ref, tar)
    icgn.prepare()
    want = icgn.compute(f.compute(host.copy()))
    dref, dtar = torch.from_numpy(ref).cuda(), torch.from_numpy(tar).cuda()
    dpois = torch.from_numpy(host).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    f2 = eng.FFTCC2D(16, 16)
    f2.set_stream(stream)
    f2.set_images(dref, dtar)
    i2 = eng.ICGN2D1(16, 16, 0.001, 10)
    i2.set_stream(stream)
    i2.share_images(f2)
    i2.prepare()
    f2.compute(dpois)
    i2.compute(dpois)
    torch.cuda.synchronize()
    assert np.array_equal(_bits(dpois.cpu().numpy()), _bits(want))


def test_golden_oht_on_gpu(eng, golden):
    """The reference's own example end to end on the GPU (FFTCC2D -> ICGN2D1, 30 000 POIs)."""
    import oracle
    P = oracle.P2
    tab, de = golden["table"], golden["deformation"]
    pois = oracle.make_pois2d(tab[:, 0], tab[:, 1])
    f = eng.FFTCC2D(golden["rx"], golden["ry"])
    f.set_images(golden["ref"], golden["tar"])
    f.compute(pois)
    same = (pois[:, P["u"]] == tab[:, 4]) & (pois[:, P["v"]] == tab[:, 5])
    assert same.mean() >= 0.999
    icgn = eng.ICGN2D1(golden["rx"], golden["ry"], golden["conv"], golden["stop"])
    icgn.share_images(f)
    icgn.prepare()
    after_fftcc = pois.copy()
    icgn.compute(pois)
    m = (tab[:, 7] < golden["stop"]) & same
    assert np.abs(pois[m, P["u"]] - tab[m, 2]).max() <= 2e-4
    assert np.abs(pois[m, P["v"]] - tab[m, 3]).max() <= 2e-4
    assert np.abs(pois[m, P["zncc"]] - tab[m, 6]).max() <= 1e-5
    assert (pois[m, P["iteration"]] == tab[m, 7]).mean() >= 0.99
    got = pois[m][:, [P["ux"], P["uy"], P["vx"], P["vy"]]]
    assert np.abs(got - de[m][:, [3, 4, 6, 7]]).max() <= 5e-5
    # and bit-exact against the oracle fed with the GPU's own FFTCC output
    want = after_fftcc.copy()
    prep = oracle.Prepared2D(golden["ref"], golden["tar"])
    oracle.icgn2d1(prep, golden["rx"], golden["ry"], golden["conv"], golden["stop"], want, order=oracle.ORDER_LANES)
    assert np.array_equal(_bits(pois), _bits(want))


@pytest.mark.parametrize("rx,ry", [(20, 20), (12, 16)])
def test_icgn2d2_bit_exact_vs_oracle(eng, rx, ry):
    """ICGN2D2 (12 DoF, src/oc_icgn.cpp:685-898) on a pair with a second-order displacement field."""
    import oracle
    from opencorr_amd import synth
    so = dict(uxx=4e-5, uxy=-2e-5, uyy=3e-5, vxx=-3e-5, vxy=2e-5, vyy=-4e-5)
    ref, tar = synth.speckle_pair_2d(320, 340, seed=11, second_order=so)
    xs, ys = synth.poi_grid_2d(320, 340, 17, 15, 32)
    pois = oracle.make_pois2d(xs, ys)
    oracle.fftcc2d(ref, tar, rx, ry, pois)
    extra = oracle.make_pois2d([4, 170, 170], [160, 160, 160])
    extra[1, oracle.P2["u"]] = 250.0
    extra[2, oracle.P2["zncc"]] = -2.0
    pois = np.concatenate([pois, extra]).astype(np.float32)
    want = pois.copy()
    prep = oracle.Prepared2D(ref, tar)
    oracle.icgn2d2(prep, rx, ry, 0.001, 10, want, order=oracle.ORDER_LANES, lanes=64)
    icgn = eng.ICGN2D2(rx, ry, 0.001, 10)
    icgn.set_images(ref, tar)
    icgn.prepare()
    got = icgn.compute(pois.copy())
    P = oracle.P2
    assert np.array_equal(got[:, P["iteration"]], want[:, P["iteration"]])
    mism = np.argwhere(_bits(got) != _bits(want))
    assert mism.size == 0, "first mismatches (poi, field): %s" % mism[:10].tolist()
    ok = want[:-3, P["zncc"]] > 0.9
    if min(rx, ry) < 20:
        return  # small subsets with 12 DoF: bit-exactness only
    assert ok.mean() > 0.95
    # the second-order terms are recovered (analytic field, tolerance set by image noise)
    m = np.flatnonzero(ok)
    assert np.abs(np.median(want[m, P["uxx"]]) - so["uxx"]) < 2e-5
    assert np.abs(np.median(want[m, P["vyy"]]) - so["vyy"]) < 2e-5
    assert want[-3, P["zncc"]] == -3.0 and want[-2, P["zncc"]] == -3.0 and want[-1, P["zncc"]] == -2.0


@pytest.mark.parametrize("dof", [6, 12])
def test_icgn2d_center_offsets_bit_exact(eng, dof):
    """compute(poi_queue, center_offset_queue) (src/oc_icgn.cpp:353-557, 910-1136): float local
    coordinates shifted by a per-POI offset, target subset centred at POI + offset."""
    import oracle
    from opencorr_amd import synth
    ref, tar = synth.speckle_pair_2d(320, 340, seed=5)
    xs, ys = synth.poi_grid_2d(320, 340, 14, 12, 36)
    pois = oracle.make_pois2d(xs, ys)
    oracle.fftcc2d(ref, tar, 16, 16, pois)
    rng = np.random.default_rng(9)
    off = rng.uniform(-3.0, 3.0, (len(pois), 2)).astype(np.float32)
    off[:5] = 0.0                      # zero offsets must reproduce the plain overload
    off[5] = [0.5, -0.25]
    prep = oracle.Prepared2D(ref, tar)
    want = pois.copy()
    (oracle.icgn2d1 if dof == 6 else oracle.icgn2d2)(prep, 16, 16, 0.001, 10, want, order=oracle.ORDER_LANES,
                                                     lanes=64, center_offsets=off)
    icgn = (eng.ICGN2D1 if dof == 6 else eng.ICGN2D2)(16, 16, 0.001, 10)
    icgn.set_images(ref, tar)
    icgn.prepare()
    got = icgn.compute_with_offsets(pois.copy(), off)
    assert np.array_equal(_bits(got), _bits(want))
    plain = icgn.compute(pois.copy())
    assert np.array_equal(_bits(got[:5]), _bits(plain[:5]))
    assert not np.array_equal(got[5:, 2], plain[5:, 2])
    assert (want[:, oracle.P2["zncc"]] > 0.9).mean() > 0.95
    # device-resident queue + offsets
    import torch
    dp, do = torch.from_numpy(pois.copy()).cuda(), torch.from_numpy(off).cuda()
    icgn.compute_with_offsets(dp, do)
    icgn.synchronize()
    assert np.array_equal(_bits(dp.cpu().numpy()), _bits(want))


@pytest.mark.parametrize("dof", [6, 12])
def test_icgn2d_self_adaptive_radius_bit_exact(eng, dof):
    """DIC::setSelfAdaptive(true): every POI brings its own subset radius (src/oc_icgn.cpp:152-158)."""
    import oracle
    from opencorr_amd import synth
    ref, tar = synth.speckle_pair_2d(330, 350, seed=6)
    xs, ys = synth.poi_grid_2d(330, 350, 13, 11, 40)
    pois = oracle.make_pois2d(xs, ys)
    oracle.fftcc2d(ref, tar, 16, 16, pois)
    rng = np.random.default_rng(10)
    pois[:, 23] = rng.integers(9, 25, len(pois)).astype(np.float32)
    pois[:, 24] = rng.integers(9, 25, len(pois)).astype(np.float32)
    pois[0, 23:25] = [24.9, 9.9]   # fractional radii truncate like the reference's float -> int argument
    pois[1, 23] = -3.0             # unusable radius: rejected in-band
    prep = oracle.Prepared2D(ref, tar)
    want = pois.copy()
    fn = oracle.icgn2d1 if dof == 6 else oracle.icgn2d2
    fn(prep, 5, 5, 0.001, 10, want, order=oracle.ORDER_LANES, lanes=64, self_adaptive=True)
    icgn = (eng.ICGN2D1 if dof == 6 else eng.ICGN2D2)(5, 5, 0.001, 10)   # the engine's own radius is ignored
    icgn.set_images(ref, tar)
    icgn.prepare()
    icgn.set_self_adaptive(True)
    got = icgn.compute(pois.copy())
    assert np.array_equal(_bits(got), _bits(want))
    P = oracle.P2
    assert got[1, P["zncc"]] == -3.0
    ok = np.ones(len(pois), bool)
    ok[1] = False
    assert np.array_equal(got[ok, P["srx"]], np.trunc(pois[ok, 23])) and np.array_equal(got[ok, P["sry"]], np.trunc(pois[ok, 24]))
    assert (got[ok, P["zncc"]] > 0.9).mean() > 0.9
    # and switched off again the engine's radius applies
    icgn.set_self_adaptive(False)
    icgn.set_subset(16, 16)
    base = pois.copy()
    want2 = base.copy()
    fn(prep, 16, 16, 0.001, 10, want2, order=oracle.ORDER_LANES, lanes=64)
    assert np.array_equal(_bits(icgn.compute(base)), _bits(want2))


def test_icgn2d2_golden_soft_anchor_on_gpu(eng, golden, golden_icgn2):
    """ICGN2D2 on the GPU against the reference authors' CUDA ICGN2D2 CSV (soft anchor: same (u0, v0),
    converged values only) and bit-exact against the oracle."""
    import oracle
    from test_oracle_golden import icgn2_soft_anchor_check
    tab = golden_icgn2
    pois = oracle.make_pois2d(tab[:, 0], tab[:, 1])
    pois[:, oracle.P2["u"]], pois[:, oracle.P2["v"]] = tab[:, 4], tab[:, 5]
    icgn = eng.ICGN2D2(16, 16, golden["conv"], golden["stop"])
    icgn.set_images(golden["ref"], golden["tar"])
    icgn.prepare()
    got = icgn.compute(pois.copy())
    icgn2_soft_anchor_check(got, tab)
    want = pois.copy()
    oracle.icgn2d2(oracle.Prepared2D(golden["ref"], golden["tar"]), 16, 16, golden["conv"], golden["stop"], want,
                   order=oracle.ORDER_LANES, lanes=64)
    assert np.array_equal(_bits(got), _bits(want))


def test_edge_cases_of_the_boundary(eng, speckle_small):
    """Empty queue, a queue embedded in wider records (stride_bytes), API misuse reported as errors."""
    import torch
    import oracle
    from opencorr_amd import synth
    ref, tar = speckle_small
    f = eng.FFTCC2D(16, 16)
    f.set_images(ref, tar)
    icgn = eng.ICGN2D1(16, 16, 0.001, 10)
    # compute before set_images / prepare is an error, not a crash or a silent no-op
    with pytest.raises(eng.capi.OpenCorrHipError):
        icgn.compute(eng.make_pois2d([100.0], [100.0]))
    icgn.share_images(f)
    with pytest.raises(eng.capi.OpenCorrHipError):
        icgn.compute(eng.make_pois2d([100.0], [100.0]))
    icgn.prepare()
    # empty queue
    empty = np.zeros((0, 25), np.float32)
    assert f.compute(empty).shape == (0, 25) and icgn.compute(empty).shape == (0, 25)
    # POI records embedded in 32-float rows: only the first 25 floats of every row are touched
    xs, ys = synth.poi_grid_2d(ref.shape[0], ref.shape[1], 7, 5, 40)
    plain = eng.make_pois2d(xs, ys)
    wide = torch.full((len(xs), 32), 7.5, dtype=torch.float32, device="cuda")
    wide[:, :25] = torch.from_numpy(plain).cuda()
    f.compute(wide)
    icgn.compute(wide)
    f.synchronize(); icgn.synchronize()
    got = wide.cpu().numpy()
    want = icgn.compute(f.compute(plain.copy()))
    assert np.array_equal(_bits(got[:, :25]), _bits(want)) and (got[:, 25:] == 7.5).all()
    # bad stride
    with pytest.raises(eng.capi.OpenCorrHipError):
        eng.capi.check(eng.capi.lib().oc_hip_compute(icgn._h, wide.data_ptr(), 3, 99, eng.capi.DEVICE))


def test_large_subsets_and_the_on_chip_limit(eng):
    """r = 36 (73 x 73 = 5329 samples, 84 passes) still runs -- through the variant with the LDS share it needs --
    and stays bit-exact; a subset beyond the on-chip capacity is refused with OC_HIP_ERR_UNSUPPORTED."""
    import oracle
    from opencorr_amd import synth
    ref, tar = synth.speckle_pair_2d(260, 280, seed=12)
    xs, ys = synth.poi_grid_2d(260, 280, 4, 3, 60)
    pois = oracle.make_pois2d(xs, ys)
    oracle.fftcc2d(ref, tar, 16, 16, pois)
    want = pois.copy()
    prep = oracle.Prepared2D(ref, tar)
    oracle.icgn2d1(prep, 36, 36, 0.001, 10, want, order=oracle.ORDER_LANES, lanes=64)
    icgn = eng.ICGN2D1(36, 36, 0.001, 10)
    icgn.set_images(ref, tar)
    icgn.prepare()
    got = icgn.compute(pois.copy())
    assert np.array_equal(_bits(got), _bits(want))
    assert (want[:, oracle.P2["zncc"]] > 0.9).all()
    big = eng.ICGN2D1(72, 72, 0.001, 10)   # 145 x 145 = 21 025 samples > 20 480 (160 KB of LDS / 2 arrays)
    big.set_images(ref, tar)
    big.prepare()
    with pytest.raises(eng.capi.OpenCorrHipError) as err:
        big.compute(pois.copy())
    assert err.value.status == eng.capi.ERR_UNSUPPORTED


def test_candidate_batching_and_select_best(eng, speckle_small):
    """The EpipolarSearch pattern (src/oc_epipolar_search.cpp:150-190) as one batch: several trial guesses per POI are
    refined in ONE ICGN2D1 launch, then oc_hip_select_best keeps the highest ZNCC per POI -- identical to doing it
    candidate by candidate and sorting on the host."""
    import torch
    from opencorr_amd import synth
    ref, tar = speckle_small
    P = {"u": 2, "v": 8, "zncc": 16}
    xs, ys = synth.poi_grid_2d(ref.shape[0], ref.shape[1], 14, 12, 30)
    n = len(xs)
    rng = np.random.default_rng(6)
    counts = rng.integers(0, 6, n)          # 0..5 trials per POI, some POIs get none
    counts[:3] = [5, 1, 0]
    starts = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint32)
    cx = np.repeat(xs, counts).astype(np.float32)
    cy = np.repeat(ys, counts).astype(np.float32)
    cand = eng.make_pois2d(cx, cy)
    trial = np.concatenate([np.arange(c) for c in counts]).astype(np.float32) if counts.sum() else np.zeros(0, np.float32)
    cand[:, P["u"]] = 2.0 + 3.0 * (trial - 2)     # trials along a line, like the epipolar search; trial 2 is close
    cand[:, P["v"]] = -2.0
    icgn = eng.ICGN2D1(16, 16, 0.001, 10)
    icgn.set_images(ref, tar)
    icgn.prepare()
    icgn.compute(cand)                            # one launch for all trials of all POIs
    pois = eng.make_pois2d(xs, ys)
    pois[:, 20:23] = 4.5                          # strain fields: not touched by the selection
    want = pois.copy()
    for s in range(n):
        seg = cand[starts[s]:starts[s + 1]]
        z = seg[:, P["zncc"]]
        if len(seg) == 0 or np.all(np.isnan(z)):
            continue
        k = int(np.nanargmax(z))                  # first maximum
        want[s, 2:20] = seg[k, 2:20]
    got = icgn.select_best(cand, starts, pois.copy())
    assert np.array_equal(_bits(got), _bits(want))
    have = counts >= 3
    assert (got[have, P["zncc"]] > 0.9).mean() > 0.95 and np.abs(got[have, P["u"]] - 2.3).max() < 0.5
    # device-resident queues
    d = icgn.select_best(torch.from_numpy(cand).cuda(), torch.from_numpy(starts.astype(np.int32)).cuda(),
                         torch.from_numpy(pois.copy()).cuda())
    assert np.array_equal(_bits(d.cpu().numpy()), _bits(want))


def test_oversized_images_are_refused_not_wrapped():
    """The 2D solvers address image-sized arrays with 32-bit byte offsets: an image beyond the limits must come back as
    OC_HIP_ERR_UNSUPPORTED from prepare() (and compute()), never as silently wrong table look-ups."""
    import torch
    import opencorr_amd
    from opencorr_amd import capi
    dev = torch.device("cuda", 0)
    # width >= 2^22: the row offset (row * width * 4) no longer fits the 24-bit multiply
    wide = torch.zeros((5, 1 << 22), dtype=torch.float32, device=dev)
    icgn = opencorr_amd.ICGN2D1(2, 2, 0.001, 10)
    icgn.set_images(wide, wide)
    with pytest.raises(capi.OpenCorrHipError) as e:
        icgn.prepare()
    assert e.value.status == capi.ERR_UNSUPPORTED
    # NR2D1 reads three tables through one descriptor each: 2^26 pixels is its limit (8192 x 8192 passes, one row more does not)
    big = torch.zeros((8193, 8192), dtype=torch.float32, device=dev)
    nr = opencorr_amd.NR2D1(16, 16, 0.001, 10)
    nr.set_images(big, big)
    with pytest.raises(capi.OpenCorrHipError) as e:
        nr.prepare()
    assert e.value.status == capi.ERR_UNSUPPORTED
    # ... while ICGN2D1 (one descriptor per table plane) accepts that image
    icgn2 = opencorr_amd.ICGN2D1(16, 16, 0.001, 10)
    icgn2.set_images(big, big)
    icgn2.prepare()
    icgn2.synchronize()


def test_engine_lifecycle_does_not_leak(eng, speckle_small):
    """create -> setImages -> prepare -> compute -> destroy, many times, with every engine kind: the device's free memory
    comes back (the engines allocate with hipMalloc, so torch's mem_get_info sees it)."""
    import gc
    import torch
    from opencorr_amd import synth
    ref, tar = speckle_small
    xs, ys = synth.poi_grid_2d(ref.shape[0], ref.shape[1], 9, 9, 30)
    pois = eng.make_pois2d(xs, ys)

    def cycle():
        f = eng.FFTCC2D(16, 16)
        f.set_images(ref, tar)
        q = f.compute(pois.copy())
        for cls in (eng.ICGN2D1, eng.ICGN2D2, eng.NR2D1, eng.ICLM2D1):
            g = cls(16, 16, 0.001, 10)
            g.share_images(f) if cls is not eng.NR2D1 else g.set_images(ref, tar)
            g.prepare()
            g.compute(q.copy())
            del g
        st = eng.Strain(30.0, 5)
        st.prepare(q)
        st.compute(q)
        del st, f

    def free_after(n):
        for _ in range(n):
            cycle()
        gc.collect()
        torch.cuda.synchronize()
        return torch.cuda.mem_get_info()[0]

    free_after(5)            # one-off allocations of the runtime (code objects, scratch, pools) happen here
    f1 = free_after(10)
    f2 = free_after(15)
    assert f1 - f2 < 4 << 20, (f1, f2)   # steady state: 15 more cycles of every engine kind cost nothing


def test_engines_are_usable_from_several_host_threads(eng, speckle_small):
    """Four host threads, each with its own FFTCC2D + ICGN2D1 pair, plus two more threads that share ONE pair (the library
    serialises calls on an engine; the reference's compute(POI2D*) is called from OpenMP regions,
    src/oc_epipolar_search.cpp:184-188): every thread gets the bits a lone run produces."""
    import threading
    from opencorr_amd import synth
    ref, tar = speckle_small
    xs, ys = synth.poi_grid_2d(ref.shape[0], ref.shape[1], 13, 11, 30)
    base = eng.make_pois2d(xs, ys)

    def make():
        f = eng.FFTCC2D(16, 16)
        f.set_images(ref, tar)
        g = eng.ICGN2D1(16, 16, 0.001, 10)
        g.share_images(f)
        g.prepare()
        return f, g

    f0, g0 = make()
    want = g0.compute(f0.compute(base.copy()))
    shared = make()
    results, errors = {}, []

    def work(k, pair):
        try:
            f, g = pair if pair is not None else make()
            for rep in range(6):
                q = base[k % 3:].copy()      # different queue lengths per thread
                q = g.compute(f.compute(q))
                results[(k, rep)] = q
        except Exception as e:  # noqa: BLE001
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=work, args=(k, None)) for k in range(4)]
    threads += [threading.Thread(target=work, args=(k, shared)) for k in (4, 5)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert len(results) == 36
    for (k, rep), q in results.items():
        assert np.array_equal(_bits(q), _bits(want[k % 3:])), (k, rep)


def test_stream_switches_are_device_ordered_and_survive_a_dead_stream(eng, speckle_small):
    """oc_hip_set_stream / oc_hip_reset_stream: the outgoing stream is ordered before the incoming one on
    the DEVICE (event), not drained from the host; a stream the caller has ALREADY DESTROYED has nothing left to drain, so
    naming another stream -- or going back to the engine's own -- succeeds and the engine keeps working.  Results are the
    bits a lone engine produces, whichever stream each call ran on."""
    import ctypes
    import torch
    from opencorr_amd import synth
    ref, tar = speckle_small
    xs, ys = synth.poi_grid_2d(ref.shape[0], ref.shape[1], 11, 9, 30)
    base = eng.make_pois2d(xs, ys)
    f0 = eng.FFTCC2D(16, 16)
    f0.set_images(ref, tar)
    g0 = eng.ICGN2D1(16, 16, 0.001, 10)
    g0.share_images(f0)
    g0.prepare()
    want = g0.compute(f0.compute(base.copy()))

    from opencorr_amd import capi
    hip = capi.hip_runtime()  # THE runtime of this process, never a second copy by name
    hip.hipStreamCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
    hip.hipStreamDestroy.argtypes = [ctypes.c_void_p]
    hip.hipStreamSynchronize.argtypes = [ctypes.c_void_p]

    def new_stream():
        h = ctypes.c_void_p()
        assert hip.hipStreamCreate(ctypes.byref(h)) == 0
        return h

    dev = torch.device("cuda", 0)
    f = eng.FFTCC2D(16, 16)
    g = eng.ICGN2D1(16, 16, 0.001, 10)
    s1 = new_stream()
    f.set_stream(s1.value)
    g.set_stream(s1.value)
    f.set_images(ref, tar)
    g.share_images(f)
    g.prepare()                    # gradients + table are still in flight on s1 ...
    s2 = new_stream()
    f.set_stream(s2.value)         # ... when both engines hop to s2: ordered by an event, no host wait
    g.set_stream(s2.value)
    q = torch.from_numpy(base).to(dev)
    torch.cuda.synchronize()
    g.compute(f.compute(q))
    assert hip.hipStreamSynchronize(s2) == 0
    assert np.array_equal(_bits(q.cpu().numpy()), _bits(want))
    # destroy s1 (no longer in use), then s2 WHILE the engines still name it; set_stream(new) must install the new one
    assert hip.hipStreamSynchronize(s1) == 0 and hip.hipStreamDestroy(s1) == 0
    assert hip.hipStreamDestroy(s2) == 0
    s3 = new_stream()
    f.set_stream(s3.value)
    g.set_stream(s3.value)
    q = torch.from_numpy(base).to(dev)
    torch.cuda.synchronize()
    g.compute(f.compute(q))
    assert hip.hipStreamSynchronize(s3) == 0
    assert np.array_equal(_bits(q.cpu().numpy()), _bits(want))
    # and reset_stream always succeeds, dead outgoing stream or not
    assert hip.hipStreamDestroy(s3) == 0
    f.reset_stream()
    g.reset_stream()
    got = g.compute(f.compute(base.copy()))
    assert np.array_equal(_bits(got), _bits(want))


def test_share_images_joins_the_donors_stream(eng, speckle_small):
    """`icgn.share_images(fftcc); icgn.prepare()` -- prepare() receives no tensor to take a stream from, so
    the borrower joins the donor's (adopted) stream; the in-place device images produced on a NON-default torch stream are
    then read in order.  A tensor on the wrong device is refused."""
    import torch
    from opencorr_amd import synth
    ref, tar = speckle_small
    xs, ys = synth.poi_grid_2d(ref.shape[0], ref.shape[1], 9, 9, 30)
    base = eng.make_pois2d(xs, ys)
    f0 = eng.FFTCC2D(16, 16)
    f0.set_images(ref, tar)
    g0 = eng.ICGN2D1(16, 16, 0.001, 10)
    g0.share_images(f0)
    g0.prepare()
    want = g0.compute(f0.compute(base.copy()))
    dev = torch.device("cuda", 0)
    side = torch.cuda.Stream(device=dev)
    ref_h, tar_h = torch.from_numpy(ref).pin_memory(), torch.from_numpy(tar).pin_memory()
    with torch.cuda.stream(side):
        # a long chain of work on the side stream ends in the images: anything not ordered behind it reads garbage
        junk = torch.zeros((4096, 4096), device=dev)
        for _ in range(20):
            junk = junk @ junk
        d_ref = torch.empty(ref.shape, device=dev).copy_(ref_h, non_blocking=True)
        d_tar = torch.empty(tar.shape, device=dev).copy_(tar_h, non_blocking=True)
        f = eng.FFTCC2D(16, 16)
        f.set_images(d_ref, d_tar)        # adopts `side`
        g = eng.ICGN2D1(16, 16, 0.001, 10)
        g.share_images(f)                  # joins `side` too
        assert g._auto_stream == side.cuda_stream and not g._on_private_stream
        g.prepare()
        q = torch.from_numpy(base).to(dev, non_blocking=False)
        g.compute(f.compute(q))
    side.synchronize()
    assert np.array_equal(_bits(q.cpu().numpy()), _bits(want))
    if torch.cuda.device_count() > 1:
        with pytest.raises(ValueError):
            f.compute(torch.zeros((4, 25), device=torch.device("cuda", 1)))


def test_compute_chain_matches_separate_calls(eng, speckle_small):
    """oc_hip_compute_chain (FFTCC2D -> ICGN2D1 over one queue, one PCIe round trip for host queues): the bits of the two
    separate calls -- host queue in one piece, host queue in pipeline chunks, device queue; a follower whose prepare() is
    still in flight on its own stream; bad chains are refused."""
    import torch
    from opencorr_amd import capi, synth
    ref, tar = speckle_small
    xs, ys = synth.poi_grid_2d(ref.shape[0], ref.shape[1], 13, 11, 30)
    base = eng.make_pois2d(xs, ys)
    f = eng.FFTCC2D(16, 16)
    f.set_images(ref, tar)
    g = eng.ICGN2D1(16, 16, 0.001, 10)
    g.share_images(f)
    g.prepare()
    want = g.compute(f.compute(base.copy()))
    assert np.array_equal(_bits(eng.compute_chain([f, g], base.copy())), _bits(want))
    q = torch.from_numpy(base).to("cuda:0")
    eng.compute_chain([f, g], q)
    torch.cuda.synchronize()
    assert np.array_equal(_bits(q.cpu().numpy()), _bits(want))
    # three engines: FFTCC -> ICGN2D1 -> ICGN2D2 (refines the first-order result further)
    g2 = eng.ICGN2D2(16, 16, 0.001, 10)
    g2.share_images(f)
    g2.prepare()                                   # no synchronisation: the chain orders itself behind it
    got3 = eng.compute_chain([f, g, g2], base.copy())
    want3 = g2.compute(want.copy())
    assert np.array_equal(_bits(got3), _bits(want3))
    # a queue long enough for the chunked pipeline (2 x 16384 POIs and a remainder)
    rng = np.random.default_rng(3)
    many = eng.make_pois2d(rng.uniform(30, ref.shape[1] - 31, 40000).astype(np.float32),
                           rng.uniform(30, ref.shape[0] - 31, 40000).astype(np.float32))
    for e in (f, g):
        e.set_tuning("host_chunk", 16384)
    want_many = g.compute(f.compute(many.copy()))
    assert np.array_equal(_bits(eng.compute_chain([f, g], many.copy())), _bits(want_many))
    # refused: an engine twice, 2D with 3D, nothing
    with pytest.raises(capi.OpenCorrHipError):
        eng.compute_chain([g, g], base.copy())
    f3 = eng.FFTCC3D(8, 8, 8)
    with pytest.raises(capi.OpenCorrHipError):
        eng.compute_chain([f, f3], base.copy())
    with pytest.raises(ValueError):
        eng.compute_chain([], base.copy())
    # the engines are still usable on their own afterwards
    assert np.array_equal(_bits(g.compute(f.compute(base.copy()))), _bits(want))
