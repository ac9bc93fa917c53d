"""GPU parity of Strain against the oracle and the reference's golden table.
Bars: bit-exact vs the oracle (same neighbour sets, same row order, same double sums -> same float bits); the OHT
table within the tolerance of the oracle's own golden test; untouched fields stay untouched."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def test_strain2d_on_the_reference_table(golden_strain):
    import opencorr_amd as eng
    import oracle
    from test_oracle_strain import strain_golden_check, strain_queue_from_golden
    g = golden_strain
    p = strain_queue_from_golden(g)
    want = p.copy()
    oracle.strain2d(want, g["radius"], g["neighbors"], g["zncc_threshold"], g["approximation"])
    st = eng.Strain(g["radius"], g["neighbors"])
    st.prepare(p)
    got = st.compute(p.copy())
    strain_golden_check(got, g)
    assert np.array_equal(_bits(got), _bits(want))


@pytest.mark.parametrize("approximation", [1, 2])
def test_strain2d_bit_exact_on_scattered_pois(approximation):
    import opencorr_amd as eng
    import oracle
    from test_oracle_strain import affine_queue_2d
    P = oracle.P2
    p, _ = affine_queue_2d(n=20000, seed=11, extent=900.0)
    rng = np.random.default_rng(2)
    p[:, P["u"]] += rng.normal(0, 0.02, len(p)).astype(np.float32)
    p[:, P["v"]] += rng.normal(0, 0.02, len(p)).astype(np.float32)
    bad = rng.random(len(p)) < 0.15
    p[bad, P["zncc"]] = rng.choice(np.array([-3.0, -4.0, 0.5], dtype=np.float32), bad.sum())
    p[rng.random(len(p)) < 0.01, P["u"]] = np.nan  # NaN displacement with a good ZNCC poisons its neighbours alike
    p[:, P["exx"]] = 7.0  # sentinel: untouched fields must survive
    want = p.copy()
    oracle.strain2d(want, 17.5, 6, 0.9, approximation)
    st = eng.Strain(17.5, 6)
    st.set_approximation(approximation)
    st.prepare(p)
    got = st.compute(p.copy())
    both_nan = np.isnan(got) & np.isnan(want)
    assert np.array_equal(_bits(got)[~both_nan], _bits(want)[~both_nan])
    assert (got[bad, P["exx"]] == 7.0).all()
    assert (got[~bad, P["exx"]] != 7.0).mean() > 0.99


def test_strain2d_knn_path_and_device_resident_queue():
    import torch
    import opencorr_amd as eng
    import oracle
    P = oracle.P2
    rng = np.random.default_rng(4)
    # sparse cloud + a dense blob: with radius 6 most sparse POIs see fewer than 5 neighbours -> KNN path
    xs = np.concatenate([rng.random(3000) * 2000, 500 + rng.random(3000) * 60]).astype(np.float32)
    ys = np.concatenate([rng.random(3000) * 1500, 400 + rng.random(3000) * 60]).astype(np.float32)
    p = oracle.make_pois2d(xs, ys)
    p[:, P["u"]] = (0.002 * xs - 0.001 * ys + rng.normal(0, 0.01, len(xs))).astype(np.float32)
    p[:, P["v"]] = (0.001 * xs + 0.003 * ys).astype(np.float32)
    p[:, P["zncc"]] = np.where(rng.random(len(xs)) < 0.1, 0.3, 0.97).astype(np.float32)
    want = p.copy()
    oracle.strain2d(want, 6.0, 5)
    st = eng.Strain(6.0, 5)
    d = torch.from_numpy(p.copy()).cuda()
    st.set_stream(torch.cuda.current_stream().cuda_stream)
    st.prepare(d)
    st.compute(d)
    torch.cuda.synchronize()
    got = d.cpu().numpy()
    assert np.array_equal(_bits(got), _bits(want))
    touched = got[:, P["exx"]] != 0
    assert 0.3 < touched[:3000].mean() < 1.0  # the KNN path fitted some and left others (filtered neighbours) alone
    # a second compute on new displacements reuses the prepared grid
    p2 = p.copy()
    p2[:, P["u"]] *= 2
    want2 = p2.copy()
    oracle.strain2d(want2, 6.0, 5)
    got2 = st.compute(torch.from_numpy(p2).cuda()).cpu().numpy()
    assert np.array_equal(_bits(got2), _bits(want2))


@pytest.mark.parametrize("approximation", [1, 2])
def test_strain3d_bit_exact(approximation):
    import opencorr_amd as eng
    import oracle
    from test_oracle_strain import affine_queue_3d
    P = oracle.P3
    p, _ = affine_queue_3d(n=12000, seed=9, extent=200.0)
    rng = np.random.default_rng(6)
    for k in ("u", "v", "w"):
        p[:, P[k]] += rng.normal(0, 0.02, len(p)).astype(np.float32)
    bad = rng.random(len(p)) < 0.1
    p[bad, P["zncc"]] = -4.0
    want = p.copy()
    oracle.strain3d(want, 21.0, 10, 0.9, approximation)
    st = eng.Strain(21.0, 10)
    st.set_approximation(approximation)
    st.prepare(p)
    got = st.compute(p.copy())
    assert np.array_equal(_bits(got), _bits(want))
    assert (got[~bad, P["ezz"]] != 0).mean() > 0.9


def test_strain_errors_and_lifecycle():
    import opencorr_amd as eng
    import oracle
    p = oracle.make_pois2d(np.arange(50, dtype=np.float32), np.arange(50, dtype=np.float32))
    st = eng.Strain(10.0, 5)
    with pytest.raises(eng.capi.OpenCorrHipError):
        st.compute(p)  # no prepare
    st.prepare(p)
    with pytest.raises(eng.capi.OpenCorrHipError):
        st.compute(p[:40].copy())  # a different queue
    st.set_subregion_radius(12.0)
    with pytest.raises(eng.capi.OpenCorrHipError):
        st.compute(p)  # radius changed: the grid is stale
    with pytest.raises(eng.capi.OpenCorrHipError):
        st.set_neighbor_min(1000)
    with pytest.raises(eng.capi.OpenCorrHipError):
        eng.Strain(-1.0, 5)
    st.prepare(p[:0])  # empty queue is a no-op
    st.compute(p[:0])


# ---- RegionFit2D / RegionFit3D --------------------------------------------------------------------------------------
def test_region_fit2d_bit_exact_including_knn_and_outside_queries():
    import opencorr_amd as eng
    import oracle
    from test_oracle_strain import affine_queue_2d
    P = oracle.P2
    cloud, _ = affine_queue_2d(n=15000, seed=31, extent=800.0)
    rng = np.random.default_rng(12)
    cloud[:, P["u"]] += rng.normal(0, 0.02, len(cloud)).astype(np.float32)
    # a hole in the cloud (the unreliable region) -> queries there take the KNN path
    hole = (np.abs(cloud[:, 0] - 400) < 60) & (np.abs(cloud[:, 1] - 300) < 60)
    cloud = np.ascontiguousarray(cloud[~hole])
    qx = np.concatenate([rng.random(4000) * 800, 340 + rng.random(500) * 120, [-500.0, 5000.0]]).astype(np.float32)
    qy = np.concatenate([rng.random(4000) * 560, 240 + rng.random(500) * 120, [100.0, -900.0]]).astype(np.float32)
    q = oracle.make_pois2d(qx, qy)
    q[:, P["zncc"]] = -4.0
    q[:, P["uxx"]] = 5.0
    want = q.copy()
    oracle.region_fit(cloud, want, 14.0, 7)
    rf = eng.RegionFit(14.0, 7)
    rf.set_neighbor(cloud)
    rf.prepare()
    got = rf.compute(q.copy())
    assert np.array_equal(_bits(got), _bits(want))
    assert (got[:, P["zncc"]] == 0).all() and (got[:, P["uxx"]] == 5.0).all()


def test_region_fit3d_bit_exact_and_device_queues():
    import torch
    import opencorr_amd as eng
    import oracle
    from test_oracle_strain import affine_queue_3d
    P = oracle.P3
    cloud, _ = affine_queue_3d(n=9000, seed=17, extent=150.0)
    rng = np.random.default_rng(5)
    xyz = (rng.random((2500, 3)) * 170 - 10).astype(np.float32)  # some outside the cloud's bounding box
    q = oracle.make_pois3d(xyz[:, 0], xyz[:, 1], xyz[:, 2])
    want = q.copy()
    oracle.region_fit(cloud, want, 16.0, 9)
    rf = eng.RegionFit(16.0, 9)
    dc, dq = torch.from_numpy(cloud).cuda(), torch.from_numpy(q.copy()).cuda()
    rf.set_stream(torch.cuda.current_stream().cuda_stream)
    rf.set_neighbor(dc)
    rf.prepare()
    rf.compute(dq)
    torch.cuda.synchronize()
    assert np.array_equal(_bits(dq.cpu().numpy()), _bits(want))


def test_region_fit_then_icgn_recovers_bad_initial_guesses(speckle_small):
    """The loop of examples/test_3d_reconstruction_sift_icgn2_regfit.cpp:225-245: POIs that failed are re-initialised
    from their reliable neighbours and refined again."""
    import opencorr_amd as eng
    import oracle
    from opencorr_amd import synth
    ref, tar = speckle_small
    P = oracle.P2
    xs, ys = synth.poi_grid_2d(ref.shape[0], ref.shape[1], 30, 26, 30)
    fftcc = eng.FFTCC2D(16, 16)
    fftcc.set_images(ref, tar)
    pois = eng.make_pois2d(xs, ys)
    fftcc.compute(pois)
    rng = np.random.default_rng(9)
    spoiled = rng.random(len(pois)) < 0.2
    pois[spoiled, P["u"]] += 9.0  # a guess far outside the convergence radius
    icgn = eng.ICGN2D1(16, 16, 0.001, 10)
    icgn.share_images(fftcc)
    icgn.prepare()
    icgn.compute(pois)
    good = pois[:, P["zncc"]] > 0.9
    assert (~good).sum() >= 0.5 * spoiled.sum()
    reliable, unreliable = np.ascontiguousarray(pois[good]), np.ascontiguousarray(pois[~good])
    rf = eng.RegionFit(40.0, 6)
    rf.set_neighbor(reliable)
    rf.prepare()
    rf.compute(unreliable)
    icgn.compute(unreliable)
    assert (unreliable[:, P["zncc"]] > 0.9).mean() > 0.95


def _host_split(q, low, high, crit, P):
    """The selection loop of examples/test_3d_reconstruction_sift_icgn2_regfit.cpp:216-229 in NumPy."""
    z, c = q[:, P["zncc"]], q[:, P["convergence"]]
    unr = (z < low) | (c > crit)
    rel = ~unr & (z >= high)
    return np.ascontiguousarray(q[rel]), np.ascontiguousarray(q[unr]), np.flatnonzero(unr).astype(np.uint32)


@pytest.mark.parametrize("resident", [True, False])
def test_region_fit_icgn_loop_without_a_host_hop(speckle_small, resident):
    """The whole RegionFit -> re-ICGN loop (examples/test_3d_reconstruction_sift_icgn2_regfit.cpp:214-260) with the queue
    resident in HBM: reliable / unreliable selection (oc_hip_split_reliable), RegionFit over the reliable set, ICGN over the
    unreliable one, merge of the recovered POIs (oc_hip_merge_recovered) -- two rounds, no record ever crosses PCIe.  Must
    equal, bit for bit, the same loop with the selections done on the host in NumPy (`resident` False: the C-ABI's
    OC_HIP_HOST form of the two calls against NumPy as well)."""
    import torch
    import opencorr_amd as eng
    import oracle
    from opencorr_amd import synth
    ref, tar = speckle_small
    P = oracle.P2
    xs, ys = synth.poi_grid_2d(ref.shape[0], ref.shape[1], 30, 26, 30)
    fftcc = eng.FFTCC2D(16, 16)
    fftcc.set_images(ref, tar)
    start = eng.make_pois2d(xs, ys)
    fftcc.compute(start)
    rng = np.random.default_rng(9)
    spoiled = rng.random(len(start)) < 0.2
    start[spoiled, P["u"]] += 9.0          # guesses far outside the convergence radius
    start[::37, P["v"]] = np.nan            # and a few NaN guesses: rejected with -5 / guard, in neither set
    icgn = eng.ICGN2D1(16, 16, 0.001, 10)
    icgn.share_images(fftcc)
    icgn.prepare()
    low, high, crit = 0.7, 0.9, 0.001
    rf = eng.RegionFit(40.0, 6)

    # ---- the reference loop, selections on the host
    want = icgn.compute(start.copy())
    rel_w, unr_w, idx_w = _host_split(want, low, high, crit, P)
    assert len(unr_w) >= 0.5 * spoiled.sum() and len(rel_w) > 0.6 * len(want)
    history = []
    for _ in range(2):
        rf.set_neighbor(rel_w)
        rf.prepare()
        rf.compute(unr_w)
        icgn.compute(unr_w)
        ok = (unr_w[:, P["zncc"]] >= high) & (unr_w[:, P["convergence"]] <= crit)
        want[idx_w[ok]] = unr_w[ok]
        rel_w = np.ascontiguousarray(np.concatenate([rel_w, unr_w[ok]]))
        unr_w, idx_w = np.ascontiguousarray(unr_w[~ok]), idx_w[~ok]
        history.append((int(ok.sum()), len(unr_w)))
    assert history[0][0] > 0.8 * spoiled.sum()   # the first round repairs most of the spoiled POIs

    # ---- the same loop through the engine's own selections
    dev = torch.device("cuda", 0)
    q = torch.from_numpy(start).to(dev) if resident else start.copy()
    icgn.compute(q)
    rel, n_rel, unr, idx, n_unr = icgn.split_reliable(q, low, high, crit)
    got_history = []
    for _ in range(2):
        rf.set_neighbor(rel[:n_rel])
        rf.prepare()
        rf.compute(unr[:n_unr])
        icgn.compute(unr[:n_unr])
        n_rec, n_rem = icgn.merge_recovered(q, unr, idx, n_unr, high, crit, rel, n_rel)
        n_rel, n_unr = n_rel + n_rec, n_rem
        got_history.append((n_rec, n_rem))
    if resident:
        torch.cuda.synchronize()
        q, rel, unr, idx = (t.cpu().numpy() for t in (q, rel, unr, idx))
    assert got_history == history
    assert np.array_equal(_bits(q), _bits(want))
    assert np.array_equal(_bits(rel[:n_rel]), _bits(rel_w))
    assert np.array_equal(_bits(unr[:n_unr]), _bits(unr_w)) and np.array_equal(np.asarray(idx[:n_unr], dtype=np.uint32), idx_w)


def test_split_reliable_edge_cases():
    """Empty queues, queues where one set is empty, 3D records, a queue longer than one scan block row (> 256 * 1024 POIs)."""
    import torch
    import opencorr_amd as eng
    import oracle
    P3 = oracle.P3
    g = eng.ICGN3D1(8, 8, 8, 0.001, 10)
    rng = np.random.default_rng(4)
    n = 300000
    q = np.zeros((n, 31), np.float32)
    q[:, 0] = np.arange(n)
    q[:, P3["zncc"]] = rng.uniform(0.5, 1.0, n)
    q[:, P3["convergence"]] = rng.uniform(0, 0.002, n)
    q[::101, P3["zncc"]] = np.nan
    rel_w, unr_w, idx_w = _host_split(q, 0.7, 0.9, 0.001, P3)
    for queue in (q, torch.from_numpy(q).to("cuda:0")):
        rel, n_rel, unr, idx, n_unr = g.split_reliable(queue, 0.7, 0.9, 0.001)
        if not isinstance(rel, np.ndarray):
            rel, unr, idx = rel.cpu().numpy(), unr.cpu().numpy(), idx.cpu().numpy()
        assert n_rel == len(rel_w) and n_unr == len(unr_w)
        assert np.array_equal(_bits(rel[:n_rel]), _bits(rel_w)) and np.array_equal(_bits(unr[:n_unr]), _bits(unr_w))
        assert np.array_equal(np.asarray(idx[:n_unr], dtype=np.uint32), idx_w)
    # nothing unreliable / nothing reliable / empty
    rel, n_rel, unr, idx, n_unr = g.split_reliable(q[:1000], -1.0, 0.0, 1.0)
    assert n_unr == 0 and n_rel == int((~np.isnan(q[:1000, P3["zncc"]])).sum())
    rel, n_rel, unr, idx, n_unr = g.split_reliable(q[:1000], 2.0, 3.0, 1.0)
    assert n_rel == 0 and n_unr == int((~np.isnan(q[:1000, P3["zncc"]])).sum())
    assert g.split_reliable(q[:0], 0.7, 0.9, 0.001)[1::3] == (0, 0)
    assert g.merge_recovered(q, unr, idx, 0, 0.9, 0.001, rel, 0) == (0, 0)


def test_split_merge_padded_rows_bad_indices_and_buffer_checks():
    """Round-3 advisor findings on the two selection calls:
    * the record type is the ENGINE's, not guessed from the row width: a POI2D queue padded to 32 floats per row is split
      on result.zncc = float 16 / convergence = float 18 (a POI3D reading would look at floats 18 / 20),
    * HOST queues with stride > record size: the bytes between records in `reliable` / `unreliable` / the main queue are
      the caller's and stay untouched (the DEVICE path never touched them),
    * an `unreliable_index` entry outside the main queue is refused -- nothing is written through it, on either path,
    * CPU torch tensors, wrong dtypes, mismatched strides or memory kinds raise ValueError instead of reaching a kernel."""
    import torch
    import opencorr_amd as eng
    import oracle
    P = oracle.P2
    g = eng.ICGN2D1(8, 8, 0.001, 10)
    rng = np.random.default_rng(12)
    n, width = 5000, 32
    q = np.full((n, width), 7.5, np.float32)       # floats 25..31: the caller's own payload
    q[:, :25] = 0
    q[:, 0] = np.arange(n)
    q[:, P["zncc"]] = rng.uniform(0.5, 1.0, n)
    q[:, P["convergence"]] = rng.uniform(0, 0.002, n)
    q[:, 20] = 5.0                                  # POI3D's convergence slot: must play no role
    q[::53, P["zncc"]] = np.nan
    rel_w, unr_w, idx_w = _host_split(q[:, :25], 0.7, 0.9, 0.001, P)
    for resident in (False, True):
        queue = torch.from_numpy(q).to("cuda:0") if resident else q.copy()
        if resident:
            rel0 = torch.full((n + 3, width), -9.0, dtype=torch.float32, device="cuda:0")
        else:
            rel0 = np.full((n + 3, width), -9.0, np.float32)
        rel, n_rel, unr, idx, n_unr = g.split_reliable(queue, 0.7, 0.9, 0.001, reliable=rel0, reliable_offset=3)
        relh, unrh, idxh = (t.cpu().numpy() if resident else t for t in (rel, unr, idx))
        assert n_rel == len(rel_w) and n_unr == len(unr_w)
        assert np.array_equal(_bits(relh[3:3 + n_rel, :25]), _bits(rel_w)) and np.array_equal(_bits(unrh[:n_unr, :25]), _bits(unr_w))
        assert np.array_equal(np.asarray(idxh[:n_unr], dtype=np.uint32), idx_w)
        assert (relh[:3] == -9.0).all() and (relh[3:3 + n_rel, 25:] == -9.0).all()   # padding and the records before the offset
        # merge: make every second unreliable POI pass, mark their payload; the main queue's padding must survive
        unr_m = unrh.copy()
        unr_m[:n_unr:2, P["zncc"]] = 0.95
        unr_m[:n_unr:2, P["convergence"]] = 0.0005
        unr_m[:, 25:] = 3.25
        unr_in = torch.from_numpy(unr_m).to("cuda:0") if resident else unr_m.copy()
        main_before = queue.cpu().numpy().copy() if resident else queue.copy()
        n_rec, n_rem = g.merge_recovered(queue, unr_in, idx, n_unr, 0.9, 0.001, rel, 3 + n_rel)
        main = queue.cpu().numpy() if resident else queue
        relh = rel.cpu().numpy() if resident else rel
        ok = np.zeros(n_unr, bool)
        ok[::2] = True
        assert (n_rec, n_rem) == (int(ok.sum()), int((~ok).sum()))
        assert np.array_equal(_bits(main[idx_w[ok], :25]), _bits(unr_m[:n_unr][ok][:, :25]))
        assert np.array_equal(_bits(main[:, 25:]), _bits(main_before[:, 25:]))       # only the record's own bytes are written back
        assert np.array_equal(_bits(relh[3 + n_rel:3 + n_rel + n_rec, :25]), _bits(unr_m[:n_unr][ok][:, :25]))
        assert (relh[3 + n_rel:3 + n_rel + n_rec, 25:] == -9.0).all()
        # an index past the main queue is refused and writes nothing
        bad_idx = idxh.copy()
        bad_idx[0] = n + 10
        bad = torch.from_numpy(bad_idx.astype(np.int32)).to("cuda:0") if resident else bad_idx.astype(np.uint32)
        snap = queue.cpu().numpy().copy() if resident else queue.copy()
        unr_again = torch.from_numpy(unr_m).to("cuda:0") if resident else unr_m.copy()
        with pytest.raises(RuntimeError, match="unreliable_index"):
            g.merge_recovered(queue[:n], unr_again, bad, n_unr, 0.9, 0.001, rel, 0)
        after = queue.cpu().numpy() if resident else queue
        assert np.array_equal(_bits(after[idx_w[0]]), _bits(snap[idx_w[0]]))
    # argument checks: no kernel may see these
    dq = torch.from_numpy(q).to("cuda:0")
    with pytest.raises(ValueError):
        g.split_reliable(torch.from_numpy(q), 0.7, 0.9, 0.001)                      # CPU torch tensor
    with pytest.raises(ValueError):
        g.split_reliable(q.astype(np.float64), 0.7, 0.9, 0.001)
    with pytest.raises(ValueError):
        g.split_reliable(q[:, :20].copy(), 0.7, 0.9, 0.001)                         # shorter than a POI2D record
    with pytest.raises(ValueError):
        g.split_reliable(dq, 0.7, 0.9, 0.001, reliable=np.zeros((n, width), np.float32))   # mixed memory kinds
    with pytest.raises(ValueError):
        g.merge_recovered(dq, torch.from_numpy(unr_m), idx, 1, 0.9, 0.001, rel, 0)   # CPU tensor among device queues
    with pytest.raises(ValueError):
        g.merge_recovered(q, unr_m, idx_w.astype(np.int64), 1, 0.9, 0.001, rel_w, 0)
    with pytest.raises(ValueError):
        g.merge_recovered(q, unr_m[:, :25].copy(), idx_w, 1, 0.9, 0.001, np.zeros((n, width), np.float32), 0)  # stride mismatch
