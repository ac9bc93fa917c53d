#!/usr/bin/env python
"""Full-size checks (part of the test infrastructure: lives under tests/ because it uses the oracle as its checker).

Runs the BASELINE.json configurations that are not the bench line (A, C, D on one GPU, E) at
their full sizes on one MI355X and checks them through size-independent properties plus a strided
oracle sample.  One JSON line per configuration.

    python tests/fullsize/run_configs.py [--configs A,C,D1,E] [--out gpurun_out/configs.json]

Checks per configuration
  * converged fraction and recovery of the analytic displacement field the synthetic pair was
    rendered with (median / max error over converged POIs),
  * GPU == oracle on a strided sample of the queue, stage by stage (the oracle is the checker; test/bench
    infrastructure only): FFTCC -- integer displacement and guess identical, ZNCC of the peak within 1e-5 (2D) / 1e-4
    (32^3 windows); ICGN on the GPU's FFTCC output -- every bit of every record,
  * idempotence of the sharding: the first and second half of the queue computed separately
    give the same bits as the whole queue,
  * the fused arithmetic contract (oc_hip_set_tuning "arith_fma", round 5): the same FFTCC output refined a second time by
    the kernels whose per-sample multiply-adds are fused -- every bit equal to the oracle in OC_ORDER_LANES_FMA, and against
    the REFERENCE's loop order (OC_ORDER_SEQ, separately rounded) the same bars as the default build; the record's "fma"
    entry also carries the solver's time in both modes (the same queue, interleaved).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


# threads of the workgroup that owns a POI in the 3D kernels (the reduction order itself is oracle.GPU_ORDER_3D)
LANES3D = 512


def timed(fn, sync, reps=3):
    fn()
    sync()
    best = 1e30
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        sync()
        best = min(best, time.perf_counter() - t0)
    return best


def vs_reference_order(gpu, seq, cols_disp, col_zncc, col_iter):
    """north_star's literal statement, GPU against the oracle in the REFERENCE's own loop order (OC_ORDER_SEQ, bit-identical
    to the reference's compiled sources: tests/test_oracle_vs_ref.py) on the same FFTCC output: "u/v/w and ZNCC within 1e-4,
    iteration counts and convergence flags bit-identical".  The GPU sums a subset in lane order (OC_ORDER_LANES), a
    re-association of the reference's sequential sums; this is what that re-association costs.
    Returns the number of POIs whose failure flag (zncc < 0) differs, the fraction of commonly converged POIs with equal
    iteration counts, and -- over the POIs with equal iteration counts -- max |d displacement| and max |d ZNCC|."""
    fa, fb = gpu[:, col_zncc] < 0, seq[:, col_zncc] < 0
    both = ~fa & ~fb
    same_it = both & (gpu[:, col_iter] == seq[:, col_iter])
    dd = np.abs(gpu[same_it][:, cols_disp].astype(np.float64) - seq[same_it][:, cols_disp].astype(np.float64))
    dz = np.abs(gpu[same_it, col_zncc].astype(np.float64) - seq[same_it, col_zncc].astype(np.float64))
    # failed POIs keep their codes: -3 / -4 / -5 must be the SAME code on both sides
    codes_same = bool(np.array_equal(gpu[fa & fb, col_zncc], seq[fa & fb, col_zncc]))
    over = int((dd.max(axis=1) > 1e-4).sum()) if dd.size else 0
    return dict(seq_flag_mismatches=int((fa != fb).sum()) + (0 if codes_same else 1),
                seq_iteration_agreement=float(same_it.sum() / max(1, both.sum())),
                seq_max_abs_d_disp=float(dd.max()) if dd.size else 0.0,
                seq_pois_over_1e4=over, seq_frac_within_1e4=float(1.0 - over / max(1, int(same_it.sum()))),
                seq_max_abs_d_zncc=float(dz.max()) if dz.size else 0.0, seq_sample=int(len(gpu)))


def run_2d(name, side, r, nside, engine, oracle_sample, so=None, with_fma=True):
    import torch
    import opencorr_amd as oc
    import oracle
    from opencorr_amd import synth
    dev = torch.device("cuda", 0)
    ref, tar = synth.speckle_pair_2d(side, side, seed=20260925, device=dev, second_order=so)
    xs, ys = synth.poi_grid_2d(side, side, nside, nside, r + 8)
    n = len(xs)
    stream = torch.cuda.current_stream().cuda_stream
    f = oc.FFTCC2D(r, r)
    f.set_stream(stream)
    f.set_images(ref, tar)
    Icgn = {1: oc.ICGN2D1, 2: oc.ICGN2D2, 3: oc.NR2D1}[engine]
    g = Icgn(r, r, 0.001, 10.0)
    g.set_stream(stream)
    g.share_images(f)
    t0 = time.perf_counter()
    g.prepare()
    torch.cuda.synchronize()
    prepare_s = time.perf_counter() - t0
    pristine = torch.from_numpy(oc.make_pois2d(xs, ys)).to(dev)
    pois = pristine.clone()

    def step():
        pois.copy_(pristine)
        f.compute(pois)
        g.compute(pois)

    secs = timed(step, torch.cuda.synchronize)
    after = pois.cpu().numpy()
    # sharding idempotence
    halves = pristine.clone()
    h = n // 2
    f.compute(halves[:h]); g.compute(halves[:h]); f.compute(halves[h:]); g.compute(halves[h:])
    torch.cuda.synchronize()
    same_split = bool(np.array_equal(halves.cpu().numpy().view(np.uint32), after.view(np.uint32)))
    # analytic field
    eu, ev = synth.expected_deformation_2d(xs, ys, side, side, second_order=so)
    conv = after[:, 16] >= 0
    du, dv = np.abs(after[conv, 2] - eu[conv]), np.abs(after[conv, 8] - ev[conv])
    # oracle sample, stage by stage: FFTCC by the oracle on the pristine records vs the GPU's FFTCC output (integer
    # displacement and guess identical; the ZNCC of the peak within 1e-5: the surface passes through a float FFT on the
    # GPU and a double DFT in the oracle), THEN the oracle refines the GPU's FFTCC output and must reproduce every bit
    step_s = max(1, n // oracle_sample)
    t_oracle = time.perf_counter()
    fin = pristine.clone()
    f.compute(fin)
    torch.cuda.synchronize()
    sample = fin.cpu().numpy()[::step_s].copy()
    ref_h, tar_h = ref.cpu().numpy(), tar.cpu().numpy()
    fo = pristine.cpu().numpy()[::step_s].copy()
    oracle.fftcc2d(ref_h, tar_h, r, r, fo)
    fftcc_same = bool(np.array_equal(fo[:, [2, 8, 14, 15]], sample[:, [2, 8, 14, 15]]))
    fftcc_zncc = float(np.abs(fo[:, 16] - sample[:, 16]).max())
    untouched = np.delete(np.arange(25), [2, 8, 14, 15, 16])
    fftcc_same = fftcc_same and bool(np.array_equal(fo[:, untouched].view(np.uint32), sample[:, untouched].view(np.uint32)))
    seq = sample.copy()  # the same FFTCC output, refined in the reference's own loop order
    guess_sample = sample.copy()
    if engine == 3:
        prep = oracle.PreparedNR2D(ref_h, tar_h)
        oracle.nr2d1(prep, r, r, 0.001, 10.0, sample, order=oracle.ORDER_LANES, lanes=64)
        oracle.nr2d1(prep, r, r, 0.001, 10.0, seq, order=oracle.ORDER_SEQ)
    else:
        prep = oracle.Prepared2D(ref_h, tar_h)
        solve = oracle.icgn2d1 if engine == 1 else oracle.icgn2d2
        solve(prep, r, r, 0.001, 10.0, sample, order=oracle.ORDER_LANES, lanes=64)
        solve(prep, r, r, 0.001, 10.0, seq, order=oracle.ORDER_SEQ)
    bit_exact = bool(np.array_equal(sample.view(np.uint32), after[::step_s].view(np.uint32)))
    vs_seq = vs_reference_order(after[::step_s], seq, [2, 8], 16, 17)
    oracle_s = time.perf_counter() - t_oracle
    fma = None
    if with_fma and engine != 3:
        # the fused arithmetic contract on the same FFTCC output: GPU(arith_fma) == oracle(LANES_FMA) bit for bit, and
        # GPU(arith_fma) against the reference's separately rounded loop order; solver time in both modes
        def solve_only():
            pois.copy_(fin)
            g.compute(pois)
        g.set_tuning("arith_fma", 1)
        t_fma = timed(solve_only, torch.cuda.synchronize)
        after_fma = pois.cpu().numpy()
        g.set_tuning("arith_fma", 0)
        t_sep = timed(solve_only, torch.cuda.synchronize)
        g.set_tuning("arith_fma", 1)
        t_fma = min(t_fma, timed(solve_only, torch.cuda.synchronize))
        g.set_tuning("arith_fma", 0)
        want_fma = guess_sample.copy()
        solve(prep, r, r, 0.001, 10.0, want_fma, order=oracle.ORDER_LANES_FMA, lanes=64)
        conv_f = after_fma[:, 16] >= 0
        fma = dict(icgn_seconds_fma=t_fma, icgn_seconds_sep=t_sep, converged=int(conv_f.sum()),
                   mean_iterations=float(after_fma[conv_f, 17].astype(np.float64).mean()),
                   oracle_bit_exact=bool(np.array_equal(want_fma.view(np.uint32), after_fma[::step_s].view(np.uint32))),
                   max_abs_d_disp_vs_default_build=float(np.abs(after_fma[conv_f & conv][:, [2, 8]].astype(np.float64)
                                                                 - after[conv_f & conv][:, [2, 8]].astype(np.float64)).max()),
                   **vs_reference_order(after_fma[::step_s], seq, [2, 8], 16, 17))
    del prep
    return dict(config=name, fma=fma, engine="FFTCC2D+" + {1: "ICGN2D1", 2: "ICGN2D2", 3: "NR2D1"}[engine], image="%dx%d" % (side, side), radius=r, pois=n,
                seconds=secs, pois_per_s=float(conv.sum() / secs), converged=int(conv.sum()),
                mean_iterations=float(after[conv, 17].astype(np.float64).mean()), prepare_s=prepare_s,
                median_abs_err_u=float(np.median(du)), max_abs_err_u=float(du.max()), max_abs_err_v=float(dv.max()),
                oracle_sample=len(sample), oracle_stride=step_s, oracle_seconds_fftcc_lanes_seq=oracle_s, oracle_cores=oracle.max_threads(),
                oracle_bit_exact=bit_exact, split_queue_same_bits=same_split,
                fftcc_oracle_same_integers=fftcc_same, fftcc_oracle_max_zncc_diff=fftcc_zncc, **vs_seq)


def run_strain(name, side, r, nside, radius, nmin):
    """FFTCC2D -> ICGN2D1 -> Strain on the device-resident queue."""
    import torch
    import opencorr_amd as oc
    import oracle
    from opencorr_amd import synth
    dev = torch.device("cuda", 0)
    ref, tar = synth.speckle_pair_2d(side, side, seed=20260925, device=dev)
    xs, ys = synth.poi_grid_2d(side, side, nside, nside, r + 8)
    stream = torch.cuda.current_stream().cuda_stream
    f = oc.FFTCC2D(r, r)
    f.set_stream(stream)
    f.set_images(ref, tar)
    g = oc.ICGN2D1(r, r, 0.001, 10.0)
    g.set_stream(stream)
    g.share_images(f)
    g.prepare()
    pois = torch.from_numpy(oc.make_pois2d(xs, ys)).to(dev)
    f.compute(pois)
    g.compute(pois)
    st = oc.Strain(radius, nmin)
    st.set_stream(stream)
    t_prep = timed(lambda: st.prepare(pois), torch.cuda.synchronize)
    t_comp = timed(lambda: st.compute(pois), torch.cuda.synchronize)
    got = pois.cpu().numpy()
    want = got.copy()
    want[:, 20:23] = 0
    t0 = time.perf_counter()
    oracle.strain2d(want, radius, nmin)
    oracle_s = time.perf_counter() - t0
    fitted = got[:, 16] >= 0.9
    exx = got[fitted, 20]
    return dict(config=name, engine="FFTCC2D+ICGN2D1 -> Strain", pois=len(xs), subregion_radius=radius, neighbor_min=nmin,
                neighbours_per_poi=float(np.pi * radius * radius / ((side - 2 * (r + 8)) / nside) ** 2),
                prepare_seconds=t_prep, compute_seconds=t_comp, pois_per_s=float(fitted.sum() / t_comp),
                exx_median=float(np.median(exx)), exx_expected=1e-3, oracle_seconds=oracle_s, oracle_cores=oracle.max_threads(),
                oracle_bit_exact=bool(np.array_equal(got.view(np.uint32), want.view(np.uint32))))


def run_3d(name, dim, r, nside, oracle_sample, with_fma=True):
    import torch
    import opencorr_amd as oc
    import oracle
    from opencorr_amd import synth
    dev = torch.device("cuda", 0)
    t0 = time.perf_counter()
    ref, tar = synth.speckle_pair_3d(dim, dim, dim, seed=20260927, device=dev)
    gen_s = time.perf_counter() - t0
    xs, ys, zs = synth.poi_grid_3d(dim, dim, dim, nside, nside, nside, r + 8)
    n = len(xs)
    stream = torch.cuda.current_stream().cuda_stream
    f = oc.FFTCC3D(r, r, r)
    f.set_stream(stream)
    f.set_images(ref, tar)
    g = oc.ICGN3D1(r, r, r, 0.001, 20.0)
    g.set_stream(stream)
    g.share_images(f)
    t0 = time.perf_counter()
    g.prepare()
    torch.cuda.synchronize()
    prepare_s = time.perf_counter() - t0
    pristine = torch.from_numpy(oc.make_pois3d(xs, ys, zs)).to(dev)
    pois = pristine.clone()
    t_f = timed(lambda: (pois.copy_(pristine), f.compute(pois)), torch.cuda.synchronize, reps=3)
    guess = pois.clone()
    t_g = timed(lambda: (pois.copy_(guess), g.compute(pois)), torch.cuda.synchronize, reps=3)
    after = pois.cpu().numpy()
    P = oracle.P3
    conv = after[:, P["zncc"]] >= 0
    w = synth.DEFAULT_WARP_3D
    xp, yp, zp = xs - (dim - 1) * 0.5, ys - (dim - 1) * 0.5, zs - (dim - 1) * 0.5
    eu = w["u"] + w["ux"] * xp + w["uy"] * yp + w["uz"] * zp
    ev = w["v"] + w["vx"] * xp + w["vy"] * yp + w["vz"] * zp
    ew = w["w"] + w["wx"] * xp + w["wy"] * yp + w["wz"] * zp
    err = np.maximum.reduce([np.abs(after[conv, P["u"]] - eu[conv]), np.abs(after[conv, P["v"]] - ev[conv]),
                             np.abs(after[conv, P["w"]] - ew[conv])])
    step_s = max(1, n // oracle_sample)
    sample = guess.cpu().numpy()[::step_s].copy()
    ref_h, tar_h = ref.cpu().numpy(), tar.cpu().numpy()
    # FFTCC3D by the oracle on the pristine sample vs the GPU's guess: integer u, v, w and u0, v0, w0 identical, ZNCC within
    # north_star's 1e-4 (at 32^3 voxels the oracle's sequential float sums of means and norms carry ~8e-5)
    fo = pristine.cpu().numpy()[::step_s].copy()
    oracle.fftcc3d(ref_h, tar_h, r, r, r, fo)
    ints = [P["u"], P["v"], P["w"], P["u0"], P["v0"], P["w0"]]
    fftcc_same = bool(np.array_equal(fo[:, ints], sample[:, ints]))
    fftcc_zncc = float(np.abs(fo[:, P["zncc"]] - sample[:, P["zncc"]]).max())
    # ... and against the oracle with exactly summed means and norms (double accumulators): the yardstick for north_star's
    # 1e-4 at every window size; `fo` keeps the reference's sequential float running sums, whose own rounding is 1e-4 at
    # 32^3 and 3e-4 at 60^3
    fe = pristine.cpu().numpy()[::step_s].copy()
    oracle.fftcc3d(ref_h, tar_h, r, r, r, fe, exact_sums=True)
    fftcc_exact_same = bool(np.array_equal(fe[:, ints], sample[:, ints]))
    fftcc_exact_zncc = float(np.abs(fe[:, P["zncc"]] - sample[:, P["zncc"]]).max())
    prep = oracle.Prepared3D(ref_h, tar_h)
    t0 = time.perf_counter()
    seq = sample.copy()
    oracle.icgn3d1(prep, r, r, r, 0.001, 20.0, sample, order=oracle.GPU_ORDER_3D, lanes=LANES3D)
    oracle_s = time.perf_counter() - t0
    oracle.icgn3d1(prep, r, r, r, 0.001, 20.0, seq, order=oracle.ORDER_SEQ)
    bit_exact = bool(np.array_equal(sample.view(np.uint32), after[::step_s].view(np.uint32)))
    vs_seq = vs_reference_order(after[::step_s], seq, [P["u"], P["v"], P["w"]], P["zncc"], P["iteration"])
    fma = None
    if with_fma:
        def solve_only():
            pois.copy_(guess)
            g.compute(pois)
        g.set_tuning("arith_fma", 1)
        t_fma = timed(solve_only, torch.cuda.synchronize, reps=2)
        after_fma = pois.cpu().numpy()
        g.set_tuning("arith_fma", 0)
        want_fma = guess.cpu().numpy()[::step_s].copy()
        oracle.icgn3d1(prep, r, r, r, 0.001, 20.0, want_fma, order=oracle.ORDER_LANES_FMA, lanes=LANES3D)
        conv_f = after_fma[:, P["zncc"]] >= 0
        cols = [P["u"], P["v"], P["w"]]
        fma = dict(icgn_seconds_fma=t_fma, icgn_seconds_sep=t_g, converged=int(conv_f.sum()),
                   mean_iterations=float(after_fma[conv_f, P["iteration"]].astype(np.float64).mean()),
                   oracle_bit_exact=bool(np.array_equal(want_fma.view(np.uint32), after_fma[::step_s].view(np.uint32))),
                   max_abs_d_disp_vs_default_build=float(np.abs(after_fma[conv_f & conv][:, cols].astype(np.float64)
                                                                 - after[conv_f & conv][:, cols].astype(np.float64)).max()),
                   **vs_reference_order(after_fma[::step_s], seq, cols, P["zncc"], P["iteration"]))
    return dict(config=name, fma=fma, engine="FFTCC3D+ICGN3D1", volume="%d^3" % dim, radius=r, pois=n, fftcc_seconds=t_f,
                icgn_seconds=t_g, pois_per_s=float(conv.sum() / (t_f + t_g)), converged=int(conv.sum()),
                mean_iterations=float(after[conv, P["iteration"]].astype(np.float64).mean()), prepare_s=prepare_s, generate_s=gen_s,
                median_abs_err=float(np.median(err)), max_abs_err=float(err.max()), oracle_sample=len(sample),
                oracle_seconds=oracle_s, oracle_pois_per_s=len(sample) / oracle_s, oracle_cores=oracle.max_threads(),
                oracle_bit_exact=bit_exact, fftcc_oracle_same_integers=fftcc_same, fftcc_oracle_max_zncc_diff=fftcc_zncc,
                fftcc_exact_sums_same_integers=fftcc_exact_same, fftcc_exact_sums_max_zncc_diff=fftcc_exact_zncc, **vs_seq)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="A,C,D1,E")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    out = []
    for c in a.configs.split(","):
        if c == "A":
            rec = run_2d("A (2048^2, r=15, 100x100 POIs)", 2048, 15, 100, 1, 10000)
        elif c == "B":
            rec = run_2d("B, the bench line's workload (4096^2, r=16, 500x500 POIs)", 4096, 16, 500, 1, 250000)
        elif c == "C":
            rec = run_2d("C (4096^2, r=20, ICGN2D2, 316x316 POIs)", 4096, 20, 316, 2, 99856, so=dict(uxx=2e-6, vyy=-1e-6))
        elif c == "D1":
            rec = run_2d("D on ONE GPU (8192^2, r=16, 1414x1414 POIs)", 8192, 16, 1414, 1, 200000)
        elif c == "BNR":
            rec = run_2d("B with NR2D1 (4096^2, r=16, 500x500 POIs)", 4096, 16, 500, 3, 4000)
        elif c == "BST":
            rec = run_strain("B + Strain (4096^2, 500x500 POIs, subregion radius 40 px, >= 5 neighbours)", 4096, 16, 500, 40.0, 5)
        elif c == "E":
            rec = run_3d("E on ONE GPU (512^3, r=16, 37^3 POIs)", 512, 16, 37, 2000)
        elif c == "E30":
            # the radius of the reference's own DVC example (examples/test_dvc_fftcc_icgn1.cpp:45-47)
            rec = run_3d("DVC example shape (256^3, r=30, 8^3 POIs)", 256, 30, 8, 256)
        elif c == "Es":
            rec = run_3d("E-small (256^3, r=16, 12^3 POIs)", 256, 16, 12, 48)
        else:
            continue
        print(json.dumps(rec), flush=True)
        out.append(rec)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
