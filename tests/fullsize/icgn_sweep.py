#!/usr/bin/env python
"""Times every ICGN2D kernel variant (x XCD mapping on/off) on one workload and checks that
all of them produce the same bits (against variant 0 on the GPU, and against the CPU oracle
on a strided sample of the queue).

    python tests/fullsize/icgn_sweep.py [--size 4096 --pois 500 --radius 16 --engine 1 --launches 3]
                               [--variants 0,3,4] [--out gpurun_out/sweep.json]

Test infrastructure (lives under tests/ because the oracle import is its checker).
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--pois", type=int, default=500)
    ap.add_argument("--radius", type=int, default=16)
    ap.add_argument("--engine", type=int, default=1, help="1 = ICGN2D1, 2 = ICGN2D2")
    ap.add_argument("--launches", type=int, default=3)
    ap.add_argument("--variants", default="")
    ap.add_argument("--xcd", default="0,1")
    ap.add_argument("--oracle-sample", type=int, default=4000)
    ap.add_argument("--out", default="")
    ap.add_argument("--tile-px", default="", help="comma list of engine-side tile schedules to time (icgn2d_tile_px)")
    ap.add_argument("--tile", type=int, default=0, help="reorder the POI queue into T x T tiles of the grid (locality experiment)")
    args = ap.parse_args()

    import torch
    import opencorr_amd
    from opencorr_amd import synth

    dev = torch.device("cuda", 0)
    r = args.radius
    so = dict(uxx=2e-6, vyy=-1e-6) if args.engine == 2 else None
    ref, tar = synth.speckle_pair_2d(args.size, args.size, seed=20260925, device=dev, second_order=so)
    xs, ys = synth.poi_grid_2d(args.size, args.size, args.pois, args.pois, r + 8)
    if args.tile:
        gi, gj = np.divmod(np.arange(len(xs)), args.pois)  # grid row, column of each POI
        key = ((gi // args.tile) * (-(-args.pois // args.tile)) + gj // args.tile) * (args.tile * args.tile) + \
              (gi % args.tile) * args.tile + gj % args.tile
        order = np.argsort(key, kind="stable")
        xs, ys = xs[order], ys[order]
    stream = torch.cuda.current_stream().cuda_stream
    fftcc = opencorr_amd.FFTCC2D(r, r)
    fftcc.set_stream(stream)
    fftcc.set_images(ref, tar)
    Icgn = opencorr_amd.ICGN2D1 if args.engine == 1 else opencorr_amd.ICGN2D2
    icgn = Icgn(r, r, 0.001, 10.0)
    icgn.set_stream(stream)
    icgn.share_images(fftcc)
    icgn.prepare()
    start = torch.from_numpy(opencorr_amd.make_pois2d(xs, ys)).to(dev)
    fftcc.compute(start)
    torch.cuda.synchronize()
    pois = start.clone()

    # the variants the shipped library contains (0, 6 and 8 live in the A/B build: OPENCORR_HIP_LIB=.../lib/ab/libopencorr_hip_ab.so)
    variants = [int(v) for v in args.variants.split(",")] if args.variants else [1, 2, 3, 4, 5, 7]
    xcds = [int(v) for v in args.xcd.split(",")]
    tiles = [int(v) for v in args.tile_px.split(",")] if args.tile_px else [None]
    base = None
    rows = []
    for v, tpx in [(v_, t_) for v_ in variants for t_ in tiles]:
        for x in xcds:
            icgn.set_tuning("icgn2d_variant", v)
            icgn.set_tuning("icgn2d_xcd", x)
            if tpx is not None:
                icgn.set_tuning("icgn2d_tile_px", tpx)
            try:
                pois.copy_(start)
                icgn.compute(pois)  # warm-up (code object load, LDS attribute)
                torch.cuda.synchronize()
            except opencorr_amd.capi.OpenCorrHipError as e:
                rows.append(dict(variant=v, xcd=x, error=str(e)))
                print(rows[-1], flush=True)
                continue
            icgn.profile_reset()
            icgn.profile_enable(True)
            for _ in range(args.launches):
                pois.copy_(start)
                icgn.compute(pois)
            ms, n = icgn.profile_read()
            icgn.profile_enable(False)
            got = pois.cpu().numpy()
            if base is None:
                base = got
            same = bool(np.array_equal(got.view(np.uint32), base.view(np.uint32)))
            rows.append(dict(variant=v, xcd=x, tile_px=tpx, ms=ms / max(n, 1), launches=n, same_bits_as_first=same,
                             converged=int((got[:, 16] >= 0).sum()), mean_iter=float(got[:, 17].mean())))
            print(rows[-1], flush=True)

    # oracle check of the common result on a strided sample
    import oracle
    step = max(1, len(xs) // args.oracle_sample)
    sample = start.cpu().numpy()[::step].copy()
    prep = oracle.Prepared2D(ref.cpu().numpy(), tar.cpu().numpy())
    fn = oracle.icgn2d1 if args.engine == 1 else oracle.icgn2d2
    fn(prep, r, r, 0.001, 10.0, sample, order=oracle.ORDER_LANES, lanes=64)
    ok = bool(np.array_equal(sample.view(np.uint32), base[::step].view(np.uint32)))
    summary = dict(workload="%dx%d r=%d %dx%d POIs engine ICGN2D%d" % (args.size, args.size, r, args.pois, args.pois,
                                                                        args.engine),
                   oracle_sample=len(sample), oracle_bit_exact=ok, rows=rows)
    print(json.dumps(summary), flush=True)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(summary, f, indent=1)
    if not ok or not all(r_.get("same_bits_as_first", True) for r_ in rows):
        sys.exit(1)


if __name__ == "__main__":
    main()
