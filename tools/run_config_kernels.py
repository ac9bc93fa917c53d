#!/usr/bin/env python
"""Runs the engines of one BASELINE config a few times and nothing else (no oracle, no checks): the command rocprofv3 wraps when
tools/gpu_profiles.sh collects the kernel trace and the PMC traffic of configs C and E (config B is bench.py itself).

    python tools/run_config_kernels.py C|E|E30 [--reps 3]          (OC_BENCH_ARITH_FMA=1: oc_hip_set_tuning arith_fma = 1)
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opencorr_amd as oc
from opencorr_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("config")
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--warm", type=int, default=1, help="untimed passes before the `reps` the profile averages")
a = ap.parse_args()
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
if a.config == "C":
    r = 20
    ref, tar = synth.speckle_pair_2d(4096, 4096, seed=20260925, device=dev, second_order=dict(uxx=2e-6, vyy=-1e-6))
    xs, ys = synth.poi_grid_2d(4096, 4096, 316, 316, r + 8)
    f, g = oc.FFTCC2D(r, r), oc.ICGN2D2(r, r, 0.001, 10.0)
    pristine = torch.from_numpy(oc.make_pois2d(xs, ys)).to(dev)
else:
    dim, r, nside = (512, 16, 37) if a.config == "E" else (256, 30, 8)
    ref, tar = synth.speckle_pair_3d(dim, dim, dim, seed=20260927, device=dev)
    xs, ys, zs = synth.poi_grid_3d(dim, dim, dim, nside, nside, nside, r + 8)
    f, g = oc.FFTCC3D(r, r, r), oc.ICGN3D1(r, r, r, 0.001, 20.0)
    pristine = torch.from_numpy(oc.make_pois3d(xs, ys, zs)).to(dev)
f.set_stream(stream)
f.set_images(ref, tar)
if os.environ.get("OC_FFTCC3D_TILE_VOX") and a.config != "C":   # A/B of the FFTCC3D visiting order (0 = queue order)
    f.set_tuning("fftcc3d_tile_vox", int(os.environ["OC_FFTCC3D_TILE_VOX"]))
g.set_stream(stream)
g.share_images(f)
if os.environ.get("OC_BENCH_ARITH_FMA") == "1":   # the fused arithmetic contract (profiles/*_fma.*)
    g.set_tuning("arith_fma", 1)
g.prepare()
q = pristine.clone()
for _ in range(a.reps + a.warm):   # warm-up passes first (tools/pmc_traffic.py averages the last `reps` launches)
    q.copy_(pristine)
    f.compute(q)
    g.compute(q)
torch.cuda.synchronize()
res = q.cpu().numpy()
zc, ic = (16, 17) if res.shape[1] == 25 else (18, 19)
print("config %s: %d POIs, %d converged, mean iterations %.3f" % (a.config, len(res), int((res[:, zc] >= 0).sum()), float(res[res[:, ic] > 0, ic].mean())))
