#!/usr/bin/env python
"""ICGN2D1 compute() on config B against the tile size of the visiting order (oc_hip_set_tuning "icgn2d_tile_px"):
32 px 3.60 ms, 48 3.50, 64 (default) 3.42, 96 3.40, 128 3.39, 192 3.45 -- flat between 64 and 128."""
import sys, json, numpy as np, torch
sys.path.insert(0, ".")
import opencorr_amd as oc
from opencorr_amd import synth
dev = torch.device("cuda", 0)
side, r, ns = 4096, 16, 500
ref, tar = synth.speckle_pair_2d(side, side, seed=20260925, device=dev)
xs, ys = synth.poi_grid_2d(side, side, ns, ns, r + 8)
stream = torch.cuda.current_stream().cuda_stream
f = oc.FFTCC2D(r, r); f.set_stream(stream); f.set_images(ref, tar)
g = oc.ICGN2D1(r, r, 0.001, 10.0); g.set_stream(stream); g.share_images(f); g.prepare()
guess = torch.from_numpy(oc.make_pois2d(xs, ys)).to(dev); f.compute(guess); q = guess.clone()
out = {}
for rd in range(3):
    for t in (48, 64, 80, 96, 128, 192, 256):
        g.set_tuning("icgn2d_tile_px", t)
        tot = 0.0
        for _ in range(10):
            q.copy_(guess); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); g.compute(q); b.record(); b.synchronize(); tot += a.elapsed_time(b)
        if rd: out.setdefault(t, []).append(round(tot / 10, 4))
print(json.dumps(out))
