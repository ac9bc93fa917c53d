#!/usr/bin/env python
"""Host-queue step (FFTCC2D + ICGN2D1 on a pageable host vector, config B) against the pipeline chunk size
(oc_hip_set_tuning "host_chunk"): 32768 -> 5.32 ms, 65536 (default) 5.43, 98304 5.44, 131072 6.21, one piece 6.42."""
import sys, time, json, numpy as np, torch
sys.path.insert(0, ".")
import opencorr_amd as oc
from opencorr_amd import synth
dev = torch.device("cuda", 0)
side, r, ns = 4096, 16, 500
ref, tar = synth.speckle_pair_2d(side, side, seed=20260925, device=dev)
xs, ys = synth.poi_grid_2d(side, side, ns, ns, r + 8)
f = oc.FFTCC2D(r, r); f.set_images(ref, tar)
g = oc.ICGN2D1(r, r, 0.001, 10.0); g.share_images(f); g.prepare()
host0 = oc.make_pois2d(xs, ys)
out = {}
for chunk in (32768, 65536, 98304, 131072, 250000):
    f.set_tuning("host_chunk", chunk); g.set_tuning("host_chunk", chunk)
    best = 1e9
    for _ in range(6):
        q = host0.copy()
        t0 = time.perf_counter(); f.compute(q); g.compute(q); best = min(best, time.perf_counter() - t0)
    out[chunk] = round(best * 1e3, 3)
print(json.dumps(out))
