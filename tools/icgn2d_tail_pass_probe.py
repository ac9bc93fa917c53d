#!/usr/bin/env python
"""What the lone 1 089th sample of a 33 x 33 subset costs: ICGN2D1 on config B's grid with subsets of 33 x 33 (17 full passes of 64
samples + ONE sample in an 18th), 33 x 31 (15 full passes + 63 samples) and 31 x 33.  Time per launch, per sample and per pass.
python tools/icgn2d_tail_pass_probe.py   (GPU box)"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opencorr_amd as oc
from opencorr_amd import synth

dev = torch.device("cuda", 0)
ref, tar = synth.speckle_pair_2d(4096, 4096, seed=20260925, device=dev)
xs, ys = synth.poi_grid_2d(4096, 4096, 500, 500, 24)
out = []
for rx, ry in ((16, 16), (16, 15), (15, 16), (15, 15)):
    f = oc.FFTCC2D(16, 16); f.set_images(ref, tar)
    g = oc.ICGN2D1(rx, ry, 0.001, 10.0); g.share_images(f); g.prepare()
    p = torch.from_numpy(oc.make_pois2d(xs, ys)).to(dev)
    f.compute(p); q = p.clone()
    for _ in range(3):
        q.copy_(p); g.compute(q)
    torch.cuda.synchronize(); g.profile_enable(True)
    for _ in range(10):
        q.copy_(p); g.compute(q)
    torch.cuda.synchronize(); ms, n = g.profile_read()
    res = q.cpu().numpy()
    it = float(res[res[:, 17] > 0, 17].mean())
    N = (2 * rx + 1) * (2 * ry + 1)
    passes = (N + 63) // 64
    out.append({"subset": "%d x %d" % (2 * rx + 1, 2 * ry + 1), "samples": N, "passes": passes, "icgn2d1_ms": round(ms / n, 4), "mean_iterations": round(it, 4),
                "ps_per_sample_iteration": round(ms / n * 1e9 / (250000 * N * it), 3), "ns_per_pass_iteration_poi": round(ms / n * 1e6 / (250000 * passes * it), 4)})
print(json.dumps(out))
