#!/usr/bin/env python
"""hipEvent-free timing of ICGN2D1::prepare() pieces on a 4096^2 pair: prepare_ref (gradients) and prepare_tar (table), best of 20."""
import sys, time, json, torch
sys.path.insert(0, ".")
import opencorr_amd as oc
from opencorr_amd import synth
dev = torch.device("cuda", 0)
side = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ref, tar = synth.speckle_pair_2d(side, side, seed=20260925, device=dev)
g = oc.ICGN2D1(16, 16, 0.001, 10.0); g.set_images(ref, tar); g.prepare(); torch.cuda.synchronize()
out = {}
for name, fn in (("prepare_ref_us", g.prepare_ref), ("prepare_tar_us", g.prepare_tar), ("prepare_us", g.prepare)):
    best = 1e9
    for _ in range(20):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    out[name] = round(best * 1e6, 1)
print(json.dumps(out))
