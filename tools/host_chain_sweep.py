#!/usr/bin/env python
"""Host-queue step through oc_hip_compute_chain (FFTCC2D + ICGN2D1, config B) against the pipeline chunk size, with a
pageable and with a pinned host queue, next to the device-resident chain."""
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
pinned = torch.empty(host0.shape, dtype=torch.float32).pin_memory()
out = {}
d = torch.from_numpy(host0).to(dev); d0 = d.clone()
best = 1e9
for _ in range(8):
    d.copy_(d0); torch.cuda.synchronize()
    t0 = time.perf_counter(); oc.compute_chain([f, g], d); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
out["device_resident"] = round(best * 1e3, 3)
for chunk in [int(a) for a in (sys.argv[1].split(",") if len(sys.argv) > 1 else "16384,32768,65536,98304,131072,0".split(","))]:
    f.set_tuning("host_chunk", chunk); g.set_tuning("host_chunk", chunk)
    for name in ("pageable", "pinned"):
        best = 1e9
        for _ in range(8):
            if name == "pageable":
                q = host0.copy()
            else:
                pinned.copy_(torch.from_numpy(host0)); q = pinned.numpy()
            t0 = time.perf_counter(); oc.compute_chain([f, g], q); best = min(best, time.perf_counter() - t0)
        out["%s_%d" % (name, chunk)] = round(best * 1e3, 3)
print(json.dumps(out))
