#!/usr/bin/env python
"""How much of ICGN3D1's time on config E is the L2-miss path?  The same 50 653-POI queue, solved (a) on the regular 37^3 grid
and (b) with every POI drawn from a small set of K^3 grid positions in the volume's centre (the queue order keeps cycling through
them), so that the subvolume neighbourhoods of all POIs in flight stay in the L2s.  Same arithmetic per POI (iteration counts are
reported), same kernel, same launch shape -- the difference is memory.   python tools/icgn3d_locality_probe.py [K ...]   (GPU box)
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opencorr_amd as oc
from opencorr_amd import synth

ks = [int(k) for k in sys.argv[1:]] or [2, 4]
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
dim, r, nside = 512, 16, 37
ref, tar = synth.speckle_pair_3d(dim, dim, dim, seed=20260927, device=dev)
xs, ys, zs = synth.poi_grid_3d(dim, dim, dim, nside, nside, nside, r + 8)
f, g = oc.FFTCC3D(r, r, r), oc.ICGN3D1(r, r, r, 0.001, 20.0)
f.set_stream(stream); f.set_images(ref, tar)
g.set_stream(stream); g.share_images(f); g.prepare()


def run(px, py, pz, tag):
    pristine = torch.from_numpy(oc.make_pois3d(px, py, pz)).to(dev)
    f.compute(pristine)
    q = pristine.clone()
    for _ in range(2):
        q.copy_(pristine); g.compute(q)
    torch.cuda.synchronize()
    g.profile_reset(); g.profile_enable(True)
    for _ in range(5):
        q.copy_(pristine); g.compute(q)
    torch.cuda.synchronize()
    ms, n = g.profile_read()
    g.profile_enable(False)
    res = q.cpu().numpy()
    it = res[res[:, 19] > 0, 19]
    return {"queue": tag, "icgn3d1_ms": round(float(ms / n), 3), "pois": len(px), "converged": int((res[:, 18] >= 0).sum()),
            "mean_iterations": round(float(it.mean()), 4), "ms_per_1e6_iteration_pois": round(float(ms / n / (float(it.sum()) / 1e6)), 3)}


out = [run(xs, ys, zs, "config E: 37^3 grid")]
gx = np.unique(xs)
mid = len(gx) // 2
for k in ks:
    sel = gx[mid - k // 2: mid - k // 2 + k]
    cx, cy, cz = np.meshgrid(sel, sel, sel, indexing="ij")
    cx, cy, cz = cx.ravel(), cy.ravel(), cz.ravel()
    idx = np.arange(len(xs)) % len(cx)
    out.append(run(cx[idx].astype(xs.dtype), cy[idx].astype(ys.dtype), cz[idx].astype(zs.dtype), "the same count drawn from %d^3 central grid positions" % k))
print(json.dumps(out))
