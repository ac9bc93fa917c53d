#!/usr/bin/env python
"""Phase timeline of icgn3d1_kernel (OC_ABLATE bit 16: results stay valid, the six strain floats of every POI record receive the
kilocycles its workgroup spent per phase).  Run under a library built with tools/ab_build.py icgn3d[:fma] "tl:-DOC_ABLATE=16":
    OPENCORR_HIP_LIB=tools/ab_build/libab_icgn3d_tl.so python tools/icgn3d_timeline.py            (ARITH_FMA=1 for the fused build)"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import opencorr_amd as oc
from opencorr_amd import synth

dev = torch.device("cuda", 0)
dim, r, ns = int(os.environ.get("DIM", 256)), 16, int(os.environ.get("NS", 20))
ref, tar = synth.speckle_pair_3d(dim, dim, dim, seed=20260927, device=dev)
xs, ys, zs = synth.poi_grid_3d(dim, dim, dim, ns, ns, ns, r + 8)
g = oc.ICGN3D1(r, r, r, 0.001, 20.0)
g.set_images(ref, tar)
g.prepare()
if os.environ.get("ARITH_FMA") == "1":
    g.set_tuning("arith_fma", 1)
pr = oc.make_pois3d(xs, ys, zs)
w = synth.DEFAULT_WARP_3D
pr[:, 3], pr[:, 7], pr[:, 11] = round(w["u"]), round(w["v"]), round(w["w"])
pristine = torch.from_numpy(pr).to(dev)
q = pristine.clone()
best = 1e9
for _ in range(4):
    q.copy_(pristine)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.compute(q)
    torch.cuda.synchronize()
    best = min(best, time.perf_counter() - t0)
res = q.cpu().numpy()
tl = res[:, 22:28].astype(np.float64).mean(0)
names = ["reference stats", "Hessian sweep + reduction", "LU inverse", "warped-subvolume sweeps (boxes, staging, taps)",
         "mean / norm / numerator sweeps + reductions", "solve + warp update"]
print(json.dumps(dict(ms=round(best * 1e3, 3), pois=len(xs), mean_iter=float(res[:, 19].mean()), arith_fma=os.environ.get("ARITH_FMA") == "1",
                      kcycles_per_poi={n: round(float(v), 1) for n, v in zip(names, tl)},
                      share={n: round(float(v / tl.sum()), 3) for n, v in zip(names, tl)})))
