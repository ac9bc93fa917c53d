#!/usr/bin/env python
"""FFTCC3D, r = 16, config E's queue (512^3, 37^3 POIs): the fused 32^3 kernel (fftcc3d_fused.hip) against the kernel of rounds 1 - 5
(fftcc3d_fused_r5.hip, tuning "fftcc3d_fused" = 2) and the rocFFT pipeline (= 0), in the A/B build of the library:
    OPENCORR_HIP_LIB=opencorr_amd/lib/ab/libopencorr_hip_ab.so python tools/fftcc3d_fused32_vs_r5.py          (GPU box)"""
import json
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opencorr_amd as oc  # noqa: E402
from opencorr_amd import synth  # noqa: E402

dev = torch.device("cuda", 0)
dim, r, ns = 512, 16, 37
ref, tar = synth.speckle_pair_3d(dim, dim, dim, seed=20260927, device=dev)
xs, ys, zs = synth.poi_grid_3d(dim, dim, dim, ns, ns, ns, r + 8)
f = oc.FFTCC3D(r, r, r)
f.set_images(ref, tar)
p = torch.from_numpy(oc.make_pois3d(xs, ys, zs)).to(dev)
q = p.clone()
out, res = {}, {}
order = [("round 6 (fftcc3d_fused = 1)", 1), ("rounds 1 - 5 (fftcc3d_fused = 2)", 2), ("rocFFT pipeline (fftcc3d_fused = 0)", 0)]
if len(sys.argv) > 1 and sys.argv[1] == "reversed":
    order.reverse()
for name, value in order:
    f.set_tuning("fftcc3d_fused", value)
    for _ in range(2):
        q.copy_(p)
        f.compute(q)
    torch.cuda.synchronize()
    f.profile_reset()
    f.profile_enable(True)
    for _ in range(8):
        q.copy_(p)
        f.compute(q)
    torch.cuda.synchronize()
    ms, n = f.profile_read()
    f.profile_enable(False)
    out[name] = round(ms / n, 4)
    res[value] = q.cpu().numpy()
zc = 18
other = [c for c in range(res[1].shape[1]) if c != zc]
print(json.dumps({"workload": "512^3 pair, r = 16, 37^3 = 50 653 POIs, FFTCC3D compute() incl. the block-order kernels, HIP events, 8 launches",
                  "ms": out,
                  "same_integers_round6_vs_round5": bool(np.array_equal(res[1][:, other].view(np.uint32), res[2][:, other].view(np.uint32))),
                  "same_integers_round6_vs_pipeline": bool(np.array_equal(res[1][:, other].view(np.uint32), res[0][:, other].view(np.uint32))),
                  "max_abs_d_zncc_round6_vs_round5": float(np.abs(res[1][:, zc] - res[2][:, zc]).max()),
                  "max_abs_d_zncc_round6_vs_pipeline": float(np.abs(res[1][:, zc] - res[0][:, zc]).max())}))
