"""FFTCC3D per window side: the single-kernel paths (fftcc3d_fusedn.hip / fftcc3d_fused.hip / fftcc3d_planes.hip) against the
rocFFT pipeline on one MI355X:   python tools/fftcc3d_sizes.py [8,12,16,20,30] [pois_per_side=8] [dim=256]
A radius may be a non-cubic triple "rx.ry.rz" (fftcc3d_box.hip): python tools/fftcc3d_sizes.py 8,8.8.4,16.16.8,12
One JSON object: radius -> {fused_ms, rocfft_ms, us_per_poi_fused, us_per_poi_rocfft, same_integers, max_zncc_diff}."""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import opencorr_amd as oc
from opencorr_amd import synth

radii = [tuple(int(c) for c in v.split(".")) * (1 if "." in v else 3) for v in (sys.argv[1] if len(sys.argv) > 1 else "8,12,16,20,30").split(",")]
nside = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dim = int(sys.argv[3]) if len(sys.argv) > 3 else 256
dev = torch.device("cuda", 0)
ref, tar = synth.speckle_pair_3d(dim, dim, dim, seed=20260927, device=dev)
out = {}
for r in radii:
    xs, ys, zs = synth.poi_grid_3d(dim, dim, dim, nside, nside, nside, max(r) + 8)
    f = oc.FFTCC3D(*r)
    f.set_images(ref, tar)
    q0 = torch.from_numpy(oc.make_pois3d(xs, ys, zs)).to(dev)
    q = q0.clone()
    res, rec = {}, {}
    for fused in (1, 0):
        f.set_tuning("fftcc3d_fused", fused)
        ts = []
        for _ in range(4):
            q.copy_(q0)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            f.compute(q)
            b.record()
            b.synchronize()
            ts.append(a.elapsed_time(b))
        res[fused] = q.cpu().numpy()
        rec["fused_ms" if fused else "rocfft_ms"] = round(min(ts), 4)
    rec["pois"] = len(xs)
    rec["us_per_poi_fused"] = round(rec["fused_ms"] * 1e3 / len(xs), 3)
    rec["us_per_poi_rocfft"] = round(rec["rocfft_ms"] * 1e3 / len(xs), 3)
    ints = [3, 7, 11, 15, 16, 17]
    rec["same_integers"] = bool(np.array_equal(res[1][:, ints], res[0][:, ints]))
    rec["max_zncc_diff"] = float(np.abs(res[1][:, 18] - res[0][:, 18]).max())
    out[r[0] if r[0] == r[1] == r[2] else "%d.%d.%d" % r] = rec
    del f
print(json.dumps(out))
