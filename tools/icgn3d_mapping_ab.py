"""ICGN3D1: row mapping (icgn3d_rows.hip, "icgn3d_mapping" = 1) against the s-mod-512 mapping (icgn3d.hip, = 0) on one MI355X, launches
interleaved on ONE volume pair and ONE FFTCC result:   python tools/icgn3d_mapping_ab.py [dim=512] [nside=37] [r=16] [reps=4]
Prints one JSON object: ms per launch for both, iteration statistics, and how far apart the two results are (a re-association)."""
import json
import os as _os
# the partners this script compares live in the A/B build of the library only (python -m opencorr_amd.build --ab)
_os.environ.setdefault("OPENCORR_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "opencorr_amd", "lib", "ab", "libopencorr_hip_ab.so"))
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import opencorr_amd as oc
from opencorr_amd import synth

dim = int(sys.argv[1]) if len(sys.argv) > 1 else 512
nside = int(sys.argv[2]) if len(sys.argv) > 2 else 37
r = int(sys.argv[3]) if len(sys.argv) > 3 else 16
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 4
dev = torch.device("cuda", 0)
ref, tar = synth.speckle_pair_3d(dim, dim, dim, seed=20260927, device=dev)
xs, ys, zs = synth.poi_grid_3d(dim, dim, dim, nside, nside, nside, r + 8)
f = oc.FFTCC3D(r, r, r)
f.set_images(ref, tar)
g = oc.ICGN3D1(r, r, r, 0.001, 20.0)
g.share_images(f)
g.prepare()
guess = torch.from_numpy(oc.make_pois3d(xs, ys, zs)).to(dev)
f.compute(guess)
q = guess.clone()
times = {0: [], 1: []}
res = {}
for rep in range(reps + 1):
    for mapping in (1, 0):
        g.set_tuning("icgn3d_mapping", mapping)
        q.copy_(guess)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.compute(q)
        b.record()
        b.synchronize()
        if rep:
            times[mapping].append(a.elapsed_time(b))
        res[mapping] = q.cpu().numpy()
it = res[1][:, 19]
d = np.abs(res[1][:, [3, 7, 11]].astype(np.float64) - res[0][:, [3, 7, 11]].astype(np.float64))
print(json.dumps({"volume": "%d^3" % dim, "radius": r, "pois": len(xs), "rows_ms": [round(t, 3) for t in times[1]], "lanes_ms": [round(t, 3) for t in times[0]],
                  "rows_best_ms": min(times[1]), "lanes_best_ms": min(times[0]), "mean_iterations": float(it[it > 0].mean()),
                  "converged_rows": int((res[1][:, 18] >= 0).sum()), "converged_lanes": int((res[0][:, 18] >= 0).sum()),
                  "same_iteration_counts": float((res[1][:, 19] == res[0][:, 19]).mean()), "max_abs_d_disp_between_mappings": float(d.max())}))
