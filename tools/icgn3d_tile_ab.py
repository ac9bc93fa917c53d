"""ICGN3D1 block schedule ("icgn3d_tile_vox": the queue is visited in compact cubic blocks) against queue order, launches interleaved on ONE
volume pair and ONE FFTCC result:   python tools/icgn3d_tile_ab.py [dim=512] [nside=37] [r=16] [tiles=0,32,48,64,96] [reps=3]"""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import opencorr_amd as oc
from opencorr_amd import synth

dim = int(sys.argv[1]) if len(sys.argv) > 1 else 512
nside = int(sys.argv[2]) if len(sys.argv) > 2 else 37
r = int(sys.argv[3]) if len(sys.argv) > 3 else 16
tiles = [int(v) for v in (sys.argv[4] if len(sys.argv) > 4 else "0,32,48,64,96").split(",")]
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
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
times = {t: [] for t in tiles}
first = None
same = {}
for rep in range(reps + 1):
    for t in tiles:
        g.set_tuning("icgn3d_tile_vox", t)
        q.copy_(guess)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.compute(q)
        b.record()
        b.synchronize()
        if rep:
            times[t].append(round(a.elapsed_time(b), 3))
        res = q.cpu().numpy()
        if first is None:
            first = res
        same[t] = bool(np.array_equal(res.view(np.uint32), first.view(np.uint32)))
print(json.dumps({"volume": "%d^3" % dim, "radius": r, "pois": len(xs), "ms_by_tile_vox": times, "best_ms": {t: min(v) for t, v in times.items()},
                  "same_bits_as_first": same}))
