#!/usr/bin/env python
"""FFTCC3D on config E's queue (512^3, r = 16, 37^3 POIs) visited in queue order and in cubic blocks of several sizes
(oc_hip_set_tuning "fftcc3d_tile_vox", round 5): kernel time by hipEvents, records compared bit for bit with queue order.
    python tools/fftcc3d_tile_ab.py [sizes=0,32,48,64,96,128] [rounds=3]        (GPU box)"""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import opencorr_amd as oc
from opencorr_amd import synth

sizes = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,32,48,64,96,128").split(",")]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
dim, r, ns = 512, 16, 37
ref, tar = synth.speckle_pair_3d(dim, dim, dim, seed=20260927, device=dev)
xs, ys, zs = synth.poi_grid_3d(dim, dim, dim, ns, ns, ns, r + 8)
f = oc.FFTCC3D(r, r, r)
f.set_stream(torch.cuda.current_stream().cuda_stream)
f.set_images(ref, tar)
p = torch.from_numpy(oc.make_pois3d(xs, ys, zs)).to(dev)
q = p.clone()
times = {s: [] for s in sizes}
bits = {}
for rd in range(rounds + 1):
    for s in sizes:
        f.set_tuning("fftcc3d_tile_vox", s)
        q.copy_(p); f.compute(q); torch.cuda.synchronize()
        f.profile_reset(); f.profile_enable(True)
        for _ in range(4):
            q.copy_(p); f.compute(q)
        torch.cuda.synchronize()
        ms, n = f.profile_read(); f.profile_enable(False)
        if rd:
            times[s].append(round(ms / n, 4))
        bits[s] = q.cpu().numpy().view(np.uint32)
print(json.dumps({"workload": "config E: 512^3, r = 16, %d POIs, fftcc3d_fused32_kernel alone (hipEvents), tile order kernels not included (~20 us)" % len(xs),
                  "ms_by_tile_vox": {str(s): t for s, t in times.items()},
                  "same_bits_as_queue_order": {str(s): bool(np.array_equal(bits[s], bits[sizes[0]])) for s in sizes}}))
