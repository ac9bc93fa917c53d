#!/usr/bin/env python
"""A/B timing of ICGN2D kernel variants on config B with HIP events, variants interleaved round by round so that clock
drift hits all of them alike.  usage: [ENGINE=2 R=20 NS=316] python tools/variant_ab.py 2,4,5 [rounds] [launches]   (GPU box;
ENGINE=2 R=20 NS=316 is config C: ICGN2D2; ARITH_FMA=1: oc_hip_set_tuning arith_fma = 1)"""
import json
import os as _os
# the partners this script compares live in the A/B build of the library only (python -m opencorr_amd.build --ab)
_os.environ.setdefault("OPENCORR_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "opencorr_amd", "lib", "ab", "libopencorr_hip_ab.so"))
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import opencorr_amd as oc
from opencorr_amd import synth

# "8/4" = variant 8 with icgn2d_split_chunks = 4 (the two-stream pipeline of the split launch shape)
labels = sys.argv[1].split(",")
variants = labels
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
launches = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = torch.device("cuda", 0)
side, r, ns = 4096, int(os.environ.get("R", 16)), int(os.environ.get("NS", 500))
engine = int(os.environ.get("ENGINE", 1))
so = dict(uxx=2e-6, vyy=-1e-6) if engine == 2 else None
ref, tar = synth.speckle_pair_2d(side, side, seed=20260925, device=dev, second_order=so)
xs, ys = synth.poi_grid_2d(side, side, ns, ns, r + 8)
stream = torch.cuda.current_stream().cuda_stream
f = oc.FFTCC2D(r, r); f.set_stream(stream); f.set_images(ref, tar)
g = (oc.ICGN2D1 if engine == 1 else oc.ICGN2D2)(r, r, 0.001, 10.0); g.set_stream(stream); g.share_images(f); g.prepare()
if os.environ.get("ARITH_FMA") == "1":   # the fused arithmetic contract (every variant exists in both modes)
    g.set_tuning("arith_fma", 1)
guess = torch.from_numpy(oc.make_pois2d(xs, ys)).to(dev)
f.compute(guess)
q = guess.clone()
times = {v: [] for v in variants}
bits = {}
for rd in range(rounds + 1):
    for v in variants:
        g.set_tuning("icgn2d_variant", int(v.split("/")[0]))
        g.set_tuning("icgn2d_split_chunks", int(v.split("/")[1]) if "/" in v else 0)
        tot = 0.0
        for _ in range(launches):
            q.copy_(guess)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); g.compute(q); b.record(); b.synchronize()
            tot += a.elapsed_time(b)
        if rd:  # round 0 warms up
            times[v].append(tot / launches)
        bits[v] = q.cpu().numpy().view(np.uint32)
first = bits[variants[0]]
print(json.dumps({"workload": "4096^2, r = %d, %d x %d POIs, ICGN2D%d compute() incl. the tile-order kernels, HIP events%s" % (r, ns, ns, engine, ", arith_fma = 1" if os.environ.get("ARITH_FMA") == "1" else ""), "launches_per_round": launches,
                  "ms": {str(v): [round(t, 4) for t in ts] for v, ts in times.items()},
                  "mean_ms": {str(v): round(float(np.mean(ts)), 4) for v, ts in times.items()},
                  "same_bits": {str(v): bool(np.array_equal(bits[v], first)) for v in variants}}))
