"""ICGN2D1 launch shape for small queues: which variant wins at 2 500 ... 40 000 POIs?  (config A is 10 000 POIs on 2048^2, r = 15)
   python tools/small_queue_variants.py"""
import json, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opencorr_amd as oc
from opencorr_amd import synth

dev = torch.device("cuda", 0)
side, r = 2048, 15
ref, tar = synth.speckle_pair_2d(side, side, seed=20260925, device=dev)
f = oc.FFTCC2D(r, r); f.set_images(ref, tar)
g = oc.ICGN2D1(r, r, 0.001, 10.0); g.share_images(f); g.prepare()
out = {}
for ns in (50, 70, 100, 140, 200):
    xs, ys = synth.poi_grid_2d(side, side, ns, ns, r + 8)
    pristine = torch.from_numpy(oc.make_pois2d(xs, ys)).to(dev)
    f.compute(pristine); torch.cuda.synchronize()
    q = pristine.clone()
    row = {}
    first = None
    for variant in (2, 4, 5):
        g.set_tuning("icgn2d_variant", variant)
        for _ in range(5):
            q.copy_(pristine); g.compute(q)
        torch.cuda.synchronize()
        g.profile_reset(); g.profile_enable(True)
        for _ in range(30):
            q.copy_(pristine); g.compute(q)
        torch.cuda.synchronize()
        ms, n = g.profile_read(); g.profile_enable(False)
        res = q.cpu().numpy().view(np.uint32)
        if first is None: first = res
        row["variant %d" % variant] = round(ms / n, 4)
        assert np.array_equal(first, res)
    out["%d POIs" % len(xs)] = row
    print(len(xs), row, flush=True)
print(json.dumps(out))
