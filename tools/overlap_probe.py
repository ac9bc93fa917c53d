#!/usr/bin/env python
"""Does FFTCC2D of chunk k+1 hide behind ICGN2D1 of chunk k?  Config B, device-resident queue, K chunks, FFTCC on stream A,
ICGN on stream B, one event per chunk.  Prints ms per step for K = 1 (sequential, one stream) and K = 2, 4, 8."""
import sys, time, json, numpy as np, torch
sys.path.insert(0, ".")
import opencorr_amd as oc
from opencorr_amd import synth
dev = torch.device("cuda", 0)
side, r, ns = 4096, 16, 500
ref, tar = synth.speckle_pair_2d(side, side, seed=20260925, device=dev)
xs, ys = synth.poi_grid_2d(side, side, ns, ns, r + 8)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
f = oc.FFTCC2D(r, r); f.set_images(ref, tar)
g = oc.ICGN2D1(r, r, 0.001, 10.0); g.share_images(f); g.prepare()
torch.cuda.synchronize()
pristine = torch.from_numpy(oc.make_pois2d(xs, ys)).to(dev)
q = pristine.clone()
n = len(xs)
out = {}
def run(K):
    cuts = [n * k // K for k in range(K + 1)]
    if K == 1:
        f.set_stream(sa.cuda_stream); g.set_stream(sa.cuda_stream)
        f.compute(q); g.compute(q)
        return
    f.set_stream(sa.cuda_stream); g.set_stream(sb.cuda_stream)
    for k in range(K):
        part = q[cuts[k]:cuts[k + 1]]
        f.compute(part)
        ev = torch.cuda.Event(); ev.record(sa)
        sb.wait_event(ev)
        g.compute(part)
for K in (1, 2, 4, 8, 1):
    best = 1e9
    for _ in range(8):
        q.copy_(pristine); torch.cuda.synchronize()
        t0 = time.perf_counter(); run(K); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    out["K%d" % K] = round(best * 1e3, 3)
    if K == 1: want = q.clone()
    else: assert torch.equal(q.view(torch.int32), want.view(torch.int32)), K
print(json.dumps(out))
