"""What a small batch costs end to end (host queue and device queue, n = 1 ... 1024 POIs): the per-call floor that the combining front end of
oc_hip_compute_one amortises.  GPU box: python tools/small_batch_latency.py"""
import sys, time, numpy as np
sys.path.insert(0, ".")
import opencorr_amd as oc, torch
from opencorr_amd import synth
ref, tar = synth.speckle_pair_2d(600, 640, seed=3)
xs, ys = synth.poi_grid_2d(600, 640, 64, 64, 28)
f = oc.FFTCC2D(16, 16); f.set_images(ref, tar)
g = oc.ICGN2D1(16, 16, 0.001, 10); g.share_images(f); g.prepare()
start = oc.make_pois2d(xs, ys); f.compute(start)
for n in (1, 8, 32, 64, 256, 1024):
    q = start[:n].copy()
    g.compute(q.copy())
    t0 = time.perf_counter()
    reps = 200
    for _ in range(reps):
        g.compute(q.copy())
    dt = (time.perf_counter() - t0) / reps
    qd = torch.from_numpy(start[:n].copy()).cuda()
    g.compute(qd); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.compute(qd)
    torch.cuda.synchronize()
    dd = (time.perf_counter() - t0) / reps
    print("n = %5d: host queue %.1f us per call, device queue %.1f us per call" % (n, dt * 1e6, dd * 1e6))
