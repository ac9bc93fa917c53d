import sys, json, numpy as np, torch
sys.path.insert(0, ".")
import opencorr_amd as oc
from opencorr_amd import synth
dev = torch.device("cuda", 0)
side = 4096
ref, tar = synth.speckle_pair_2d(side, side, seed=20260925, device=dev)
out = {}
shapes = [(r, r) for r in (8, 9, 25, 30, 32, 16)]
if len(sys.argv) > 1:   # e.g. "8x16,16x8,20x16,24x32,32x12": rx x ry pairs (rectangular windows, fftcc2d_fusedr.hip)
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1].split(",")]
for rx, ry in shapes:
    r = "%dx%d" % (rx, ry) if rx != ry else rx
    xs, ys = synth.poi_grid_2d(side, side, 500, 500, max(rx, ry) + 8)
    f = oc.FFTCC2D(rx, ry); f.set_images(ref, tar)
    q0 = torch.from_numpy(oc.make_pois2d(xs, ys)).to(dev); q = q0.clone()
    res = {}
    for fused in (1, 0):
        f.set_tuning("fftcc2d_fused", fused)
        ts = []
        for _ in range(4):
            q.copy_(q0); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); f.compute(q); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
        res["fused" if fused else "rocfft"] = round(min(ts), 3)
    out[r] = res
print(json.dumps(out))
