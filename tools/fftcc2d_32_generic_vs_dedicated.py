import sys, json, numpy as np, torch
sys.path.insert(0, ".")
import opencorr_amd as oc
from opencorr_amd import synth
dev = torch.device("cuda", 0)
side = 4096
ref, tar = synth.speckle_pair_2d(side, side, seed=20260925, device=dev)
xs, ys = synth.poi_grid_2d(side, side, 500, 500, 24)
f = oc.FFTCC2D(16, 16); f.set_images(ref, tar)
q0 = torch.from_numpy(oc.make_pois2d(xs, ys)).to(dev); q = q0.clone()
res = {}; outs = {}
for rep in range(2):
  for mode in (1, 2):
    f.set_tuning("fftcc2d_fused", mode)
    for _ in range(3): q.copy_(q0); f.compute(q)
    torch.cuda.synchronize()
    f.profile_reset(); f.profile_enable(True)
    for _ in range(20): q.copy_(q0); f.compute(q)
    torch.cuda.synchronize()
    ms, n = f.profile_read(); f.profile_enable(False)
    res.setdefault("dedicated x2" if mode == 1 else "generic <32,32>", []).append(round(ms / n, 4))
    outs[mode] = q.cpu().numpy()
same_int = all(np.array_equal(outs[1][:, c], outs[2][:, c]) for c in (2, 8, 14, 15))
print(json.dumps(dict(ms=res, same_integers=same_int, max_zncc_diff=float(np.abs(outs[1][:, 16] - outs[2][:, 16]).max()))))
