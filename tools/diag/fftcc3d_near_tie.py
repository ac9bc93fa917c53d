"""Near-ties of the FFTCC3D peak: which POIs of the r = 13 timing case (tools/fftcc3d_sizes.py 13 12 256, volumes generated on the device) get different integers from fftcc3d_fusedn<26> and
the rocFFT pipeline, and what the oracle says there."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import opencorr_amd as oc
import oracle
from opencorr_amd import synth
dim, nside = 256, 12
dev = torch.device('cuda', 0)
for r in (13,):
    ref_d, tar_d = synth.speckle_pair_3d(dim, dim, dim, seed=20260927, device=dev)
    ref, tar = ref_d.cpu().numpy(), tar_d.cpu().numpy()
    xs, ys, zs = synth.poi_grid_3d(dim, dim, dim, nside, nside, nside, r + 8)
    pois = oc.make_pois3d(xs, ys, zs)
    f = oc.FFTCC3D(r, r, r)
    f.set_images(ref_d, tar_d)
    a = f.compute(pois.copy())
    a2 = f.compute(pois.copy())
    f.set_tuning("fftcc3d_fused", 0)
    b = f.compute(pois.copy())
    ints = [3, 7, 11]
    bad = np.where((a[:, ints] != b[:, ints]).any(axis=1))[0]
    print("r", r, "fused deterministic", np.array_equal(a.view(np.uint32), a2.view(np.uint32)), "differing POIs", bad.tolist())
    if len(bad):
        w = pois[bad].copy()
        oracle.fftcc3d(ref, tar, r, r, r, w)
        for n, i in enumerate(bad):
            print("  poi", i, pois[i, :3].tolist(), "fused", a[i, ints].tolist(), a[i, 18], "rocfft", b[i, ints].tolist(), b[i, 18], "oracle", w[n, ints].tolist(), w[n, 18])
