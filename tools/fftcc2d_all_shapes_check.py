"""Every fused FFTCC2D window shape (12 square sides, 42 rectangular pairs) against the oracle and the rocFFT pipeline on one
small pair: integers identical, ZNCC within 3e-5, guard trippers untouched, odd queue length.   python tools/fftcc2d_all_shapes_check.py"""
import sys, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import opencorr_amd as oc
from opencorr_amd import synth
import oracle

h, w = 520, 600
ref, tar = synth.speckle_pair_2d(h, w, seed=7, device=None) if hasattr(synth, "speckle_pair_2d") else (None, None)
ref = np.ascontiguousarray(np.asarray(ref.cpu() if hasattr(ref, "cpu") else ref, dtype=np.float32))
tar = np.ascontiguousarray(np.asarray(tar.cpu() if hasattr(tar, "cpu") else tar, dtype=np.float32))
sides = [16, 20, 24, 32, 40, 48, 64]
shapes = [(r, r) for r in (8, 9, 10, 12, 15, 16, 18, 20, 24, 25, 30, 32)] + [(a // 2, b // 2) for a in sides for b in sides if a != b]
P = oracle.P2
bad = []
worst = 0.0
for rx, ry in shapes:
    rng = np.random.default_rng(rx * 100 + ry)
    n = 203
    m = max(rx, ry) + 6
    xs = rng.uniform(m, w - m, n).astype(np.float32)
    ys = rng.uniform(m, h - m, n).astype(np.float32)
    xs[::2] = np.floor(xs[::2]); ys[::2] = np.floor(ys[::2])
    base = oc.make_pois2d(xs, ys)
    base[:, P["u"]] = rng.integers(-3, 4, n).astype(np.float32)
    base[:, P["v"]] = rng.integers(-3, 4, n).astype(np.float32)
    base[7, P["x"]] = 2.0; base[100, P["y"]] = h - 1.0; base[n - 1, P["u"]] = 5000.0
    f = oc.FFTCC2D(rx, ry); f.set_images(ref, tar)
    fused = f.compute(base.copy())
    f.set_tuning("fftcc2d_fused", 0)
    piped = f.compute(base.copy())
    want = base.copy(); oracle.fftcc2d(ref, tar, rx, ry, want)
    ok = all(np.array_equal(fused[:, P[c]], want[:, P[c]]) and np.array_equal(fused[:, P[c]], piped[:, P[c]]) for c in ("u", "v", "u0", "v0"))
    dz = float(np.abs(fused[:, P["zncc"]] - want[:, P["zncc"]]).max())
    worst = max(worst, dz)
    trip = [7, 100, n - 1]
    ok = ok and dz <= 3e-5 and np.array_equal(fused[trip].view(np.uint32), base[trip].view(np.uint32))
    if not ok: bad.append((rx, ry, dz))
print(json.dumps(dict(shapes=len(shapes), failed=bad, worst_zncc_diff=worst)))
