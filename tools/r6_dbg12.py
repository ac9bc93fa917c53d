import os, sys, numpy as np
sys.path.insert(0, ".")
os.environ.setdefault("OPENCORR_HIP_LIB", os.path.join("opencorr_amd", "lib", "ab", "libopencorr_hip_ab.so"))
import opencorr_amd as oc, oracle
from opencorr_amd import synth
ref, tar = synth.speckle_pair_2d(300, 340, seed=5)
h, w = ref.shape
P = oracle.P2
for dof in (12, 6):
    r = 16
    xs, ys = synth.poi_grid_2d(h, w, 32, 30, 26)
    pois = oracle.make_pois2d(xs, ys)
    oracle.fftcc2d(ref, tar, r, r, pois)
    prep = oracle.Prepared2D(ref, tar)
    fn = oracle.icgn2d1 if dof == 6 else oracle.icgn2d2
    solved = pois.copy()
    fn(prep, r, r, 0.001, 10, solved, order=oracle.ORDER_LANES, lanes=64)
    for kinds in ([0], [1], [2], [3], [4], [0, 1, 2, 3, 4]):
        q = pois.copy()
        slot = np.arange(len(q)) % 8
        if 0 in kinds: q[slot == 0, P["zncc"]] = -2.0
        if 1 in kinds: q[slot == 1, P["u"]] = w - 30.0
        if 2 in kinds: q[slot == 2, P["v"]] = np.nan
        if 3 in kinds: q[slot == 3, 2:14] = solved[slot == 3, 2:14]
        if 4 in kinds:
            q[slot == 4, P["u"]] += 6.5
            q[slot == 4, P["v"]] -= 5.5
        want = q.copy()
        fn(prep, r, r, 0.001, 10, want, order=oracle.ORDER_LANES, lanes=64)
        g = (oc.ICGN2D1 if dof == 6 else oc.ICGN2D2)(r, r, 0.001, 10)
        g.set_images(ref, tar); g.prepare(); g.set_tuning("icgn2d_variant", 9)
        got = g.compute(q.copy())
        bad = np.argwhere(got.view(np.uint32) != want.view(np.uint32))
        rows = sorted(set(bad[:, 0]))
        print("dof", dof, "kinds", kinds, "ablate", os.environ.get("OC_BAND_ABLATE"), "mismatching POIs", len(rows), "slots", sorted(set(int(i) % 8 for i in rows)))
        for i in rows[:3]:
            print("   poi", i, "got ", got[i, [2, 8, 16, 17, 18]], "want", want[i, [2, 8, 16, 17, 18]])
