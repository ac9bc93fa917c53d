#!/usr/bin/env python
"""Re-runs one seed of tests/test_gpu_fuzz.py::test_fuzz_icgn2d1_icgn2d2_nr2d1_iclm for NR2D1 and prints the records that differ from the oracle."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import opencorr_amd
import oracle
from opencorr_amd import synth
import test_gpu_fuzz as F

seed = int(sys.argv[1])
rng = np.random.default_rng(1000 + seed)
h, w = int(rng.integers(150, 260)), int(rng.integers(150, 260))
warp = dict(u=float(rng.uniform(-3, 3)), ux=float(rng.uniform(-4e-3, 4e-3)), uy=float(rng.uniform(-4e-3, 4e-3)),
            v=float(rng.uniform(-3, 3)), vx=float(rng.uniform(-4e-3, 4e-3)), vy=float(rng.uniform(-4e-3, 4e-3)))
ref, tar = synth.speckle_pair_2d(h, w, seed=500 + seed, warp=warp)
rx, ry = int(rng.integers(4, 22)), int(rng.integers(4, 22))
conv = float(rng.choice([1e-3, 1e-4, 5e-3]))
stop = float(rng.choice([10, 6, 15]))
pois, P = F._queue2d(rng, h, w, rx, ry, 150)
pois = F._guess2d(rng, pois, P, warp["u"], warp["v"], spread=[0.05, 0.4, 1.0][seed % 3])
print("case", dict(h=h, w=w, rx=rx, ry=ry, conv=conv, stop=stop))
nr = opencorr_amd.NR2D1(rx, ry, conv, stop)
nr.set_images(ref, tar)
nr.prepare()
got = nr.compute(pois.copy())
inv = {v: k for k, v in P.items()}
for order, name in ((oracle.ORDER_LANES, "LANES"), (oracle.ORDER_SEQ, "SEQ")):
    want = pois.copy()
    oracle.nr2d1(oracle.PreparedNR2D(ref, tar), rx, ry, conv, stop, want, order=order, lanes=64)
    bad = np.flatnonzero(~(F._bits(got) == F._bits(want)).all(axis=1))
    print(name, "mismatching POIs:", bad.tolist())
    for i in bad[:4]:
        cols = np.flatnonzero(F._bits(got[i]) != F._bits(want[i]))
        print("  poi", i, "input", {inv.get(c, c): float(pois[i, c]) for c in (0, 1, 2, 3, 4, 8, 9, 10, 16)})
        print("      ", {inv.get(c, c): (float(got[i, c]), float(want[i, c])) for c in cols})
