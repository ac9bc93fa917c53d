#!/usr/bin/env python
"""ICGN2D1 on config B with 3, 2 and 1 workgroups per CU (the LDS request padded: OC_ICGN2D_LDS_PAD), everything else equal: is
the kernel bound by a shared resource (time per POI barely moves with the number of resident workgroups) or by the latency of a
workgroup's own phase sequence (time ~ 1 / workgroups)?   python tools/icgn2d_occupancy_probe.py   (GPU box; one process per point)"""
import json
import os as _os
# the OC_ICGN*_ knobs are honoured by the A/B build of the library only (python -m opencorr_amd.build --ab)
_os.environ.setdefault("OPENCORR_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "opencorr_amd", "lib", "ab", "libopencorr_hip_ab.so"))
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json, torch
sys.path.insert(0, %r)
import opencorr_amd as oc
from opencorr_amd import synth
dev = torch.device("cuda", 0)
ref, tar = synth.speckle_pair_2d(4096, 4096, seed=20260925, device=dev)
xs, ys = synth.poi_grid_2d(4096, 4096, 500, 500, 24)
f = oc.FFTCC2D(16, 16); f.set_images(ref, tar)
g = oc.ICGN2D1(16, 16, 0.001, 10.0); g.share_images(f); g.prepare()
p = torch.from_numpy(oc.make_pois2d(xs, ys)).to(dev)
f.compute(p); q = p.clone()
for _ in range(3):
    q.copy_(p); g.compute(q)
torch.cuda.synchronize(); g.profile_enable(True)
for _ in range(10):
    q.copy_(p); g.compute(q)
torch.cuda.synchronize(); ms, n = g.profile_read()
print(json.dumps({"icgn2d1_ms": round(ms / n, 4)}))
''' % ROOT
out = []
for wgs, pad in ((3, 0), (2, 60 * 1024), (1, 100 * 1024)):
    env = dict(os.environ)
    if pad:
        env["OC_ICGN2D_LDS_PAD"] = str(pad)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    rec = json.loads(line[-1]) if line else {"error": r.stderr[-300:]}
    rec.update({"workgroups_per_cu": wgs, "waves_per_simd": 2 * wgs, "lds_request_bytes": pad or 52736})
    out.append(rec)
print(json.dumps(out))
