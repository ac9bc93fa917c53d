#!/usr/bin/env python
"""ICGN3D1 on config E with two (the default) and ONE persistent workgroup per CU (OC_ICGN3D_BLOCKS), everything else equal: how much
of a workgroup's phase sequence does the second workgroup hide?   python tools/icgn3d_occupancy_probe.py   (GPU box)"""
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
dim, r, ns = 512, 16, 37
ref, tar = synth.speckle_pair_3d(dim, dim, dim, seed=20260927, device=dev)
xs, ys, zs = synth.poi_grid_3d(dim, dim, dim, ns, ns, ns, r + 8)
f = oc.FFTCC3D(r, r, r); f.set_images(ref, tar)
g = oc.ICGN3D1(r, r, r, 0.001, 20.0); g.share_images(f); g.prepare()
p = torch.from_numpy(oc.make_pois3d(xs, ys, zs)).to(dev)
f.compute(p); q = p.clone()
for _ in range(2):
    q.copy_(p); g.compute(q)
torch.cuda.synchronize(); g.profile_enable(True)
for _ in range(4):
    q.copy_(p); g.compute(q)
torch.cuda.synchronize(); ms, n = g.profile_read()
print(json.dumps({"icgn3d1_ms": round(ms / n, 3)}))
''' % ROOT
out = []
for blocks in (512, 256):
    env = dict(os.environ)
    env["OC_ICGN3D_BLOCKS"] = str(blocks)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=900)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    rec = json.loads(line[-1]) if line else {"error": r.stderr[-300:]}
    rec.update({"persistent_workgroups": blocks, "workgroups_per_cu": blocks // 256})
    out.append(rec)
print(json.dumps(out))
