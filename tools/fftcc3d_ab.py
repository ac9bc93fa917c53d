#!/usr/bin/env python
"""Times FFTCC3D (r = 16, config E's queue) under each library of tools/ab_build/ named on the command line and compares the records
bit for bit with the first:   python tools/fftcc3d_ab.py fftcc3d_fused base wavexy      (GPU box; one process per library)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json, numpy as np, torch
sys.path.insert(0, %r)
import opencorr_amd as oc
from opencorr_amd import synth
dev = torch.device("cuda", 0)
dim, r, ns = 512, 16, 37
# the GPU renderer of the synthetic pair adds its speckles with float atomics: the volumes differ in a few voxels from process to
# process, so every library is run on ONE pair, rendered by the first process
import os
pair = "/tmp/fftcc3d_ab_pair.pt"
if os.path.exists(pair):
    ref, tar = torch.load(pair)
else:
    ref, tar = synth.speckle_pair_3d(dim, dim, dim, seed=20260927, device=dev)
    torch.save((ref, tar), pair)
xs, ys, zs = synth.poi_grid_3d(dim, dim, dim, ns, ns, ns, r + 8)
f = oc.FFTCC3D(r, r, r); f.set_images(ref, tar)
p = torch.from_numpy(oc.make_pois3d(xs, ys, zs)).to(dev)
q = p.clone()
for _ in range(2):
    q.copy_(p); f.compute(q)
torch.cuda.synchronize(); f.profile_enable(True)
for _ in range(8):
    q.copy_(p); f.compute(q)
torch.cuda.synchronize(); ms, n = f.profile_read()
np.save(sys.argv[1], q.cpu().numpy())
print(json.dumps({"fftcc3d_ms": round(ms / n, 4)}))
''' % ROOT
src, names = sys.argv[1], sys.argv[2:]
out, first = [], None
for rep in (1, 2):
    for name in names:
        env = dict(os.environ, OPENCORR_HIP_LIB=os.path.join(ROOT, "tools", "ab_build", "libab_%s_%s.so" % (src, name)))
        res = "/tmp/fftcc3d_ab_%s.npy" % name
        r = subprocess.run([sys.executable, "-c", CHILD, res], env=env, capture_output=True, text=True, timeout=900)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        rec = json.loads(line[-1]) if line else {"error": r.stderr[-300:]}
        import numpy as np
        if line:
            a = np.load(res)
            if first is None:
                first = a
            rec["same_bits_as_first"] = bool(np.array_equal(a.view(np.uint32), first.view(np.uint32)))
            zc = 18  # POI3D: ZNCC (opencorr_amd/io.py TABLE3D); everything else FFTCC3D writes is an integer-valued float
            other = [c for c in range(a.shape[1]) if c != zc]
            rec["same_integers_as_first"] = bool(np.array_equal(a[:, other].view(np.uint32), first[:, other].view(np.uint32)))
            rec["max_abs_d_zncc_vs_first"] = float(np.abs(a[:, zc] - first[:, zc]).max())
            if os.environ.get("OC_AB_TIMELINE"):   # builds with -DOC_F32_TIMELINE leave thread 0's cycles per phase in fields 19 ... 30
                rec["timeline_mean_cycles"] = [round(float(x), 1) for x in a[:, 19:31].mean(axis=0)]
        rec.update({"lib": name, "run": rep})
        out.append(rec)
        print(json.dumps(rec), flush=True)
