#!/bin/bash
# Phase ablation of icgn3d1_kernel on the GPU box: builds the library with -DOC_ABLATE=<mask> (icgn3d.hip) and times
# ICGN3D1 on a 256^3 volume, r = 16, 20^3 POIs, three forced iterations.  usage: bash tools/ablate_icgn3d.sh <tag>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-ablate3d}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
LIB=opencorr_amd/lib
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize"
OBJS=$(ls $LIB/*.o | grep -v "/icgn3d\.o")   # (icgn3d_fma.o, the fused-arithmetic build of the same file, stays)
cat > /tmp/time3d.py <<'PY'
import sys, time, json, numpy as np, torch
sys.path.insert(0, ".")
import opencorr_amd as oc
from opencorr_amd import synth
dev = torch.device("cuda", 0)
import os
dim, r, ns = int(os.environ.get('DIM', 256)), 16, int(os.environ.get('NS', 20))
ref, tar = synth.speckle_pair_3d(dim, dim, dim, seed=20260927, device=dev)
xs, ys, zs = synth.poi_grid_3d(dim, dim, dim, ns, ns, ns, r + 8)
g = oc.ICGN3D1(r, r, r, 0.001, 20.0)
g.set_images(ref, tar); g.prepare()
pr = oc.make_pois3d(xs, ys, zs)
w = synth.DEFAULT_WARP_3D
pr[:, 3], pr[:, 7], pr[:, 11] = round(w["u"]), round(w["v"]), round(w["w"])
pristine = torch.from_numpy(pr).to(dev); q = pristine.clone()
best = 1e9
for _ in range(4):
    q.copy_(pristine); torch.cuda.synchronize(); t0 = time.perf_counter(); g.compute(q); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
res = q.cpu().numpy()
print(json.dumps(dict(ms=best * 1e3, pois=len(xs), mean_iter=float(res[:, 19].mean()), timeline_kcycles=[float(v) for v in res[:, 22:28].astype(np.float64).mean(0)])))
PY
for mask in ${MASKS:-0 16}; do
  hipcc --offload-arch=gfx950 -c opencorr_amd/csrc/icgn3d.hip -o /tmp/icgn3d_ab.o $FLAGS -DOC_ABLATE=$mask || exit 1
  hipcc --offload-arch=gfx950 -shared -o /tmp/libablate_$mask.so $OBJS /tmp/icgn3d_ab.o -L/opt/rocm/lib -lrocfft -ldl -lpthread || exit 1
  echo -n "mask $mask: " | tee -a $OUT/ablate.txt
  OPENCORR_HIP_LIB=/tmp/libablate_$mask.so timeout 300 python /tmp/time3d.py 2>&1 | tail -1 | tee -a $OUT/ablate.txt
done
