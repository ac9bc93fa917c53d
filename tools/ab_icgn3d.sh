#!/bin/bash
# A/B of compile-time variants of icgn3d1_kernel on the GPU box: every variant is built with its own -D flags, run on the
# same 256^3 / r = 16 / NS^3-POI case (natural convergence), timed (best of 4) and compared BIT FOR BIT with the first one.
#   VARIANTS="base: ;g1:-DOC_GTAPS=1" bash tools/ab_icgn3d.sh <tag>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-ab3d}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
LIB=opencorr_amd/lib
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize"
OBJS=$(ls $LIB/*.o | grep -v "/icgn3d\.o")   # (icgn3d_fma.o, the fused-arithmetic build of the same file, stays)
cat > /tmp/time3d_ab.py <<'PY'
import sys, time, json, os, numpy as np, torch
sys.path.insert(0, ".")
import opencorr_amd as oc
from opencorr_amd import synth
dev = torch.device("cuda", 0)
dim, r, ns = int(os.environ.get('DIM', 256)), int(os.environ.get('RAD', 16)), int(os.environ.get('NS', 20))
# the GPU renderer of the synthetic pair adds its speckles with float atomics: the images differ in a few pixels from process
# to process (and with them ~40 of 250 000 POIs in the last bits).  Variants are compared on ONE pair, rendered by the first.
pair = "/tmp/ab_pair_%s.pt" % os.environ.get("AB_PAIR_TAG", "x")
if os.path.exists(pair):
    ref, tar = torch.load(pair)
else:
    ref, tar = synth.speckle_pair_3d(dim, dim, dim, seed=20260927, device=dev)
    torch.save((ref, tar), pair)
xs, ys, zs = synth.poi_grid_3d(dim, dim, dim, ns, ns, ns, r + 8)
f = oc.FFTCC3D(r, r, r); f.set_images(ref, tar)
g = oc.ICGN3D1(r, r, r, 0.001, 20.0)
g.share_images(f); g.prepare()
if os.environ.get("ARITH_FMA") == "1":   # variants built with tools/ab_build.py "icgn3d:fma" replace the fused-arithmetic object
    g.set_tuning("arith_fma", 1)
pristine = torch.from_numpy(oc.make_pois3d(xs, ys, zs)).to(dev)
f.compute(pristine); torch.cuda.synchronize()
q = pristine.clone()
best = 1e9
for _ in range(4):
    q.copy_(pristine); torch.cuda.synchronize(); t0 = time.perf_counter(); g.compute(q); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
res = q.cpu().numpy()
np.save(sys.argv[1], res)
same = None
if len(sys.argv) > 2 and os.path.exists(sys.argv[2]):
    same = bool(np.array_equal(np.load(sys.argv[2]).view(np.uint32), res.view(np.uint32)))
print(json.dumps(dict(ms=round(best * 1e3, 3), pois=len(xs), mean_iter=float(res[:, 19].mean()), converged=int((res[:, 18] >= 0).sum()), same_bits_as_first=same)))
PY
first=""
IFS=';' read -ra VS <<< "${VARIANTS:-base: }"
for v in "${VS[@]}"; do
  name=${v%%:*}; defs=${v#*:}
  hipcc --offload-arch=gfx950 -c opencorr_amd/csrc/icgn3d.hip -o /tmp/icgn3d_$name.o $FLAGS $defs || exit 1
  hipcc --offload-arch=gfx950 -shared -o /tmp/libab_$name.so $OBJS /tmp/icgn3d_$name.o -L/opt/rocm/lib -lrocfft -ldl -lpthread || exit 1
  [ -z "$first" ] && first=/tmp/res_$name.npy
  echo -n "$name [$defs]: " | tee -a $OUT/ab.txt
  OPENCORR_HIP_LIB=/tmp/libab_$name.so timeout 300 python /tmp/time3d_ab.py /tmp/res_$name.npy $first 2>&1 | tail -1 | tee -a $OUT/ab.txt
done
