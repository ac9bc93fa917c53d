#!/bin/bash
# Ablation of icgn2d_kernel on the GPU box: builds the library with -DOC_ABLATE2D=<mask> (icgn2d.hip) and times ICGN2D1 on
# config B (4096^2, r = 16, 500 x 500 POIs; ENGINE=2 R=20 NS=316 = config C).  usage: MASKS="1 3 5" bash tools/ablate_icgn2d.sh <tag>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-ablate2d}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
LIB=opencorr_amd/lib
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -DOC_BUILD_AB=1 ${EXTRA_FLAGS}"
OBJS=$(ls $LIB/*.o | grep -v "icgn2d\.o")
cat > /tmp/time2d.py <<'PY'
import sys, time, json, os, numpy as np, torch
sys.path.insert(0, ".")
import opencorr_amd as oc
from opencorr_amd import synth
dev = torch.device("cuda", 0)
side, r, ns = 4096, int(os.environ.get('R', 16)), int(os.environ.get('NS', 500))
engine = int(os.environ.get('ENGINE', 1))  # 1 = ICGN2D1, 2 = ICGN2D2 (config C: ENGINE=2 R=20 NS=316)
so = dict(uxx=2e-6, vyy=-1e-6) if engine == 2 else None
ref, tar = synth.speckle_pair_2d(side, side, seed=20260925, device=dev, second_order=so)
xs, ys = synth.poi_grid_2d(side, side, ns, ns, r + 8)
f = oc.FFTCC2D(r, r); f.set_images(ref, tar)
g = (oc.ICGN2D1 if engine == 1 else oc.ICGN2D2)(r, r, 0.001, 10.0); g.share_images(f); g.prepare()
for k, v in [kv.split("=") for kv in os.environ.get("TUNING", "").split(",") if kv]:
    g.set_tuning(k, int(v))
pristine = torch.from_numpy(oc.make_pois2d(xs, ys)).to(dev)
f.compute(pristine); torch.cuda.synchronize()
q = pristine.clone()
ts = []
for _ in range(6):
    q.copy_(pristine); torch.cuda.synchronize(); t0 = time.perf_counter(); g.compute(q); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
res = q.cpu().numpy()
print(json.dumps(dict(ms_best=min(ts) * 1e3, ms_median=sorted(ts)[len(ts) // 2] * 1e3, pois=len(xs), mean_iter=float(res[:, 17].mean()),
                      converged=int((res[:, 16] >= 0).sum()),
                      timeline_kcycles=dict(zip(['ref_and_hessian_sweep', 'hessian_reduce_inverse', 'interpolation_sweeps', 'rest_of_iterations'],
                                                [float(v) for v in res[:, [20, 21, 22, 19]].astype(np.float64).mean(0)])))))
PY
for mask in ${MASKS:-0 1 3 5}; do
  hipcc --offload-arch=gfx950 -c opencorr_amd/csrc/icgn2d.hip -o /tmp/icgn2d_ab.o $FLAGS -DOC_ABLATE2D=$mask || exit 1
  hipcc --offload-arch=gfx950 -shared -o /tmp/libablate2d_$mask.so $OBJS /tmp/icgn2d_ab.o -L/opt/rocm/lib -lrocfft -ldl -lpthread || exit 1
  echo -n "mask $mask: " | tee -a $OUT/ablate2d.txt
  OPENCORR_HIP_LIB=/tmp/libablate2d_$mask.so timeout 300 python /tmp/time2d.py 2>&1 | tail -1 | tee -a $OUT/ablate2d.txt
done
