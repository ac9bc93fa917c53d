#!/bin/bash
# A/B of extra compiler flags on ONE kernel file: builds a second library whose <file>.o got EXTRA_FLAGS and times config B
# (FFTCC2D and ICGN2D1 separately, HIP events) with both libraries, interleaved.
#   usage: FILE=fftcc2d_fused EXTRA_FLAGS="-Xclang -target-feature -Xclang -packed-fp32-ops" bash tools/ab_flags.sh <tag>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-abflags}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
LIB=opencorr_amd/lib
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize"
OBJS=$(ls $LIB/*.o | grep -v "/$FILE\.o")
hipcc --offload-arch=gfx950 -c opencorr_amd/csrc/$FILE.hip -o /tmp/ab_$FILE.o $FLAGS $EXTRA_FLAGS || exit 1
hipcc --offload-arch=gfx950 -shared -o /tmp/libab_$FILE.so $OBJS /tmp/ab_$FILE.o -L/opt/rocm/lib -lrocfft -ldl -lpthread || exit 1
cat > /tmp/time_ab.py <<'PY'
import sys, json, numpy as np, torch
sys.path.insert(0, ".")
import opencorr_amd as oc
from opencorr_amd import synth
dev = torch.device("cuda", 0)
side, r, ns = 4096, 16, 500
ref, tar = synth.speckle_pair_2d(side, side, seed=20260925, device=dev)
xs, ys = synth.poi_grid_2d(side, side, ns, ns, r + 8)
stream = torch.cuda.current_stream().cuda_stream
f = oc.FFTCC2D(r, r); f.set_stream(stream); f.set_images(ref, tar)
g = oc.ICGN2D1(r, r, 0.001, 10.0); g.set_stream(stream); g.share_images(f); g.prepare()
pristine = torch.from_numpy(oc.make_pois2d(xs, ys)).to(dev)
q = pristine.clone()
def timed(fn, n=20):
    tot = 0.0
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize(); tot += a.elapsed_time(b)
    return tot / n
q.copy_(pristine); f.compute(q); guess = q.clone(); g.compute(q)
tf = timed(lambda: (q.copy_(pristine), f.compute(q)))
tg = timed(lambda: (q.copy_(guess), g.compute(q)))
print(json.dumps(dict(fftcc_ms=round(tf, 4), icgn_ms=round(tg, 4))))
PY
for rep in 1 2 3; do
  echo -n "base:  "; timeout 300 python /tmp/time_ab.py 2>&1 | tail -1 | tee -a $OUT/ab_$FILE.txt
  echo -n "flags: "; OPENCORR_HIP_LIB=/tmp/libab_$FILE.so timeout 300 python /tmp/time_ab.py 2>&1 | tail -1 | tee -a $OUT/ab_$FILE.txt
done
