#!/bin/bash
# round 4, session b: the new FFTCC kernels -- parity tests, then timings against the rocFFT pipeline
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r4b}
mkdir -p $OUT
cd $ROOT
echo "== parity: FFTCC 2D / 3D, split-merge"
timeout 1200 python -m pytest tests/test_gpu_parity_3d.py tests/test_gpu_parity_2d.py tests/test_gpu_parity_strain.py -m gpu -q --timeout 600 -p no:cacheprovider \
   -k "fftcc or split or merge or lockstep" 2>&1 | tail -40 > $OUT/pytest.log
tail -6 $OUT/pytest.log
echo "== FFTCC3D sizes"
timeout 600 python tools/fftcc3d_sizes.py 4,8,10,12,13,14,15,16,20,24,25,30,32 8 256 2>&1 | tail -1 > $OUT/fftcc3d_sizes.json
cut -c1-1500 $OUT/fftcc3d_sizes.json
echo "== FFTCC3D planes: workgroup counts at r = 30"
for b in 64 128 256 512; do
  OC_PLANES_BLOCKS=$b timeout 300 python - <<PY 2>&1 | tail -1
import os, sys, json, numpy as np, torch
sys.path.insert(0, ".")
import opencorr_amd as oc
from opencorr_amd import synth
dev = torch.device("cuda", 0)
ref, tar = synth.speckle_pair_3d(256, 256, 256, seed=20260927, device=dev)
out = {}
for r, nside in ((30, 8), (30, 18), (20, 12)):
    xs, ys, zs = synth.poi_grid_3d(256, 256, 256, nside, nside, nside, r + 8)
    f = oc.FFTCC3D(r, r, r); f.set_images(ref, tar); f.set_tuning("fftcc3d_planes_blocks", int(os.environ["OC_PLANES_BLOCKS"]))
    q0 = torch.from_numpy(oc.make_pois3d(xs, ys, zs)).to(dev); q = q0.clone(); ts = []
    for _ in range(4):
        q.copy_(q0); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f.compute(q); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
    out["r%d_n%d" % (r, len(xs))] = round(min(ts) * 1e3 / len(xs), 3)
print(json.dumps({"blocks": int(os.environ["OC_PLANES_BLOCKS"]), "us_per_poi": out}))
PY
done | tee $OUT/fftcc3d_planes_blocks.txt
echo "== FFTCC2D sizes (new sides)"
timeout 600 python tools/fftcc2d_sizes.py 4x4,5x5,6x6,7x7,11x11,13x13,14x14,17x17,19x19,21x21,22x22,23x23,26x26,27x27,28x28,29x29,31x31 2>&1 | tail -1 > $OUT/fftcc2d_sizes.json
cut -c1-1500 $OUT/fftcc2d_sizes.json
echo "== config E30 (the reference's DVC example shape)"
timeout 600 python tests/fullsize/run_configs.py --configs E30 --out $OUT/configs_E30.json 2>&1 | tail -1 | cut -c1-900
