#!/bin/bash
# round 4, session d: packed products A/B (ICGN3D1 taps, ICGN2D polynomial), co-issue ubench, the two failing tests again
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r4d}
mkdir -p $OUT
cd $ROOT
echo "== tests"
timeout 900 python -m pytest tests/test_gpu_parity_3d.py -m gpu -q --timeout 600 -p no:cacheprovider -k "fused_cube or both_mappings" 2>&1 | tail -30 > $OUT/pytest.log
tail -5 $OUT/pytest.log
echo "== A/B ICGN3D1 taps (256^3, r = 16, 20^3 POIs)"
DIM=256 RAD=16 NS=20 bash tools/ab_run.sh $1 3d icgn3d base pk novc
echo "== A/B ICGN2D1 polynomial (config B)"
bash tools/ab_run.sh $1 2d icgn2d base pk
echo "== A/B ICGN2D2 polynomial (config C shape)"
SIDE=4096 RAD=20 NS=316 ORDER=2 bash tools/ab_run.sh ${1}_C 2d icgn2d base pk
echo "== co-issue micro-benchmark"
timeout 300 tools/ubench/coissue_ubench 2>&1 | tee $OUT/coissue_ubench.json
