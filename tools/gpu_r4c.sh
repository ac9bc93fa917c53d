#!/bin/bash
# round 4, session c: ICGN3D1 row mapping -- parity (both mappings), then A/B timing on config E and on the DVC example's radius
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r4c}
mkdir -p $OUT
cd $ROOT
echo "== parity 3D"
timeout 1500 python -m pytest tests/test_gpu_parity_3d.py tests/test_gpu_fuzz.py -m gpu -q --timeout 900 -p no:cacheprovider 2>&1 | tail -25 > $OUT/pytest3d.log
tail -8 $OUT/pytest3d.log
echo "== A/B Es (256^3, 12^3 POIs)"
timeout 600 python tools/icgn3d_mapping_ab.py 256 12 16 4 2>&1 | tail -1 | tee $OUT/ab_Es.json | cut -c1-700
echo "== A/B E (512^3, 37^3 POIs)"
timeout 900 python tools/icgn3d_mapping_ab.py 512 37 16 3 2>&1 | tail -1 | tee $OUT/ab_E.json | cut -c1-700
echo "== A/B E30 (256^3, r = 30, 8^3 POIs)"
timeout 600 python tools/icgn3d_mapping_ab.py 256 8 30 3 2>&1 | tail -1 | tee $OUT/ab_E30.json | cut -c1-700
echo "== A/B r = 20 (256^3, 10^3 POIs)"
timeout 600 python tools/icgn3d_mapping_ab.py 256 10 20 3 2>&1 | tail -1 | tee $OUT/ab_r20.json | cut -c1-700
echo "== co-issue micro-benchmark"
timeout 300 tools/ubench/coissue_ubench 2>&1 | tee $OUT/coissue_ubench.json
