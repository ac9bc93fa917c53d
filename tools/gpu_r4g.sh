#!/bin/bash
# round 4, session g: ICGN3D1 block schedule -- parity, A/B on config E / Es / r = 30, PMC traffic with the schedule on
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r4g}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
echo "== parity"
timeout 900 python -m pytest tests/test_gpu_parity_3d.py -m gpu -q --timeout 600 -p no:cacheprovider -k "icgn3d1 or chain" 2>&1 | tail -12 > $OUT/pytest.log
tail -4 $OUT/pytest.log
echo "== A/B E"
timeout 900 python tools/icgn3d_tile_ab.py 512 37 16 0,24,32,48,64,96 3 2>&1 | tail -1 | tee $OUT/tile_ab_E.json | cut -c1-900
echo "== A/B Es"
timeout 600 python tools/icgn3d_tile_ab.py 256 20 16 0,32,48,64 3 2>&1 | tail -1 | tee $OUT/tile_ab_Es.json | cut -c1-700
echo "== A/B r = 30, 18^3 POIs on 256^3"
timeout 600 python tools/icgn3d_tile_ab.py 256 18 30 0,48,64,96 2 2>&1 | tail -1 | tee $OUT/tile_ab_r30.json | cut -c1-700
echo "== PMC traffic of config E with the schedule"
OC_PROFILE_REPS=4 bash tools/gpu_profiles.sh $TAG "E" 2>&1 | grep -v "^$" | cut -c1-300
