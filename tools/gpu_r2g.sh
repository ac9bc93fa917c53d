#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02g}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
echo "== 3D parity"; 
timeout 900 python -m pytest tests/test_gpu_parity_3d.py tests/test_cpp_shim.py tests/test_gpu_groups.py -m gpu -x -q --timeout 600 2>&1 | tail -6 | tee $OUT/pytest3d.log
echo "== timeline"
MASKS="16" bash tools/ablate_icgn3d.sh $TAG 2>&1 | tail -2
echo "== config E"
timeout 900 python tests/fullsize/run_configs.py --configs E --out $OUT/configE.json 2>&1 | tail -2
