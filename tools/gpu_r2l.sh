#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02l}
bash $ROOT/tools/gpu_round.sh $TAG
cd $ROOT
OUT=$ROOT/gpurun_out/$TAG
echo "== configs A C D1 E E30 BNR BST"
timeout 900 python tests/fullsize/run_configs.py --configs A,C,D1,E,E30,BNR,BST --out $OUT/configs.json 2>&1 | grep -v amdgpu.ids | cut -c1-420
