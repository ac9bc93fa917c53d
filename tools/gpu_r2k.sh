#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02k}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
echo "== timeline config E shape"
DIM=512 NS=37 MASKS="16" bash tools/ablate_icgn3d.sh $TAG 2>&1 | tail -1
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -5 | tee $OUT/pytest.log
echo "== bench (all side measurements)"
timeout 900 python bench.py --steps 10 --warmup 2 2>&1 | tail -1 | tee $OUT/bench.json
