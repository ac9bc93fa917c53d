#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02p}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
echo "== 2D parity"
timeout 1200 python -m pytest tests/test_gpu_parity_2d.py tests/test_gpu_parity_iclm.py tests/test_gpu_fullsize.py -m gpu -x -q --timeout 600 2>&1 | tail -4 | tee $OUT/pytest2d.log
echo "== sweep ICGN2D1 config B"
timeout 600 python tests/fullsize/icgn_sweep.py --xcd 1 --oracle-sample 500 --out $OUT/sweep_2d1.json 2>&1 | grep -v "^{\"workload" | grep variant | cut -c1-110
echo "== bench"
timeout 900 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-1500 | tee $OUT/bench.log
