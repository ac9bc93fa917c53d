#!/bin/bash
# Round-2 GPU session D: group + host-pipeline tests first (new code), then the whole suite and the bench line.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02d}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
echo "== new tests"
timeout 600 python -m pytest tests/test_gpu_groups.py -x -q --timeout 300 2>&1 | tail -30 | tee $OUT/pytest_groups.log
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -15 | tee $OUT/pytest.log
echo "== bench"
timeout 900 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -3 | tee $OUT/bench.log
