#!/bin/bash
# Round-2 GPU session C: parity suite, ICGN2D variant sweeps, bench line, config E timing (after the unpacked sweep).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02c}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 600 2>&1 | tail -25 | tee $OUT/pytest.log
echo "== sweep ICGN2D1 config B"
timeout 600 python tests/fullsize/icgn_sweep.py --xcd 1 --oracle-sample 500 --out $OUT/sweep_2d1.json 2>&1 | grep -v "^{\"workload" | tail -12
echo "== sweep ICGN2D2 config C"
timeout 600 python tests/fullsize/icgn_sweep.py --engine 2 --radius 20 --pois 316 --xcd 1 --oracle-sample 300 --out $OUT/sweep_2d2.json 2>&1 | grep -v "^{\"workload" | tail -12
echo "== bench"
timeout 900 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -3 | tee $OUT/bench.log
echo "== configs Es, BNR"
timeout 900 python tests/fullsize/run_configs.py --configs Es,BNR --out $OUT/configs.json 2>&1 | tail -4
