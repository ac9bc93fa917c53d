#!/bin/bash
# Round-2 GPU session A: parity suite, micro-benchmarks (VALU issue, gather layouts), ICGN2D variant sweeps, bench line.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02a}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 600 2>&1 | tail -25 | tee $OUT/pytest.log
echo "== ubench"
hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_ubench.hip -o /tmp/valu_ubench && timeout 120 /tmp/valu_ubench > $OUT/valu_ubench.json; tail -c 600 $OUT/valu_ubench.json
hipcc --offload-arch=gfx950 -O3 tools/ubench/gather_ubench.hip -o /tmp/gather_ubench && timeout 120 /tmp/gather_ubench | tee $OUT/gather_ubench.txt
echo "== sweep ICGN2D1 config B"
timeout 600 python tests/fullsize/icgn_sweep.py --xcd 1 --oracle-sample 500 --out $OUT/sweep_2d1.json 2>&1 | grep -v "^{\"workload" | tail -12
echo "== sweep ICGN2D2 config C"
timeout 600 python tests/fullsize/icgn_sweep.py --engine 2 --radius 20 --pois 316 --xcd 1 --oracle-sample 300 --out $OUT/sweep_2d2.json 2>&1 | grep -v "^{\"workload" | tail -12
echo "== bench"
timeout 900 python bench.py --steps 10 --warmup 2 2>&1 | tail -3 | tee $OUT/bench.log
