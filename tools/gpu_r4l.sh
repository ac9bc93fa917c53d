#!/bin/bash
# round 4, session l (final state of the tree): the GPU suite, smoke, the bench line with the driver's command line, every config at full size
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r4l}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider 2>&1 | tail -15 > $OUT/pytest.log
tail -4 $OUT/pytest.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== bench (driver's command line)"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_n1.json 2> $OUT/bench.err
tail -c 300 $OUT/bench.err; cut -c1-400 $OUT/bench_n1.json
echo "== configs at full size"
timeout 1500 python tests/fullsize/run_configs.py --configs A,B,C,D1,E,E30,BNR,BST --out $OUT/configs.json 2>&1 | grep -v amdgpu | cut -c1-160
