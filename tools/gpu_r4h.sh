#!/bin/bash
# round 4, session h (final evidence): the GPU suite, config E's profile with the block schedule, the bench line, every config at full size
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r4h}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider 2>&1 | tail -15 > $OUT/pytest.log
tail -4 $OUT/pytest.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== profile E"
OC_PROFILE_REPS=6 bash tools/gpu_profiles.sh $TAG "E" 2>&1 | grep -v "^$" | cut -c1-300
cp $OUT/${TAG}_traffic_configE.json profiles/traffic_configE.json
echo "== bench (driver's command line)"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_n1.json 2> $OUT/bench.err
tail -c 300 $OUT/bench.err; cut -c1-300 $OUT/bench_n1.json
echo "== configs at full size"
timeout 1500 python tests/fullsize/run_configs.py --configs A,B,C,D1,E,E30,BNR,BST --out $OUT/configs.json 2>&1 | grep -v amdgpu | cut -c1-160
