#!/bin/bash
# round 4, session e: the whole GPU suite, the bench line, the profiles of configs B, C, E (kernel stats + PMC traffic), all configs at full size
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r4e}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider 2>&1 | tail -15 > $OUT/pytest.log
tail -4 $OUT/pytest.log
echo "== profiles B C E"
bash tools/gpu_profiles.sh $TAG "B C E" 2>&1 | grep -v "^$" | cut -c1-300
echo "== bench (driver's command line)"
mkdir -p profiles_tmp
cp $OUT/${TAG}_traffic_configB.json profiles/traffic_configB.json 2>/dev/null
cp $OUT/${TAG}_traffic_configC.json profiles/traffic_configC.json 2>/dev/null
cp $OUT/${TAG}_traffic_configE.json profiles/traffic_configE.json 2>/dev/null
cp $OUT/${TAG}_traffic_configB_second_run.json profiles/traffic_configB_second_run.json 2>/dev/null
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_n1.json 2> $OUT/bench.err
tail -c 300 $OUT/bench.err; cut -c1-600 $OUT/bench_n1.json
echo "== configs at full size"
timeout 1200 python tests/fullsize/run_configs.py --configs A,C,D1,E,E30,BNR,BST --out $OUT/configs.json 2>&1 | grep -v amdgpu | cut -c1-200
