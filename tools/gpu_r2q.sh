#!/bin/bash
# final-state evidence: full suite, bench (all side fields), rocprofv3 kernel trace + HBM PMC of the bench, strong-scaling mode at N = 1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02q}
bash $ROOT/tools/gpu_round.sh $TAG
cd $ROOT
OUT=$ROOT/gpurun_out/$TAG
echo "== bench --scaling strong (config D on one GPU)"
timeout 600 python bench.py --scaling strong --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-700 | tee $OUT/bench_strong_n1.json
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
