#!/bin/bash
# round 4, session f: kernel stats + PMC traffic of configs B, C, E, E30 (tools/gpu_profiles.sh), then the bench line with those records in place
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r4f}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
bash tools/gpu_profiles.sh $TAG "B C E E30" 2>&1 | grep -v "^$" | cut -c1-400
for c in B C E; do cp $OUT/${TAG}_traffic_config$c.json profiles/traffic_config$c.json 2>/dev/null; done
cp $OUT/${TAG}_traffic_configB_second_run.json profiles/traffic_configB_second_run.json 2>/dev/null
echo "== bench (driver's command line)"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_n1.json 2> $OUT/bench.err
tail -c 300 $OUT/bench.err; cut -c1-300 $OUT/bench_n1.json
echo "== rocprofv3 kernel trace of the driver's bench command"
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
python $ROOT/tools/rocpd_summary.py $(ls $OUT/prof/*.db | head -1) $OUT/${TAG}_bench_n1_kernel_stats.csv 2>&1 | head -6
rm -rf $OUT/prof $OUT/cfg*/pmc_*/*.db
