#!/bin/bash
# round 4, session n: the bench line and the rocprofv3 kernel trace of the SAME command on the final tree, one box, one session
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r4n}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_n1.json 2> $OUT/bench.err
cut -c1-300 $OUT/bench_n1.json
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_n1_under_rocprof.json 2> $OUT/trace.err
python $ROOT/tools/rocpd_summary.py $(ls $OUT/trace/*.db | head -1) $OUT/bench_n1_kernel_stats.csv 2>&1 | head -8
rm -rf $OUT/trace
