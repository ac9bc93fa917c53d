#!/bin/bash
# round 4, session t (after the FFTCC3D barrier work): config E's kernel trace + PMC traffic, the bench line + its kernel trace, configs at full size
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r4t}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
echo "== profile E"
OC_PROFILE_REPS=6 bash tools/gpu_profiles.sh $TAG "E" 2>&1 | grep -v "^$" | cut -c1-300
echo "== bench (driver's command line) + kernel trace of the same command"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_n1.json 2> $OUT/bench.err
cut -c1-300 $OUT/bench_n1.json
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_n1_under_rocprof.json 2> $OUT/trace.err
python $ROOT/tools/rocpd_summary.py $(ls $OUT/trace/*.db | head -1) $OUT/bench_n1_kernel_stats.csv 2>&1 | head -4
rm -rf $OUT/trace
cd $ROOT
echo "== configs at full size"
timeout 1500 python tests/fullsize/run_configs.py --configs A,B,C,D1,E,E30,BNR,BST --out $OUT/configs.json 2>&1 | grep -v amdgpu | cut -c1-160
