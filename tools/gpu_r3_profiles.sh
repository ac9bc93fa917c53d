#!/bin/bash
# round-3 evidence session: bench line, rocprofv3 kernel-trace summary of the same command, PMC traffic record, all five
# BASELINE configs at full size, kernel trace of config E.   bash tools/gpu_r3_profiles.sh <tag>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r3p}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
echo "== bench (driver's command line)"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_n1.json 2> $OUT/bench.err
tail -c 300 $OUT/bench.err
cd /tmp
echo "== rocprofv3 kernel trace of the bench"
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
python $ROOT/tools/rocpd_summary.py $(ls $OUT/prof/*.db | head -1) $OUT/bench_n1_kernel_stats.csv 2>&1 | head -10
cd $ROOT
echo "== PMC traffic"
bash tools/gpu_traffic.sh ${TAG}_traffic 2>&1 | tail -3 | cut -c1-600
cp $ROOT/gpurun_out/${TAG}_traffic/icgn2d1_traffic_configB.json $OUT/ 2>/dev/null
cp $ROOT/gpurun_out/${TAG}_traffic/pmc_tcc/tcc_counter_collection.csv $OUT/icgn2d1_pmc_tcc.csv 2>/dev/null
echo "== configs at full size"
timeout 900 python tests/fullsize/run_configs.py --configs A,C,D1,E,E30,BNR,BST --out $OUT/configs.json 2>&1 | grep -v amdgpu | cut -c1-260
cd /tmp
echo "== rocprofv3 kernel trace of config E"
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/profE -o cfgE -- python $ROOT/tests/fullsize/run_configs.py --configs E > $OUT/configE.log 2>&1
python $ROOT/tools/rocpd_summary.py $(ls $OUT/profE/*.db | head -1) $OUT/configE_kernel_stats.csv 2>&1 | head -8
rm -rf $OUT/prof $OUT/profE
