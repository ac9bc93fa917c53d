#!/bin/bash
# One parameterised GPU-box session (round 5; replaces the per-session gpu_r*.sh scripts of rounds 3 - 4):
#   bash tools/gpu_session.sh <tag> <step> [<step> ...]          results under gpurun_out/<tag>/
# steps:
#   test:<pytest args>     python -m pytest <args> -x -q           ("+" stands for a space: test:tests/test_gpu_arith_fma.py, test:-m+gpu)
#   configs:<A,B,..>       tests/fullsize/run_configs.py --configs ... (both arithmetic modes, oracle samples)  -> configs.json
#   bench                  the driver's command line                                                              -> bench_n1.json
#   benchtrace             rocprofv3 --kernel-trace --stats of the same command                                   -> bench_n1_kernel_stats.csv
#   profiles:<"B C E">     tools/gpu_profiles.sh (kernel stats + PMC traffic per config)
#   py:<script and args>   python <script> <args> (underscores for spaces are NOT translated; quote the step)
#   sh:<command>           bash -c <command>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r5}
shift
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
for step in "$@"; do
  kind=${step%%:*}; arg=${step#*:}
  echo "== $step"
  case $kind in
    test) timeout 1500 python -m pytest ${arg//+/ } -x -q 2>&1 | grep -v amdgpu.ids | tail -15;;
    configs) timeout 2400 python tests/fullsize/run_configs.py --configs $arg --out $OUT/configs_${arg//,/_}.json 2>&1 | grep -v amdgpu | cut -c1-2500;;
    bench) timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_n1.json 2> $OUT/bench.err; cut -c1-600 $OUT/bench_n1.json; tail -3 $OUT/bench.err;;
    benchtrace)
      cd /tmp
      timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_n1_under_rocprof.json 2> $OUT/trace.err
      python $ROOT/tools/rocpd_summary.py $(ls $OUT/trace/*.db | head -1) $OUT/bench_n1_kernel_stats.csv 2>&1 | head -6
      rm -rf $OUT/trace
      cd $ROOT;;
    profiles) OC_PROFILE_REPS=${OC_PROFILE_REPS:-8} bash tools/gpu_profiles.sh $TAG "$arg" 2>&1 | grep -v "^$" | cut -c1-400;;
    py) timeout 1800 python $arg 2>&1 | grep -v amdgpu.ids | tail -40;;
    sh) timeout 1800 bash -c "$arg" 2>&1 | tail -40;;
    *) echo "unknown step $step";;
  esac
done
