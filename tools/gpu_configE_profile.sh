#!/bin/bash
# 3D parity tests + rocprofv3 kernel trace of BASELINE config E (prepare kernels, FFTCC3D, ICGN3D1): bash tools/gpu_configE_profile.sh <tag>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-configE}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
echo "== 3D parity"
timeout 900 python -m pytest tests/test_gpu_parity_3d.py -m gpu -x -q --timeout 600 2>&1 | tail -4 | tee $OUT/pytest3d.log
cd /tmp
echo "== rocprofv3 kernel trace of config E"
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof -o cfgE -- python $ROOT/tests/fullsize/run_configs.py --configs E > $OUT/configE.log 2>&1
tail -1 $OUT/configE.log | cut -c1-400
python $ROOT/tools/rocpd_summary.py $(ls $OUT/prof/*.db | head -1) $OUT/kernel_stats_configE.csv 2>&1 | head -12
