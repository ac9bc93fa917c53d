#!/bin/bash
# roofline.traffic of bench.py as a counter statement: PMC passes over the EXACT bench command (one counter set per
# rocprofv3 run, --pmc never combined with a trace), plus a calibration of the L2 request size on a stream of known size.
#   bash tools/gpu_traffic.sh <tag>       -> gpurun_out/<tag>/icgn2d1_traffic_configB.json  (copy to profiles/)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-traffic}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
STEPS=5; WARM=2
BENCH="python $ROOT/bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline"
cd /tmp
for c in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "tcc:TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
  name=${c%%:*}; ctr=${c#*:}
  timeout 600 rocprofv3 --pmc $ctr --kernel-include-regex "icgn2d_kernel" --output-format csv -d $OUT/pmc_$name -o $name -- $BENCH > $OUT/pmc_$name.log 2>&1
  echo "pmc $name rc=$? $(tail -c 300 $OUT/pmc_$name.log | tr '\n' ' ' | cut -c1-200)"
done
# calibration: bytes per TCC_REQ on a read-once stream of 1 GiB
if [ ! -x $ROOT/tools/ubench/l2_req_calib ]; then
  hipcc --offload-arch=gfx950 -O3 $ROOT/tools/ubench/l2_req_calib.hip -o $ROOT/tools/ubench/l2_req_calib
fi
timeout 300 rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --kernel-include-regex "stream_read" --output-format csv \
    -d $OUT/pmc_calib -o calib -- $ROOT/tools/ubench/l2_req_calib > $OUT/pmc_calib.log 2>&1
echo "pmc calib rc=$?"; tail -2 $OUT/pmc_calib.log
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "stream_read" --output-format csv \
    -d $OUT/pmc_calib_fetch -o calibf -- $ROOT/tools/ubench/l2_req_calib > $OUT/pmc_calib_fetch.log 2>&1
python $ROOT/tools/pmc_traffic.py $OUT "icgn2d_kernel" $OUT/icgn2d1_traffic_configB.json --launches $STEPS \
    --command "bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline"
