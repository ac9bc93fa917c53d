#!/bin/bash
# Every roofline fraction of the bench line from ONE script (round 4; tools/gpu_traffic.sh generalised): for configs B (bench.py),
# C and E -- a rocprofv3 kernel trace (--kernel-trace --stats) and, in SEPARATE runs (one counter set each, --pmc never combined
# with a trace), the PMC passes FETCH_SIZE / WRITE_SIZE / TCC_REQ+HIT+MISS, reduced per dominant kernel to
#     profiles/<tag>_traffic_config{B,C,E}.json      (HBM bytes = FETCH_SIZE x 2 + WRITE_SIZE; L2-side bytes = TCC_REQ x calibrated size)
#     profiles/<tag>_config{B,C,E}_kernel_stats.csv
# which bench.py reads for roofline.traffic and roofline_secondary[*].traffic.  `spread` runs config B's HBM passes a second time
# and records both values: the run-to-run spread of the counter (2.29 vs 2.60 GB were seen in round 3).
#   bash tools/gpu_profiles.sh <tag> [configs="B C E"]        (on the GPU box; results under gpurun_out/<tag>/)
# OC_BENCH_ARITH_FMA=1 (round 5): the same collection with the solvers under the fused arithmetic contract (bench.py and
# tools/run_config_kernels.py read the variable); files get the suffix _fma.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r4p}
CONFIGS=${2:-"B C E"}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPS=${OC_PROFILE_REPS:-10}   # launches averaged per kernel (the first ones -- warm-up -- are dropped)
SUF=""; [ "$OC_BENCH_ARITH_FMA" = "1" ] && SUF="_fma"
cd /tmp
if [ ! -x $ROOT/tools/ubench/l2_req_calib ]; then
  hipcc --offload-arch=gfx950 -O3 $ROOT/tools/ubench/l2_req_calib.hip -o $ROOT/tools/ubench/l2_req_calib
fi
# calibration of the L2 request size and of the FETCH_SIZE correction on a read-once stream of 1 GiB
timeout 300 rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --kernel-include-regex "stream_read" --output-format csv \
    -d $OUT/pmc_calib -o calib -- $ROOT/tools/ubench/l2_req_calib > $OUT/pmc_calib.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "stream_read" --output-format csv \
    -d $OUT/pmc_calib_fetch -o calibf -- $ROOT/tools/ubench/l2_req_calib > $OUT/pmc_calib_fetch.log 2>&1
for cfg in $CONFIGS; do
  case $cfg in
    B) CMD="python $ROOT/bench.py --steps $REPS --warmup 5 --no-cpu-baseline"; KERNELS="icgn2d_kernel fftcc2d_fused32x2_kernel";;
    C) CMD="python $ROOT/tools/run_config_kernels.py C --reps $REPS --warm 3"; KERNELS="icgn2d_kernel fftcc2d_fusedn_kernel";;
    E) CMD="python $ROOT/tools/run_config_kernels.py E --reps $REPS --warm 2"; KERNELS="icgn3d1 fftcc3d_fused32_kernel";;
    E30) CMD="python $ROOT/tools/run_config_kernels.py E30 --reps $REPS --warm 2"; KERNELS="icgn3d1 fftcc3d_planes_kernel";;
  esac
  D=$OUT/cfg$cfg$SUF
  mkdir -p $D
  REGEX=$(echo $KERNELS | tr ' ' '|')
  echo "== config $cfg: kernel trace"
  timeout 900 rocprofv3 --kernel-trace --stats -d $D/trace -o trace -- $CMD > $D/trace.log 2>&1
  python $ROOT/tools/rocpd_summary.py $(ls $D/trace/*.db | head -1) $OUT/${TAG}_config${cfg}${SUF}_kernel_stats.csv 2>&1 | head -6
  for c in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "tcc:TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" \
           "sq:SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU2 SQ_LDS_BANK_CONFLICT"; do
    name=${c%%:*}; ctr=${c#*:}
    # OC_PROFILE_PASSES="sq" (or "fetch write", ...): only these counter sets (default: all four)
    if [ -n "$OC_PROFILE_PASSES" ] && ! echo " $OC_PROFILE_PASSES " | grep -q " $name "; then continue; fi
    timeout 900 rocprofv3 --pmc $ctr --kernel-include-regex "$REGEX" --output-format csv -d $D/pmc_$name -o $name -- $CMD > $D/pmc_$name.log 2>&1
    echo "config $cfg pmc $name rc=$?"
  done
  ln -sfn $OUT/pmc_calib $D/pmc_calib; ln -sfn $OUT/pmc_calib_fetch $D/pmc_calib_fetch
  python $ROOT/tools/pmc_traffic.py $D "$REGEX" $OUT/${TAG}_traffic_config${cfg}${SUF}.json --launches $REPS --kernels $KERNELS \
      --stats $OUT/${TAG}_config${cfg}${SUF}_kernel_stats.csv --command "$(echo $CMD | sed "s#$ROOT/##g")" | cut -c1-400
  if [ "$cfg" = "B" ] && [ -z "$SUF" ] && [ -z "$OC_PROFILE_NO_SPREAD" ]; then
    # the HBM passes once more: run-to-run spread of FETCH_SIZE / WRITE_SIZE on the same command
    mkdir -p $D/again
    for c in "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
      name=${c%%:*}; ctr=${c#*:}
      timeout 900 rocprofv3 --pmc $ctr --kernel-include-regex "$REGEX" --output-format csv -d $D/again/pmc_$name -o $name -- $CMD > $D/again/pmc_$name.log 2>&1
    done
    python $ROOT/tools/pmc_traffic.py $D/again "$REGEX" $OUT/${TAG}_traffic_configB_second_run.json --launches $REPS --kernels $KERNELS \
        --command "$(echo $CMD | sed "s#$ROOT/##g") (second collection: spread of the HBM counters)" | cut -c1-300
  fi
  rm -rf $D/trace
done
