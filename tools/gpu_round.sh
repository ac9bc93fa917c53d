#!/bin/bash
# GPU-box session: parity tests, bench line, rocprofv3 kernel-trace summary of the bench.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-round}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 900 python -m pytest tests -m gpu -x -q --timeout 120 2>&1 | tail -15 | tee $OUT/pytest.log
echo "== bench"
timeout 900 python bench.py --steps 10 --warmup 2 2>&1 | tail -3 | tee $OUT/bench.log
cd /tmp
echo "== rocprofv3 kernel trace of the bench"
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1
tail -2 $OUT/bench_under_rocprof.log
ls $OUT/prof | head
python $ROOT/tools/rocpd_summary.py $(ls $OUT/prof/*.db | head -1) $OUT/kernel_stats.csv 2>&1 | head -14
echo "== PMC passes (HBM traffic of the ICGN2D1 kernel; counters in their own runs)"
for c in fetch:FETCH_SIZE write:WRITE_SIZE; do
  name=${c%%:*}; ctr=${c##*:}
  timeout 600 rocprofv3 --pmc $ctr --kernel-include-regex "icgn2d_kernel" --output-format csv -d $OUT/pmc_$name -o $name -- \
      python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/pmc_$name.log 2>&1
  echo "pmc $name rc=$?"
done
python $ROOT/tools/pmc_traffic.py $OUT "icgn2d_kernel" $OUT/icgn2d1_hbm_traffic_configB.json --launches 5
