#!/bin/bash
# One GPU-box session: parity tests, ICGN2D variant sweep, PMC passes over the sweep.
# Usage (from the repo root on the GPU box): bash tools/gpu_sweep.sh
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/sweep1
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
echo "== pytest -m gpu"; 
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $OUT/pytest.log
echo "== sweep ICGN2D1 config B"
timeout 600 python tests/fullsize/icgn_sweep.py --out $OUT/sweep_2d1.json 2>&1 | tail -40 | tee $OUT/sweep_2d1.log
echo "== sweep ICGN2D2 config C"
timeout 600 python tests/fullsize/icgn_sweep.py --engine 2 --radius 20 --pois 316 --out $OUT/sweep_2d2.json 2>&1 | tail -30 | tee $OUT/sweep_2d2.log
cd /tmp
pmc() {  # name, counters...
  name=$1; shift
  timeout 600 rocprofv3 --pmc "$@" --kernel-include-regex "icgn2d_kernel" --output-format csv -d $OUT/pmc_$name -o $name -- \
      python $ROOT/tests/fullsize/icgn_sweep.py --launches 1 --oracle-sample 200 > $OUT/pmc_$name.log 2>&1
  echo "pmc $name rc=$?"
}
pmc sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
pmc sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
pmc tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
pmc fetch FETCH_SIZE TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_TA_BUSY_sum
pmc write WRITE_SIZE TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum
find $OUT -name "*.csv" | head -30
du -sh $OUT
