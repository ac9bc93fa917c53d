#!/bin/bash
# PMC passes over the ICGN3D1 kernel (tests/fullsize/run_configs.py --configs Es): bash tools/gpu_pmc3d.sh <tag>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-pmc3d}
CFG=${2:-Es}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
pmc() {
  name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-include-regex "icgn3d1_kernel" --output-format csv -d $OUT/pmc_$name -o $name -- \
      python $ROOT/tests/fullsize/run_configs.py --configs $CFG > $OUT/pmc_$name.log 2>&1
  echo "pmc $name rc=$?"
}
pmc sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
pmc sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
pmc tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
pmc fetch FETCH_SIZE TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_TA_BUSY_sum
pmc write WRITE_SIZE TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
