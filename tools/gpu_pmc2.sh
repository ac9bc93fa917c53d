#!/bin/bash
# Second-level PMC passes for icgn2d_kernel (issue / fetch / TA fifo counters): bash tools/gpu_pmc2.sh <tag> <variant>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-pmc2}
VARS=${2:-5}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
pmc() {
  name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-include-regex "icgn2d_kernel" --output-format csv -d $OUT/pmc_$name -o $name -- \
      python $ROOT/tests/fullsize/icgn_sweep.py --launches 1 --oracle-sample 200 --variants $VARS --xcd 1 > $OUT/pmc_$name.log 2>&1
  echo "pmc $name rc=$?"
}
pmc a SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
pmc b SQ_IFETCH SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_SALU
pmc c SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_LDS
pmc d SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/pmc_*/*counter_collection.csv")):
    acc = collections.defaultdict(float); n = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        if "icgn2d_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for k in acc: print(f.split("/")[-2], k, "%.6g" % acc[k], "rows", n[k])
PY
