#!/bin/bash
# Issue-level PMC passes over the ICGN3D1 kernel (dual issue, LDS pipe): bash tools/gpu_pmc3d_issue.sh <tag> [config]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-pmc3di}
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
pmc a SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU2 SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY
pmc b SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_SALU
pmc c SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_SCA
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/pmc_*/*counter_collection.csv")):
    acc = collections.defaultdict(float); n = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        if "icgn3d1_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for k in acc: print(f.split("/")[-2], k, "%.6g" % (acc[k] / max(n[k], 1)), "per launch over", n[k], "launches")
PY
