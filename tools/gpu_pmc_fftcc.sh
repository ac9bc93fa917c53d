#!/bin/bash
# Issue-level PMC passes for the 32 x 32 FFTCC2D kernel over a short bench run (one counter set per rocprofv3 run, --pmc never
# combined with a trace): bash tools/gpu_pmc_fftcc.sh <tag>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-pmc_fftcc}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
pmc() {
  name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-include-regex "fftcc2d_fused32" --output-format csv -d $OUT/pmc_$name -o $name -- \
      python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_$name.log 2>&1
  echo "pmc $name rc=$?"
}
pmc a SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU2
pmc b SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM
pmc c SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/pmc_*/*counter_collection.csv")):
    acc = collections.defaultdict(float); n = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        if "fftcc2d_fused32" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for k in acc: print(f.split("/")[-2], k, "per launch %.6g" % (acc[k] / max(n[k], 1)), "launches", n[k])
PY
