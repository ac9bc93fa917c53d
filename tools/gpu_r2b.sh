#!/bin/bash
# Round-2 GPU session B: VALU ubench (long form) + PMC passes over the ICGN2D1 kernel (variant 2, config B).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02b}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_ubench.hip -o /tmp/valu_ubench && timeout 300 /tmp/valu_ubench > $OUT/valu_ubench.json; tail -c 400 $OUT/valu_ubench.json
bash tools/gpu_pmc.sh $TAG/pmc_v2 2
python tools/pmc_table.py $OUT/pmc_v2 $OUT/icgn2d1_pmc_table.csv 2>&1 | tail -5
