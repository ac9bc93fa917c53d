#!/bin/bash
# Runs prebuilt library variants (tools/ab_build.py) on the GPU box: every variant on ONE synthetic pair, timed, compared BIT FOR BIT
# with the first.   bash tools/ab_run.sh <tag> 2d|3d <src> name1 name2 ...      (env: SIDE RAD NS ORDER / DIM RAD NS as in ab_icgn2d.sh / ab_icgn3d.sh)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; KIND=$2; SRC=$3; shift 3
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
# the timing scripts live inside the A/B build scripts: extract them without running a build
if [ "$KIND" = "2d" ]; then
  sed -n "/^cat > \/tmp\/time2d_ab.py <<'PY'$/,/^PY$/p" tools/ab_icgn2d.sh | sed '1d;$d' > /tmp/time_ab.py
else
  sed -n "/^cat > \/tmp\/time3d_ab.py <<'PY'$/,/^PY$/p" tools/ab_icgn3d.sh | sed '1d;$d' > /tmp/time_ab.py
fi
export AB_PAIR_TAG=${TAG}_${KIND}_${SRC}
first=""
for rep in 1 2; do
 for name in "$@"; do
  [ -z "$first" ] && first=/tmp/res_${SRC}_$name.npy
  echo -n "$name (run $rep): " | tee -a $OUT/ab_${SRC}.txt
  OPENCORR_HIP_LIB=$ROOT/tools/ab_build/libab_${SRC}_$name.so timeout 600 python /tmp/time_ab.py /tmp/res_${SRC}_$name.npy $first 2>&1 | tail -1 | tee -a $OUT/ab_${SRC}.txt
 done
done
