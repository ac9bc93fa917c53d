#!/bin/bash
# round-3 session A: parity tests (incl. configs D, E, RCCL one-rank, stream switches), bench line, PMC traffic record
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3a
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 1200 python -m pytest tests -m gpu -q --timeout 300 2>&1 | tail -40 | tee $OUT/pytest.log
echo "== bench"
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
tail -c 1500 $OUT/bench.err; cut -c1-3000 $OUT/bench.json
echo "== traffic"
echo skipped
