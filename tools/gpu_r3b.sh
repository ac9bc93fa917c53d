#!/bin/bash
# round-3 check session: full GPU suite, bench line, the timing table of the reference's unmodified example main
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3b
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
export OC_EXAMPLE_OUT=$OUT
echo "== pytest -m gpu"
timeout 1200 python -m pytest tests -m gpu -q --timeout 300 2>&1 | tail -30 | tee $OUT/pytest.log
echo "== bench"
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.err
python - <<'P'
import json,os
r=json.load(open(os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r3b/bench.json")))
print("value",r["value"],"ms",r["ms_per_step"],"pcie",r["pcie_inclusive"]["ms_per_step"],r["pcie_inclusive"]["two_calls_ms_per_step"],"icgn",r["stage_ms"])
P
cat $OUT/example_test_2d_dic_fftcc_icgn1_time_mi355x.csv
