#!/bin/bash
# Samples clocks and power (rocm-smi) while the ICGN2D1 A/B timing loop keeps the GPU busy.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-power}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
rocm-smi --showclocks --showpower --showtemp > $OUT/smi_idle.txt 2>&1
python tools/variant_ab.py 5 12 40 > $OUT/ab.json 2>&1 &
PID=$!
sleep 12
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower --showtemp 2>&1 | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (edge|junction|hot)" >> $OUT/smi_busy.txt
  echo "---" >> $OUT/smi_busy.txt
  sleep 1
done
wait $PID
tail -1 $OUT/ab.json | cut -c1-300
echo "== idle"; grep -E "sclk|Power" $OUT/smi_idle.txt | head -5
echo "== busy"; head -30 $OUT/smi_busy.txt
