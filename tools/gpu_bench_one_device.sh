#!/bin/bash
# the N = 2 control flow of bench.py on ONE GPU (gloo, both ranks on device 0): weak and strong mode, small workload
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
export OC_BENCH_ONE_DEVICE=1 OC_BENCH_STRICT=1
for mode in weak strong; do
  echo "== N=2 $mode"
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --scaling $mode 2>&1 | grep -v "amdgpu.ids\|^W0\|^\*\*\*\|Setting OMP" | tail -3 | cut -c1-900
done
