#!/bin/bash
# soak: the randomised differential parity tests with OC_FUZZ_EXTRA further seeds each (every float of every record against the oracle)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-soak}
mkdir -p $OUT
cd $ROOT
OC_FUZZ_EXTRA=${2:-40} timeout 2400 python -m pytest tests/test_gpu_fuzz.py -m gpu -q --timeout 900 -p no:cacheprovider 2>&1 | tail -6 | tee $OUT/fuzz_soak.log
