#!/bin/bash
# round-4 session i: 4-wave lockstep workgroups (variant 6) against the shipped 8-wave ones (variant 5) on config B
out=gpurun_out/${1:-r4i}
mkdir -p $out
echo "== parity of the new variant"
timeout 600 python -m pytest tests/test_gpu_parity_2d.py -q -m gpu -k "variants_identical_bits or coordinate_table_variants or lockstep_barriers" 2>&1 | tail -5 | tee $out/pytest_variant6.log
echo "== A/B on config B"
timeout 600 python tools/variant_ab.py 5,6,2 4 10 2>&1 | tail -2 | tee $out/icgn2d1_variant_ab_4wave_lockstep.json
