export OPENCORR_HIP_LIB=$PWD/opencorr_amd/lib/libopencorr_hip.so
timeout 1500 python -m pytest tests/test_gpu_parity_2d.py tests/test_gpu_arith_fma.py tests/test_gpu_ab_build.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -6
timeout 300 python tools/variant_ab.py 5,2,4 3 10 2>/dev/null
ENGINE=2 R=20 NS=316 timeout 300 python tools/variant_ab.py 4 3 8 2>/dev/null
