set -x
export OPENCORR_HIP_LIB=$PWD/opencorr_amd/lib/libopencorr_hip.so
timeout 900 python -m pytest tests/test_gpu_parity_2d.py tests/test_gpu_arith_fma.py -x -q -m gpu -k "variant or table or lockstep" 2>&1 | tail -5 > gpurun_out/r6a_band_parity.txt
cat gpurun_out/r6a_band_parity.txt
timeout 300 python tools/variant_ab.py 5,9 3 8 > gpurun_out/r6a_band_ab_B.json 2>gpurun_out/r6a_band_ab_B.err; cat gpurun_out/r6a_band_ab_B.json
ENGINE=2 R=20 NS=316 timeout 300 python tools/variant_ab.py 4,9 3 8 > gpurun_out/r6a_band_ab_C.json 2>gpurun_out/r6a_band_ab_C.err; cat gpurun_out/r6a_band_ab_C.json
