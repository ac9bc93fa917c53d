export OPENCORR_HIP_LIB=$PWD/opencorr_amd/lib/libopencorr_hip.so
timeout 900 python -m pytest tests/test_gpu_parity_2d.py tests/test_gpu_arith_fma.py -x -q -m gpu -k "variant or table or lockstep" 2>&1 | tail -5
for a in 0 16 17 18 20 24 27; do
echo "ablate $a: $(OC_BAND_ABLATE=$a timeout 300 python tools/variant_ab.py 9 2 6 2>/dev/null | python -c 'import sys,json; print(json.load(sys.stdin)["mean_ms"])')"
done
timeout 300 python tools/variant_ab.py 5,9 3 8 2>/dev/null
ENGINE=2 R=20 NS=316 timeout 300 python tools/variant_ab.py 4,9 3 8 2>/dev/null
