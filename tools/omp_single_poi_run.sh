# Builds tests/cpp/omp_single_poi.cpp and runs it with 8 / 16 / 64 OpenMP threads (GPU box): the combining front end of oc_hip_compute_one against one
# launch per call (profiles/r6k_omp_single_poi_combining_front_end.txt).
set -e
python - <<'PY'
import struct, numpy as np, sys
sys.path.insert(0, ".")
from opencorr_amd import synth
ref, tar = synth.speckle_pair_2d(600, 640, seed=3)
h, w = ref.shape
xs, ys = synth.poi_grid_2d(h, w, 100, 100, 28)
with open("/tmp/in.bin", "wb") as f:
    f.write(struct.pack("<5i2f", h, w, 16, 16, len(xs), 0.001, 10.0))
    f.write(np.ascontiguousarray(ref, np.float32).tobytes()); f.write(np.ascontiguousarray(tar, np.float32).tobytes())
    f.write(xs.astype(np.float32).tobytes()); f.write(ys.astype(np.float32).tobytes())
PY
g++ -std=c++17 -O2 -fopenmp -Iinclude tests/cpp/omp_single_poi.cpp -o /tmp/omp_single_poi -Lopencorr_amd/lib -lopencorr_hip -Wl,-rpath,$PWD/opencorr_amd/lib -Wl,-rpath,/opt/rocm/lib
for t in 8 16 64; do OC_HIP_QUIET=1 OC_HIP_SINGLE_DEBUG=1 /tmp/omp_single_poi /tmp/in.bin /tmp/out.bin $t; done
