"""The register FFTs of opencorr_amd/csrc/fft_device.h on the HOST (no GPU): the mixed-radix transforms are plain arithmetic
on register arrays and compile as host code too; tests/cpp/fft_host_check.hip runs every window side the fused FFTCC2D
kernels are instantiated for (every even side from 8 to 64 -- factors 2, 3, 4, 5 and the generic odd-prime butterfly for
7 ... 31) plus the prime and odd sizes on their own, forward and inverse, against a double-precision DFT.  What the GPU tests
then add is the kernel around the transforms (gather, transposes, spectrum product, arg-max), compared with the oracle and
the rocFFT pipeline."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_mixed_radix_ffts_match_a_double_precision_dft(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not found")
    exe = str(tmp_path / "fft_host_check")
    src = os.path.join(ROOT, "tests", "cpp", "fft_host_check.hip")
    # -ffp-contract=off like the library (the butterflies opt back in with their own pragma)
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O1", "-std=c++17", "-ffp-contract=off", src, "-o", exe], check=True,
                   cwd=str(tmp_path), timeout=900)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    rows = [ln.split() for ln in out.stdout.splitlines() if ln.strip()]
    assert out.returncode == 0, out.stdout
    sizes = {int(r[0]) for r in rows}
    assert set(range(8, 65, 2)) <= sizes and {7, 11, 13, 17, 19, 23, 29, 31} <= sizes
    assert all(r[3] == "ok" for r in rows)
    assert max(float(r[1]) for r in rows) < 5e-7 and max(float(r[2]) for r in rows) < 5e-7
