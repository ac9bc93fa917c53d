"""Runs tests/ab/ -- the bit-exactness checks of the A/B partners (icgn2d variants 0, 6, 8 -- the split launch shape -- and 9 -- the LDS-band kernel --, the ICGN3D1 row mapping, the fused 32^3 FFTCC3D kernel of rounds 1 - 5) -- in a
process of its own against the A/B build of the library (lib/ab/libopencorr_hip_ab.so, -DOC_BUILD_AB=1).  The library that
ships contains none of them and refuses their tuning values, which is asserted here as well."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ab_partners_in_the_ab_build():
    from opencorr_amd import build as hip_build
    if not os.path.exists(hip_build.AB_LIB):
        pytest.skip("the A/B build is missing (python -m opencorr_amd.build --ab)")
    env = dict(os.environ, OPENCORR_HIP_LIB=hip_build.AB_LIB, OC_AB_RUN="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "ab"), "-x", "-q", "-p", "no:cacheprovider"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "skipped" not in r.stdout.splitlines()[-1], tail


def test_the_product_library_refuses_the_ab_partners():
    import opencorr_amd
    icgn = opencorr_amd.ICGN2D1(16, 16, 0.001, 10)
    for v in (0, 6, 8, 9):
        with pytest.raises(Exception, match="A/B"):
            icgn.set_tuning("icgn2d_variant", v)
    icgn.set_tuning("icgn2d_variant", 7)
    g3 = opencorr_amd.ICGN3D1(8, 8, 8, 0.001, 20)
    with pytest.raises(Exception, match="A/B"):
        g3.set_tuning("icgn3d_mapping", 1)
    g3.set_tuning("icgn3d_mapping", 0)
    f3 = opencorr_amd.FFTCC3D(16, 16, 16)
    with pytest.raises(Exception, match="A/B"):
        f3.set_tuning("fftcc3d_fused", 2)
    f3.set_tuning("fftcc3d_fused", 1)
