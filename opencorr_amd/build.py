"""Builds opencorr_amd/lib/libopencorr_hip.so (gfx950 only) with hipcc.

    python -m opencorr_amd.build [--force]

Flags that matter for parity: -ffp-contract=off (no FMA
contraction; every multiply and add rounds separately, like the oracle) and no
fast-math; hipcc's default correctly-rounded fp32 divide/sqrt is kept.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libopencorr_hip.so")
SOURCES = ["capi.hip", "prepare2d.hip", "icgn2d.hip", "nr2d.hip", "poi_order.hip", "poi_split.hip", "strain.hip", "fftcc2d.hip", "fftcc2d_fused.hip", "fftcc2d_fusedn.hip", "fftcc2d_fusedp.hip", "fftcc2d_fusedr.hip", "prepare3d.hip", "icgn3d.hip", "icgn3d_rows.hip", "fftcc3d.hip", "fftcc3d_fused.hip", "fftcc3d_fusedn.hip", "fftcc3d_planes.hip", "fftcc3d_planesb.hip"]
HEADERS = ["oc_device.h", "oc_kernels.h", "dic2d_device.h", "fft_device.h", "fftcc2d_fusedn_impl.h", "fftcc3d_planes_impl.h", "icgn3d_device.h", os.path.join("..", "..", "include", "opencorr_hip.h")]
ARCH = "gfx950"
# -fno-slp-vectorize: the SLP vectoriser pairs independent scalar fp32 operations into v_pk_mul_f32 / v_pk_add_f32.  On gfx950 a
# packed op occupies a SIMD for 4.3 cycles against 2.4 for the plain one (profiles/r02b_valu_ubench.json) -- a 10 % gain that the
# v_mov_b32 forming the register pairs turn into a loss (ICGN2D sweep: 48 moves per 3 samples, 17 % more issue cycles).
# Code that wants packed arithmetic says so with float2 vector types.
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function"]
# per-file additions (none at present; round 4 built icgn3d*.hip with `-mllvm -disable-vector-combine` for the packed tap
# products of OC_TAPS_PACKED=1 -- measured slower, see icgn3d_device.h)
EXTRA_FLAGS = {}
# the solver files that exist in two arithmetic modes (csrc/oc_device.h): compiled a second time with -DOC_FMA=1 into
# <name>_fma.o (kernels in ochip::fma; oc_hip_set_tuning("arith_fma", 1) launches them)
FMA_SOURCES = ["icgn2d.hip", "icgn3d.hip"]


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    headers = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    objs = []
    cc = hipcc()
    procs = []
    units = [(src, src.replace(".hip", ".o"), []) for src in SOURCES]
    units += [(src, src.replace(".hip", "_fma.o"), ["-DOC_FMA=1"]) for src in FMA_SOURCES]
    for src, obj, defs in units:
        s = os.path.join(CSRC, src)
        o = os.path.join(LIBDIR, obj)
        objs.append(o)
        if force or _stale(o, [s] + headers):
            cmd = [cc, "--offload-arch=" + ARCH, "-c", s, "-o", o] + FLAGS + EXTRA_FLAGS.get(src, []) + defs
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on " + src)
    if force or procs or _stale(LIB, objs):
        cmd = [cc, "--offload-arch=" + ARCH, "-shared", "-o", LIB] + objs + ["-L/opt/rocm/lib", "-lrocfft", "-ldl", "-lpthread"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
