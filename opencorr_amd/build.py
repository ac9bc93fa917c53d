"""Builds opencorr_amd/lib/libopencorr_hip.so (gfx950 only) with hipcc.

    python -m opencorr_amd.build [--force] [--ab]

--ab additionally builds lib/ab/libopencorr_hip_ab.so: the SAME library plus the measured losers that are kept as A/B
partners (-DOC_BUILD_AB=1: icgn2d variants 0 and 6, the LDS-band kernel icgn2d_band.hip (variant 9), the ICGN3D1 row mapping
icgn3d_rows.hip, the experiment environment knobs).  Test / experiment infrastructure: tests/ab/ and the tools/*_probe.py scripts load it through OPENCORR_HIP_LIB; the
library that ships does not contain any of it.

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
SOURCES = ["capi.hip", "capi_host.hip", "capi_group.hip", "capi_strain.hip", "prepare2d.hip", "icgn2d.hip", "nr2d.hip", "poi_order.hip", "poi_split.hip", "strain.hip", "fftcc2d.hip", "fftcc2d_fused.hip", "fftcc2d_fusedn.hip", "fftcc2d_fusedp.hip", "fftcc2d_fusedr.hip", "fftcc2d_rect.hip", "prepare3d.hip", "icgn3d.hip", "fftcc3d.hip", "fftcc3d_fused.hip", "fftcc3d_fusedn.hip", "fftcc3d_box.hip", "fftcc3d_planes.hip", "fftcc3d_planesb.hip"]
# the A/B build: sources that exist only there, and the product sources whose code depends on OC_BUILD_AB (recompiled with
# -DOC_BUILD_AB=1; every other object is shared with the product build)
AB_ONLY_SOURCES = ["icgn3d_rows.hip", "icgn2d_band.hip", "fftcc3d_fused_r5.hip"]
AB_DEPENDENT = ["capi.hip", "icgn2d.hip", "icgn3d.hip"]
AB_LIBDIR = os.path.join(LIBDIR, "ab")
AB_LIB = os.path.join(AB_LIBDIR, "libopencorr_hip_ab.so")
HEADERS = ["capi_internal.h", "oc_device.h", "oc_kernels.h", "dic2d_device.h", "fft_device.h", "fftcc2d_fusedn_impl.h", "fftcc3d_planes_impl.h", "icgn3d_device.h", os.path.join("..", "..", "include", "opencorr_hip.h")]
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
FMA_SOURCES = ["icgn2d.hip", "icgn2d_band.hip", "icgn3d.hip"]


# Sources a profiled kernel is built from (kernel-name regex of tools/gpu_profiles.sh -> files under csrc/): the PMC records under
# profiles/ carry kernel_fingerprint() of the day they were collected, and bench.py refuses a record whose fingerprint is not
# the current tree's (a kernel edit must not keep yesterday's counters: VERDICT r5 weak 10)
KERNEL_SOURCES = {
    "icgn2d_kernel": ["icgn2d.hip", "dic2d_device.h", "oc_device.h"],
    "fftcc2d_fused32x2_kernel": ["fftcc2d_fused.hip", "fft_device.h", "oc_device.h"],
    "fftcc2d_fusedn_kernel": ["fftcc2d_fusedn_impl.h", "fftcc2d_fusedn.hip", "fftcc2d_fusedp.hip", "fftcc2d_fusedr.hip", "fft_device.h", "oc_device.h"],
    "icgn3d1": ["icgn3d.hip", "icgn3d_device.h", "oc_device.h"],
    "fftcc3d_fused32_kernel": ["fftcc3d_fused.hip", "fft_device.h", "oc_device.h"],
    "fftcc3d_planes_kernel": ["fftcc3d_planes_impl.h", "fftcc3d_planes.hip", "fftcc3d_planesb.hip", "fft_device.h", "oc_device.h"],
}


def kernel_fingerprint(kernel_regex):
    """sha256 over the sources of the kernel family `kernel_regex` names (KERNEL_SOURCES; an unknown name: every file of csrc/)
    and the compiler flags -- what decides the instruction stream the counters were measured on."""
    import hashlib
    files = KERNEL_SOURCES.get(kernel_regex)
    if files is None:
        files = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    h = hashlib.sha256()
    h.update((" ".join(FLAGS) + " " + ARCH).encode())
    for f in files:
        h.update(f.encode())
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


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


def _compile_and_link(units, lib, force, verbose, shared_objs=()):
    """units: (source, object path, extra defines).  Compiles what is stale (in parallel) and links `lib`."""
    headers = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    cc = hipcc()
    objs, procs = list(shared_objs), []
    for src, o, defs in units:
        s = os.path.join(CSRC, src)
        objs.append(o)
        if force or _stale(o, [s] + headers):
            cmd = [cc, "--offload-arch=" + ARCH, "-c", s, "-o", o] + FLAGS + EXTRA_FLAGS.get(src, []) + defs
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on " + src)
    if force or procs or _stale(lib, objs):
        cmd = [cc, "--offload-arch=" + ARCH, "-shared", "-o", lib] + objs + ["-L/opt/rocm/lib", "-lrocfft", "-ldl", "-lpthread"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return lib


def _units(sources, libdir, defs):
    units = [(src, os.path.join(libdir, src.replace(".hip", ".o")), list(defs)) for src in sources]
    units += [(src, os.path.join(libdir, src.replace(".hip", "_fma.o")), list(defs) + ["-DOC_FMA=1"]) for src in sources if src in FMA_SOURCES]
    return units


def build(force=False, verbose=True, ab=False):
    """The product library; ab=True: also the A/B build (see the module docstring).  Returns the product library's path."""
    os.makedirs(LIBDIR, exist_ok=True)
    _compile_and_link(_units(SOURCES, LIBDIR, []), LIB, force, verbose)
    if ab:
        build_ab(force=force, verbose=verbose)
    return LIB


def build_ab(force=False, verbose=True):
    """lib/ab/libopencorr_hip_ab.so = the product objects that do not depend on OC_BUILD_AB + the dependent and A/B-only sources
    compiled with -DOC_BUILD_AB=1.  Needs the product build's objects (build() first)."""
    os.makedirs(AB_LIBDIR, exist_ok=True)
    shared = [o for src, o, _ in _units([s for s in SOURCES if s not in AB_DEPENDENT], LIBDIR, [])]
    missing = [o for o in shared if not os.path.exists(o)]
    if missing:
        raise RuntimeError("build_ab: the product build comes first (missing %s)" % ", ".join(os.path.basename(m) for m in missing))
    return _compile_and_link(_units(AB_DEPENDENT + AB_ONLY_SOURCES, AB_LIBDIR, ["-DOC_BUILD_AB=1"]), AB_LIB, force, verbose, shared_objs=shared)


if __name__ == "__main__":
    build(force="--force" in sys.argv, ab="--ab" in sys.argv)
    print(LIB)
    if "--ab" in sys.argv:
        print(AB_LIB)
