"""ctypes binding of ``oracle/_ref/liboc_ref.so``: the REFERENCE's own hot-path sources, compiled unmodified.

TEST INFRASTRUCTURE ONLY (same rule as ``oracle/__init__.py``).  The library is built by ``make -C oracle ref`` from
``/root/reference/src/*.cpp`` against the stand-in Eigen / FFTW / OpenCV headers of ``oracle/ref_stubs`` and exists
only where the reference tree is mounted (this container; the GPU box receives the prebuilt file with the snapshot
but no test there depends on it).  ``available()`` says whether it can be used; tests skip otherwise.

What it pins: the oracle's reading of the reference's LOOPS -- guards, window fills, summation order of the hand
written reductions, the steepest-descent / Hessian / numerator loops, warp composition, output and error-code logic --
bit for bit (``tests/test_oracle_vs_ref.py``).  What it cannot pin: the arithmetic INSIDE Eigen and FFTW (inverse(),
products, mean(), squaredNorm(), the FFT butterflies), which the stand-ins restate like the oracle does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_ref", "liboc_ref.so")
REFERENCE_ROOT = os.environ.get("OC_REFERENCE_ROOT", "/root/reference")

ICGN2D1, ICGN2D2, ICLM2D1, ICLM2D2, NR2D1 = 0, 1, 2, 3, 4

_lib = None


def build(force=False):
    """``make -C oracle ref`` when the reference tree is present; returns the library path or None."""
    if not os.path.isdir(os.path.join(REFERENCE_ROOT, "src")):
        return _LIB_PATH if os.path.exists(_LIB_PATH) else None
    cmd = ["make", "-C", _HERE, "ref", "REF=" + REFERENCE_ROOT] + (["-B"] if force else ["-s"])
    subprocess.check_call(cmd)
    return _LIB_PATH


def build_examples(force=False):
    """``make -C oracle examples``: the reference's own example mains compiled unmodified against the drop-in headers and
    linked with libopencorr_hip.so (tests/test_gpu_reference_examples.py runs them on the GPU box).  Needs the reference
    tree and the built HIP library; returns the list of binaries that exist afterwards."""
    if os.path.isdir(os.path.join(REFERENCE_ROOT, "examples")):
        cmd = ["make", "-C", _HERE, "examples", "REF=" + REFERENCE_ROOT] + (["-B"] if force else ["-s"])
        subprocess.check_call(cmd)
    out = os.path.join(_HERE, "_ref")
    return sorted(os.path.join(out, f) for f in os.listdir(out) if f.startswith("example_")) if os.path.isdir(out) else []


def available():
    try:
        return lib() is not None
    except Exception:
        return False


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH) and build() is None:
            return None
        L = ctypes.CDLL(_LIB_PATH)
        fp = ctypes.POINTER(ctypes.c_float)
        i, f, l = ctypes.c_int, ctypes.c_float, ctypes.c_long
        L.oc_ref_fftcc2d.argtypes = [fp, fp, i, i, i, i, fp, l, i]
        L.oc_ref_solve2d.argtypes = [i, fp, fp, i, i, i, i, f, f, fp, l, fp, i, fp, i]
        L.oc_ref_prepare2d.argtypes = [fp, fp, i, i, fp, fp, fp]
        try:
            dp = ctypes.POINTER(ctypes.c_double)
            L.oc_ref_time_icgn2d1.argtypes = [fp, fp, i, i, i, i, f, f, fp, l, i, i, dp, dp]
        except AttributeError:
            pass   # a library built before round 5
        L.oc_ref_bspline2d_eval.argtypes = [fp, i, i, fp, l, fp]
        L.oc_ref_fftcc3d.argtypes = [fp, fp, i, i, i, i, i, i, fp, l, i]
        L.oc_ref_icgn3d1.argtypes = [fp, fp, i, i, i, i, i, i, f, f, fp, l, i]
        L.oc_ref_prepare3d.argtypes = [fp, fp, i, i, i, fp, fp, fp, fp, l, fp]
        if hasattr(L, "oc_ref_epipolar_search"):
            L.oc_ref_epipolar_search.argtypes = [fp, fp, i, i, fp, fp, fp, fp, i, i, fp, fp, i, i, f, f, fp, l, i, fp]
        L.oc_ref_strain.argtypes = [i, fp, l, f, i, f, i, i]
        L.oc_ref_region_fit.argtypes = [i, fp, l, fp, l, f, i, i]
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float)) if a is not None else None


def _img(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("reference %s failed (rc %d: the reference threw std::string)" % (what, rc))


def fftcc2d(ref, tar, rx, ry, pois, threads=0):
    ref, tar = _img(ref), _img(tar)
    h, w = ref.shape
    _check(lib().oc_ref_fftcc2d(_fp(ref), _fp(tar), h, w, rx, ry, _fp(pois), pois.shape[0], threads), "FFTCC2D")


def solve2d(engine, ref, tar, rx, ry, conv, stop, pois, center_offsets=None, self_adaptive=False, damping=None, threads=0):
    """prepare() + compute(poi_queue[, center_offset_queue]) of ICGN2D1 / ICGN2D2 / ICLM2D1 / ICLM2D2 / NR2D1, in place."""
    ref, tar = _img(ref), _img(tar)
    h, w = ref.shape
    off = None if center_offsets is None else np.ascontiguousarray(center_offsets, dtype=np.float32)
    dmp = None if damping is None else np.asarray(damping, dtype=np.float32)
    _check(lib().oc_ref_solve2d(engine, _fp(ref), _fp(tar), h, w, rx, ry, float(conv), float(stop), _fp(pois), pois.shape[0],
                                _fp(off), 1 if self_adaptive else 0, _fp(dmp), threads), "2D solver %d" % engine)


def time_icgn2d1(ref, tar, rx, ry, conv, stop, pois, threads=0, reps=3):
    """The reference's own ICGN2D1 on a queue of initial guesses: (prepare seconds, best compute seconds); ``pois`` in place.
    None when the library predates this entry point."""
    L = lib()
    if L is None or not hasattr(L, "oc_ref_time_icgn2d1"):
        return None
    ref, tar = _img(ref), _img(tar)
    h, w = ref.shape
    tp, tc = ctypes.c_double(0.0), ctypes.c_double(0.0)
    _check(L.oc_ref_time_icgn2d1(_fp(ref), _fp(tar), h, w, rx, ry, float(conv), float(stop), _fp(pois), pois.shape[0], threads,
                                 reps, ctypes.byref(tp), ctypes.byref(tc)), "ICGN2D1 (timed)")
    return tp.value, tc.value


def epipolar_search(ref, tar, cam1, cam2, search_radius, search_step, parallax_x, parallax_y, rx, ry, conv, stop, pois, threads=0):
    """The reference's EpipolarSearch (src/oc_epipolar_search.cpp) end to end: cameras from (13 intrinsics, 6 extrinsics) each,
    prepare(), compute(poi_queue) in place.  Returns the fundamental matrix it used (3 x 3, row-major float32), or None when
    the library predates this entry point."""
    L = lib()
    if L is None or not hasattr(L, "oc_ref_epipolar_search"):
        return None
    ref, tar = _img(ref), _img(tar)
    h, w = ref.shape
    a = [np.ascontiguousarray(v, dtype=np.float32) for v in (cam1[0], cam1[1], cam2[0], cam2[1], parallax_x, parallax_y)]
    assert a[0].size == 13 and a[1].size == 6 and a[2].size == 13 and a[3].size == 6 and a[4].size == 3 and a[5].size == 3
    F = np.zeros(9, dtype=np.float32)
    _check(L.oc_ref_epipolar_search(_fp(ref), _fp(tar), h, w, _fp(a[0]), _fp(a[1]), _fp(a[2]), _fp(a[3]), int(search_radius),
                                    int(search_step), _fp(a[4]), _fp(a[5]), rx, ry, float(conv), float(stop), _fp(pois), pois.shape[0],
                                    threads, _fp(F)), "EpipolarSearch")
    return F.reshape(3, 3)


def gradient2d(img):
    img = _img(img)
    h, w = img.shape
    gx, gy = np.empty_like(img), np.empty_like(img)
    _check(lib().oc_ref_prepare2d(_fp(img), _fp(img), h, w, _fp(gx), _fp(gy), None), "Gradient2D4")
    return gx, gy


def bspline2d_eval(img, xy):
    img = _img(img)
    xy = np.ascontiguousarray(xy, dtype=np.float32)
    out = np.empty(len(xy), dtype=np.float32)
    _check(lib().oc_ref_bspline2d_eval(_fp(img), img.shape[0], img.shape[1], _fp(xy), len(xy), _fp(out)), "BicubicBspline")
    return out


def fftcc3d(ref, tar, rx, ry, rz, pois, threads=0):
    ref, tar = _img(ref), _img(tar)
    dz, dy, dx = ref.shape
    _check(lib().oc_ref_fftcc3d(_fp(ref), _fp(tar), dz, dy, dx, rx, ry, rz, _fp(pois), pois.shape[0], threads), "FFTCC3D")


def icgn3d1(ref, tar, rx, ry, rz, conv, stop, pois, threads=0):
    ref, tar = _img(ref), _img(tar)
    dz, dy, dx = ref.shape
    _check(lib().oc_ref_icgn3d1(_fp(ref), _fp(tar), dz, dy, dx, rx, ry, rz, float(conv), float(stop), _fp(pois), pois.shape[0],
                                threads), "ICGN3D1")


def prepare3d(ref, tar, xyz):
    """Gradient3D4 of ``ref`` and TricubicBspline(tar).compute at the points ``xyz`` (n x 3)."""
    ref, tar = _img(ref), _img(tar)
    dz, dy, dx = ref.shape
    gx, gy, gz = np.empty_like(ref), np.empty_like(ref), np.empty_like(ref)
    xyz = np.ascontiguousarray(xyz, dtype=np.float32)
    out = np.empty(len(xyz), dtype=np.float32)
    _check(lib().oc_ref_prepare3d(_fp(ref), _fp(tar), dz, dy, dx, _fp(gx), _fp(gy), _fp(gz), _fp(xyz), len(xyz), _fp(out)),
           "Gradient3D4 / TricubicBspline")
    return gx, gy, gz, out


def strain(pois, subregion_radius, neighbor_number_min, zncc_threshold=0.9, approximation=1, threads=0):
    """Strain(radius, nmin, threads) + setZnccThreshold + setApproximation + prepare(queue) + compute(queue) of the
    reference (src/oc_strain.cpp), in place; 2D or 3D by the record size (25 / 31 floats).  Built against the stand-in
    nanoflann (brute-force radius / knn search) and the stand-in float colPivHouseholderQr."""
    ndim = {25: 2, 31: 3}[pois.shape[1]]
    assert pois.dtype == np.float32 and pois.flags.c_contiguous
    _check(lib().oc_ref_strain(ndim, _fp(pois), pois.shape[0], float(subregion_radius), int(neighbor_number_min),
                               float(zncc_threshold), int(approximation), threads), "Strain")


def region_fit(reliable, pois, neighbor_search_radius, neighbor_number_min, threads=0):
    """RegionFit2D / RegionFit3D: setNeighbor(reliable) + prepare() + compute(pois) of the reference
    (src/oc_region_fit.cpp), ``pois`` in place."""
    ndim = {25: 2, 31: 3}[pois.shape[1]]
    assert reliable.shape[1] == pois.shape[1]
    reliable = np.ascontiguousarray(reliable, dtype=np.float32)
    assert pois.dtype == np.float32 and pois.flags.c_contiguous
    _check(lib().oc_ref_region_fit(ndim, _fp(reliable), reliable.shape[0], _fp(pois), pois.shape[0],
                                   float(neighbor_search_radius), int(neighbor_number_min), threads), "RegionFit")
