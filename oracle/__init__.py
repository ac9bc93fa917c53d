"""ctypes binding of the CPU oracle (``oracle/liboc_oracle.so``).

TEST INFRASTRUCTURE ONLY.  Importable from ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg -- never from ``opencorr_amd``.  The
library is a float32 CPU restatement of OpenCorr's FFTCC -> ICGN path; every
native function cites the reference file:line it follows (``oc_oracle.cpp``).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboc_oracle.so")

ORDER_SEQ = 0
ORDER_LANES = 1
ORDER_ROWS = 2   # ICGN3D1 only: the association of the row-mapping A/B kernel (one half-wave per subvolume row), lanes = 512
# OR into an order: every per-sample multiply-add is ONE fmaf (oc_oracle.h "Arithmetic contract"); the HIP engines under
# set_tuning("arith_fma", 1) are bit-exact against ORDER_LANES_FMA
ARITH_FMA = 0x100
ORDER_SEQ_FMA = ORDER_SEQ | ARITH_FMA
ORDER_LANES_FMA = ORDER_LANES | ARITH_FMA
# what the HIP kernels are bit-exact against (tests pass these to icgn2d* / icgn3d1)
GPU_ORDER_2D, GPU_LANES_2D = ORDER_LANES, 64
GPU_ORDER_3D, GPU_LANES_3D = ORDER_LANES, 512     # icgn3d.hip, the default mapping; icgn3d_rows.hip ("icgn3d_mapping" = 1): ORDER_ROWS, 512
POI2D_FLOATS = 25
POI3D_FLOATS = 31

# float offsets inside a POI2D record (src/oc_poi.h:102-136 of the reference)
P2 = dict(x=0, y=1, u=2, ux=3, uy=4, uxx=5, uxy=6, uyy=7, v=8, vx=9, vy=10, vxx=11, vxy=12, vyy=13,
          u0=14, v0=15, zncc=16, iteration=17, convergence=18, feature=19, exx=20, eyy=21, exy=22,
          srx=23, sry=24)
# float offsets inside a POI3D record (src/oc_poi.h:187-222)
P3 = dict(x=0, y=1, z=2, u=3, ux=4, uy=5, uz=6, v=7, vx=8, vy=9, vz=10, w=11, wx=12, wy=13, wz=14,
          u0=15, v0=16, w0=17, zncc=18, iteration=19, convergence=20, feature=21,
          exx=22, eyy=23, ezz=24, exy=25, eyz=26, ezx=27, srx=28, sry=29, srz=30)


def _host_has_fma():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("flags"):
                    return " fma " in line + " "
    except OSError:
        pass
    return False


_FLAGS_NOTE = os.path.join(_HERE, ".liboc_oracle.flags")   # "-mfma" when the library holds vfmadd / VEX instructions


def build(force=False):
    """Compile the oracle with its Makefile (g++ only, no third-party deps).  The library is built with -mfma where the
    building host has the instruction (the OC_ARITH_FMA orders are ~20 x faster with it, same bits); a library that
    arrived from such a host on one without it (a box snapshot) is rebuilt in the portable form before it is loaded."""
    src = os.path.join(_HERE, "oc_oracle.cpp")
    fma = "-mfma" if _host_has_fma() else ""
    try:
        with open(_FLAGS_NOTE) as fh:
            built = fh.read().strip()
    except OSError:
        built = None
    usable = built is None or built == "" or fma == "-mfma"   # unknown / portable builds run anywhere
    if (not force and usable and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(src),
                                                   os.path.getmtime(os.path.join(_HERE, "oc_oracle.h")),
                                                   os.path.getmtime(os.path.join(_HERE, "Makefile")))):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "-s", "FMA_FLAG=" + fma, "liboc_oracle.so"])
    with open(_FLAGS_NOTE, "w") as fh:
        fh.write(fma + "\n")
    return _LIB_PATH


_lib = None
_lib_path = _LIB_PATH


def use_timing_build():
    """bench.py's cpu_baseline only: switch this process to ``liboc_oracle_fast.so`` (the same source compiled
    ``-O3 -march=native`` ON THIS HOST) -- never used for parity.  Returns the build description; falls
    back to the parity build (``-O2 -march=x86-64-v2``) when the compiler is not available."""
    global _lib, _lib_path
    fast = os.path.join(_HERE, "liboc_oracle_fast.so")
    try:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "liboc_oracle_fast.so"], stdout=subprocess.DEVNULL,
                              stderr=subprocess.DEVNULL, timeout=180)
    except Exception:
        return "g++ -O2 -march=x86-64-v2 -fopenmp -ffp-contract=off (parity build: the native build failed)"
    _lib, _lib_path = None, fast
    return "g++ -O3 -march=native -fopenmp -ffp-contract=off, built on this host"


def lib():
    global _lib
    if _lib is None:
        if _lib_path == _LIB_PATH:
            build()   # no-op when the library is current and runs on this host
        L = ctypes.CDLL(_lib_path)
        fp = ctypes.POINTER(ctypes.c_float)
        i, f, l = ctypes.c_int, ctypes.c_float, ctypes.c_long
        L.oc_oracle_gradient2d.argtypes = [fp, i, i, fp, fp, i]
        L.oc_oracle_bspline2d_lut.argtypes = [fp, i, i, fp, i]
        L.oc_oracle_bspline2d_eval.argtypes = [fp, i, i, f, f]
        L.oc_oracle_bspline2d_eval.restype = f
        L.oc_oracle_bspline2d_eval_fma.argtypes = [fp, i, i, f, f]
        L.oc_oracle_bspline2d_eval_fma.restype = f
        L.oc_oracle_bspline3d_eval_fma.argtypes = [fp, i, i, i, f, f, f]
        L.oc_oracle_bspline3d_eval_fma.restype = f
        L.oc_oracle_fma_is_hardware.restype = i
        L.oc_oracle_fftcc2d.argtypes = [fp, fp, i, i, i, i, fp, l, i, fp]
        L.oc_oracle_icgn2d1.argtypes = [fp, fp, fp, fp, i, i, i, i, f, f, fp, l, i, i, i]
        L.oc_oracle_icgn2d2.argtypes = [fp, fp, fp, fp, i, i, i, i, f, f, fp, l, i, i, i]
        L.oc_oracle_icgn2d1_ex.argtypes = [fp, fp, fp, fp, i, i, i, i, f, f, fp, l, i, i, i, fp, i]
        L.oc_oracle_icgn2d2_ex.argtypes = [fp, fp, fp, fp, i, i, i, i, f, f, fp, l, i, i, i, fp, i]
        L.oc_oracle_nr2d1.argtypes = [fp, fp, fp, fp, i, i, i, i, f, f, fp, l, i, i, i]
        L.oc_oracle_nr2d1.restype = None
        L.oc_oracle_iclm2d.argtypes = [i, fp, fp, fp, fp, i, i, i, i, f, f, fp, i, fp, l, i, i, i, i]
        L.oc_oracle_iclm2d.restype = None
        L.oc_oracle_strain2d.argtypes = [fp, l, i, f, i, f, i, i]
        L.oc_oracle_strain2d.restype = None
        L.oc_oracle_strain3d.argtypes = [fp, l, i, f, i, f, i, i]
        L.oc_oracle_strain3d.restype = None
        L.oc_oracle_region_fit2d.argtypes = [fp, l, i, fp, l, i, f, i, i]
        L.oc_oracle_region_fit2d.restype = None
        L.oc_oracle_region_fit3d.argtypes = [fp, l, i, fp, l, i, f, i, i]
        L.oc_oracle_region_fit3d.restype = None
        L.oc_oracle_pow_lambda.argtypes = [f, f]
        L.oc_oracle_pow_lambda.restype = f
        L.oc_oracle_inverse.argtypes = [fp, fp, i]
        L.oc_oracle_inverse.restype = i
        L.oc_oracle_mat_mul.argtypes = [fp, fp, fp, i]
        L.oc_oracle_icgn2d1_ex.restype = None
        L.oc_oracle_icgn2d2_ex.restype = None
        L.oc_oracle_gradient3d.argtypes = [fp, i, i, i, fp, fp, fp, i]
        L.oc_oracle_bspline3d_prefilter.argtypes = [fp, i, i, i, fp, i]
        L.oc_oracle_bspline3d_eval.argtypes = [fp, i, i, i, f, f, f]
        L.oc_oracle_bspline3d_eval.restype = f
        L.oc_oracle_fftcc3d.argtypes = [fp, fp, i, i, i, i, i, i, fp, l, i]
        L.oc_oracle_fftcc3d_ex.argtypes = [fp, fp, i, i, i, i, i, i, fp, l, i, i]
        L.oc_oracle_fftcc3d_ex.restype = None
        L.oc_oracle_icgn3d1.argtypes = [fp, fp, fp, fp, fp, i, i, i, i, i, i, f, f, fp, l, i, i, i]
        L.oc_oracle_max_threads.restype = i
        for name in ("oc_oracle_gradient2d", "oc_oracle_bspline2d_lut", "oc_oracle_fftcc2d", "oc_oracle_icgn2d1",
                     "oc_oracle_icgn2d2", "oc_oracle_gradient3d", "oc_oracle_bspline3d_prefilter",
                     "oc_oracle_fftcc3d", "oc_oracle_icgn3d1"):
            getattr(L, name).restype = None
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _img(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a


def max_threads():
    return lib().oc_oracle_max_threads()


def make_pois2d(xs, ys):
    """Zero-initialised POI2D records (POI2D ctor, src/oc_poi.h:108-135)."""
    xs = np.asarray(xs, dtype=np.float32).ravel()
    ys = np.asarray(ys, dtype=np.float32).ravel()
    pois = np.zeros((xs.size, POI2D_FLOATS), dtype=np.float32)
    pois[:, 0] = xs
    pois[:, 1] = ys
    return pois


def make_pois3d(xs, ys, zs):
    xs = np.asarray(xs, dtype=np.float32).ravel()
    pois = np.zeros((xs.size, POI3D_FLOATS), dtype=np.float32)
    pois[:, 0] = xs
    pois[:, 1] = np.asarray(ys, dtype=np.float32).ravel()
    pois[:, 2] = np.asarray(zs, dtype=np.float32).ravel()
    return pois


def gradient2d(img, threads=0):
    img = _img(img)
    h, w = img.shape
    gx = np.empty_like(img)
    gy = np.empty_like(img)
    lib().oc_oracle_gradient2d(_fp(img), h, w, _fp(gx), _fp(gy), threads)
    return gx, gy


def bspline2d_lut(img, threads=0):
    img = _img(img)
    h, w = img.shape
    lut = np.empty((h, w, 16), dtype=np.float32)
    lib().oc_oracle_bspline2d_lut(_fp(img), h, w, _fp(lut), threads)
    return lut


def bspline2d_eval(lut, x, y):
    h, w, _ = lut.shape
    return lib().oc_oracle_bspline2d_eval(_fp(lut), h, w, float(x), float(y))


def fftcc2d(ref, tar, rx, ry, pois, threads=0, want_surface=False):
    """In-place FFTCC2D::compute(poi_queue) on ``pois`` (n x 25 float32)."""
    ref, tar = _img(ref), _img(tar)
    assert pois.dtype == np.float32 and pois.flags.c_contiguous and pois.shape[1] == POI2D_FLOATS
    h, w = ref.shape
    surf = np.zeros((2 * ry, 2 * rx), dtype=np.float32) if want_surface else None
    lib().oc_oracle_fftcc2d(_fp(ref), _fp(tar), h, w, rx, ry, _fp(pois), pois.shape[0], threads,
                            _fp(surf) if want_surface else None)
    return surf


class Prepared2D:
    """What ICGN2D1/2D2::prepare() builds: reference gradients + target LUT."""

    def __init__(self, ref, tar, threads=0):
        self.ref = _img(ref)
        self.tar = _img(tar)
        self.gx, self.gy = gradient2d(self.ref, threads)
        self.lut = bspline2d_lut(self.tar, threads)


def _icgn2d(fn, prep, rx, ry, conv, stop, pois, order, lanes, threads, center_offsets, self_adaptive):
    assert pois.dtype == np.float32 and pois.flags.c_contiguous and pois.shape[1] == POI2D_FLOATS
    h, w = prep.ref.shape
    off = None
    if center_offsets is not None:
        off = np.ascontiguousarray(center_offsets, dtype=np.float32)
        assert off.shape == (pois.shape[0], 2)
    fn(_fp(prep.ref), _fp(prep.gx), _fp(prep.gy), _fp(prep.lut), h, w, rx, ry, float(conv), float(stop), _fp(pois),
       pois.shape[0], order, lanes, threads, _fp(off) if off is not None else None, 1 if self_adaptive else 0)


def icgn2d1(prep, rx, ry, conv, stop, pois, order=ORDER_SEQ, lanes=64, threads=0, center_offsets=None,
            self_adaptive=False):
    """ICGN2D1::compute(poi_queue[, center_offset_queue]); ``self_adaptive`` = DIC::setSelfAdaptive(true)."""
    _icgn2d(lib().oc_oracle_icgn2d1_ex, prep, rx, ry, conv, stop, pois, order, lanes, threads, center_offsets,
            self_adaptive)


def icgn2d2(prep, rx, ry, conv, stop, pois, order=ORDER_SEQ, lanes=64, threads=0, center_offsets=None,
            self_adaptive=False):
    _icgn2d(lib().oc_oracle_icgn2d2_ex, prep, rx, ry, conv, stop, pois, order, lanes, threads, center_offsets,
            self_adaptive)


def _iclm2d(dof, prep, rx, ry, conv, stop, pois, damping, order, lanes, threads, self_adaptive):
    assert pois.dtype == np.float32 and pois.flags.c_contiguous and pois.shape[1] == POI2D_FLOATS
    h, w = prep.ref.shape
    d = np.asarray(damping, dtype=np.float32)
    assert d.shape == (3,)
    lib().oc_oracle_iclm2d(dof, _fp(prep.ref), _fp(prep.gx), _fp(prep.gy), _fp(prep.lut), h, w, rx, ry, float(conv),
                           float(stop), _fp(d), 1 if self_adaptive else 0, _fp(pois), pois.shape[0], POI2D_FLOATS, order,
                           lanes, threads)


DEFAULT_DAMPING = (100.0, 0.1, 10.0)  # struct DampingParameter, src/oc_iclm.h:33-38


def iclm2d1(prep, rx, ry, conv, stop, pois, damping=DEFAULT_DAMPING, order=ORDER_SEQ, lanes=64, threads=0,
            self_adaptive=False):
    """ICLM2D1::compute(poi_queue) (src/oc_iclm.cpp:150-368), in place on ``pois``; ``prep`` as for icgn2d1."""
    _iclm2d(6, prep, rx, ry, conv, stop, pois, damping, order, lanes, threads, self_adaptive)


def iclm2d2(prep, rx, ry, conv, stop, pois, damping=DEFAULT_DAMPING, order=ORDER_SEQ, lanes=64, threads=0,
            self_adaptive=False):
    """ICLM2D2::compute(poi_queue) (src/oc_iclm.cpp:505-741)."""
    _iclm2d(12, prep, rx, ry, conv, stop, pois, damping, order, lanes, threads, self_adaptive)


def pow_lambda(lam, q):
    return float(lib().oc_oracle_pow_lambda(float(lam), float(q)))


def inverse(a):
    """Inverse of a small square float32 matrix as the solvers compute it (cofactors for 3 x 3 / 4 x 4, LU with partial
    pivoting otherwise: what the reference takes from Eigen, src/oc_icgn.cpp:210,290,759,831,1339,1439)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    assert a.ndim == 2 and a.shape[0] == a.shape[1]
    out = np.empty_like(a)
    if lib().oc_oracle_inverse(_fp(a), _fp(out), a.shape[0]) != 0:
        raise ValueError("unsupported matrix size %d" % a.shape[0])
    return out


def mat_mul(a, b):
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    c = np.empty_like(a)
    lib().oc_oracle_mat_mul(_fp(a), _fp(b), _fp(c), a.shape[0])
    return c


def strain2d(pois, subregion_radius, neighbor_number_min, zncc_threshold=0.9, approximation=1, threads=0):
    """Strain::prepare + Strain::compute(std::vector<POI2D>&) (src/oc_strain.cpp:96-107, 149-247), in place."""
    assert pois.dtype == np.float32 and pois.flags.c_contiguous and pois.shape[1] == POI2D_FLOATS
    lib().oc_oracle_strain2d(_fp(pois), pois.shape[0], POI2D_FLOATS, float(subregion_radius), int(neighbor_number_min),
                             float(zncc_threshold), int(approximation), threads)


def strain3d(pois, subregion_radius, neighbor_number_min, zncc_threshold=0.9, approximation=1, threads=0):
    """Strain::compute(std::vector<POI3D>&) (src/oc_strain.cpp:372-488), in place."""
    assert pois.dtype == np.float32 and pois.flags.c_contiguous and pois.shape[1] == POI3D_FLOATS
    lib().oc_oracle_strain3d(_fp(pois), pois.shape[0], POI3D_FLOATS, float(subregion_radius), int(neighbor_number_min),
                             float(zncc_threshold), int(approximation), threads)


def region_fit(reliable, pois, neighbor_search_radius, neighbor_number_min, threads=0):
    """RegionFit2D / RegionFit3D: setNeighbor(reliable); prepare(); compute(pois) (src/oc_region_fit.cpp), in place on
    ``pois``.  Both arrays are (n, 25) POI2D or (n, 31) POI3D records."""
    for a in (reliable, pois):
        assert a.dtype == np.float32 and a.flags.c_contiguous and a.ndim == 2
    assert reliable.shape[1] == pois.shape[1] and pois.shape[1] in (POI2D_FLOATS, POI3D_FLOATS)
    fn = lib().oc_oracle_region_fit2d if pois.shape[1] == POI2D_FLOATS else lib().oc_oracle_region_fit3d
    fn(_fp(reliable), reliable.shape[0], reliable.shape[1], _fp(pois), pois.shape[0], pois.shape[1], float(neighbor_search_radius),
       int(neighbor_number_min), threads)


class PreparedNR2D:
    """What NR2D1::prepare() builds (src/oc_nr.cpp:119-158): the target's gradients and three LUTs."""

    def __init__(self, ref, tar, threads=0):
        self.ref = _img(ref)
        self.tar = _img(tar)
        gx, gy = gradient2d(self.tar, threads)
        self.lut = bspline2d_lut(self.tar, threads)
        self.lut_gx = bspline2d_lut(gx, threads)
        self.lut_gy = bspline2d_lut(gy, threads)


def nr2d1(prep, rx, ry, conv, stop, pois, order=ORDER_SEQ, lanes=64, threads=0):
    """NR2D1::compute(poi_queue), in place on ``pois`` (n x 25 float32)."""
    assert pois.dtype == np.float32 and pois.flags.c_contiguous and pois.shape[1] == POI2D_FLOATS
    h, w = prep.ref.shape
    lib().oc_oracle_nr2d1(_fp(prep.ref), _fp(prep.lut), _fp(prep.lut_gx), _fp(prep.lut_gy), h, w, rx, ry, float(conv),
                          float(stop), _fp(pois), pois.shape[0], order, lanes, threads)


def gradient3d(vol, threads=0):
    vol = _img(vol)
    dz, dy, dx = vol.shape
    gx, gy, gz = np.empty_like(vol), np.empty_like(vol), np.empty_like(vol)
    lib().oc_oracle_gradient3d(_fp(vol), dz, dy, dx, _fp(gx), _fp(gy), _fp(gz), threads)
    return gx, gy, gz


def bspline3d_prefilter(vol, threads=0):
    vol = _img(vol)
    dz, dy, dx = vol.shape
    coef = np.empty_like(vol)
    lib().oc_oracle_bspline3d_prefilter(_fp(vol), dz, dy, dx, _fp(coef), threads)
    return coef


def bspline3d_eval(coef, x, y, z):
    dz, dy, dx = coef.shape
    return lib().oc_oracle_bspline3d_eval(_fp(coef), dz, dy, dx, float(x), float(y), float(z))


def fftcc3d(ref, tar, rx, ry, rz, pois, threads=0, exact_sums=False):
    """FFTCC3D::compute(poi_queue) in place.  ``exact_sums``: means, norms and the ZNCC quotient in double instead of the
    reference's sequential float32 running sums (src/oc_fftcc.cpp:340-376) -- the yardstick for the ZNCC of large windows,
    where those running sums alone carry 1e-4 (32^3) ... 3e-4 (60^3)."""
    ref, tar = _img(ref), _img(tar)
    assert pois.dtype == np.float32 and pois.flags.c_contiguous and pois.shape[1] == POI3D_FLOATS
    dz, dy, dx = ref.shape
    lib().oc_oracle_fftcc3d_ex(_fp(ref), _fp(tar), dz, dy, dx, rx, ry, rz, _fp(pois), pois.shape[0], threads,
                               1 if exact_sums else 0)


class Prepared3D:
    def __init__(self, ref, tar, threads=0):
        self.ref = _img(ref)
        self.tar = _img(tar)
        self.gx, self.gy, self.gz = gradient3d(self.ref, threads)
        self.coef = bspline3d_prefilter(self.tar, threads)


def icgn3d1(prep, rx, ry, rz, conv, stop, pois, order=ORDER_SEQ, lanes=256, threads=0):
    assert pois.dtype == np.float32 and pois.flags.c_contiguous and pois.shape[1] == POI3D_FLOATS
    dz, dy, dx = prep.ref.shape
    lib().oc_oracle_icgn3d1(_fp(prep.ref), _fp(prep.gx), _fp(prep.gy), _fp(prep.gz), _fp(prep.coef), dz, dy, dx,
                            rx, ry, rz, float(conv), float(stop), _fp(pois), pois.shape[0], order, lanes, threads)
