"""ctypes binding of the C-ABI in ``include/opencorr_hip.h``.

Nothing here computes: every call goes to ``lib/libopencorr_hip.so`` (hand-written
HIP kernels for gfx950 + rocFFT).  If the library is missing the import fails
loudly -- there is no CPU fallback anywhere in ``opencorr_amd``.
"""
import ctypes
import importlib.util
import os
import warnings

_HERE = os.path.dirname(os.path.abspath(__file__))
# OPENCORR_HIP_LIB: developer override (tools/ablate_icgn3d.sh loads instrumented builds of the same library)
LIB_PATH = os.environ.get("OPENCORR_HIP_LIB") or os.path.join(_HERE, "lib", "libopencorr_hip.so")

OK = 0
ERR_INVALID, ERR_HIP, ERR_ROCFFT, ERR_NOMEM, ERR_UNSUPPORTED = 1, 2, 3, 4, 5
HOST, DEVICE = 0, 1
ROW_MAJOR, COL_MAJOR = 0, 1
FFTCC2D, ICGN2D1, ICGN2D2, FFTCC3D, ICGN3D1, NR2D1, ICLM2D1, ICLM2D2, STRAIN, REGION_FIT = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10
POI2D_BYTES, POI3D_BYTES = 100, 124
POI2D_FLOATS, POI3D_FLOATS = 25, 31

# every symbol include/opencorr_hip.h declares (tests check the .so exports them all)
SYMBOLS = [
    "oc_hip_last_error", "oc_hip_device_count", "oc_hip_abi_version",
    "oc_hip_fftcc2d_create", "oc_hip_icgn2d1_create", "oc_hip_icgn2d2_create", "oc_hip_nr2d1_create",
    "oc_hip_iclm2d1_create", "oc_hip_iclm2d2_create", "oc_hip_set_damping",
    "oc_hip_strain_create", "oc_hip_strain_set", "oc_hip_strain_prepare", "oc_hip_strain_compute",
    "oc_hip_region_fit_create", "oc_hip_region_fit_set", "oc_hip_region_fit_prepare", "oc_hip_region_fit_compute",
    "oc_hip_fftcc3d_create", "oc_hip_icgn3d1_create", "oc_hip_destroy",
    "oc_hip_set_images2d", "oc_hip_set_images3d", "oc_hip_share_images", "oc_hip_set_subset",
    "oc_hip_set_iteration", "oc_hip_set_stream", "oc_hip_reset_stream", "oc_hip_set_tuning",
    "oc_hip_prepare", "oc_hip_prepare_ref", "oc_hip_prepare_tar",
    "oc_hip_compute", "oc_hip_compute_chain", "oc_hip_compute_one", "oc_hip_compute_with_offsets", "oc_hip_compute_one_with_offset", "oc_hip_single_stats",
    "oc_hip_set_self_adaptive", "oc_hip_synchronize", "oc_hip_select_best", "oc_hip_split_reliable", "oc_hip_merge_recovered",
    "oc_hip_get_kind", "oc_hip_get_field", "oc_hip_read_field",
    "oc_hip_profile_enable", "oc_hip_profile_read", "oc_hip_profile_reset",
    "oc_hip_set_devices", "oc_hip_get_devices", "oc_hip_group_queue",
]


class OpenCorrHipError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("opencorr_hip status %d: %s" % (status, message))
        self.status = status


_lib = None
_hip = None


def _mapped(name):
    """Paths of the shared objects mapped into this process whose file name contains ``name``."""
    try:
        with open("/proc/self/maps") as f:
            return sorted({ln.split()[-1] for ln in f if name in ln and "/" in ln})
    except OSError:
        return []


def _one_hip_runtime():
    """A process must hold ONE HIP runtime: stream and event handles of one copy of libamdhip64 crash the other
    (measured on the MI355X box: `std::bad_variant_access` inside hipStreamWaitEvent).  PyTorch wheels bundle their own
    copy (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7); libopencorr_hip.so asks for "libamdhip64.so.7".  When torch
    is imported FIRST the dynamic loader hands us torch's copy (SONAME match) and all is well; when this module is loaded
    first it would bind /opt/rocm's copy and a later `import torch` would add a second runtime beside it.  So, if torch
    is installed and no HIP runtime is mapped yet, torch's copy is loaded by path before our library -- without importing
    torch.  C++ hosts (no Python, no torch) simply get the ROCm installation's runtime."""
    if _mapped("libamdhip64"):
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)


def hip_runtime():
    """ctypes handle of THE HIP runtime this process uses (the copy libopencorr_hip.so is bound to) -- for helpers that
    need a raw hipMemcpy / hipStreamCreate next to the engines.  Never `ctypes.CDLL("libamdhip64.so")`: by name that may
    load a second copy (see _one_hip_runtime)."""
    global _hip
    if _hip is None:
        lib()
        paths = _mapped("libamdhip64")
        if not paths:
            raise ImportError("no libamdhip64 is mapped into this process")
        _hip = ctypes.CDLL(paths[0], mode=ctypes.RTLD_GLOBAL)
    return _hip


def lib():
    """Loads libopencorr_hip.so (built by ``python -m opencorr_amd.build``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing: build it with `python -m opencorr_amd.build` (hipcc, gfx950). "
            "opencorr_amd has no CPU fallback." % LIB_PATH)
    _one_hip_runtime()
    L = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    copies = _mapped("libamdhip64")
    if len(copies) > 1:
        warnings.warn("more than one HIP runtime is mapped into this process (%s): stream handles must not cross between "
                      "them -- import opencorr_amd before anything that loads another libamdhip64, or torch first"
                      % ", ".join(copies), RuntimeWarning)
    vp, i, f, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t
    pp = ctypes.POINTER(vp)
    L.oc_hip_last_error.restype = ctypes.c_char_p
    L.oc_hip_last_error.argtypes = []
    L.oc_hip_device_count.argtypes = [ctypes.POINTER(i)]
    L.oc_hip_abi_version.argtypes = []
    L.oc_hip_fftcc2d_create.argtypes = [i, i, i, pp]
    L.oc_hip_icgn2d1_create.argtypes = [i, i, f, f, i, pp]
    L.oc_hip_icgn2d2_create.argtypes = [i, i, f, f, i, pp]
    L.oc_hip_nr2d1_create.argtypes = [i, i, f, f, i, pp]
    L.oc_hip_iclm2d1_create.argtypes = [i, i, f, f, i, pp]
    L.oc_hip_iclm2d2_create.argtypes = [i, i, f, f, i, pp]
    L.oc_hip_set_damping.argtypes = [vp, f, f, f]
    L.oc_hip_select_best.argtypes = [vp, vp, sz, sz, vp, sz, vp, sz, i]
    psz = ctypes.POINTER(sz)
    L.oc_hip_split_reliable.argtypes = [vp, vp, sz, sz, i, f, f, f, vp, sz, vp, vp, psz, psz, i]
    L.oc_hip_merge_recovered.argtypes = [vp, vp, sz, sz, i, vp, vp, sz, f, f, vp, sz, psz, psz, i]
    L.oc_hip_strain_create.argtypes = [f, i, i, pp]
    L.oc_hip_strain_set.argtypes = [vp, f, i, f, i]
    L.oc_hip_strain_prepare.argtypes = [vp, vp, sz, sz, i, i]
    L.oc_hip_strain_compute.argtypes = [vp, vp, sz, sz, i, i]
    L.oc_hip_region_fit_create.argtypes = [f, i, i, pp]
    L.oc_hip_region_fit_set.argtypes = [vp, f, i]
    L.oc_hip_region_fit_prepare.argtypes = [vp, vp, sz, sz, i, i]
    L.oc_hip_region_fit_compute.argtypes = [vp, vp, sz, sz, i, i]
    L.oc_hip_fftcc3d_create.argtypes = [i, i, i, i, pp]
    L.oc_hip_icgn3d1_create.argtypes = [i, i, i, f, f, i, pp]
    L.oc_hip_destroy.argtypes = [vp]
    L.oc_hip_set_images2d.argtypes = [vp, vp, vp, i, i, i, i]
    L.oc_hip_set_images3d.argtypes = [vp, vp, vp, i, i, i, i]
    L.oc_hip_share_images.argtypes = [vp, vp]
    L.oc_hip_set_subset.argtypes = [vp, i, i, i]
    L.oc_hip_set_iteration.argtypes = [vp, f, f]
    L.oc_hip_set_stream.argtypes = [vp, vp]
    L.oc_hip_reset_stream.argtypes = [vp]
    L.oc_hip_set_tuning.argtypes = [vp, ctypes.c_char_p, i]
    L.oc_hip_prepare.argtypes = [vp]
    L.oc_hip_prepare_ref.argtypes = [vp]
    L.oc_hip_prepare_tar.argtypes = [vp]
    L.oc_hip_compute.argtypes = [vp, vp, sz, sz, i]
    L.oc_hip_compute_chain.argtypes = [pp, i, vp, sz, sz, i]
    L.oc_hip_compute_one.argtypes = [vp, vp]
    L.oc_hip_compute_with_offsets.argtypes = [vp, vp, vp, sz, sz, i]
    L.oc_hip_compute_one_with_offset.argtypes = [vp, vp, vp]
    L.oc_hip_single_stats.argtypes = [vp, ctypes.POINTER(ctypes.c_ulonglong), ctypes.POINTER(ctypes.c_ulonglong)]
    L.oc_hip_set_self_adaptive.argtypes = [vp, i]
    L.oc_hip_synchronize.argtypes = [vp]
    L.oc_hip_get_kind.argtypes = [vp, ctypes.POINTER(i)]
    L.oc_hip_get_field.argtypes = [vp, ctypes.c_char_p, pp, ctypes.POINTER(sz)]
    L.oc_hip_read_field.argtypes = [vp, ctypes.c_char_p, vp, sz]
    L.oc_hip_profile_enable.argtypes = [vp, i]
    L.oc_hip_profile_read.argtypes = [vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_long)]
    L.oc_hip_profile_reset.argtypes = [vp]
    L.oc_hip_set_devices.argtypes = [vp, ctypes.POINTER(i), i]
    L.oc_hip_get_devices.argtypes = [vp, ctypes.POINTER(i), i, ctypes.POINTER(i)]
    L.oc_hip_group_queue.argtypes = [vp, i, pp, ctypes.POINTER(sz)]
    for name in SYMBOLS:
        if name != "oc_hip_last_error":
            getattr(L, name).restype = i
    _lib = L
    return L


def check(status):
    if status != OK:
        raise OpenCorrHipError(status, lib().oc_hip_last_error().decode("utf-8", "replace"))


def device_count():
    n = ctypes.c_int(0)
    check(lib().oc_hip_device_count(ctypes.byref(n)))
    return n.value
