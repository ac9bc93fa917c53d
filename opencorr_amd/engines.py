"""Python mirror of OpenCorr's hot-path classes over the HIP C-ABI.

Same names, constructor arguments and call order as the reference
(``src/oc_fftcc.h:54-89``, ``src/oc_icgn.h:45-180``, ``src/oc_dic.h:43-84``):

    fftcc = FFTCC2D(rx, ry);            fftcc.set_images(ref, tar); fftcc.compute(pois)
    icgn  = ICGN2D1(rx, ry, conv, stop); icgn.set_images(ref, tar);  icgn.prepare(); icgn.compute(pois)

``pois`` is the reference's POI2D/POI3D AoS as a float32 array of shape (n, 25) /
(n, 31), updated in place: a NumPy array takes the host path (H2D, kernels,
D2H), a CUDA torch tensor is used in place on the device.  Images are NumPy
arrays (uploaded) or CUDA torch tensors (row-major, used in place).
"""
import ctypes

import numpy as np

from . import capi


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _buf(x, floats_per_row=None):
    """(pointer, memory kind, keep-alive) of a float32 NumPy array or CUDA torch tensor."""
    if _is_torch(x):
        import torch
        if x.dtype != torch.float32 or not x.is_contiguous():
            raise ValueError("torch buffers must be contiguous float32")
        if not x.is_cuda:
            raise ValueError("torch buffers must live on the GPU (pass NumPy arrays for host data)")
        return ctypes.c_void_p(x.data_ptr()), capi.DEVICE, x
    a = np.ascontiguousarray(x, dtype=np.float32)
    return ctypes.c_void_p(a.ctypes.data), capi.HOST, a


class _Engine:
    _ndim = 2

    def __init__(self, device=0):
        self._h = ctypes.c_void_p()
        self._keep = []
        self._device = int(device)   # where the engine lives (set_devices()[0] moves it)
        self._stream_pinned = False  # set_stream() was called by the user
        self._auto_stream = None     # torch stream adopted for CUDA-tensor arguments
        self._on_private_stream = True  # False once any caller-owned stream (pinned or adopted) is in use

    # -- lifecycle ---------------------------------------------------------
    def close(self):
        if self._h:
            capi.lib().oc_hip_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- DIC::setImages / DVC::setImages ------------------------------------
    def set_images(self, ref, tar, layout=capi.ROW_MAJOR):
        """``layout=COL_MAJOR``: the arrays have the logical shape (height, width) but Fortran
        (Eigen::MatrixXf) memory order; NumPy inputs are converted with ``np.asfortranarray``."""
        if layout == capi.COL_MAJOR and not _is_torch(ref):
            ref = np.asfortranarray(ref, dtype=np.float32)
            tar = np.asfortranarray(tar, dtype=np.float32)
            rp, rmem, rkeep = ctypes.c_void_p(ref.ctypes.data), capi.HOST, ref
            tp, tmem, tkeep = ctypes.c_void_p(tar.ctypes.data), capi.HOST, tar
        else:
            self._adopt_stream_of(ref)
            rp, rmem, rkeep = _buf(ref)
            tp, tmem, tkeep = _buf(tar)
        if rmem != tmem:
            raise ValueError("reference and target image must live in the same memory space")
        shape = tuple(rkeep.shape)
        if tuple(tkeep.shape) != shape or len(shape) != self._ndim:
            raise ValueError("bad image shapes %r / %r" % (shape, tuple(tkeep.shape)))
        if self._ndim == 2:
            h, w = shape
            capi.check(capi.lib().oc_hip_set_images2d(self._h, rp, tp, h, w, layout, rmem))
        else:
            dz, dy, dx = shape
            capi.check(capi.lib().oc_hip_set_images3d(self._h, rp, tp, dx, dy, dz, rmem))
        # device images are used in place: keep them alive
        self._keep = [rkeep, tkeep] if rmem == capi.DEVICE else []
        self.shape = shape

    def share_images(self, donor):
        """Use the donor's device-resident image pair (one upload for an FFTCC + ICGN pair).  The borrower also joins the
        stream the donor runs on (pinned with ``set_stream`` or adopted from a CUDA tensor), unless it was given a
        stream of its own: the donor's images may still be in flight on that stream, and ``prepare()`` receives no tensor
        it could take the stream from -- on its private stream it would read them too early."""
        capi.check(capi.lib().oc_hip_share_images(self._h, donor._h))
        self._keep = [donor]
        self.shape = donor.shape
        if not self._stream_pinned and not getattr(donor, "_on_private_stream", True):
            s = donor._auto_stream
            capi.check(capi.lib().oc_hip_set_stream(self._h, ctypes.c_void_p(s or None)))
            self._auto_stream = s
            self._on_private_stream = False

    def set_subset(self, rx, ry, rz=0):
        capi.check(capi.lib().oc_hip_set_subset(self._h, rx, ry, rz))

    def set_stream(self, stream_handle):
        """Run on a caller-owned hipStream_t (0 / None = HIP's default stream, as torch reports it)."""
        capi.check(capi.lib().oc_hip_set_stream(self._h, ctypes.c_void_p(stream_handle or None)))
        self._stream_pinned = True
        self._auto_stream = stream_handle or None
        self._on_private_stream = False

    def _adopt_stream_of(self, x):
        """CUDA torch tensors are produced on torch's current stream: unless the user pinned a stream, the engine
        runs on that stream too, so the tensor is read after whatever wrote it and torch ops that follow see the
        results (stream order instead of timing).  From then on the engine STAYS on that stream (until
        ``reset_stream``): DEVICE-queue computes are asynchronous, and later NumPy / host-queue calls run on it too.
        A tensor that lives on another GPU than the engine is refused (its stream belongs to that device)."""
        if not _is_torch(x) or not x.is_cuda:
            return
        dev = getattr(self, "_device", None)
        if dev is not None and x.device.index is not None and x.device.index != dev:
            raise ValueError("tensor lives on cuda:%d but the engine on device %d" % (x.device.index, dev))
        if self._stream_pinned:
            return
        import torch
        s = torch.cuda.current_stream(x.device).cuda_stream or None
        if getattr(self, "_on_private_stream", True) or s != self._auto_stream:
            capi.check(capi.lib().oc_hip_set_stream(self._h, ctypes.c_void_p(s)))
            self._auto_stream = s
            self._on_private_stream = False

    def set_devices(self, device_ids):
        """Spread this engine over several GPUs of the node (``oc_hip_set_devices``): contiguous blocks of every
        queue, one per device; setters, images, prepare() and compute() fan out.  ``device_ids[0]`` is the engine's
        own device (it moves there if needed -- set the images again).  Results do not depend on the group size."""
        ids = (ctypes.c_int * len(device_ids))(*[int(d) for d in device_ids])
        capi.check(capi.lib().oc_hip_set_devices(self._h, ids, len(device_ids)))
        if int(device_ids[0]) != self._device:
            # the engine moved: it is back on a fresh private stream of the new device
            self._device = int(device_ids[0])
            self._stream_pinned = False
            self._auto_stream = None
            self._on_private_stream = True

    def devices(self):
        n = ctypes.c_int()
        capi.check(capi.lib().oc_hip_get_devices(self._h, None, 0, ctypes.byref(n)))
        ids = (ctypes.c_int * n.value)()
        capi.check(capi.lib().oc_hip_get_devices(self._h, ids, n.value, ctypes.byref(n)))
        return list(ids)

    def group_queue(self, member):
        """(device pointer, bytes per block) of a member's all-gathered copy of the last DEVICE queue."""
        ptr = ctypes.c_void_p()
        blk = ctypes.c_size_t()
        capi.check(capi.lib().oc_hip_group_queue(self._h, int(member), ctypes.byref(ptr), ctypes.byref(blk)))
        return ptr.value, blk.value

    def set_tuning(self, key, value):
        """Knob of the C-ABI (``oc_hip_set_tuning``).  Every key but one selects among kernels that compute the same bits;
        ``"arith_fma"`` = 1 switches the ICGN / IC-LM solvers to the fused arithmetic contract (one rounding fewer per
        per-sample multiply-add: oracle order ``ORDER_LANES_FMA``; results move by rounding, inside the 1e-4 tolerance)."""
        capi.check(capi.lib().oc_hip_set_tuning(self._h, key.encode(), int(value)))

    def reset_stream(self):
        capi.check(capi.lib().oc_hip_reset_stream(self._h))
        self._stream_pinned = False
        self._auto_stream = None
        self._on_private_stream = True

    def prepare(self):
        capi.check(capi.lib().oc_hip_prepare(self._h))

    def prepare_ref(self):
        capi.check(capi.lib().oc_hip_prepare_ref(self._h))

    def prepare_tar(self):
        capi.check(capi.lib().oc_hip_prepare_tar(self._h))

    # -- compute(std::vector<POI>&) -------------------------------------------
    def compute(self, pois):
        floats = capi.POI2D_FLOATS if self._ndim == 2 else capi.POI3D_FLOATS
        self._adopt_stream_of(pois)
        if _is_torch(pois):
            p, mem, _ = _buf(pois)
            n, stride = pois.shape[0], pois.stride(0) * 4
            if pois.shape[1] < floats:
                raise ValueError("POI records need %d floats" % floats)
        else:
            if pois.dtype != np.float32 or not pois.flags.c_contiguous or pois.ndim != 2 or pois.shape[1] < floats:
                raise ValueError("pois must be a C-contiguous float32 array of shape (n, >=%d)" % floats)
            p, mem = ctypes.c_void_p(pois.ctypes.data), capi.HOST
            n, stride = pois.shape[0], pois.strides[0]
        capi.check(capi.lib().oc_hip_compute(self._h, p, n, stride, mem))
        return pois

    def compute_with_offsets(self, pois, center_offsets):
        """compute(poi_queue, center_offset_queue) of ICGN2D1/2D2 (src/oc_icgn.h:76,131);
        ``center_offsets`` is (n, 2) float32 (Point2D x, y) in the same memory space as ``pois``."""
        floats = capi.POI2D_FLOATS
        self._adopt_stream_of(pois)
        if _is_torch(pois):
            p, mem, _ = _buf(pois)
            o, omem, _ = _buf(center_offsets)
            if omem != mem:
                raise ValueError("POIs and center offsets must live in the same memory space")
            n, stride = pois.shape[0], pois.stride(0) * 4
        else:
            if pois.dtype != np.float32 or not pois.flags.c_contiguous or pois.ndim != 2 or pois.shape[1] < floats:
                raise ValueError("pois must be a C-contiguous float32 array of shape (n, >=%d)" % floats)
            off = np.ascontiguousarray(center_offsets, dtype=np.float32)
            p, mem = ctypes.c_void_p(pois.ctypes.data), capi.HOST
            o = ctypes.c_void_p(off.ctypes.data)
            n, stride = pois.shape[0], pois.strides[0]
        if tuple(center_offsets.shape) != (n, 2):
            raise ValueError("center_offsets must have shape (n, 2)")
        capi.check(capi.lib().oc_hip_compute_with_offsets(self._h, p, o, n, stride, mem))
        return pois

    def select_best(self, candidates, segment_starts, pois):
        """Keeps, for POI s, the candidate with the highest ZNCC among candidates[segment_starts[s]:segment_starts[s+1]]
        (deformation and result vectors are copied into pois[s]) -- the selection step of
        EpipolarSearch::compute (src/oc_epipolar_search.cpp:181-190) for a batched candidate queue."""
        self._adopt_stream_of(pois)
        if _is_torch(pois):
            import torch
            cp, mem, _ = _buf(candidates)
            pp_, pmem, _ = _buf(pois)
            if segment_starts.dtype != torch.int32 or not segment_starts.is_cuda or mem != pmem:
                raise ValueError("device queues need an int32 CUDA tensor of segment starts")
            sp = ctypes.c_void_p(segment_starts.data_ptr())
            nc, cs, ns, ps = candidates.shape[0], candidates.stride(0) * 4, pois.shape[0], pois.stride(0) * 4
        else:
            seg = np.ascontiguousarray(segment_starts, dtype=np.uint32)
            self._keep_seg = seg
            cp, mem = ctypes.c_void_p(candidates.ctypes.data), capi.HOST
            pp_ = ctypes.c_void_p(pois.ctypes.data)
            sp = ctypes.c_void_p(seg.ctypes.data)
            nc, cs, ns, ps = candidates.shape[0], candidates.strides[0], pois.shape[0], pois.strides[0]
        if len(segment_starts) != ns + 1:
            raise ValueError("segment_starts needs len(pois) + 1 entries")
        capi.check(capi.lib().oc_hip_select_best(self._h, cp, nc, cs, sp, ns, pp_, ps, mem))
        return pois

    # -- the two selections of the RegionFit -> re-ICGN loop, on the device --------------------------------------
    def _record_floats(self):
        return capi.POI2D_FLOATS if self._ndim == 2 else capi.POI3D_FLOATS

    def _queue_buf(self, x, name, want_mem=None, rows=None, stride_bytes=None):
        """(pointer, memory kind, row stride in bytes) of a POI queue argument, checked the way compute() checks its
        queue: float32, C-contiguous rows of at least one record (the record type is the ENGINE's: a 2D queue padded to
        32 floats per row is still a POI2D queue), CUDA tensor or NumPy array, and -- when other queues of the call are
        already known -- the same memory kind and row stride."""
        floats = self._record_floats()
        if _is_torch(x):
            p, mem, _ = _buf(x)  # float32, contiguous, on the GPU -- or ValueError
            ndim, width, n = x.dim(), x.shape[1] if x.dim() == 2 else 0, x.shape[0]
        else:
            if not isinstance(x, np.ndarray) or x.dtype != np.float32 or not x.flags.c_contiguous:
                raise ValueError("%s must be a C-contiguous float32 NumPy array or a contiguous float32 CUDA tensor" % name)
            p, mem = ctypes.c_void_p(x.ctypes.data), capi.HOST
            ndim, width, n = x.ndim, x.shape[1] if x.ndim == 2 else 0, x.shape[0]
        if ndim != 2 or width < floats:
            raise ValueError("%s must have shape (n, >= %d): POI%dD records" % (name, floats, self._ndim))
        stride = width * 4  # contiguous rows (checked above); an EMPTY array reports a stride of 0, its width does not
        if want_mem is not None and mem != want_mem:
            raise ValueError("%s must live in the same memory space as the POI queue" % name)
        if stride_bytes is not None and stride != stride_bytes:
            raise ValueError("%s must have the POI queue's row stride (%d bytes, got %d)" % (name, stride_bytes, stride))
        if rows is not None and n < rows:
            raise ValueError("%s needs room for %d records (has %d)" % (name, rows, n))
        return p, mem, stride

    @staticmethod
    def _index_buf(index, name, want_mem, rows):
        """pointer of a queue-index list: one 32-bit integer per record, same memory kind as the queues"""
        if _is_torch(index):
            import torch
            # (torch.uint32 exists from torch 2.3 on)
            ok_dtypes = tuple(d for d in (torch.int32, getattr(torch, "uint32", None)) if d is not None)
            if index.dtype not in ok_dtypes or not index.is_contiguous() or index.dim() != 1 or not index.is_cuda:
                raise ValueError("%s must be a contiguous 1-d int32 CUDA tensor" % name)
            mem, n, p = capi.DEVICE, index.shape[0], ctypes.c_void_p(index.data_ptr())
        else:
            if not isinstance(index, np.ndarray) or index.dtype not in (np.uint32, np.int32) or index.ndim != 1 or not index.flags.c_contiguous:
                raise ValueError("%s must be a contiguous 1-d uint32 / int32 NumPy array" % name)
            mem, n, p = capi.HOST, index.shape[0], ctypes.c_void_p(index.ctypes.data)
        if mem != want_mem:
            raise ValueError("%s must live in the same memory space as the POI queue" % name)
        if n < rows:
            raise ValueError("%s needs room for %d entries (has %d)" % (name, rows, n))
        return p

    def split_reliable(self, pois, zncc_threshold_low, zncc_threshold_high, conv_criterion, reliable=None, reliable_offset=0):
        """``oc_hip_split_reliable``: order-preserving partition of a finished queue (examples/
        test_3d_reconstruction_sift_icgn2_regfit.cpp:216-229).  Returns ``(reliable, n_reliable, unreliable, index,
        n_unreliable)``: ``reliable[reliable_offset : reliable_offset + n_reliable]`` and ``unreliable[:n_unreliable]`` hold
        the records, ``index[:n_unreliable]`` the queue positions of the unreliable ones.  CUDA tensors stay on the device
        (the buffers are allocated with ``len(pois)`` records unless ``reliable`` is passed in).  The record type is the
        engine's (POI2D for 2D engines, POI3D for 3D ones), whatever the row width."""
        self._adopt_stream_of(pois)
        reliable_offset = int(reliable_offset)
        if reliable_offset < 0:
            raise ValueError("reliable_offset must not be negative")
        p, mem, stride = self._queue_buf(pois, "pois")
        n, width = pois.shape
        if _is_torch(pois):
            import torch
            if reliable is None:
                reliable = torch.empty((reliable_offset + n, width), dtype=torch.float32, device=pois.device)
            unreliable = torch.empty((n, width), dtype=torch.float32, device=pois.device)
            index = torch.empty((n,), dtype=torch.int32, device=pois.device)
            if not _is_torch(reliable) or reliable.device != pois.device:
                raise ValueError("reliable must be a CUDA tensor on the POI queue's device")
        else:
            if reliable is None:
                reliable = np.zeros((reliable_offset + n, width), dtype=np.float32)
            unreliable = np.zeros((n, width), dtype=np.float32)
            index = np.zeros((n,), dtype=np.uint32)
        rp, _, _ = self._queue_buf(reliable, "reliable", mem, reliable_offset + n, stride)
        up, _, _ = self._queue_buf(unreliable, "unreliable", mem, n, stride)
        ip = self._index_buf(index, "index", mem, n)
        n_rel, n_unr = ctypes.c_size_t(), ctypes.c_size_t()
        capi.check(capi.lib().oc_hip_split_reliable(self._h, p, n, stride, self._ndim, zncc_threshold_low, zncc_threshold_high, conv_criterion,
                                                    rp, reliable_offset, up, ip, ctypes.byref(n_rel), ctypes.byref(n_unr), mem))
        return reliable, n_rel.value, unreliable, index, n_unr.value

    def merge_recovered(self, pois, unreliable, index, n_unreliable, zncc_threshold_high, conv_criterion, reliable, reliable_offset):
        """``oc_hip_merge_recovered``: after RegionFit + ICGN over ``unreliable[:n_unreliable]`` -- the POIs that now pass go
        back into ``pois`` (at their ``index``) and to ``reliable[reliable_offset:]``, the rest move to the front of
        ``unreliable`` / ``index``.  Returns ``(n_recovered, n_remaining)``.  All four buffers must share one memory kind
        (all NumPy or all CUDA tensors on one device), dtype and row stride; an ``index`` entry outside ``pois`` is refused
        (``OC_HIP_ERR_INVALID``) instead of written through."""
        self._adopt_stream_of(pois)
        n_unreliable, reliable_offset = int(n_unreliable), int(reliable_offset)
        if n_unreliable < 0 or reliable_offset < 0:
            raise ValueError("n_unreliable and reliable_offset must not be negative")
        pp_, mem, stride = self._queue_buf(pois, "pois")
        up, _, _ = self._queue_buf(unreliable, "unreliable", mem, n_unreliable, stride)
        rp, _, _ = self._queue_buf(reliable, "reliable", mem, reliable_offset + n_unreliable, stride)
        ip = self._index_buf(index, "index", mem, n_unreliable)
        if mem == capi.DEVICE and not (pois.device == unreliable.device == reliable.device == index.device):
            raise ValueError("pois, unreliable, index and reliable must live on one device")
        n_rec, n_rem = ctypes.c_size_t(), ctypes.c_size_t()
        capi.check(capi.lib().oc_hip_merge_recovered(self._h, pp_, pois.shape[0], stride, self._ndim, up, ip, n_unreliable, zncc_threshold_high,
                                                     conv_criterion, rp, reliable_offset, ctypes.byref(n_rec), ctypes.byref(n_rem), mem))
        return n_rec.value, n_rem.value

    def compute_one(self, poi):
        assert poi.dtype == np.float32 and poi.flags.c_contiguous
        capi.check(capi.lib().oc_hip_compute_one(self._h, ctypes.c_void_p(poi.ctypes.data)))
        return poi

    def synchronize(self):
        capi.check(capi.lib().oc_hip_synchronize(self._h))

    # -- introspection ----------------------------------------------------------
    def read_field(self, name):
        ptr = ctypes.c_void_p()
        count = ctypes.c_size_t()
        capi.check(capi.lib().oc_hip_get_field(self._h, name.encode(), ctypes.byref(ptr), ctypes.byref(count)))
        out = np.empty(count.value, dtype=np.float32)
        capi.check(capi.lib().oc_hip_read_field(self._h, name.encode(), ctypes.c_void_p(out.ctypes.data), count.value))
        if name in ("lut", "lut_gx", "lut_gy"):
            # stored planar [k][y][x][l]; returned in the reference's order coef[y][x][k][l] (src/oc_cubic_bspline.cpp:123-129)
            h, w = self.shape
            return np.ascontiguousarray(out.reshape(4, h, w, 4).transpose(1, 2, 0, 3)).reshape(h, w, 16)
        return out.reshape(self.shape)

    def profile_enable(self, on=True):
        capi.check(capi.lib().oc_hip_profile_enable(self._h, 1 if on else 0))

    def profile_reset(self):
        capi.check(capi.lib().oc_hip_profile_reset(self._h))

    def profile_read(self):
        ms = ctypes.c_double()
        n = ctypes.c_long()
        capi.check(capi.lib().oc_hip_profile_read(self._h, ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value


class _IcgnMixin:
    def set_iteration(self, conv_criterion, stop_condition):
        capi.check(capi.lib().oc_hip_set_iteration(self._h, conv_criterion, stop_condition))

    def set_self_adaptive(self, is_self_adaptive):
        """DIC::setSelfAdaptive (src/oc_dic.cpp:34-37): per-POI subset radius from poi.subset_radius."""
        capi.check(capi.lib().oc_hip_set_self_adaptive(self._h, 1 if is_self_adaptive else 0))


class FFTCC2D(_Engine):
    """FFTCC2D(subset_radius_x, subset_radius_y, thread_number) -- src/oc_fftcc.h:54-68."""

    def __init__(self, subset_radius_x, subset_radius_y, thread_number=1, device=0):
        super().__init__(device)
        self.thread_number = thread_number  # kept for signature compatibility
        capi.check(capi.lib().oc_hip_fftcc2d_create(subset_radius_x, subset_radius_y, device, ctypes.byref(self._h)))


class ICGN2D1(_Engine, _IcgnMixin):
    """ICGN2D1(rx, ry, conv_criterion, stop_condition, thread_number) -- src/oc_icgn.h:45-77."""

    def __init__(self, subset_radius_x, subset_radius_y, conv_criterion, stop_condition, thread_number=1, device=0):
        super().__init__(device)
        self.thread_number = thread_number
        capi.check(capi.lib().oc_hip_icgn2d1_create(subset_radius_x, subset_radius_y, conv_criterion, stop_condition,
                                                    device, ctypes.byref(self._h)))


class ICGN2D2(_Engine, _IcgnMixin):
    """ICGN2D2(rx, ry, conv_criterion, stop_condition, thread_number) -- src/oc_icgn.h:100-132."""

    def __init__(self, subset_radius_x, subset_radius_y, conv_criterion, stop_condition, thread_number=1, device=0):
        super().__init__(device)
        self.thread_number = thread_number
        capi.check(capi.lib().oc_hip_icgn2d2_create(subset_radius_x, subset_radius_y, conv_criterion, stop_condition,
                                                    device, ctypes.byref(self._h)))


class NR2D1(_Engine, _IcgnMixin):
    """NR2D1(rx, ry, conv_criterion, stop_condition, thread_number) -- src/oc_nr.h, src/oc_nr.cpp:75-91."""

    def __init__(self, subset_radius_x, subset_radius_y, conv_criterion, stop_condition, thread_number=1, device=0):
        super().__init__(device)
        self.thread_number = thread_number
        capi.check(capi.lib().oc_hip_nr2d1_create(subset_radius_x, subset_radius_y, conv_criterion, stop_condition,
                                                  device, ctypes.byref(self._h)))


class _IclmMixin(_IcgnMixin):
    def set_damping(self, lambda_, alpha, beta):
        """ICLM2D*::setDamping(lambda, alpha, beta) (src/oc_iclm.cpp:114-119); defaults 100, 0.1, 10."""
        capi.check(capi.lib().oc_hip_set_damping(self._h, lambda_, alpha, beta))


class ICLM2D1(_Engine, _IclmMixin):
    """ICLM2D1(rx, ry, conv_criterion, stop_condition, thread_number) -- src/oc_iclm.h:56-85."""

    def __init__(self, subset_radius_x, subset_radius_y, conv_criterion, stop_condition, thread_number=1, device=0):
        super().__init__(device)
        self.thread_number = thread_number
        capi.check(capi.lib().oc_hip_iclm2d1_create(subset_radius_x, subset_radius_y, conv_criterion, stop_condition,
                                                    device, ctypes.byref(self._h)))


class ICLM2D2(_Engine, _IclmMixin):
    """ICLM2D2(rx, ry, conv_criterion, stop_condition, thread_number) -- src/oc_iclm.h:104-133."""

    def __init__(self, subset_radius_x, subset_radius_y, conv_criterion, stop_condition, thread_number=1, device=0):
        super().__init__(device)
        self.thread_number = thread_number
        capi.check(capi.lib().oc_hip_iclm2d2_create(subset_radius_x, subset_radius_y, conv_criterion, stop_condition,
                                                    device, ctypes.byref(self._h)))


class Strain:
    """Strain(subregion_radius, neighbor_number_min, thread_number) -- src/oc_strain.h:34-73, src/oc_strain.cpp:31-46.

    ``prepare(pois)`` builds the neighbour search over the queue's coordinates, ``compute(pois)`` writes the strain
    fields of every POI that can be fitted, in place (POI2D: exx, eyy, exy; POI3D: exx, eyy, ezz, exy, eyz, ezx).
    ``pois`` is a float32 (n, 25) / (n, 31) NumPy array (host path) or CUDA torch tensor (used in place)."""

    def __init__(self, subregion_radius, neighbor_number_min, thread_number=1, device=0):
        self._h = ctypes.c_void_p()
        self.thread_number = thread_number
        self._device = int(device)
        self._p = dict(radius=float(subregion_radius), nmin=int(neighbor_number_min), zncc=0.9, approx=1)
        capi.check(capi.lib().oc_hip_strain_create(self._p["radius"], self._p["nmin"], device, ctypes.byref(self._h)))

    def close(self):
        if self._h:
            capi.lib().oc_hip_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _push(self, **kw):
        p = dict(self._p, **kw)
        capi.check(capi.lib().oc_hip_strain_set(self._h, p["radius"], p["nmin"], p["zncc"], p["approx"]))
        self._p = p

    # src/oc_strain.cpp:72-95
    def set_subregion_radius(self, subregion_radius):
        self._push(radius=float(subregion_radius))

    def set_neighbor_min(self, neighbor_number_min):
        self._push(nmin=int(neighbor_number_min))

    def set_zncc_threshold(self, zncc_threshold):
        self._push(zncc=float(zncc_threshold))

    def set_approximation(self, approximation):
        """1 = Cauchy strain, 2 = Green strain."""
        self._push(approx=int(approximation))

    set_stream = _Engine.set_stream
    _adopt_stream_of = _Engine._adopt_stream_of
    _stream_pinned = False
    _auto_stream = None
    _on_private_stream = True

    def synchronize(self):
        capi.check(capi.lib().oc_hip_synchronize(self._h))

    @staticmethod
    def _queue(pois):
        if _is_torch(pois):
            p, mem, _ = _buf(pois)
            n, floats, stride = pois.shape[0], pois.shape[1], pois.stride(0) * 4
        else:
            if pois.dtype != np.float32 or not pois.flags.c_contiguous or pois.ndim != 2:
                raise ValueError("pois must be a C-contiguous float32 array of shape (n, 25) or (n, 31)")
            p, mem = ctypes.c_void_p(pois.ctypes.data), capi.HOST
            n, floats, stride = pois.shape[0], pois.shape[1], pois.strides[0]
        if floats not in (capi.POI2D_FLOATS, capi.POI3D_FLOATS):
            raise ValueError("POI records have 25 (POI2D) or 31 (POI3D) floats, got %d" % floats)
        return p, n, stride, (2 if floats == capi.POI2D_FLOATS else 3), mem

    def prepare(self, pois):
        self._adopt_stream_of(pois)
        p, n, stride, ndim, mem = self._queue(pois)
        capi.check(capi.lib().oc_hip_strain_prepare(self._h, p, n, stride, ndim, mem))

    def compute(self, pois):
        self._adopt_stream_of(pois)
        p, n, stride, ndim, mem = self._queue(pois)
        capi.check(capi.lib().oc_hip_strain_compute(self._h, p, n, stride, ndim, mem))
        return pois

    def profile_enable(self, on=True):
        capi.check(capi.lib().oc_hip_profile_enable(self._h, 1 if on else 0))

    def profile_reset(self):
        capi.check(capi.lib().oc_hip_profile_reset(self._h))

    def profile_read(self):
        ms = ctypes.c_double()
        n = ctypes.c_long()
        capi.check(capi.lib().oc_hip_profile_read(self._h, ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value


class RegionFit:
    """RegionFit2D / RegionFit3D(neighbor_search_radius, neighbor_number_min, thread_number) -- src/oc_region_fit.h.

    ``set_neighbor(reliable)`` + ``prepare()`` build the neighbour search over the reliable POIs (their displacements
    are snapshotted at ``prepare``); ``compute(pois)`` gives every POI of a second queue the plane fitted through the
    reliable POIs around it as its deformation, with ``zncc = 0`` -- ready for another ICGN pass.  2D or 3D by the
    record size of the queues."""

    def __init__(self, neighbor_search_radius, neighbor_number_min, thread_number=1, device=0):
        self._h = ctypes.c_void_p()
        self.thread_number = thread_number
        self._device = int(device)
        self._radius, self._nmin = float(neighbor_search_radius), int(neighbor_number_min)
        self._reliable = None
        capi.check(capi.lib().oc_hip_region_fit_create(self._radius, self._nmin, device, ctypes.byref(self._h)))

    close = Strain.close
    __del__ = Strain.__del__
    set_stream = Strain.set_stream
    _adopt_stream_of = _Engine._adopt_stream_of
    _stream_pinned = False
    _auto_stream = None
    _on_private_stream = True
    synchronize = Strain.synchronize
    profile_enable = Strain.profile_enable
    profile_reset = Strain.profile_reset
    profile_read = Strain.profile_read

    def set_search_radius(self, neighbor_search_radius):
        capi.check(capi.lib().oc_hip_region_fit_set(self._h, float(neighbor_search_radius), self._nmin))
        self._radius = float(neighbor_search_radius)

    def set_neighbor_min(self, neighbor_number_min):
        capi.check(capi.lib().oc_hip_region_fit_set(self._h, self._radius, int(neighbor_number_min)))
        self._nmin = int(neighbor_number_min)

    def set_neighbor(self, reliable_pois):
        self._reliable = reliable_pois

    def prepare(self):
        if self._reliable is None:
            raise ValueError("RegionFit.prepare: call set_neighbor(reliable_pois) first")
        self._adopt_stream_of(self._reliable)
        p, n, stride, ndim, mem = Strain._queue(self._reliable)
        capi.check(capi.lib().oc_hip_region_fit_prepare(self._h, p, n, stride, ndim, mem))

    def compute(self, pois):
        self._adopt_stream_of(pois)
        p, n, stride, ndim, mem = Strain._queue(pois)
        capi.check(capi.lib().oc_hip_region_fit_compute(self._h, p, n, stride, ndim, mem))
        return pois


class FFTCC3D(_Engine):
    """FFTCC3D(rx, ry, rz, thread_number) -- src/oc_fftcc.h:75-89."""
    _ndim = 3

    def __init__(self, subset_radius_x, subset_radius_y, subset_radius_z, thread_number=1, device=0):
        super().__init__(device)
        self.thread_number = thread_number
        capi.check(capi.lib().oc_hip_fftcc3d_create(subset_radius_x, subset_radius_y, subset_radius_z, device,
                                                    ctypes.byref(self._h)))


class ICGN3D1(_Engine, _IcgnMixin):
    """ICGN3D1(rx, ry, rz, conv_criterion, stop_condition, thread_number) -- src/oc_icgn.h:155-180."""
    _ndim = 3

    def __init__(self, subset_radius_x, subset_radius_y, subset_radius_z, conv_criterion, stop_condition,
                 thread_number=1, device=0):
        super().__init__(device)
        self.thread_number = thread_number
        capi.check(capi.lib().oc_hip_icgn3d1_create(subset_radius_x, subset_radius_y, subset_radius_z, conv_criterion,
                                                    stop_condition, device, ctypes.byref(self._h)))


def compute_chain(engines, pois):
    """Several engines over ONE queue, in order (``oc_hip_compute_chain``): ``compute_chain([fftcc, icgn], pois)`` does what
    ``fftcc.compute(pois); icgn.compute(pois)`` does -- same bits -- but a host (NumPy) queue crosses PCIe once in each
    direction instead of once per engine."""
    engines = list(engines)
    if not engines:
        raise ValueError("compute_chain needs at least one engine")
    floats = capi.POI2D_FLOATS if engines[0]._ndim == 2 else capi.POI3D_FLOATS
    for e in engines:
        e._adopt_stream_of(pois)
    if _is_torch(pois):
        p, mem, _ = _buf(pois)
        n, stride = pois.shape[0], pois.stride(0) * 4
        if pois.shape[1] < floats:
            raise ValueError("POI records need %d floats" % floats)
    else:
        if pois.dtype != np.float32 or not pois.flags.c_contiguous or pois.ndim != 2 or pois.shape[1] < floats:
            raise ValueError("pois must be a C-contiguous float32 array of shape (n, >=%d)" % floats)
        p, mem = ctypes.c_void_p(pois.ctypes.data), capi.HOST
        n, stride = pois.shape[0], pois.strides[0]
    handles = (ctypes.c_void_p * len(engines))(*[e._h for e in engines])
    capi.check(capi.lib().oc_hip_compute_chain(handles, len(engines), p, n, stride, mem))
    return pois


def make_pois2d(xs, ys):
    """Zero-initialised POI2D records (POI2D ctor, src/oc_poi.h:108-135)."""
    xs = np.asarray(xs, dtype=np.float32).ravel()
    pois = np.zeros((xs.size, capi.POI2D_FLOATS), dtype=np.float32)
    pois[:, 0] = xs
    pois[:, 1] = np.asarray(ys, dtype=np.float32).ravel()
    return pois


def make_pois3d(xs, ys, zs):
    xs = np.asarray(xs, dtype=np.float32).ravel()
    pois = np.zeros((xs.size, capi.POI3D_FLOATS), dtype=np.float32)
    pois[:, 0] = xs
    pois[:, 1] = np.asarray(ys, dtype=np.float32).ravel()
    pois[:, 2] = np.asarray(zs, dtype=np.float32).ravel()
    return pois
