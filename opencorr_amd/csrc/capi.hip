// capi.hip -- implementation of the C-ABI declared in include/opencorr_hip.h.
//
// Host-side engine objects: device-resident image pair, precomputed fields,
// POI staging, rocFFT plans, stream and profiling events.  No CPU compute path
// exists here: every compute call ends in HIP kernel launches or fails.
#include "../../include/opencorr_hip.h"

#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

// RCCL: types and enumerators only -- the library itself is loaded with dlopen on first use (struct Rccl), and a
// single-GPU host needs neither the library nor its development headers: without <rccl/rccl.h> the few declarations the
// binding uses are restated here (the stable NCCL 2.x C API: opaque communicator handle, ncclResult_t with ncclSuccess = 0,
// ncclUint8 = 1 in ncclDataType_t) and the version check against the loaded library is what guards them.
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#define OC_HIP_RCCL_HEADER 1
#else
#define OC_HIP_RCCL_HEADER 0
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1 } ncclDataType_t;
ncclResult_t ncclCommInitAll(ncclComm_t* comm, int ndev, const int* devlist);
ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclGroupStart();
ncclResult_t ncclGroupEnd();
ncclResult_t ncclCommDestroy(ncclComm_t comm);
const char* ncclGetErrorString(ncclResult_t result);
ncclResult_t ncclGetVersion(int* version);
}
#define NCCL_MAJOR 2
#endif

#include "oc_kernels.h"

namespace {

thread_local std::string g_last_error;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define OC_HIP_TRY(expr)                                                                              \
    do {                                                                                              \
        hipError_t err__ = (expr);                                                                    \
        if (err__ != hipSuccess)                                                                      \
            return fail(err__ == hipErrorOutOfMemory ? OC_HIP_ERR_NOMEM : OC_HIP_ERR_HIP,             \
                        "%s failed: %s (%s:%d)", #expr, hipGetErrorString(err__), __FILE__, __LINE__); \
    } while (0)

#define OC_FFT_TRY(expr)                                                                               \
    do {                                                                                               \
        rocfft_status st__ = (expr);                                                                   \
        if (st__ != rocfft_status_success)                                                             \
            return fail(OC_HIP_ERR_ROCFFT, "%s failed: rocfft_status %d (%s:%d)", #expr, (int)st__, __FILE__, __LINE__); \
    } while (0)

#define OC_TRY(expr)                  \
    do {                              \
        int rc__ = (expr);            \
        if (rc__ != OC_HIP_OK) return rc__; \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    // grow-only allocation
    int reserve(size_t n) {
        if (n <= bytes) return OC_HIP_OK;
        release();
        hipError_t err = hipMalloc(&p, n);
        if (err != hipSuccess) {
            p = nullptr;
            return fail(OC_HIP_ERR_NOMEM, "hipMalloc(%zu bytes) failed: %s", n, hipGetErrorString(err));
        }
        bytes = n;
        return OC_HIP_OK;
    }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// device copy of a reference/target image pair (2D) or volume pair (3D), row-major, x fastest
struct ImagePair {
    int ndim = 0;
    int dx = 0, dy = 0, dz = 1;  // width, height, depth
    DevBuf ref, tar;
    const float* ref_ext = nullptr;  // used in place when the caller handed device memory
    const float* tar_ext = nullptr;
    const float* ref_ptr() const { return ref_ext ? ref_ext : ref.as<float>(); }
    const float* tar_ptr() const { return tar_ext ? tar_ext : tar.as<float>(); }
    size_t count() const { return (size_t)dx * dy * dz; }
};

struct FftPlans {
    int n0 = 0, n1 = 0, n2 = 0;  // slowest .. fastest (n2 == 0 for 2D)
    size_t chunk = 0;
    rocfft_plan fwd = nullptr, inv = nullptr;
    rocfft_execution_info info_fwd = nullptr, info_inv = nullptr;
    DevBuf work_fwd, work_inv;
    void destroy() {
        if (fwd) rocfft_plan_destroy(fwd);
        if (inv) rocfft_plan_destroy(inv);
        if (info_fwd) rocfft_execution_info_destroy(info_fwd);
        if (info_inv) rocfft_execution_info_destroy(info_inv);
        fwd = inv = nullptr;
        info_fwd = info_inv = nullptr;
        chunk = 0;
    }
    ~FftPlans() { destroy(); }
};

std::once_flag g_rocfft_once;
void rocfft_init_once() {
    std::call_once(g_rocfft_once, [] { rocfft_setup(); });
}

}  // namespace

struct oc_hip_engine {
    int kind = 0;
    int device = 0;
    int rx = 0, ry = 0, rz = 0;
    float conv = 0.001f, stop = 10.f;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t order_ev = nullptr;  // orders the private stream behind the caller's default-stream work
    // Orders a newly chosen stream behind the work this engine left on the previous one.  On a CALLER-owned stream the
    // event is recorded at the end of every entry point that returns with work still enqueued (mark_tail): the caller may
    // destroy its stream at any time afterwards, and HIP aborts the process when a destroyed stream is handed to ANY
    // API call -- so a stream switch, destroy() and set_devices() never touch a caller's stream again, they wait for
    // this event instead.
    hipEvent_t switch_ev = nullptr;
    bool tail_marked = false;  // switch_ev holds the tail of this engine's work on the current (caller-owned) stream
    std::shared_ptr<ImagePair> img;
    DevBuf gx, gy, gz, coef;  // coef: 2D LUT (16 floats / px) or 3D coefficient volume
    DevBuf coef_gx, coef_gy;  // NR2D1: LUTs of the target gradients
    DevBuf tmp;               // scratch for layout conversion / the warped subvolumes of ICGN3D1
    DevBuf prefilter_tmp;     // second volume of the 3D B-spline prefilter (x pass -> here -> y pass -> coef -> z pass)
    bool ref_ready = false, tar_ready = false;
    DevBuf poi_stage, off_stage;
    DevBuf cursors;  // small device scratch (batch maxima)
    DevBuf perm, tiles, perm_slots;  // locality schedule of the ICGN2D queue (poi_order.hip)
    DevBuf setup_recs;               // icgn2d variant 8 (split launch shape): mean, norm, H^-1 per POI between the two kernels
    DevBuf split_scratch, split_tmp; // oc_hip_split_reliable / oc_hip_merge_recovered (poi_split.hip)
    // Strain (src/oc_strain.cpp:31-46: radius, min neighbours; ZNCC threshold 0.9, Cauchy approximation)
    float st_radius = 0.f, st_zncc = 0.9f;
    int st_nmin = 0, st_approx = 1, st_ndim = 0;
    size_t st_count = 0;  // queue length the grid was prepared for (0 = not prepared)
    ochip::StrainGrid st_grid{};
    DevBuf st_box, st_counts, st_start, st_cursor, st_slots, st_order, st_recs, st_fallback;
    float lm_lambda = 100.f, lm_alpha = 0.1f, lm_beta = 10.f;  // DampingParameter defaults, src/oc_iclm.h:33-38
    int icgn2d_tile_px = 128;  // 0 = visit the queue in its own order (64 until round 3; 128 suits the lockstep sweeps: 3.29 vs 3.34 ms)
    // FFTCC working set
    FftPlans fft;
    DevBuf win, freq, norms, flags;
    // kernel selection (oc_hip_set_tuning); every choice computes the same bits
    int icgn2d_variant = -1;  // -1 = automatic (run_icgn2d; MI355X sweeps), else the variant oc_hip_set_tuning chose
    bool self_adaptive = false;  // DIC::setSelfAdaptive
    int icgn2d_xcd = 1;
    // ICGN2D1 / ICGN2D2 / ICLM2D1 / ICLM2D2 / ICGN3D1: 0 = every multiply and add of the solver rounds on its own (oracle
    // OC_ORDER_LANES; the reference built for baseline x86-64), 1 = the per-sample multiply-adds are fused (oc_device.h
    // OC_FMA; oracle OC_ORDER_LANES_FMA) -- the only tuning key that changes result bits (by rounding, inside north_star's
    // tolerance: DESIGN.md section 3)
    int arith_fma = 0;
    int fftcc2d_fused = 1;    // single-kernel FFTCC2D when the window is 32 x 32
    int fftcc3d_fused = 1;    // single-kernel FFTCC3D for cubic windows of side 8 ... 64 (three kernels by size)
    int fftcc3d_planes_blocks = 0;  // persistent workgroups (= scratch volumes) of the plane-wise kernel; 0 = 256
    int fftcc3d_tile_vox = 64; // FFTCC3D single-kernel paths: queues of >= 2048 POIs are visited in cubic blocks of this many voxels (0 = queue order)
    int icgn3d_tile_vox = 64; // ICGN3D1: queues of >= 2048 POIs are visited in cubic blocks of this many voxels (0 = queue order; config E: 78.8 -> 75.5 ms, profiles/r4g_icgn3d1_ab_block_schedule.txt)
    int icgn3d_mapping = 0;   // ICGN3D1: 0 = sample s owned by thread s mod 512 (icgn3d.hip; oracle order OC_ORDER_LANES) -- the default:
                              // 1 = one half-wave per subvolume row (icgn3d_rows.hip; OC_ORDER_ROWS), built and measured in round 4:
                              // bit-exact against its own order, 12 - 25 % SLOWER -- kept as the A/B partner
    // host-queue pipeline (compute_host): the queue travels in chunks, copies of one chunk overlap the kernels of
    // its neighbours; one event per chunk orders the copy-out stream behind the kernels
    hipStream_t copy_stream = nullptr, copy_in_stream = nullptr;
    // icgn2d variant 8 with "icgn2d_split_chunks" >= 2: the set-up kernels run on this second stream, one or two chunks ahead
    // of the iteration kernels on the engine's stream, so that workgroups of both kinds are resident together
    hipStream_t aux_stream = nullptr;
    std::vector<hipEvent_t> split_ev;
    int icgn2d_split_chunks = 0;
    std::vector<hipEvent_t> chunk_done, chunk_in;
    size_t chunks_fed = 0;  // chunks whose kernels (and event) are enqueued; (size_t)-1: the feeder failed.  Guarded by feed_mu
    std::mutex feed_mu;
    std::condition_variable feed_cv;  // the copy-out thread sleeps here until the next chunk has been handed over
    int host_chunk = 65536;  // POIs per chunk ("host_chunk" tuning key; 0 = the whole queue at once)
    std::atomic<unsigned> single_calls{0};  // compute(POI*) calls on this engine (one hint on stderr when a caller loops over them)
    // device group (oc_hip_set_devices): this engine leads, replicas[i] is a full engine of the same kind on
    // group_devices[i + 1]; every setter, set_images, prepare and compute fans out
    std::vector<oc_hip_engine*> replicas;
    std::vector<int> group_devices;
    bool is_replica = false;
    int group_allgather = 0;     // DEVICE queues: leave the complete result queue in every member's mirror
    int group_force_rccl = 0;    // the all-gather goes through RCCL even for a group of ONE (a one-rank communicator)
    DevBuf group_mirror;         // full-size copy of a DEVICE queue (members other than the leader work in theirs)
    DevBuf group_off_mirror;
    size_t group_mirror_block = 0;  // bytes per member block of the last all-gathered queue
    hipEvent_t group_ev = nullptr;
    void* rccl_comm = nullptr;   // ncclComm_t of this member (group_allgather with distinct devices)
    // profiling
    bool prof = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    mutable std::mutex mu;

    bool is3d() const { return kind == OC_HIP_FFTCC3D || kind == OC_HIP_ICGN3D1; }
    bool is_iclm() const { return kind == OC_HIP_ICLM2D1 || kind == OC_HIP_ICLM2D2; }
    bool is_icgn2d() const { return kind == OC_HIP_ICGN2D1 || kind == OC_HIP_ICGN2D2 || is_iclm(); }
    bool is_icgn() const { return is_icgn2d() || kind == OC_HIP_ICGN3D1 || kind == OC_HIP_NR2D1; }
    size_t poi_bytes() const { return is3d() ? OC_HIP_POI3D_BYTES : OC_HIP_POI2D_BYTES; }
};

namespace {

void group_drop_comms(oc_hip_engine* e);  // RCCL communicators of a device group (defined with the group code)

int check_engine(const oc_hip_engine* e) {
    if (!e) return fail(OC_HIP_ERR_INVALID, "null engine handle");
    return OC_HIP_OK;
}

int activate(const oc_hip_engine* e) {
    OC_TRY(check_engine(e));
    OC_HIP_TRY(hipSetDevice(e->device));
    return OC_HIP_OK;
}

// Entry points make the engine's device current for the calling thread and give the caller's device back on the way
// out (a host that drives several GPUs from one thread -- torch with more than one device, say -- must not find its
// current device changed by a library call).
struct DeviceScope {
    int saved = -1;
    DeviceScope() {
        if (hipGetDevice(&saved) != hipSuccess) saved = -1;
    }
    ~DeviceScope() {
        int now = -1;
        if (saved >= 0 && hipGetDevice(&now) == hipSuccess && now != saved) (void)hipSetDevice(saved);
    }
    DeviceScope(const DeviceScope&) = delete;
    DeviceScope& operator=(const DeviceScope&) = delete;
};
#define OC_ACTIVATE(e)        \
    DeviceScope device_scope; \
    OC_TRY(activate(e))

int create_engine(int kind, int rx, int ry, int rz, float conv, float stop, int device, oc_hip_engine** out) {
    if (!out) return fail(OC_HIP_ERR_INVALID, "null output handle");
    *out = nullptr;
    if (rx < 1 || ry < 1 || ((kind == OC_HIP_FFTCC3D || kind == OC_HIP_ICGN3D1) && rz < 1))
        return fail(OC_HIP_ERR_INVALID, "subset radius must be >= 1 (got %d, %d, %d)", rx, ry, rz);
    int ndev = 0;
    hipError_t err = hipGetDeviceCount(&ndev);
    if (err != hipSuccess || ndev <= 0)
        return fail(OC_HIP_ERR_HIP, "no usable HIP device: %s", err == hipSuccess ? "device count is 0" : hipGetErrorString(err));
    if (device < 0 || device >= ndev) return fail(OC_HIP_ERR_INVALID, "device %d out of range [0,%d)", device, ndev);
    DeviceScope device_scope;
    OC_HIP_TRY(hipSetDevice(device));
    std::unique_ptr<oc_hip_engine> e(new oc_hip_engine);
    e->kind = kind;
    e->device = device;
    e->rx = rx;
    e->ry = ry;
    e->rz = rz;
    e->conv = conv;
    e->stop = stop;
    // MI355X sweep (profiles/r01b_icgn2d*_variant_sweep.json): 12 DoF keeps more registers live, so
    // the single-wave G = 4 variant wins there
    OC_HIP_TRY(hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking));
    e->stream = e->own_stream;
    *out = e.release();
    return OC_HIP_OK;
}

// The engine's private stream is non-blocking: nothing the caller enqueued is ordered against it.  Inputs that
// live on the device (OC_HIP_DEVICE queues, offsets, images used in place) are usually produced on the legacy
// default stream -- hipMemcpy, plain launches, torch unless told otherwise -- so every entry point that enqueues work
// on the private stream first makes it wait for what the default stream holds at that moment.  Producers on OTHER
// streams must be complete, or be named with oc_hip_set_stream (documented in opencorr_hip.h).
int order_after_default_stream(oc_hip_engine* e) {
    if (e->stream != e->own_stream) return OC_HIP_OK;  // caller-chosen stream: stream order is the caller's
    if (!e->order_ev) OC_HIP_TRY(hipEventCreateWithFlags(&e->order_ev, hipEventDisableTiming));
    OC_HIP_TRY(hipEventRecord(e->order_ev, nullptr));
    OC_HIP_TRY(hipStreamWaitEvent(e->own_stream, e->order_ev, 0));
    return OC_HIP_OK;
}

// See oc_hip_engine::switch_ev.  Entry points that return with work enqueued on a caller-owned stream end with this.
int mark_tail(oc_hip_engine* e) {
    if (e->stream == e->own_stream) return OC_HIP_OK;  // the engine's own stream can always be asked later
    if (!e->switch_ev) OC_HIP_TRY(hipEventCreateWithFlags(&e->switch_ev, hipEventDisableTiming));
    OC_HIP_TRY(hipEventRecord(e->switch_ev, e->stream));
    e->tail_marked = true;
    return OC_HIP_OK;
}

// Scope guard of the entry points that enqueue work: whatever way the function leaves -- also an error exit after some
// kernels or copies were already enqueued on a CALLER-owned stream -- the engine's tail event covers that work, so a later
// set_stream / destroy / set_devices drains it through the event instead of relying on hipFree's implicit device
// synchronisation.  Quiet: a failure to record never replaces the error message the function itself reports, and the
// success paths' own mark_tail (finish_device_call) simply records the same point twice.
struct TailGuard {
    oc_hip_engine* e;
    explicit TailGuard(oc_hip_engine* engine) : e(engine) {}
    TailGuard(const TailGuard&) = delete;
    TailGuard& operator=(const TailGuard&) = delete;
    ~TailGuard() {
        if (!e || e->stream == e->own_stream) return;
        if (!e->switch_ev && hipEventCreateWithFlags(&e->switch_ev, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            return;
        }
        if (hipEventRecord(e->switch_ev, e->stream) == hipSuccess) e->tail_marked = true;
        else (void)hipGetLastError();
    }
};

// Host-side wait for everything this engine has enqueued, without ever touching a caller's (possibly destroyed) stream.
void drain_engine(oc_hip_engine* e) {
    if (e->stream == e->own_stream) {
        if (e->own_stream) (void)hipStreamSynchronize(e->own_stream);
    } else {
        if (e->tail_marked && e->switch_ev) (void)hipEventSynchronize(e->switch_ev);
        if (e->own_stream) (void)hipStreamSynchronize(e->own_stream);
    }
    (void)hipGetLastError();
}

void clear_events(oc_hip_engine* e) {
    for (auto& ev : e->events) {
        (void)hipEventDestroy(ev.first);
        (void)hipEventDestroy(ev.second);
    }
    e->events.clear();
}

struct ProfScope {
    oc_hip_engine* e;
    hipEvent_t stop = nullptr;
    explicit ProfScope(oc_hip_engine* e_) : e(e_) {
        if (!e->prof) return;
        hipEvent_t start = nullptr;
        if (hipEventCreate(&start) != hipSuccess) return;
        if (hipEventCreate(&stop) != hipSuccess) { (void)hipEventDestroy(start); stop = nullptr; return; }
        (void)hipEventRecord(start, e->stream);
        e->events.emplace_back(start, stop);
    }
    ~ProfScope() {
        if (stop) (void)hipEventRecord(stop, e->stream);
    }
};

// ---------------------------------------------------------------------------
// FFTCC2D pipeline
// ---------------------------------------------------------------------------
size_t fftcc_chunk_limit() {
    const char* s = getenv("OC_HIP_FFTCC_CHUNK");
    if (s && *s) {
        long v = atol(s);
        if (v > 0) return (size_t)v;
    }
    return 32768;
}

int ensure_fft(oc_hip_engine* e, size_t chunk) {
    // The reference plans fftwf_plan_dft_r2c_2d(width, height, ...) over a buffer filled
    // [r*width + c] (src/oc_fftcc.cpp:40-42, 204-221): n0 = 2rx is the slow dimension of
    // the transform, n1 = 2ry the fast one; in 3D fftwf_plan_dft_r2c_3d(dim_x, dim_y, dim_z)
    // over [(i*dim_y + j)*dim_x + k] (:68-70, 349-360): n0 = 2rx, n1 = 2ry, n2 = 2rz (fastest).
    // rocFFT takes lengths fastest-first.
    const int n0 = 2 * e->rx, n1 = 2 * e->ry, n2 = e->is3d() ? 2 * e->rz : 0;
    FftPlans& f = e->fft;
    if (f.fwd && f.n0 == n0 && f.n1 == n1 && f.n2 == n2 && f.chunk == chunk) return OC_HIP_OK;
    rocfft_init_once();
    f.destroy();
    const size_t lengths2[2] = {(size_t)n1, (size_t)n0};
    const size_t lengths3[3] = {(size_t)n2, (size_t)n1, (size_t)n0};
    const size_t* lengths = n2 ? lengths3 : lengths2;
    const size_t dims = n2 ? 3 : 2;
    OC_FFT_TRY(rocfft_plan_create(&f.fwd, rocfft_placement_notinplace, rocfft_transform_type_real_forward,
                                  rocfft_precision_single, dims, lengths, 2 * chunk, nullptr));
    OC_FFT_TRY(rocfft_plan_create(&f.inv, rocfft_placement_notinplace, rocfft_transform_type_real_inverse,
                                  rocfft_precision_single, dims, lengths, chunk, nullptr));
    OC_FFT_TRY(rocfft_execution_info_create(&f.info_fwd));
    OC_FFT_TRY(rocfft_execution_info_create(&f.info_inv));
    size_t wf = 0, wi = 0;
    OC_FFT_TRY(rocfft_plan_get_work_buffer_size(f.fwd, &wf));
    OC_FFT_TRY(rocfft_plan_get_work_buffer_size(f.inv, &wi));
    if (wf) {
        OC_TRY(f.work_fwd.reserve(wf));
        OC_FFT_TRY(rocfft_execution_info_set_work_buffer(f.info_fwd, f.work_fwd.p, wf));
    }
    if (wi) {
        OC_TRY(f.work_inv.reserve(wi));
        OC_FFT_TRY(rocfft_execution_info_set_work_buffer(f.info_inv, f.work_inv.p, wi));
    }
    f.n0 = n0;
    f.n1 = n1;
    f.n2 = n2;
    f.chunk = chunk;
    return OC_HIP_OK;
}

int run_fftcc2d(oc_hip_engine* e, float* d_pois, int stride_f, size_t count) {
    if (!e->img || e->img->ndim != 2) return fail(OC_HIP_ERR_INVALID, "FFTCC2D: set_images2d has not been called");
    const ImagePair& im = *e->img;
    // ("fftcc2d_fused" = 2: the generic NR x NC kernel also for 32 x 32 windows -- an A/B switch)
    const bool fused32 = e->fftcc2d_fused != 2 && ochip::fftcc2d_fused_supported(e->rx, e->ry);
    const bool rect = ochip::fftcc2d_rect_supported(e->rx, e->ry);   // rectangular windows without an instantiation (fftcc2d_rect.hip)
    if (e->fftcc2d_fused && (fused32 || rect || ochip::fftcc2d_fusedn_supported(e->rx, e->ry))) {
        ochip::Fftcc2dParams P = {im.ref_ptr(), im.tar_ptr(), im.dy, im.dx, e->rx, e->ry};
        ProfScope prof(e);
        const size_t kMaxBatch = 1u << 30;
        for (size_t first = 0; first < count; first += kMaxBatch) {
            const size_t n = (count - first) < kMaxBatch ? (count - first) : kMaxBatch;
            float* q = d_pois + first * (size_t)stride_f;
            hipError_t err = fused32 ? ochip::launch_fftcc2d_fused(P, q, stride_f, n, e->icgn2d_xcd != 0, e->stream)
                             : rect  ? ochip::launch_fftcc2d_rect(P, q, stride_f, n, e->icgn2d_xcd != 0, e->stream)
                                     : ochip::launch_fftcc2d_fusedn(P, q, stride_f, n, e->icgn2d_xcd != 0, e->stream);
            if (err != hipSuccess) return fail(OC_HIP_ERR_HIP, "fused FFTCC2D launch failed: %s", hipGetErrorString(err));
        }
        return OC_HIP_OK;
    }
    const size_t chunk = count < fftcc_chunk_limit() ? count : fftcc_chunk_limit();
    if (chunk == 0) return OC_HIP_OK;
    OC_TRY(ensure_fft(e, chunk));
    const size_t M = (size_t)4 * e->rx * e->ry;                   // 2rx * 2ry
    const size_t F = (size_t)(2 * e->rx) * (size_t)(e->ry + 1);   // n0 * (n1/2 + 1)
    OC_TRY(e->win.reserve(2 * chunk * M * sizeof(float)));
    OC_TRY(e->freq.reserve(2 * chunk * F * sizeof(float2)));
    OC_TRY(e->norms.reserve(2 * chunk * sizeof(float)));
    OC_TRY(e->flags.reserve(chunk * sizeof(int)));
    OC_FFT_TRY(rocfft_execution_info_set_stream(e->fft.info_fwd, e->stream));
    OC_FFT_TRY(rocfft_execution_info_set_stream(e->fft.info_inv, e->stream));
    ochip::Fftcc2dParams P = {im.ref_ptr(), im.tar_ptr(), im.dy, im.dx, e->rx, e->ry};
    float* ref_win = e->win.as<float>();
    float* tar_win = ref_win + chunk * M;
    float2* ref_freq = e->freq.as<float2>();
    float2* tar_freq = ref_freq + chunk * F;
    ProfScope prof(e);
    for (size_t first = 0; first < count; first += chunk) {
        const size_t n = (count - first) < chunk ? (count - first) : chunk;
        float* pois = d_pois + first * (size_t)stride_f;
        if (n < chunk) {
            // the plans are built for a full chunk: clear the tail so stale windows stay finite
            OC_HIP_TRY(hipMemsetAsync(ref_win + n * M, 0, (chunk - n) * M * sizeof(float), e->stream));
            OC_HIP_TRY(hipMemsetAsync(tar_win + n * M, 0, (chunk - n) * M * sizeof(float), e->stream));
        }
        OC_HIP_TRY(ochip::launch_fftcc2d_gather(P, pois, stride_f, n, ref_win, tar_win, e->norms.as<float>(),
                                                e->flags.as<int>(), e->stream));
        void* in_fwd[1] = {ref_win};
        void* out_fwd[1] = {ref_freq};
        OC_FFT_TRY(rocfft_execute(e->fft.fwd, in_fwd, out_fwd, e->fft.info_fwd));
        OC_HIP_TRY(ochip::launch_fftcc_conjmul(ref_freq, tar_freq, ref_freq, n * F, e->stream));
        void* in_inv[1] = {ref_freq};
        void* out_inv[1] = {ref_win};  // correlation surfaces overwrite the reference windows
        OC_FFT_TRY(rocfft_execute(e->fft.inv, in_inv, out_inv, e->fft.info_inv));
        OC_HIP_TRY(ochip::launch_fftcc2d_argmax(P, ref_win, e->norms.as<float>(), e->flags.as<int>(), pois, stride_f,
                                                n, e->stream));
    }
    return OC_HIP_OK;
}

// Locality schedule of a 2D queue (poi_order.hip): worth four tiny kernels once the queue is much larger than what
// is in flight.  *perm = nullptr when the queue is short or the schedule is switched off.
int tile_order(oc_hip_engine* e, const float* pois, int stride_f, size_t n, const unsigned** perm) {
    *perm = nullptr;
    if (e->icgn2d_tile_px <= 0 || n < 16384) return OC_HIP_OK;
    const ImagePair& im = *e->img;
    OC_TRY(e->perm.reserve(n * sizeof(unsigned)));
    OC_TRY(e->tiles.reserve(ochip::poi2d_tile_count(im.dy, im.dx, e->icgn2d_tile_px) * sizeof(unsigned)));
    OC_TRY(e->perm_slots.reserve(n * sizeof(unsigned)));
    OC_HIP_TRY(ochip::launch_poi2d_tile_order(pois, stride_f, n, im.dy, im.dx, e->icgn2d_tile_px, e->tiles.as<unsigned>(),
                                              e->perm_slots.as<unsigned>(), e->perm.as<unsigned>(), e->stream));
    *perm = e->perm.as<unsigned>();
    return OC_HIP_OK;
}

// The 2D solvers address image-sized arrays with 32-bit byte offsets (buffer resources, dic2d_device.h): a plane of
// the coefficient table is 16 B per pixel and must stay below 4 GiB (ICGN2D / ICLM: one descriptor per plane ->
// 2^28 pixels; NR2D1: one descriptor per table -> 2^26 pixels), and the row index times the width goes through a
// 24-bit multiply (width * 4 < 2^24).  Larger images are refused instead of wrapping silently.
int check_image2d_limits(const char* who, const ImagePair& im, unsigned long long max_pixels) {
    if ((unsigned long long)im.dy * (unsigned long long)im.dx > max_pixels || im.dx >= (1 << 22) || im.dy >= (1 << 22))
        return fail(OC_HIP_ERR_UNSUPPORTED, "%s: image %d x %d exceeds the engine's limit of %llu pixels (width, height < 2^22)", who,
                    im.dx, im.dy, max_pixels);
    return OC_HIP_OK;
}

// ---------------------------------------------------------------------------
// ICGN2D
// ---------------------------------------------------------------------------
int run_icgn2d(oc_hip_engine* e, float* d_pois, int stride_f, size_t count, const float* d_offsets) {
    if (!e->img || e->img->ndim != 2) return fail(OC_HIP_ERR_INVALID, "ICGN2D: set_images2d has not been called");
    if (!e->ref_ready || !e->tar_ready)
        return fail(OC_HIP_ERR_INVALID, "ICGN2D: prepare() has not been called since the last set_images");
    const ImagePair& im = *e->img;
    OC_TRY(check_image2d_limits("ICGN2D", im, 1ull << 28));
    const int dof = (e->kind == OC_HIP_ICGN2D1 || e->kind == OC_HIP_ICLM2D1) ? 6 : 12;
    const bool lm = e->is_iclm();
    int rx = e->rx, ry = e->ry;
    if (e->self_adaptive) {
        // every POI brings its own radius (src/oc_icgn.cpp:152-158): size the on-chip arrays for the largest
        OC_TRY(e->cursors.reserve(2 * sizeof(int)));
        OC_HIP_TRY(ochip::launch_poi2d_max_radius(d_pois, stride_f, count, e->cursors.as<int>(), e->stream));
        int mx[2] = {0, 0};
        OC_HIP_TRY(hipMemcpyAsync(mx, e->cursors.p, sizeof(mx), hipMemcpyDeviceToHost, e->stream));
        OC_HIP_TRY(hipStreamSynchronize(e->stream));
        rx = mx[0] > 0 ? mx[0] : 1;
        ry = mx[1] > 0 ? mx[1] : 1;
        if (rx > 4096 || ry > 4096) return fail(OC_HIP_ERR_UNSUPPORTED, "ICGN2D: self-adaptive subset radius %d x %d is not plausible", rx, ry);
    }
    ochip::Icgn2dParams P = {im.ref_ptr(), e->gx.as<float>(), e->gy.as<float>(), e->coef.as<float>(),
                             im.dy,        im.dx,              rx,                ry,
                             e->conv,      e->stop,            d_offsets,         nullptr,
                             e->self_adaptive ? 1 : 0,
                             lm ? std::log((double)e->lm_lambda) : 0.0,
                             e->lm_alpha,  e->lm_beta,         e->arith_fma,      nullptr};
    const long long N = (2LL * rx + 1) * (2LL * ry + 1);
    // fall back to the LDS-light single-wave variant when the tuned one cannot hold the subset
    int variant = e->icgn2d_variant;
    if (variant < 0) {
        // ICGN2D1: the coordinate-table variant at 6 waves per SIMD while three of its workgroups fit a CU's LDS (subsets up
        // to 19 passes of 64 samples, i.e. 35 x 34), the LDS-light 4-wave workgroups beyond; ICGN2D2: see below
        // (queues that fill the chip only a couple of times over finish sooner in the finer-grained 4-wave workgroups:
        // config A, 10 000 POIs, 0.33 against 0.38 ms)
        const long long passes = ((2LL * rx + 1) * (2LL * ry + 1) + 63) / 64;
        // ICGN2D2: the table variant in 8-wave workgroups while two of them fit a CU (up to 28 passes: 41 x 41), measured
        // 3.65 against 3.81 ms on config C (profiles/r03q_icgn2d2_variant_ab_configC.json)
        variant = dof == 12 ? (passes <= 28 && count >= 32768 ? 4 : 3) : (passes <= 19 && count >= 32768 ? 5 : 2);
    }
    // per-POI radii: no shared coordinate table, and the per-wave arrays are sized for the LARGEST subset of the batch -- the
    // target-array-only shape (variant 7) keeps four workgroups on a CU where variant 2 holds two (bench.py paths_8f_row1)
    if (e->self_adaptive && (e->icgn2d_variant < 0 || ochip::icgn2d_variant_uses_table(variant))) variant = 7;
    if (lm) variant = 1;  // the IC-LM launch shape has the LDS footprint of variant 1
    // variant 9 = icgn2d_band.hip (A/B build only: the workgroup's band of the table staged in LDS, the warped subset in registers --
    // bit-exact, measured 1.7 x slower than variant 5): one radius per launch and at most the passes it unrolls
#if OC_BUILD_AB
    const bool band = variant == 9 && !lm && !e->self_adaptive && ochip::icgn2d_band_supported(dof, rx, ry);
#else
    const bool band = false;
#endif
    if (variant == 9 && !band) variant = dof == 12 ? 3 : 2;
    if (!band && N > ochip::icgn2d_max_samples(variant)) variant = 1;
    if (!band && N > ochip::icgn2d_max_samples(variant))
        return fail(OC_HIP_ERR_UNSUPPORTED, "ICGN2D%d: subset %dx%d (%lld samples) exceeds the on-chip limit of %d samples",
                    dof == 6 ? 1 : 2, 2 * rx + 1, 2 * ry + 1, N, ochip::icgn2d_max_samples(variant));
    if (variant == 8 && !lm) {
        OC_TRY(e->setup_recs.reserve(count * (size_t)ochip::icgn2d_setup_record_floats(dof) * sizeof(float)));
        P.setup = e->setup_recs.as<float>();
    }
    // one wave per POI; grid.x is limited to 2^31-1
    const size_t kMaxGrid = 1u << 30;
    for (size_t first = 0; first < count; first += kMaxGrid) {
        const size_t n = (count - first) < kMaxGrid ? (count - first) : kMaxGrid;
        float* pois = d_pois + first * (size_t)stride_f;
        if (d_offsets) P.offsets = d_offsets + 2 * first;
        if (P.setup) P.setup = e->setup_recs.as<float>() + first * (size_t)ochip::icgn2d_setup_record_floats(dof);
        OC_TRY(tile_order(e, pois, stride_f, n, &P.perm));
        ProfScope prof(e);  // the solver kernel alone (what rocprofv3 reports for it)
        hipError_t err;
        if (variant == 8 && !lm && e->icgn2d_split_chunks >= 2 && n >= 16384) {
            // the split launch shape as a two-stream pipeline over chunks of the visiting order: set-up kernel of chunk c on
            // the auxiliary stream (at most two chunks ahead), iteration kernel of chunk c on the engine's stream behind it
            const size_t C = (size_t)e->icgn2d_split_chunks;
            const size_t m = ((n + C - 1) / C + 7) / 8 * 8;
            if (!e->aux_stream) OC_HIP_TRY(hipStreamCreateWithFlags(&e->aux_stream, hipStreamNonBlocking));
            while (e->split_ev.size() < 2 * C + 1) {
                hipEvent_t ev;
                OC_HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
                e->split_ev.push_back(ev);
            }
            OC_HIP_TRY(hipEventRecord(e->split_ev[2 * C], e->stream));         // the queue, its visiting order, the previous call
            OC_HIP_TRY(hipStreamWaitEvent(e->aux_stream, e->split_ev[2 * C], 0));
            const size_t sf = (size_t)ochip::icgn2d_setup_record_floats(dof);
            for (size_t c = 0; c * m < n; c++) {
                const size_t lo = c * m, cnt = (n - lo) < m ? (n - lo) : m;
                ochip::Icgn2dParams Pc = P;
                float* pc = pois;
                if (P.perm) {
                    Pc.perm = P.perm + lo;   // the records, offsets and set-up records stay addressed by POI index
                } else {
                    pc = pois + lo * (size_t)stride_f;
                    if (Pc.offsets) Pc.offsets += 2 * lo;
                    Pc.setup += lo * sf;
                }
                if (c >= 2) OC_HIP_TRY(hipStreamWaitEvent(e->aux_stream, e->split_ev[2 * (c - 2) + 1], 0));
                err = dof == 6 ? ochip::launch_icgn2d1(Pc, pc, stride_f, cnt, 8, e->icgn2d_xcd != 0, e->aux_stream, 1)
                               : ochip::launch_icgn2d2(Pc, pc, stride_f, cnt, 8, e->icgn2d_xcd != 0, e->aux_stream, 1);
                if (err != hipSuccess) return fail(OC_HIP_ERR_HIP, "ICGN2D set-up kernel launch failed: %s", hipGetErrorString(err));
                OC_HIP_TRY(hipEventRecord(e->split_ev[2 * c], e->aux_stream));
                OC_HIP_TRY(hipStreamWaitEvent(e->stream, e->split_ev[2 * c], 0));
                err = dof == 6 ? ochip::launch_icgn2d1(Pc, pc, stride_f, cnt, 8, e->icgn2d_xcd != 0, e->stream, 2)
                               : ochip::launch_icgn2d2(Pc, pc, stride_f, cnt, 8, e->icgn2d_xcd != 0, e->stream, 2);
                if (err != hipSuccess) return fail(OC_HIP_ERR_HIP, "ICGN2D iteration kernel launch failed: %s", hipGetErrorString(err));
                OC_HIP_TRY(hipEventRecord(e->split_ev[2 * c + 1], e->stream));
            }
            continue;
        }
        if (lm)
            err = dof == 6 ? ochip::launch_iclm2d1(P, pois, stride_f, n, e->icgn2d_xcd != 0, e->stream)
                           : ochip::launch_iclm2d2(P, pois, stride_f, n, e->icgn2d_xcd != 0, e->stream);
#if OC_BUILD_AB
        else if (band)
            err = dof == 6 ? ochip::launch_icgn2d1_band(P, pois, stride_f, n, e->icgn2d_xcd != 0, e->stream)
                           : ochip::launch_icgn2d2_band(P, pois, stride_f, n, e->icgn2d_xcd != 0, e->stream);
#endif
        else
            err = dof == 6 ? ochip::launch_icgn2d1(P, pois, stride_f, n, variant, e->icgn2d_xcd != 0, e->stream)
                           : ochip::launch_icgn2d2(P, pois, stride_f, n, variant, e->icgn2d_xcd != 0, e->stream);
        if (err != hipSuccess) return fail(OC_HIP_ERR_HIP, "ICGN2D kernel launch failed: %s", hipGetErrorString(err));
    }
    return OC_HIP_OK;
}

// ---------------------------------------------------------------------------
// NR2D1
// ---------------------------------------------------------------------------
int run_nr2d1(oc_hip_engine* e, float* d_pois, int stride_f, size_t count) {
    if (!e->img || e->img->ndim != 2) return fail(OC_HIP_ERR_INVALID, "NR2D1: set_images2d has not been called");
    if (!e->tar_ready) return fail(OC_HIP_ERR_INVALID, "NR2D1: prepare() has not been called since the last set_images");
    const ImagePair& im = *e->img;
    OC_TRY(check_image2d_limits("NR2D1", im, 1ull << 26));
    const long long N = (2LL * e->rx + 1) * (2LL * e->ry + 1);
    if (N > ochip::nr2d1_max_samples())
        return fail(OC_HIP_ERR_UNSUPPORTED, "NR2D1: subset %dx%d (%lld samples) exceeds the on-chip limit of %d samples",
                    2 * e->rx + 1, 2 * e->ry + 1, N, ochip::nr2d1_max_samples());
    ochip::Nr2dParams P = {im.ref_ptr(), e->coef.as<float>(), e->coef_gx.as<float>(), e->coef_gy.as<float>(),
                           im.dy,        im.dx,               e->rx,                  e->ry,
                           e->conv,      e->stop,             nullptr};
    const size_t kMaxGrid = 1u << 30;
    for (size_t first = 0; first < count; first += kMaxGrid) {
        const size_t n = (count - first) < kMaxGrid ? (count - first) : kMaxGrid;
        OC_TRY(tile_order(e, d_pois + first * (size_t)stride_f, stride_f, n, &P.perm));
        ProfScope prof(e);
        hipError_t err = ochip::launch_nr2d1(P, d_pois + first * (size_t)stride_f, stride_f, n, e->stream);
        if (err != hipSuccess) return fail(OC_HIP_ERR_HIP, "NR2D1 kernel launch failed: %s", hipGetErrorString(err));
    }
    return OC_HIP_OK;
}

// ---------------------------------------------------------------------------
// FFTCC3D pipeline / ICGN3D1
// ---------------------------------------------------------------------------
size_t fftcc3d_chunk_limit() {
    const char* s = getenv("OC_HIP_FFTCC3D_CHUNK");
    if (s && *s) {
        long v = atol(s);
        if (v > 0) return (size_t)v;
    }
    return 1024;
}

// visiting order of a POI3D queue in compact cubic blocks (poi_order.hip): FFTCC3D's single-kernel paths and ICGN3D1
int tile_order3d(oc_hip_engine* e, const float* d_pois, int stride_f, size_t count, int tile_vox, const unsigned** perm) {
    *perm = nullptr;
    if (tile_vox <= 0 || count < 2048 || count > 0xffffffffull) return OC_HIP_OK;
    const ImagePair& im = *e->img;
    OC_TRY(e->perm.reserve(count * sizeof(unsigned)));
    OC_TRY(e->perm_slots.reserve(count * sizeof(unsigned)));
    OC_TRY(e->tiles.reserve(ochip::poi3d_tile_count(im.dz, im.dy, im.dx, tile_vox) * sizeof(unsigned)));
    OC_HIP_TRY(ochip::launch_poi3d_tile_order(d_pois, stride_f, count, im.dz, im.dy, im.dx, tile_vox, e->tiles.as<unsigned>(),
                                              e->perm_slots.as<unsigned>(), e->perm.as<unsigned>(), e->stream));
    *perm = e->perm.as<unsigned>();
    return OC_HIP_OK;
}

int run_fftcc3d(oc_hip_engine* e, float* d_pois, int stride_f, size_t count) {
    if (!e->img || e->img->ndim != 3) return fail(OC_HIP_ERR_INVALID, "FFTCC3D: set_images3d has not been called");
    const ImagePair& im = *e->img;
    const bool fused32 = ochip::fftcc3d_fused_supported(e->rx, e->ry, e->rz);
    if (e->fftcc3d_fused && ochip::fftcc3d_planes_supported(e->rx, e->ry, e->rz)) {
        // cubes too large for the chip: persistent workgroups, each with a private complex N^3 scratch volume (fftcc3d_planes.hip)
        ochip::Fftcc3dParams P = {im.ref_ptr(), im.tar_ptr(), im.dz, im.dy, im.dx, e->rx, e->ry, e->rz};
        int blocks = e->fftcc3d_planes_blocks > 0 ? e->fftcc3d_planes_blocks : 256;
        blocks = (blocks + 7) / 8 * 8;
        if ((size_t)blocks > (count + 7) / 8 * 8) blocks = (int)((count + 7) / 8 * 8);
        OC_TRY(e->win.reserve(ochip::fftcc3d_planes_scratch_bytes(e->rx, blocks)));
        OC_TRY(tile_order3d(e, d_pois, stride_f, count, e->fftcc3d_tile_vox, &P.perm));
        ProfScope prof(e);
        hipError_t err = ochip::launch_fftcc3d_planes(P, d_pois, stride_f, count, e->win.p, blocks, e->stream);
        if (err != hipSuccess) return fail(OC_HIP_ERR_HIP, "plane-wise FFTCC3D kernel launch failed: %s", hipGetErrorString(err));
        return OC_HIP_OK;
    }
    const bool box = ochip::fftcc3d_box_supported(e->rx, e->ry, e->rz);   // non-cubic windows (fftcc3d_box.hip)
    if (e->fftcc3d_fused && (fused32 || box || ochip::fftcc3d_fusedn_supported(e->rx, e->ry, e->rz))) {
        ochip::Fftcc3dParams P = {im.ref_ptr(), im.tar_ptr(), im.dz, im.dy, im.dx, e->rx, e->ry, e->rz};
        if (count <= (1u << 30)) OC_TRY(tile_order3d(e, d_pois, stride_f, count, e->fftcc3d_tile_vox, &P.perm));
        ProfScope prof(e);
        const size_t kMaxGrid = 1u << 30;
        for (size_t first = 0; first < count; first += kMaxGrid) {
            const size_t n = (count - first) < kMaxGrid ? (count - first) : kMaxGrid;
            float* q = d_pois + first * (size_t)stride_f;
            if (fused32) OC_TRY(e->flags.reserve(n));
            hipError_t err = fused32 ? ochip::launch_fftcc3d_fused(P, q, stride_f, n, e->icgn2d_xcd != 0, e->flags.as<unsigned char>(), e->stream)
                             : box   ? ochip::launch_fftcc3d_box(P, q, stride_f, n, e->icgn2d_xcd != 0, e->stream)
                                     : ochip::launch_fftcc3d_fusedn(P, q, stride_f, n, e->icgn2d_xcd != 0, e->stream);
            if (err != hipSuccess) return fail(OC_HIP_ERR_HIP, "fused FFTCC3D kernel launch failed: %s", hipGetErrorString(err));
        }
        return OC_HIP_OK;
    }
    const size_t chunk = count < fftcc3d_chunk_limit() ? count : fftcc3d_chunk_limit();
    if (chunk == 0) return OC_HIP_OK;
    OC_TRY(ensure_fft(e, chunk));
    const size_t M = (size_t)8 * e->rx * e->ry * e->rz;
    const size_t F = (size_t)(2 * e->rx) * (size_t)(2 * e->ry) * (size_t)(e->rz + 1);
    OC_TRY(e->win.reserve(2 * chunk * M * sizeof(float)));
    OC_TRY(e->freq.reserve(2 * chunk * F * sizeof(float2)));
    OC_TRY(e->norms.reserve(2 * chunk * sizeof(float)));
    OC_FFT_TRY(rocfft_execution_info_set_stream(e->fft.info_fwd, e->stream));
    OC_FFT_TRY(rocfft_execution_info_set_stream(e->fft.info_inv, e->stream));
    ochip::Fftcc3dParams P = {im.ref_ptr(), im.tar_ptr(), im.dz, im.dy, im.dx, e->rx, e->ry, e->rz};
    float* ref_win = e->win.as<float>();
    float* tar_win = ref_win + chunk * M;
    float2* ref_freq = e->freq.as<float2>();
    float2* tar_freq = ref_freq + chunk * F;
    ProfScope prof(e);
    for (size_t first = 0; first < count; first += chunk) {
        const size_t n = (count - first) < chunk ? (count - first) : chunk;
        float* pois = d_pois + first * (size_t)stride_f;
        if (n < chunk) {
            OC_HIP_TRY(hipMemsetAsync(ref_win + n * M, 0, (chunk - n) * M * sizeof(float), e->stream));
            OC_HIP_TRY(hipMemsetAsync(tar_win + n * M, 0, (chunk - n) * M * sizeof(float), e->stream));
        }
        OC_HIP_TRY(ochip::launch_fftcc3d_gather(P, pois, stride_f, n, ref_win, tar_win, e->norms.as<float>(), e->stream));
        void* in_fwd[1] = {ref_win};
        void* out_fwd[1] = {ref_freq};
        OC_FFT_TRY(rocfft_execute(e->fft.fwd, in_fwd, out_fwd, e->fft.info_fwd));
        OC_HIP_TRY(ochip::launch_fftcc_conjmul(ref_freq, tar_freq, ref_freq, n * F, e->stream));
        void* in_inv[1] = {ref_freq};
        void* out_inv[1] = {ref_win};
        OC_FFT_TRY(rocfft_execute(e->fft.inv, in_inv, out_inv, e->fft.info_inv));
        OC_HIP_TRY(ochip::launch_fftcc3d_argmax(P, ref_win, e->norms.as<float>(), pois, stride_f, n, e->stream));
    }
    return OC_HIP_OK;
}

int run_icgn3d1(oc_hip_engine* e, float* d_pois, int stride_f, size_t count) {
    if (!e->img || e->img->ndim != 3) return fail(OC_HIP_ERR_INVALID, "ICGN3D1: set_images3d has not been called");
    if (!e->ref_ready || !e->tar_ready)
        return fail(OC_HIP_ERR_INVALID, "ICGN3D1: prepare() has not been called since the last set_images");
    const ImagePair& im = *e->img;
    int blocks = 0;
    size_t scratch = ochip::icgn3d1_scratch_floats(e->rx, e->ry, e->rz, &blocks);
#if OC_BUILD_AB
    // the row mapping (icgn3d_rows.hip, an A/B partner: only the A/B build contains it) keeps whole steps per thread: its
    // slots are a little larger
    const size_t rows_scratch = ochip::icgn3d1_rows_slot_floats(e->rx, e->ry, e->rz) * (size_t)blocks;
    if (e->icgn3d_mapping != 0 && rows_scratch > scratch) scratch = rows_scratch;
#endif
    if (scratch) OC_TRY(e->tmp.reserve(scratch * sizeof(float)));
    ochip::Icgn3dParams P = {im.ref_ptr(), e->gx.as<float>(), e->gy.as<float>(), e->gz.as<float>(), e->coef.as<float>(),
                             im.dz, im.dy, im.dx, e->rx, e->ry, e->rz, e->conv, e->stop,
                             scratch ? e->tmp.as<float>() : nullptr, 1, 1, nullptr, e->arith_fma};
    if (e->arith_fma && e->icgn3d_mapping != 0)
        return fail(OC_HIP_ERR_UNSUPPORTED, "ICGN3D1: the row mapping (icgn3d_mapping = 1) has no fused-arithmetic build (arith_fma = 1)");
    // locality schedule: visit the queue in compact cubic blocks, so that the POIs in flight share their voxels behind the
    // L2s / the Infinity Cache (poi_order.hip); same bits for every POI
    OC_TRY(tile_order3d(e, d_pois, stride_f, count, e->icgn3d_tile_vox, &P.perm));
    ProfScope prof(e);
#if OC_BUILD_AB
    hipError_t err = e->icgn3d_mapping != 0 ? ochip::launch_icgn3d1_rows(P, d_pois, stride_f, count, blocks, e->stream)
                                            : ochip::launch_icgn3d1(P, d_pois, stride_f, count, e->stream);
#else
    hipError_t err = ochip::launch_icgn3d1(P, d_pois, stride_f, count, e->stream);
#endif
    if (err != hipSuccess) return fail(OC_HIP_ERR_HIP, "ICGN3D1 kernel launch failed: %s", hipGetErrorString(err));
    return OC_HIP_OK;
}

int run_compute_device(oc_hip_engine* e, float* d_pois, int stride_f, size_t count, const float* d_offsets = nullptr) {
    if (d_offsets && e->kind != OC_HIP_ICGN2D1 && e->kind != OC_HIP_ICGN2D2)
        return fail(OC_HIP_ERR_INVALID, "center offsets are an ICGN2D1/ICGN2D2 feature (src/oc_icgn.h:75-76,130-131)");
    switch (e->kind) {
        case OC_HIP_FFTCC2D: return run_fftcc2d(e, d_pois, stride_f, count);
        case OC_HIP_ICGN2D1:
        case OC_HIP_ICGN2D2:
        case OC_HIP_ICLM2D1:
        case OC_HIP_ICLM2D2: return run_icgn2d(e, d_pois, stride_f, count, d_offsets);
        case OC_HIP_NR2D1: return run_nr2d1(e, d_pois, stride_f, count);
        case OC_HIP_FFTCC3D: return run_fftcc3d(e, d_pois, stride_f, count);
        case OC_HIP_ICGN3D1: return run_icgn3d1(e, d_pois, stride_f, count);
        default: return fail(OC_HIP_ERR_UNSUPPORTED, "engine kind %d has no device path yet", e->kind);
    }
}

}  // namespace

// ===========================================================================
// C entry points
// ===========================================================================
extern "C" {

const char* oc_hip_last_error(void) { return g_last_error.c_str(); }

int oc_hip_abi_version(void) { return 3; }

int oc_hip_device_count(int* count) {
    if (!count) return fail(OC_HIP_ERR_INVALID, "null count");
    *count = 0;
    int n = 0;
    hipError_t err = hipGetDeviceCount(&n);
    if (err != hipSuccess) return fail(OC_HIP_ERR_HIP, "hipGetDeviceCount failed: %s", hipGetErrorString(err));
    *count = n;
    return OC_HIP_OK;
}

int oc_hip_fftcc2d_create(int rx, int ry, int device, oc_hip_engine** out) {
    return create_engine(OC_HIP_FFTCC2D, rx, ry, 0, 0.f, 0.f, device, out);
}
int oc_hip_icgn2d1_create(int rx, int ry, float conv, float stop, int device, oc_hip_engine** out) {
    return create_engine(OC_HIP_ICGN2D1, rx, ry, 0, conv, stop, device, out);
}
int oc_hip_icgn2d2_create(int rx, int ry, float conv, float stop, int device, oc_hip_engine** out) {
    return create_engine(OC_HIP_ICGN2D2, rx, ry, 0, conv, stop, device, out);
}
int oc_hip_iclm2d1_create(int rx, int ry, float conv, float stop, int device, oc_hip_engine** out) {
    return create_engine(OC_HIP_ICLM2D1, rx, ry, 0, conv, stop, device, out);
}

int oc_hip_iclm2d2_create(int rx, int ry, float conv, float stop, int device, oc_hip_engine** out) {
    return create_engine(OC_HIP_ICLM2D2, rx, ry, 0, conv, stop, device, out);
}

int oc_hip_set_damping(oc_hip_engine* e, float lambda, float alpha, float beta) {
    OC_TRY(check_engine(e));
    if (!e->is_iclm()) return fail(OC_HIP_ERR_INVALID, "set_damping: not an ICLM2D1/ICLM2D2 engine");
    // powf(lambda, q) with a non-integer q is NaN for lambda < 0 and the reference would then reject every step
    if (!(lambda > 0.f)) return fail(OC_HIP_ERR_INVALID, "set_damping: lambda must be > 0 (got %g)", (double)lambda);
    std::lock_guard<std::mutex> lock(e->mu);
    e->lm_lambda = lambda;
    e->lm_alpha = alpha;
    e->lm_beta = beta;
    for (oc_hip_engine* r : e->replicas) OC_TRY(oc_hip_set_damping(r, lambda, alpha, beta));
    return OC_HIP_OK;
}

// A call on a device-resident queue is asynchronous on a stream the CALLER chose (oc_hip_set_stream: stream-ordered
// with the caller's other work, e.g. the next engine on the same stream).  On the engine's own private stream nobody
// else could order against it, so the call completes before it returns.
static int finish_device_call(oc_hip_engine* e) {
    if (e->stream == e->own_stream) OC_HIP_TRY(hipStreamSynchronize(e->stream));
    else OC_TRY(mark_tail(e));
    return OC_HIP_OK;
}

// ---------------------------------------------------------------------------
// Strain
// ---------------------------------------------------------------------------
static int strain_check(const oc_hip_engine* e, float radius, int nmin, int approximation) {
    (void)e;
    if (!(radius > 0.f)) return fail(OC_HIP_ERR_INVALID, "Strain: subregion radius must be > 0 (got %g)", (double)radius);
    if (nmin < 1) return fail(OC_HIP_ERR_INVALID, "Strain: neighbor_number_min must be >= 1 (got %d)", nmin);
    if (nmin > ochip::strain_knn_max())
        return fail(OC_HIP_ERR_UNSUPPORTED, "Strain: neighbor_number_min %d exceeds the KNN path's limit of %d", nmin,
                    ochip::strain_knn_max());
    if (approximation != 1 && approximation != 2)
        return fail(OC_HIP_ERR_INVALID, "Strain: approximation must be 1 (Cauchy) or 2 (Green), got %d", approximation);
    return OC_HIP_OK;
}

int oc_hip_strain_create(float subregion_radius, int neighbor_number_min, int device, oc_hip_engine** out) {
    if (out) *out = nullptr;
    OC_TRY(strain_check(nullptr, subregion_radius, neighbor_number_min, 1));
    OC_TRY(create_engine(OC_HIP_STRAIN, 1, 1, 0, 0.f, 0.f, device, out));
    (*out)->st_radius = subregion_radius;
    (*out)->st_nmin = neighbor_number_min;
    return OC_HIP_OK;
}

int oc_hip_region_fit_create(float neighbor_search_radius, int neighbor_number_min, int device, oc_hip_engine** out) {
    if (out) *out = nullptr;
    OC_TRY(strain_check(nullptr, neighbor_search_radius, neighbor_number_min, 1));
    OC_TRY(create_engine(OC_HIP_REGION_FIT, 1, 1, 0, 0.f, 0.f, device, out));
    (*out)->st_radius = neighbor_search_radius;
    (*out)->st_nmin = neighbor_number_min;
    return OC_HIP_OK;
}

int oc_hip_region_fit_set(oc_hip_engine* e, float neighbor_search_radius, int neighbor_number_min) {
    OC_TRY(check_engine(e));
    if (e->kind != OC_HIP_REGION_FIT) return fail(OC_HIP_ERR_INVALID, "region_fit_set: not a RegionFit engine");
    OC_TRY(strain_check(e, neighbor_search_radius, neighbor_number_min, 1));
    std::lock_guard<std::mutex> lock(e->mu);
    if (neighbor_search_radius != e->st_radius) e->st_count = 0;
    e->st_radius = neighbor_search_radius;
    e->st_nmin = neighbor_number_min;
    return OC_HIP_OK;
}

int oc_hip_strain_set(oc_hip_engine* e, float subregion_radius, int neighbor_number_min, float zncc_threshold,
                      int approximation) {
    OC_TRY(check_engine(e));
    if (e->kind != OC_HIP_STRAIN) return fail(OC_HIP_ERR_INVALID, "strain_set: not a Strain engine");
    OC_TRY(strain_check(e, subregion_radius, neighbor_number_min, approximation));
    std::lock_guard<std::mutex> lock(e->mu);
    if (subregion_radius != e->st_radius) e->st_count = 0;  // the grid pitch follows the radius: prepare() again
    e->st_radius = subregion_radius;
    e->st_nmin = neighbor_number_min;
    e->st_zncc = zncc_threshold;
    e->st_approx = approximation;
    return OC_HIP_OK;
}

static int strain_stage(oc_hip_engine* e, const void* pois, size_t count, size_t stride_bytes, int ndim, int memory,
                        float** d_pois) {
    if (e->kind != OC_HIP_STRAIN && e->kind != OC_HIP_REGION_FIT) return fail(OC_HIP_ERR_INVALID, "not a Strain / RegionFit engine");
    if (ndim != 2 && ndim != 3) return fail(OC_HIP_ERR_INVALID, "ndim must be 2 (POI2D) or 3 (POI3D), got %d", ndim);
    if (!pois) return fail(OC_HIP_ERR_INVALID, "null POI buffer");
    const size_t rec = ndim == 2 ? OC_HIP_POI2D_BYTES : OC_HIP_POI3D_BYTES;
    if (stride_bytes < rec || (stride_bytes & 3))
        return fail(OC_HIP_ERR_INVALID, "bad POI stride %zu (record is %zu bytes, stride must be a multiple of 4)", stride_bytes, rec);
    if (count > 0x7fffffffull) return fail(OC_HIP_ERR_UNSUPPORTED, "Strain: at most 2^31-1 POIs per queue");
    if (memory == OC_HIP_DEVICE) {
        *d_pois = static_cast<float*>(const_cast<void*>(pois));
        return OC_HIP_OK;
    }
    OC_TRY(e->poi_stage.reserve(count * stride_bytes));
    OC_HIP_TRY(hipMemcpyAsync(e->poi_stage.p, pois, count * stride_bytes, hipMemcpyHostToDevice, e->stream));
    *d_pois = e->poi_stage.as<float>();
    return OC_HIP_OK;
}

// neighbour search over a queue's coordinates; gather_records: also snapshot the fit records (RegionFit's cloud)
static int plane_prepare(oc_hip_engine* e, int kind, const void* pois, size_t count, size_t stride_bytes, int ndim, int memory,
                         bool gather_records) {
    OC_ACTIVATE(e);
    if (e->kind != kind) return fail(OC_HIP_ERR_INVALID, "prepare: wrong engine kind for this entry point");
    std::lock_guard<std::mutex> lock(e->mu);
    TailGuard tail(e);  // also the error exits leave the enqueued work covered by the tail event
    e->st_count = 0;
    if (count == 0) return OC_HIP_OK;
    OC_TRY(order_after_default_stream(e));
    float* d_pois = nullptr;
    OC_TRY(strain_stage(e, pois, count, stride_bytes, ndim, memory, &d_pois));
    const int stride_f = (int)(stride_bytes / 4);
    // bounding box -> grid (the cell count is needed on the host to size the tables)
    OC_TRY(e->st_box.reserve(6 * sizeof(unsigned)));
    OC_HIP_TRY(ochip::launch_strain_bbox(ndim, d_pois, stride_f, count, e->st_box.as<unsigned>(), e->stream));
    unsigned box[6];
    OC_HIP_TRY(hipMemcpyAsync(box, e->st_box.p, sizeof(box), hipMemcpyDeviceToHost, e->stream));
    OC_HIP_TRY(hipStreamSynchronize(e->stream));
    const ochip::StrainGrid g = ochip::strain_make_grid(ndim, box, e->st_radius);
    const size_t ncell = ochip::strain_cell_count(g);
    OC_TRY(e->st_counts.reserve(ncell * sizeof(unsigned)));
    OC_TRY(e->st_cursor.reserve(ncell * sizeof(unsigned)));
    OC_TRY(e->st_start.reserve((ncell + 1) * sizeof(unsigned)));
    OC_TRY(e->st_slots.reserve(count * sizeof(unsigned)));
    OC_TRY(e->st_order.reserve(count * sizeof(unsigned)));
    OC_HIP_TRY(ochip::launch_strain_sort(ndim, d_pois, stride_f, count, g, e->st_counts.as<unsigned>(), e->st_start.as<unsigned>(),
                                         e->st_cursor.as<unsigned>(), e->st_slots.as<unsigned>(), e->st_order.as<unsigned>(),
                                         e->stream));
    if (gather_records) {
        OC_TRY(e->st_recs.reserve(count * 32));
        OC_HIP_TRY(ochip::launch_strain_gather(ndim, d_pois, stride_f, count, e->st_order.as<unsigned>(), e->st_recs.p, e->stream));
    }
    if (memory == OC_HIP_HOST) OC_HIP_TRY(hipStreamSynchronize(e->stream));  // the staging buffer is reused by compute
    else OC_TRY(finish_device_call(e));
    e->st_grid = g;
    e->st_ndim = ndim;
    e->st_count = count;
    return OC_HIP_OK;
}

int oc_hip_strain_prepare(oc_hip_engine* e, const void* pois, size_t count, size_t stride_bytes, int ndim, int memory) {
    OC_TRY(check_engine(e));
    return plane_prepare(e, OC_HIP_STRAIN, pois, count, stride_bytes, ndim, memory, false);
}

int oc_hip_region_fit_prepare(oc_hip_engine* e, const void* reliable_pois, size_t count, size_t stride_bytes, int ndim,
                              int memory) {
    OC_TRY(check_engine(e));
    return plane_prepare(e, OC_HIP_REGION_FIT, reliable_pois, count, stride_bytes, ndim, memory, true);
}

int oc_hip_region_fit_compute(oc_hip_engine* e, void* pois, size_t count, size_t stride_bytes, int ndim, int memory) {
    OC_ACTIVATE(e);
    if (count == 0) return OC_HIP_OK;
    std::lock_guard<std::mutex> lock(e->mu);
    TailGuard tail(e);  // also the error exits leave the enqueued work covered by the tail event
    if (e->kind != OC_HIP_REGION_FIT) return fail(OC_HIP_ERR_INVALID, "not a RegionFit engine");
    if (e->st_count == 0)
        return fail(OC_HIP_ERR_INVALID, "RegionFit: setNeighbor + prepare has not been called (or the radius changed since)");
    if (e->st_ndim != ndim) return fail(OC_HIP_ERR_INVALID, "RegionFit: prepared for POI%dD, compute() got POI%dD", e->st_ndim, ndim);
    OC_TRY(order_after_default_stream(e));
    float* d_pois = nullptr;
    OC_TRY(strain_stage(e, pois, count, stride_bytes, ndim, memory, &d_pois));
    const int stride_f = (int)(stride_bytes / 4);
    OC_TRY(e->st_fallback.reserve((count + 1) * sizeof(unsigned)));
    const ochip::StrainParams P = {e->st_radius * e->st_radius, 0.f, e->st_nmin, 1};
    {
        ProfScope prof(e);
        OC_HIP_TRY(ochip::launch_region_fit_compute(ndim, d_pois, stride_f, count, e->st_grid, P, e->st_start.as<unsigned>(),
                                                    e->st_recs.p, e->st_fallback.as<unsigned>(), e->stream));
    }
    if (memory == OC_HIP_HOST) {
        OC_HIP_TRY(hipMemcpyAsync(pois, d_pois, count * stride_bytes, hipMemcpyDeviceToHost, e->stream));
        OC_HIP_TRY(hipStreamSynchronize(e->stream));
    } else {
        OC_TRY(finish_device_call(e));
    }
    return OC_HIP_OK;
}

int oc_hip_strain_compute(oc_hip_engine* e, void* pois, size_t count, size_t stride_bytes, int ndim, int memory) {
    OC_ACTIVATE(e);
    if (count == 0) return OC_HIP_OK;
    std::lock_guard<std::mutex> lock(e->mu);
    TailGuard tail(e);  // also the error exits leave the enqueued work covered by the tail event
    if (e->kind != OC_HIP_STRAIN) return fail(OC_HIP_ERR_INVALID, "not a Strain engine");
    if (e->st_count == 0) return fail(OC_HIP_ERR_INVALID, "Strain: prepare(poi_queue) has not been called (or the radius changed since)");
    if (e->st_count != count || e->st_ndim != ndim)
        return fail(OC_HIP_ERR_INVALID, "Strain: prepare() saw %zu POI%dD, compute() got %zu POI%dD", e->st_count, e->st_ndim, count, ndim);
    OC_TRY(order_after_default_stream(e));
    float* d_pois = nullptr;
    OC_TRY(strain_stage(e, pois, count, stride_bytes, ndim, memory, &d_pois));
    const int stride_f = (int)(stride_bytes / 4);
    OC_TRY(e->st_recs.reserve(count * 32));
    OC_TRY(e->st_fallback.reserve((count + 1) * sizeof(unsigned)));
    const ochip::StrainParams P = {e->st_radius * e->st_radius, e->st_zncc, e->st_nmin, e->st_approx};
    {
        ProfScope prof(e);
        OC_HIP_TRY(ochip::launch_strain_compute(ndim, d_pois, stride_f, count, e->st_grid, P, e->st_start.as<unsigned>(),
                                                e->st_order.as<unsigned>(), e->st_recs.p, e->st_fallback.as<unsigned>(), e->stream));
    }
    if (memory == OC_HIP_HOST) {
        OC_HIP_TRY(hipMemcpyAsync(pois, d_pois, count * stride_bytes, hipMemcpyDeviceToHost, e->stream));
        OC_HIP_TRY(hipStreamSynchronize(e->stream));
    } else {
        OC_TRY(finish_device_call(e));
    }
    return OC_HIP_OK;
}

int oc_hip_nr2d1_create(int rx, int ry, float conv, float stop, int device, oc_hip_engine** out) {
    return create_engine(OC_HIP_NR2D1, rx, ry, 0, conv, stop, device, out);
}
int oc_hip_fftcc3d_create(int rx, int ry, int rz, int device, oc_hip_engine** out) {
    return create_engine(OC_HIP_FFTCC3D, rx, ry, rz, 0.f, 0.f, device, out);
}
int oc_hip_icgn3d1_create(int rx, int ry, int rz, float conv, float stop, int device, oc_hip_engine** out) {
    return create_engine(OC_HIP_ICGN3D1, rx, ry, rz, conv, stop, device, out);
}

int oc_hip_destroy(oc_hip_engine* e) {
    if (!e) return OC_HIP_OK;
    DeviceScope device_scope;
    group_drop_comms(e);
    for (oc_hip_engine* r : e->replicas) {
        r->is_replica = false;
        (void)oc_hip_destroy(r);
    }
    e->replicas.clear();
    (void)hipSetDevice(e->device);
    drain_engine(e);  // never through a caller's stream handle: it may be gone already
    clear_events(e);
    e->fft.destroy();
    if (e->order_ev) (void)hipEventDestroy(e->order_ev);
    if (e->switch_ev) (void)hipEventDestroy(e->switch_ev);
    if (e->group_ev) (void)hipEventDestroy(e->group_ev);
    for (hipEvent_t ev : e->chunk_done) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : e->chunk_in) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : e->split_ev) (void)hipEventDestroy(ev);
    for (hipStream_t* st : {&e->copy_stream, &e->copy_in_stream, &e->aux_stream})
        if (*st) {
            (void)hipStreamSynchronize(*st);
            (void)hipStreamDestroy(*st);
        }
    if (e->own_stream) (void)hipStreamDestroy(e->own_stream);
    delete e;
    return OC_HIP_OK;
}

// A group member on another device receives its own copy of the leader's image pair (peer copy over xGMI; a member on
// the leader's device simply shares the pair).
static int replicate_images(oc_hip_engine* leader, oc_hip_engine* r) {
    std::lock_guard<std::mutex> lock(r->mu);
    if (r->device == leader->device) {
        r->img = leader->img;
    } else {
        const ImagePair& src = *leader->img;
        auto img = std::make_shared<ImagePair>();
        img->ndim = src.ndim;
        img->dx = src.dx;
        img->dy = src.dy;
        img->dz = src.dz;
        const size_t bytes = src.count() * sizeof(float);
        OC_HIP_TRY(hipSetDevice(r->device));
        OC_TRY(img->ref.reserve(bytes));
        OC_TRY(img->tar.reserve(bytes));
        OC_HIP_TRY(hipMemcpyPeer(img->ref.p, r->device, src.ref_ptr(), leader->device, bytes));
        OC_HIP_TRY(hipMemcpyPeer(img->tar.p, r->device, src.tar_ptr(), leader->device, bytes));
        OC_HIP_TRY(hipSetDevice(leader->device));
        r->img = img;
    }
    r->ref_ready = r->tar_ready = false;
    return OC_HIP_OK;
}

static int upload_image(oc_hip_engine* e, const float* src, size_t count, int memory, DevBuf& dst) {
    OC_TRY(dst.reserve(count * sizeof(float)));
    OC_HIP_TRY(hipMemcpyAsync(dst.p, src, count * sizeof(float),
                              memory == OC_HIP_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, e->stream));
    return OC_HIP_OK;
}

int oc_hip_set_images2d(oc_hip_engine* e, const float* ref, const float* tar, int height, int width, int layout,
                        int memory) {
    OC_ACTIVATE(e);
    if (e->is3d()) return fail(OC_HIP_ERR_INVALID, "set_images2d called on a 3D engine");
    if (!ref || !tar) return fail(OC_HIP_ERR_INVALID, "null image pointer");
    if (height < 5 || width < 5) return fail(OC_HIP_ERR_INVALID, "image too small: %d x %d", width, height);
    if (layout != OC_HIP_ROW_MAJOR && layout != OC_HIP_COL_MAJOR) return fail(OC_HIP_ERR_INVALID, "bad layout %d", layout);
    std::lock_guard<std::mutex> lock(e->mu);
    TailGuard tail(e);  // also the error exits leave the enqueued work covered by the tail event
    OC_TRY(order_after_default_stream(e));
    auto img = std::make_shared<ImagePair>();
    img->ndim = 2;
    img->dx = width;
    img->dy = height;
    img->dz = 1;
    const size_t count = img->count();
    if (memory == OC_HIP_DEVICE && layout == OC_HIP_ROW_MAJOR) {
        img->ref_ext = ref;  // used in place
        img->tar_ext = tar;
    } else if (layout == OC_HIP_ROW_MAJOR) {
        OC_TRY(upload_image(e, ref, count, memory, img->ref));
        OC_TRY(upload_image(e, tar, count, memory, img->tar));
    } else {
        OC_TRY(img->ref.reserve(count * sizeof(float)));
        OC_TRY(img->tar.reserve(count * sizeof(float)));
        OC_TRY(upload_image(e, ref, count, memory, e->tmp));
        OC_HIP_TRY(ochip::launch_colmajor_to_rowmajor(e->tmp.as<float>(), height, width, img->ref.as<float>(), e->stream));
        OC_HIP_TRY(hipStreamSynchronize(e->stream));
        OC_TRY(upload_image(e, tar, count, memory, e->tmp));
        OC_HIP_TRY(ochip::launch_colmajor_to_rowmajor(e->tmp.as<float>(), height, width, img->tar.as<float>(), e->stream));
    }
    OC_HIP_TRY(hipStreamSynchronize(e->stream));  // host buffers may be released by the caller
    e->img = img;
    e->ref_ready = e->tar_ready = false;
    for (oc_hip_engine* r : e->replicas) OC_TRY(replicate_images(e, r));
    return OC_HIP_OK;
}

int oc_hip_set_images3d(oc_hip_engine* e, const float* ref, const float* tar, int dim_x, int dim_y, int dim_z,
                        int memory) {
    OC_ACTIVATE(e);
    if (!e->is3d()) return fail(OC_HIP_ERR_INVALID, "set_images3d called on a 2D engine");
    if (!ref || !tar) return fail(OC_HIP_ERR_INVALID, "null volume pointer");
    if (dim_x < 15 || dim_y < 15 || dim_z < 15)
        return fail(OC_HIP_ERR_INVALID, "volume too small: %d x %d x %d", dim_x, dim_y, dim_z);
    std::lock_guard<std::mutex> lock(e->mu);
    TailGuard tail(e);  // also the error exits leave the enqueued work covered by the tail event
    OC_TRY(order_after_default_stream(e));
    auto img = std::make_shared<ImagePair>();
    img->ndim = 3;
    img->dx = dim_x;
    img->dy = dim_y;
    img->dz = dim_z;
    const size_t count = img->count();
    if (memory == OC_HIP_DEVICE) {
        img->ref_ext = ref;
        img->tar_ext = tar;
    } else {
        OC_TRY(upload_image(e, ref, count, memory, img->ref));
        OC_TRY(upload_image(e, tar, count, memory, img->tar));
    }
    OC_HIP_TRY(hipStreamSynchronize(e->stream));
    e->img = img;
    e->ref_ready = e->tar_ready = false;
    for (oc_hip_engine* r : e->replicas) OC_TRY(replicate_images(e, r));
    return OC_HIP_OK;
}

int oc_hip_share_images(oc_hip_engine* e, oc_hip_engine* donor) {
    OC_ACTIVATE(e);
    OC_TRY(check_engine(donor));
    if (donor->device != e->device) return fail(OC_HIP_ERR_INVALID, "share_images: engines live on different devices");
    if (!donor->img) return fail(OC_HIP_ERR_INVALID, "share_images: donor has no images");
    if ((donor->img->ndim == 3) != e->is3d()) return fail(OC_HIP_ERR_INVALID, "share_images: 2D/3D mismatch");
    std::lock_guard<std::mutex> lock(e->mu);
    e->img = donor->img;
    e->ref_ready = e->tar_ready = false;
    // members of a group: share with the donor's member on the same device when there is one, copy otherwise
    for (oc_hip_engine* r : e->replicas) {
        oc_hip_engine* twin = nullptr;
        for (oc_hip_engine* d : donor->replicas)
            if (d->device == r->device && d->img && d->img->ndim == donor->img->ndim && d->img->dx == donor->img->dx &&
                d->img->dy == donor->img->dy && d->img->dz == donor->img->dz)
                twin = d;
        if (twin && r->device != e->device) {
            std::lock_guard<std::mutex> rlock(r->mu);
            r->img = twin->img;
            r->ref_ready = r->tar_ready = false;
        } else {
            OC_TRY(replicate_images(e, r));
        }
    }
    return OC_HIP_OK;
}

// Clone of an engine's configuration on another device (a group member)
static int clone_engine(const oc_hip_engine* e, int device, oc_hip_engine** out) {
    OC_TRY(create_engine(e->kind, e->rx, e->ry, e->rz, e->conv, e->stop, device, out));
    oc_hip_engine* r = *out;
    r->is_replica = true;
    r->lm_lambda = e->lm_lambda;
    r->lm_alpha = e->lm_alpha;
    r->lm_beta = e->lm_beta;
    r->icgn2d_tile_px = e->icgn2d_tile_px;
    r->icgn2d_variant = e->icgn2d_variant;
    r->self_adaptive = e->self_adaptive;
    r->icgn2d_xcd = e->icgn2d_xcd;
    r->arith_fma = e->arith_fma;
    r->icgn2d_split_chunks = e->icgn2d_split_chunks;
    r->fftcc2d_fused = e->fftcc2d_fused;
    r->fftcc3d_fused = e->fftcc3d_fused;
    r->fftcc3d_planes_blocks = e->fftcc3d_planes_blocks;
    r->icgn3d_mapping = e->icgn3d_mapping;
    r->icgn3d_tile_vox = e->icgn3d_tile_vox;
    r->fftcc3d_tile_vox = e->fftcc3d_tile_vox;
    r->host_chunk = e->host_chunk;
    r->group_allgather = e->group_allgather;
    r->group_force_rccl = e->group_force_rccl;
    return OC_HIP_OK;
}

// Moves an engine to another device: everything it holds on the old one is released (images included: the caller
// sets them again, like after construction).
static int rehome(oc_hip_engine* e, int device) {
    int ndev = 0;
    OC_HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(OC_HIP_ERR_INVALID, "device %d out of range [0,%d)", device, ndev);
    OC_HIP_TRY(hipSetDevice(e->device));
    drain_engine(e);
    e->tail_marked = false;
    clear_events(e);
    e->fft.destroy();
    e->fft.work_fwd.release();
    e->fft.work_inv.release();
    for (DevBuf* b : {&e->gx, &e->gy, &e->gz, &e->coef, &e->coef_gx, &e->coef_gy, &e->tmp, &e->poi_stage, &e->off_stage, &e->cursors,
                      &e->perm, &e->tiles, &e->perm_slots, &e->split_scratch, &e->split_tmp, &e->prefilter_tmp, &e->st_box, &e->st_counts, &e->st_start, &e->st_cursor, &e->st_slots,
                      &e->st_order, &e->st_recs, &e->st_fallback, &e->win, &e->freq, &e->norms, &e->flags, &e->group_mirror,
                      &e->group_off_mirror})
        b->release();
    e->img.reset();
    e->ref_ready = e->tar_ready = false;
    e->st_count = 0;
    if (e->order_ev) { (void)hipEventDestroy(e->order_ev); e->order_ev = nullptr; }
    if (e->switch_ev) { (void)hipEventDestroy(e->switch_ev); e->switch_ev = nullptr; }
    if (e->group_ev) { (void)hipEventDestroy(e->group_ev); e->group_ev = nullptr; }
    for (hipEvent_t ev : e->chunk_done) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : e->chunk_in) (void)hipEventDestroy(ev);
    e->chunk_done.clear();
    e->chunk_in.clear();
    if (e->copy_stream) { (void)hipStreamDestroy(e->copy_stream); e->copy_stream = nullptr; }
    if (e->copy_in_stream) { (void)hipStreamDestroy(e->copy_in_stream); e->copy_in_stream = nullptr; }
    for (hipEvent_t ev : e->split_ev) (void)hipEventDestroy(ev);
    e->split_ev.clear();
    if (e->aux_stream) { (void)hipStreamSynchronize(e->aux_stream); (void)hipStreamDestroy(e->aux_stream); e->aux_stream = nullptr; }
    (void)hipStreamDestroy(e->own_stream);
    e->own_stream = nullptr;
    OC_HIP_TRY(hipSetDevice(device));
    e->device = device;
    OC_HIP_TRY(hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking));
    e->stream = e->own_stream;
    return OC_HIP_OK;
}

int oc_hip_set_devices(oc_hip_engine* e, const int* device_ids, int n_devices) {
    OC_TRY(check_engine(e));
    DeviceScope device_scope;
    if (e->is_replica) return fail(OC_HIP_ERR_INVALID, "set_devices: this handle is a group member");
    if (!device_ids || n_devices < 1) return fail(OC_HIP_ERR_INVALID, "set_devices: need at least one device id");
    if (n_devices > 1 && (e->kind == OC_HIP_STRAIN || e->kind == OC_HIP_REGION_FIT))
        return fail(OC_HIP_ERR_UNSUPPORTED, "set_devices: Strain / RegionFit need every neighbour of a POI and stay on one device");
    int ndev = 0;
    OC_HIP_TRY(hipGetDeviceCount(&ndev));
    for (int i = 0; i < n_devices; i++)
        if (device_ids[i] < 0 || device_ids[i] >= ndev) return fail(OC_HIP_ERR_INVALID, "set_devices: device %d out of range [0,%d)", device_ids[i], ndev);
    std::lock_guard<std::mutex> lock(e->mu);
    // dissolve the current group
    group_drop_comms(e);
    for (oc_hip_engine* r : e->replicas) {
        r->is_replica = false;
        (void)oc_hip_destroy(r);
    }
    e->replicas.clear();
    e->group_devices.clear();
    if (device_ids[0] != e->device) OC_TRY(rehome(e, device_ids[0]));
    OC_HIP_TRY(hipSetDevice(e->device));
    // The new members are built aside and committed as a whole: a failure half way (allocation on member 3 of 8, say)
    // destroys what was built and leaves a plain single-device engine, never a handle that fans out over a partial group.
    std::vector<oc_hip_engine*> fresh;
    auto build = [&]() -> int {
        for (int i = 1; i < n_devices; i++) {
            oc_hip_engine* r = nullptr;
            OC_TRY(clone_engine(e, device_ids[i], &r));
            fresh.push_back(r);
            if (device_ids[i] != e->device) {
                // direct xGMI copies between the members (ignored when the platform has no peer path: copies are then staged)
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, e->device, device_ids[i]) == hipSuccess && can) {
                    (void)hipSetDevice(e->device);
                    (void)hipDeviceEnablePeerAccess(device_ids[i], 0);
                    (void)hipSetDevice(device_ids[i]);
                    (void)hipDeviceEnablePeerAccess(e->device, 0);
                }
                (void)hipGetLastError();  // "already enabled" is fine
            }
            if (e->img) OC_TRY(replicate_images(e, r));
        }
        return OC_HIP_OK;
    };
    const int rc = build();
    (void)hipSetDevice(e->device);
    if (rc != OC_HIP_OK) {
        const std::string why = g_last_error;
        for (oc_hip_engine* r : fresh) {
            r->is_replica = false;
            (void)oc_hip_destroy(r);
        }
        (void)hipSetDevice(e->device);
        e->group_devices.assign(1, e->device);
        return fail(rc, "set_devices: %s (the engine stays on device %d alone)", why.c_str(), e->device);
    }
    e->replicas.swap(fresh);
    e->group_devices.assign(device_ids, device_ids + n_devices);
    // precomputed fields exist on the leader only: prepare() again
    if (n_devices > 1) e->ref_ready = e->tar_ready = false;
    return OC_HIP_OK;
}

int oc_hip_get_devices(const oc_hip_engine* e, int* device_ids, int capacity, int* n_devices) {
    OC_TRY(check_engine(e));
    if (!n_devices) return fail(OC_HIP_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lock(e->mu);
    *n_devices = (int)e->replicas.size() + 1;
    if (device_ids) {
        if (capacity > 0) device_ids[0] = e->device;
        for (int i = 1; i < *n_devices && i < capacity; i++) device_ids[i] = e->replicas[i - 1]->device;
    }
    return OC_HIP_OK;
}

int oc_hip_group_queue(const oc_hip_engine* e, int member, const void** device_ptr, size_t* block_bytes) {
    OC_TRY(check_engine(e));
    if (!device_ptr || !block_bytes) return fail(OC_HIP_ERR_INVALID, "null argument");
    *device_ptr = nullptr;
    *block_bytes = 0;
    std::lock_guard<std::mutex> lock(e->mu);
    if (member < 0 || member > (int)e->replicas.size()) return fail(OC_HIP_ERR_INVALID, "group_queue: member %d out of range", member);
    const oc_hip_engine* m = member == 0 ? e : e->replicas[member - 1];
    if (!m->group_mirror.p || m->group_mirror_block == 0)
        return fail(OC_HIP_ERR_INVALID, "group_queue: no all-gathered queue yet (set_tuning(\"group_allgather\", 1), then compute a DEVICE queue)");
    *device_ptr = m->group_mirror.p;
    *block_bytes = m->group_mirror_block;
    return OC_HIP_OK;
}

int oc_hip_set_subset(oc_hip_engine* e, int rx, int ry, int rz) {
    OC_TRY(check_engine(e));
    if (rx < 1 || ry < 1 || (e->is3d() && rz < 1)) return fail(OC_HIP_ERR_INVALID, "subset radius must be >= 1");
    std::lock_guard<std::mutex> lock(e->mu);
    e->rx = rx;
    e->ry = ry;
    if (e->is3d()) e->rz = rz;
    for (oc_hip_engine* r : e->replicas) OC_TRY(oc_hip_set_subset(r, rx, ry, rz));
    return OC_HIP_OK;
}

int oc_hip_set_iteration(oc_hip_engine* e, float conv, float stop) {
    OC_TRY(check_engine(e));
    if (!e->is_icgn()) return fail(OC_HIP_ERR_INVALID, "set_iteration on a non-ICGN engine");
    std::lock_guard<std::mutex> lock(e->mu);
    e->conv = conv;
    e->stop = stop;
    for (oc_hip_engine* r : e->replicas) OC_TRY(oc_hip_set_iteration(r, conv, stop));
    return OC_HIP_OK;
}

// Switching streams: work already enqueued on the old stream (prepare()'s gradient and table kernels, a layout
// conversion) must not race with computes on the new one.  The two streams are ordered ON THE DEVICE: the incoming stream
// waits for an event that marks the end of this engine's work on the outgoing one -- no host-side wait, so a caller that
// hops between streams (torch's current stream changing from call to call) stays asynchronous.  The outgoing handle may
// be dead already (destroy the stream, then name another one, is a normal C-API sequence) and HIP aborts on any call that
// is handed a destroyed stream, so a caller-owned outgoing stream is never touched here: its event was recorded when the
// work was enqueued (mark_tail).  The engine's own stream is recorded at switch time.
static int switch_stream(oc_hip_engine* e, hipStream_t next, bool must_succeed) {
    if (next == e->stream) return OC_HIP_OK;
    bool have_event = false;
    if (e->stream == e->own_stream) {
        if (!e->switch_ev && hipEventCreateWithFlags(&e->switch_ev, hipEventDisableTiming) != hipSuccess) {
            e->switch_ev = nullptr;
            (void)hipGetLastError();
        }
        if (e->switch_ev && hipEventRecord(e->switch_ev, e->own_stream) == hipSuccess) have_event = true;
        else (void)hipStreamSynchronize(e->own_stream);  // no event to order with: drain the own stream from the host
    } else {
        have_event = e->tail_marked && e->switch_ev;  // nothing marked: this engine left no work on that stream
    }
    if (have_event && next != e->own_stream) {
        const hipError_t err = hipStreamWaitEvent(next, e->switch_ev, 0);
        if (err != hipSuccess) {
            (void)hipGetLastError();
            return fail(OC_HIP_ERR_HIP, "set_stream: cannot enqueue on the new stream: %s", hipGetErrorString(err));
        }
    } else if (have_event) {
        // back to the engine's own stream: always possible; should even the wait fail, wait for the event on the host
        if (hipStreamWaitEvent(e->own_stream, e->switch_ev, 0) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipEventSynchronize(e->switch_ev);
            (void)hipGetLastError();
        }
    }
    (void)must_succeed;
    e->stream = next;
    // the event still marks the end of everything this engine has enqueued anywhere: if the caller moves on to a third
    // stream before the engine has put work on this one, that stream has to wait for it as well
    e->tail_marked = have_event;
    return OC_HIP_OK;
}

int oc_hip_set_stream(oc_hip_engine* e, void* hip_stream) {
    OC_ACTIVATE(e);
    std::lock_guard<std::mutex> lock(e->mu);
    return switch_stream(e, reinterpret_cast<hipStream_t>(hip_stream), false);
}

int oc_hip_reset_stream(oc_hip_engine* e) {
    OC_ACTIVATE(e);
    std::lock_guard<std::mutex> lock(e->mu);
    return switch_stream(e, e->own_stream, true);
}

int oc_hip_set_tuning(oc_hip_engine* e, const char* key, int value) {
    OC_TRY(check_engine(e));
    if (!key) return fail(OC_HIP_ERR_INVALID, "null tuning key");
    std::lock_guard<std::mutex> lock(e->mu);
    const std::string k(key);
    if (k == "icgn2d_variant") {
        if (value < -1 || value >= ochip::icgn2d_variant_count())
            return fail(OC_HIP_ERR_INVALID, "icgn2d_variant %d out of range [-1 (automatic), %d)", value, ochip::icgn2d_variant_count());
        if (value >= 0 && !ochip::icgn2d_variant_built(value))
            return fail(OC_HIP_ERR_UNSUPPORTED, "icgn2d_variant %d is an A/B partner that only the A/B build of the library contains "
                                               "(python -m opencorr_amd.build --ab)", value);
        e->icgn2d_variant = value;
    } else if (k == "icgn2d_xcd" || k == "xcd") {
        e->icgn2d_xcd = value != 0;
    } else if (k == "arith_fma") {
        if (value != 0 && !(e->is_icgn2d() || e->kind == OC_HIP_ICGN3D1))
            return fail(OC_HIP_ERR_UNSUPPORTED, "arith_fma: only the ICGN2D1 / ICGN2D2 / ICLM2D1 / ICLM2D2 / ICGN3D1 engines have a fused-arithmetic build");
        e->arith_fma = value != 0;
    } else if (k == "icgn2d_split_chunks") {
        if (value < 0 || value > 256) return fail(OC_HIP_ERR_INVALID, "icgn2d_split_chunks must be 0 (back to back) ... 256");
        e->icgn2d_split_chunks = value;
    } else if (k == "icgn2d_tile_px") {
        if (value < 0 || (value > 0 && value < 16)) return fail(OC_HIP_ERR_INVALID, "icgn2d_tile_px must be 0 (off) or >= 16");
        e->icgn2d_tile_px = value;
    } else if (k == "fftcc2d_fused") {
        e->fftcc2d_fused = value == 2 ? 2 : (value != 0);
    } else if (k == "fftcc3d_fused") {
        e->fftcc3d_fused = value != 0;
    } else if (k == "fftcc3d_tile_vox") {
        if (value < 0 || (value > 0 && value < 8)) return fail(OC_HIP_ERR_INVALID, "fftcc3d_tile_vox must be 0 (off) or >= 8");
        e->fftcc3d_tile_vox = value;
    } else if (k == "icgn3d_tile_vox") {
        if (value < 0 || (value > 0 && value < 8)) return fail(OC_HIP_ERR_INVALID, "icgn3d_tile_vox must be 0 (off) or >= 8");
        e->icgn3d_tile_vox = value;
    } else if (k == "icgn3d_mapping") {
#if !OC_BUILD_AB
        if (value != 0)
            return fail(OC_HIP_ERR_UNSUPPORTED, "icgn3d_mapping = 1 (the row mapping, measured 12 - 25 %% slower) is an A/B partner that only "
                                               "the A/B build of the library contains (python -m opencorr_amd.build --ab)");
#endif
        e->icgn3d_mapping = value != 0;
    } else if (k == "fftcc3d_planes_blocks") {
        if (value < 0 || value > 4096) return fail(OC_HIP_ERR_INVALID, "fftcc3d_planes_blocks must be 0 (default) ... 4096");
        e->fftcc3d_planes_blocks = value;
    } else if (k == "host_chunk") {
        if (value < 0 || (value > 0 && value < 16384)) return fail(OC_HIP_ERR_INVALID, "host_chunk must be 0 (off) or >= 16384 POIs");
        e->host_chunk = value;
    } else if (k == "group_allgather") {
        e->group_allgather = value != 0;
    } else if (k == "group_force_rccl") {
        e->group_force_rccl = value != 0;
    } else {
        return fail(OC_HIP_ERR_INVALID, "unknown tuning key '%s'", key);
    }
    for (oc_hip_engine* r : e->replicas) OC_TRY(oc_hip_set_tuning(r, key, value));
    return OC_HIP_OK;
}

int oc_hip_prepare_ref(oc_hip_engine* e) {
    OC_ACTIVATE(e);
    if (!e->is_icgn()) return OC_HIP_OK;  // FFTCC::prepare() is empty in the reference
    if (!e->img) return fail(OC_HIP_ERR_INVALID, "prepare: set_images has not been called");
    std::lock_guard<std::mutex> lock(e->mu);
    TailGuard tail(e);  // also the error exits leave the enqueued work covered by the tail event
    OC_TRY(order_after_default_stream(e));  // images used in place may have been written on the default stream
    const ImagePair& im = *e->img;
    const size_t bytes = im.count() * sizeof(float);
    if (e->kind == OC_HIP_NR2D1) {
        e->ref_ready = true;  // NR2D1::prepare builds target-side tables only (src/oc_nr.cpp:119-158)
        for (oc_hip_engine* r : e->replicas) OC_TRY(oc_hip_prepare_ref(r));
        return OC_HIP_OK;
    }
    if (im.ndim == 2) {
        OC_TRY(e->gx.reserve(bytes));
        OC_TRY(e->gy.reserve(bytes));
        OC_HIP_TRY(ochip::launch_grad2d(im.ref_ptr(), im.dy, im.dx, e->gx.as<float>(), e->gy.as<float>(), e->stream));
    } else {
        OC_TRY(e->gx.reserve(bytes));
        OC_TRY(e->gy.reserve(bytes));
        OC_TRY(e->gz.reserve(bytes));
        OC_HIP_TRY(ochip::launch_grad3d(im.ref_ptr(), im.dz, im.dy, im.dx, e->gx.as<float>(), e->gy.as<float>(),
                                        e->gz.as<float>(), e->stream));
    }
    e->ref_ready = true;
    OC_TRY(mark_tail(e));
    for (oc_hip_engine* r : e->replicas) OC_TRY(oc_hip_prepare_ref(r));
    OC_HIP_TRY(hipSetDevice(e->device));
    return OC_HIP_OK;
}

int oc_hip_prepare_tar(oc_hip_engine* e) {
    OC_ACTIVATE(e);
    if (!e->is_icgn()) return OC_HIP_OK;
    if (!e->img) return fail(OC_HIP_ERR_INVALID, "prepare: set_images has not been called");
    std::lock_guard<std::mutex> lock(e->mu);
    TailGuard tail(e);  // also the error exits leave the enqueued work covered by the tail event
    OC_TRY(order_after_default_stream(e));
    const ImagePair& im = *e->img;
    if (im.ndim == 2) {
        OC_TRY(check_image2d_limits(e->kind == OC_HIP_NR2D1 ? "NR2D1" : "ICGN2D", im, e->kind == OC_HIP_NR2D1 ? 1ull << 26 : 1ull << 28));
        OC_TRY(e->coef.reserve(im.count() * 16 * sizeof(float)));
        OC_HIP_TRY(ochip::launch_bspline2d_lut(im.tar_ptr(), im.dy, im.dx, e->coef.as<float>(), e->stream));
        if (e->kind == OC_HIP_NR2D1) {
            // gradients of the TARGET and their interpolation tables (src/oc_nr.cpp:121-157)
            OC_TRY(e->gx.reserve(im.count() * sizeof(float)));
            OC_TRY(e->gy.reserve(im.count() * sizeof(float)));
            OC_TRY(e->coef_gx.reserve(im.count() * 16 * sizeof(float)));
            OC_TRY(e->coef_gy.reserve(im.count() * 16 * sizeof(float)));
            OC_HIP_TRY(ochip::launch_grad2d(im.tar_ptr(), im.dy, im.dx, e->gx.as<float>(), e->gy.as<float>(), e->stream));
            OC_HIP_TRY(ochip::launch_bspline2d_lut(e->gx.as<float>(), im.dy, im.dx, e->coef_gx.as<float>(), e->stream));
            OC_HIP_TRY(ochip::launch_bspline2d_lut(e->gy.as<float>(), im.dy, im.dx, e->coef_gy.as<float>(), e->stream));
        }
    } else {
        OC_TRY(e->coef.reserve(im.count() * sizeof(float)));
        // the y pass needs a second volume.  It stays with the engine (grow-only, like every other buffer): allocating and
        // freeing 537 MB around every prepare() of a 512^3 volume cost ~1 ms of hipMalloc / hipFree plus a stream drain --
        // as much as the three filter passes themselves -- and a DVC run prepares once per volume pair of its sequence
        OC_TRY(e->prefilter_tmp.reserve(im.count() * sizeof(float)));
        OC_HIP_TRY(ochip::launch_bspline3d_prefilter(im.tar_ptr(), im.dz, im.dy, im.dx, e->coef.as<float>(),
                                                     e->prefilter_tmp.as<float>(), e->stream));
    }
    e->tar_ready = true;
    OC_TRY(mark_tail(e));
    for (oc_hip_engine* r : e->replicas) OC_TRY(oc_hip_prepare_tar(r));
    OC_HIP_TRY(hipSetDevice(e->device));
    return OC_HIP_OK;
}

int oc_hip_prepare(oc_hip_engine* e) {
    OC_TRY(oc_hip_prepare_ref(e));
    return oc_hip_prepare_tar(e);
}

}  // extern "C"

namespace {

// ---------------------------------------------------------------------------
// host queues: H2D of the AoS, kernels, D2H (the reference's CUDA module does the same,
// examples/test_2d_dic_gpu_icgn.cpp:136-149) -- but chunk by chunk, so that the copies of one chunk overlap the
// kernels of its neighbours:   H2D(0) K(0) | H2D(1) K(1) D2H(0) | H2D(2) K(2) D2H(1) | ... | D2H(last)
// Kernels run on the engine's stream, copies out on a second stream behind a per-chunk event and from a second host
// thread (copies to / from pageable memory block their thread; PCIe is full duplex); copies in are issued on the
// engine's stream ahead of their kernels.  A POI's result does not depend on the chunk it travels in (tests: split
// queue == whole queue), chunks are large enough for the ICGN2D tile schedule.
// ---------------------------------------------------------------------------
// `chain`: further engines that process the same records right after `e` (oc_hip_compute_chain: FFTCC then ICGN, say) --
// per chunk ONE copy in, every engine's kernels in order on e's stream, ONE copy out, instead of a round trip per engine.
// The callers have moved the chained engines onto e's stream for the duration of the call.
int compute_host(oc_hip_engine* e, char* pois, const float* offsets, size_t count, size_t stride_bytes,
                 oc_hip_engine* const* chain = nullptr, int n_chain = 0) {
    const int stride_f = (int)(stride_bytes / 4);
    auto run_all = [&](float* d_pois, size_t n, const float* d_off) -> int {
        OC_TRY(run_compute_device(e, d_pois, stride_f, n, d_off));
        for (int i = 0; i < n_chain; i++) OC_TRY(run_compute_device(chain[i], d_pois, stride_f, n, nullptr));
        return OC_HIP_OK;
    };
    const size_t bytes = count * stride_bytes;
    OC_TRY(e->poi_stage.reserve(bytes));
    if (offsets) OC_TRY(e->off_stage.reserve(count * 2 * sizeof(float)));
    // Chunk schedule.  What a pipeline cannot hide is the copy-in of its FIRST chunk and the copy-out of its LAST one, and
    // every extra chunk costs a launch tail (the ICGN kernels' last workgroups run on a half-empty chip) plus an
    // inter-stream hand-over.  So: small chunks at both ends (half of "host_chunk"), few large ones (three times
    // "host_chunk") in between, whose copies hide behind the neighbours' kernels.  Measured on config B (250 000 POIs,
    // chain of FFTCC2D + ICGN2D1): uniform chunks of 65 536: 4.76 ms, one piece: 5.36 ms, this schedule: see DESIGN 4.4.
    std::vector<std::pair<size_t, size_t>> sched;  // (first POI, POIs)
    {
        const size_t unit = e->host_chunk > 0 ? (size_t)e->host_chunk : count;
        if (count < 2 * unit) {
            sched.emplace_back(0, count);  // not worth a pipeline
        } else {
            const size_t edge = std::max<size_t>(unit / 2, 1), mid_max = 3 * unit;
            sched.emplace_back(0, edge);
            size_t at = edge;
            const size_t mid_total = count - 2 * edge;
            const size_t nmid = (mid_total + mid_max - 1) / mid_max;
            for (size_t i = 0; i < nmid; i++) {
                const size_t n = mid_total / nmid + (i < mid_total % nmid ? 1 : 0);
                sched.emplace_back(at, n);
                at += n;
            }
            sched.emplace_back(at, count - at);
        }
    }
    const size_t nchunk = sched.size();
    if (nchunk > 1) {
        if (!e->copy_stream) OC_HIP_TRY(hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking));
        if (!e->copy_in_stream) OC_HIP_TRY(hipStreamCreateWithFlags(&e->copy_in_stream, hipStreamNonBlocking));
        // (two loops: a failed creation must not leave the lists at different lengths for the next call)
        while (e->chunk_done.size() < nchunk) {
            hipEvent_t ev = nullptr;
            OC_HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            e->chunk_done.push_back(ev);
        }
        while (e->chunk_in.size() < nchunk) {
            hipEvent_t ev = nullptr;
            OC_HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            e->chunk_in.push_back(ev);
        }
    }
    char* stage = e->poi_stage.as<char>();
    if (nchunk == 1) {
        OC_HIP_TRY(hipMemcpyAsync(stage, pois, bytes, hipMemcpyHostToDevice, e->stream));
        const float* d_off = nullptr;
        if (offsets) {
            OC_HIP_TRY(hipMemcpyAsync(e->off_stage.p, offsets, count * 2 * sizeof(float), hipMemcpyHostToDevice, e->stream));
            d_off = e->off_stage.as<float>();
        }
        OC_TRY(run_all(reinterpret_cast<float*>(stage), count, d_off));
        OC_HIP_TRY(hipMemcpyAsync(pois, stage, bytes, hipMemcpyDeviceToHost, e->stream));
        OC_HIP_TRY(hipStreamSynchronize(e->stream));
        return OC_HIP_OK;
    }
    // A copy between pageable host memory and the device blocks the calling thread, so the two directions get a thread
    // each (PCIe is full duplex): this thread feeds chunks in and launches their kernels, the helper waits for each
    // chunk's event and copies its records back.
    const int device = e->device;
    hipError_t out_err = hipSuccess;
    {
        std::lock_guard<std::mutex> hand(e->feed_mu);
        e->chunks_fed = 0;
    }
    auto hand_over = [&](size_t fed) {
        {
            std::lock_guard<std::mutex> hand(e->feed_mu);
            e->chunks_fed = fed;
        }
        e->feed_cv.notify_one();
    };
    auto copy_out = [&] {
        if (hipSetDevice(device) != hipSuccess) {
            out_err = hipErrorInvalidDevice;
            return;
        }
        for (size_t c = 0; c < nchunk && out_err == hipSuccess; c++) {
            const size_t first = sched[c].first, n = sched[c].second;
            // the event is recorded by the feeding thread after chunk c's kernels were enqueued; until then
            // hipStreamWaitEvent would see the event of an earlier call, so sleep until the hand-off
            {
                std::unique_lock<std::mutex> hand(e->feed_mu);
                e->feed_cv.wait(hand, [&] { return e->chunks_fed > c; });
                if (e->chunks_fed == (size_t)-1) break;  // the feeder failed: drain what was issued and stop
            }
            out_err = hipStreamWaitEvent(e->copy_stream, e->chunk_done[c], 0);
            if (out_err == hipSuccess)
                out_err = hipMemcpyAsync(pois + first * stride_bytes, stage + first * stride_bytes, n * stride_bytes, hipMemcpyDeviceToHost,
                                         e->copy_stream);
        }
        if (out_err == hipSuccess) out_err = hipStreamSynchronize(e->copy_stream);
    };
    // std::thread's constructor throws std::system_error when the process is out of threads; nothing may unwind through
    // the extern "C" boundary, so the helper is optional: without it this thread copies out after feeding
    std::thread out_thread;
    bool helper = true;
    try {
        out_thread = std::thread(copy_out);
    } catch (...) {
        helper = false;
    }
    int rc = OC_HIP_OK;
    auto feed = [&]() -> int {
        // the staging buffer may still be read by kernels of an earlier call on the engine's stream
        OC_HIP_TRY(hipEventRecord(e->chunk_in[0], e->stream));
        OC_HIP_TRY(hipStreamWaitEvent(e->copy_in_stream, e->chunk_in[0], 0));
        for (size_t c = 0; c < nchunk; c++) {
            const size_t first = sched[c].first, n = sched[c].second;
            // copies in travel on their own stream: on the kernels' stream a pageable copy would queue behind the
            // previous chunk's kernels and nothing would overlap
            OC_HIP_TRY(hipMemcpyAsync(stage + first * stride_bytes, pois + first * stride_bytes, n * stride_bytes, hipMemcpyHostToDevice,
                                      e->copy_in_stream));
            const float* d_off = nullptr;
            if (offsets) {
                float* o = e->off_stage.as<float>() + 2 * first;
                OC_HIP_TRY(hipMemcpyAsync(o, offsets + 2 * first, n * 2 * sizeof(float), hipMemcpyHostToDevice, e->copy_in_stream));
                d_off = o;
            }
            OC_HIP_TRY(hipEventRecord(e->chunk_in[c], e->copy_in_stream));
            OC_HIP_TRY(hipStreamWaitEvent(e->stream, e->chunk_in[c], 0));
            OC_TRY(run_all(reinterpret_cast<float*>(stage + first * stride_bytes), n, d_off));
            OC_HIP_TRY(hipEventRecord(e->chunk_done[c], e->stream));
            hand_over(c + 1);
        }
        return OC_HIP_OK;
    };
    rc = feed();
    const std::string feed_error = g_last_error;
    if (rc != OC_HIP_OK) hand_over((size_t)-1);
    if (helper) out_thread.join();
    else copy_out();
    OC_HIP_TRY(hipStreamSynchronize(e->stream));
    if (rc != OC_HIP_OK) return fail(rc, "%s", feed_error.c_str());
    if (out_err != hipSuccess) return fail(OC_HIP_ERR_HIP, "copying results back failed: %s", hipGetErrorString(out_err));
    return OC_HIP_OK;
}

// ---------------------------------------------------------------------------
// device groups (oc_hip_set_devices): contiguous blocks of the queue, one per member (the loop that is
// being replaced is src/oc_icgn.cpp:343-351).  Member g takes POIs [g * ceil(n / G), (g + 1) * ceil(n / G)).
// ---------------------------------------------------------------------------
struct GroupBlock {
    oc_hip_engine* e;
    size_t first, n;
};

std::vector<GroupBlock> group_blocks(oc_hip_engine* e, size_t count) {
    const size_t G = e->replicas.size() + 1, per = (count + G - 1) / G;
    std::vector<GroupBlock> b;
    for (size_t g = 0; g < G; g++) {
        const size_t first = std::min(g * per, count), n = std::min(per, count - first);
        b.push_back({g == 0 ? e : e->replicas[g - 1], first, n});
    }
    return b;
}

// RCCL, loaded on first use: the single-GPU path never needs it and should not pay for loading it (nor fail to start
// on a machine without it).  Types and enumerators come from <rccl/rccl.h> at compile time -- the datatype passed to
// ncclAllGather is the header's ncclUint8, not a number typed in here -- and the loaded library must report the same
// major version as that header.
struct Rccl {
    decltype(&ncclCommInitAll) comm_init_all = nullptr;
    decltype(&ncclAllGather) all_gather = nullptr;
    decltype(&ncclGroupStart) group_start = nullptr;
    decltype(&ncclGroupEnd) group_end = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclGetErrorString) err = nullptr;
    decltype(&ncclGetVersion) get_version = nullptr;
    int version = 0;
    std::string why;  // why the library is unusable
    bool ok = false;
    const char* text(ncclResult_t rc) const { return err ? err(rc) : "?"; }
    static Rccl& get() {
        static Rccl r;
        static std::once_flag once;
        std::call_once(once, [] {
            // OC_HIP_RCCL_LIB names a specific build; otherwise the ROCm soname, then the unversioned name
            const char* names[3] = {getenv("OC_HIP_RCCL_LIB"), "librccl.so.1", "librccl.so"};
            void* h = nullptr;
            for (const char* name : names)
                if (!h && name && *name) h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (!h) {
                const char* d = dlerror();
                r.why = std::string("librccl.so.1 could not be loaded: ") + (d ? d : "?");
                return;
            }
            r.comm_init_all = (decltype(r.comm_init_all))dlsym(h, "ncclCommInitAll");
            r.all_gather = (decltype(r.all_gather))dlsym(h, "ncclAllGather");
            r.group_start = (decltype(r.group_start))dlsym(h, "ncclGroupStart");
            r.group_end = (decltype(r.group_end))dlsym(h, "ncclGroupEnd");
            r.comm_destroy = (decltype(r.comm_destroy))dlsym(h, "ncclCommDestroy");
            r.err = (decltype(r.err))dlsym(h, "ncclGetErrorString");
            r.get_version = (decltype(r.get_version))dlsym(h, "ncclGetVersion");
            if (!(r.comm_init_all && r.all_gather && r.group_start && r.group_end && r.comm_destroy && r.get_version)) {
                r.why = "librccl lacks one of ncclCommInitAll / ncclAllGather / ncclGroupStart / ncclGroupEnd / ncclCommDestroy / ncclGetVersion";
                return;
            }
            if (r.get_version(&r.version) != ncclSuccess || r.version / 10000 != NCCL_MAJOR) {
                r.why = "librccl reports version " + std::to_string(r.version) + ", built against major " + std::to_string(NCCL_MAJOR);
                return;
            }
            r.ok = true;
        });
        return r;
    }
};

void group_drop_comms(oc_hip_engine* e) {
    bool any = e->rccl_comm != nullptr;
    for (oc_hip_engine* r : e->replicas) any = any || r->rccl_comm != nullptr;
    if (!any) return;  // never load RCCL just to find out there is nothing to drop
    Rccl& R = Rccl::get();
    auto drop = [&](oc_hip_engine* m) {
        if (m->rccl_comm && R.ok) {
            (void)hipSetDevice(m->device);
            (void)R.comm_destroy((ncclComm_t)m->rccl_comm);
        }
        m->rccl_comm = nullptr;
    };
    drop(e);
    for (oc_hip_engine* r : e->replicas) drop(r);
}

// Can this group's all-gather be ONE ncclAllGather?  Yes when its members sit on distinct devices (a communicator
// cannot hold a device twice) and there is more than one of them -- or exactly one and "group_force_rccl" is set, which
// runs the identical code (ncclCommInitAll over one device, ncclAllGather on a one-rank communicator) so that the RCCL
// binding executes on a one-GPU machine.
bool group_uses_rccl(oc_hip_engine* e, const std::vector<GroupBlock>& blocks) {
    const int G = (int)blocks.size();
    for (int a = 0; a < G; a++)
        for (int b = a + 1; b < G; b++)
            if (blocks[a].e->device == blocks[b].e->device) return false;
    if (G == 1 && !e->group_force_rccl) return false;
    return true;
}

// Every member's mirror ends up holding the whole queue (blocks of `block` bytes, the last one padded): ONE
// ncclAllGather over xGMI when the members sit on distinct devices, peer copies otherwise (a group may name a device
// twice -- that is how the sharding logic is exercised on a one-GPU box).  RCCL: every member sends its own block from
// where it lies -- the leader straight from the caller's queue (`leader_block`), the others from their slot of their
// mirror (an in-place all-gather for them).
int group_allgather(oc_hip_engine* e, const std::vector<GroupBlock>& blocks, size_t block, const char* leader_block) {
    const int G = (int)blocks.size();
    if (group_uses_rccl(e, blocks)) {
        Rccl& R = Rccl::get();
        if (!R.ok) {
            if (e->group_force_rccl) return fail(OC_HIP_ERR_HIP, "group_force_rccl: %s", R.why.c_str());
        } else {
            if (!e->rccl_comm) {
                std::vector<ncclComm_t> comms(G, nullptr);
                std::vector<int> devs;
                for (const GroupBlock& b : blocks) devs.push_back(b.e->device);
                const ncclResult_t rc = R.comm_init_all(comms.data(), G, devs.data());
                (void)hipSetDevice(e->device);
                if (rc != ncclSuccess) return fail(OC_HIP_ERR_HIP, "ncclCommInitAll over %d devices failed: %s", G, R.text(rc));
                for (int g = 0; g < G; g++) blocks[g].e->rccl_comm = comms[g];
            }
            // whatever happens between ncclGroupStart and ncclGroupEnd, ncclGroupEnd is reached: an open group would
            // poison this thread's later RCCL calls and the cached communicators
            ncclResult_t rc = R.group_start();
            hipError_t herr = hipSuccess;
            if (rc == ncclSuccess) {
                for (int g = 0; g < G && rc == ncclSuccess && herr == hipSuccess; g++) {
                    oc_hip_engine* m = blocks[g].e;
                    herr = hipSetDevice(m->device);
                    if (herr != hipSuccess) break;
                    char* mirror = m->group_mirror.as<char>();
                    const char* mine = g == 0 ? leader_block : mirror + (size_t)g * block;
                    rc = R.all_gather(mine, mirror, block, ncclUint8, (ncclComm_t)m->rccl_comm, m->stream);
                }
                const ncclResult_t rc2 = R.group_end();
                if (rc == ncclSuccess) rc = rc2;
            }
            (void)hipSetDevice(e->device);
            if (herr != hipSuccess) return fail(OC_HIP_ERR_HIP, "hipSetDevice inside the all-gather failed: %s", hipGetErrorString(herr));
            if (rc != ncclSuccess) return fail(OC_HIP_ERR_HIP, "ncclAllGather failed: %s", R.text(rc));
            return OC_HIP_OK;
        }
    }
    // peer copies: the leader's block joins its mirror, then member g fetches every other member's block once that
    // member is done (its event)
    OC_HIP_TRY(hipMemcpyAsync(e->group_mirror.p, leader_block, block, hipMemcpyDeviceToDevice, e->stream));  // blocks[0] is a full block
    OC_HIP_TRY(hipEventRecord(e->group_ev, e->stream));
    for (int g = 0; g < G; g++) {
        oc_hip_engine* m = blocks[g].e;
        OC_HIP_TRY(hipSetDevice(m->device));
        for (int h = 0; h < G; h++) {
            if (h == g) continue;
            oc_hip_engine* src = blocks[h].e;
            OC_HIP_TRY(hipStreamWaitEvent(m->stream, src->group_ev, 0));
            OC_HIP_TRY(hipMemcpyPeerAsync(m->group_mirror.as<char>() + (size_t)h * block, m->device,
                                          src->group_mirror.as<char>() + (size_t)h * block, src->device, block, m->stream));
        }
    }
    OC_HIP_TRY(hipSetDevice(e->device));
    return OC_HIP_OK;
}

// DEVICE queue of a group: the queue lives on the leader's device.  Every other member pulls its block into its own
// mirror (peer copy over xGMI), solves it there on its own stream and pushes the records back; the leader solves
// block 0 in place.  All of it is stream-ordered: members wait for the leader's stream to reach this call, the
// leader's stream waits for the members' completion events.
int compute_group_device(oc_hip_engine* e, char* pois, const float* offsets, size_t count, size_t stride_bytes) {
    const std::vector<GroupBlock> blocks = group_blocks(e, count);
    const int stride_f = (int)(stride_bytes / 4);
    const size_t per = blocks[0].n, block = per * stride_bytes;  // blocks[0] is the largest
    auto event_of = [](oc_hip_engine* m) -> int {
        if (!m->group_ev) OC_HIP_TRY(hipEventCreateWithFlags(&m->group_ev, hipEventDisableTiming));
        return OC_HIP_OK;
    };
    OC_TRY(event_of(e));
    OC_HIP_TRY(hipEventRecord(e->group_ev, e->stream));  // the queue is ready when the leader's stream gets here
    hipEvent_t ready = e->group_ev;
    for (size_t g = 1; g < blocks.size(); g++) {
        oc_hip_engine* m = blocks[g].e;
        const size_t first = blocks[g].first, n = blocks[g].n;
        std::lock_guard<std::mutex> lock(m->mu);
        OC_HIP_TRY(hipSetDevice(m->device));
        OC_TRY(event_of(m));
        OC_TRY(m->group_mirror.reserve(blocks.size() * block));
        OC_HIP_TRY(hipStreamWaitEvent(m->stream, ready, 0));
        if (n) {
            char* mine = m->group_mirror.as<char>() + g * block;
            OC_HIP_TRY(hipMemcpyPeerAsync(mine, m->device, pois + first * stride_bytes, e->device, n * stride_bytes, m->stream));
            const float* d_off = nullptr;
            if (offsets) {
                OC_TRY(m->group_off_mirror.reserve(per * 2 * sizeof(float)));
                OC_HIP_TRY(hipMemcpyPeerAsync(m->group_off_mirror.p, m->device, offsets + 2 * first, e->device, n * 2 * sizeof(float), m->stream));
                d_off = m->group_off_mirror.as<float>();
            }
            OC_TRY(run_compute_device(m, reinterpret_cast<float*>(mine), stride_f, n, d_off));
            OC_HIP_TRY(hipMemcpyPeerAsync(pois + first * stride_bytes, e->device, mine, m->device, n * stride_bytes, m->stream));
        }
        OC_HIP_TRY(hipEventRecord(m->group_ev, m->stream));
    }
    OC_HIP_TRY(hipSetDevice(e->device));
    // the members are busy; now the leader's own block, in place
    if (blocks[0].n) OC_TRY(run_compute_device(e, reinterpret_cast<float*>(pois), stride_f, blocks[0].n, offsets));
    if (e->group_allgather) {
        // one all-gather: every member ends up with every block
        OC_TRY(e->group_mirror.reserve(blocks.size() * block));
        OC_TRY(group_allgather(e, blocks, block, pois));
        for (const GroupBlock& b : blocks) {
            b.e->group_mirror_block = block;
            if (b.e != e) {
                OC_HIP_TRY(hipSetDevice(b.e->device));
                OC_HIP_TRY(hipEventRecord(b.e->group_ev, b.e->stream));
            }
        }
        OC_HIP_TRY(hipSetDevice(e->device));
    }
    for (size_t g = 1; g < blocks.size(); g++) OC_HIP_TRY(hipStreamWaitEvent(e->stream, blocks[g].e->group_ev, 0));
    return OC_HIP_OK;
}

// HOST queue of a group: every member moves and solves its own block (its own host thread, its own PCIe link), the
// results land directly in the caller's vector -- no exchange step is needed for a host-resident queue.
int compute_group_host(oc_hip_engine* e, char* pois, const float* offsets, size_t count, size_t stride_bytes) {
    const std::vector<GroupBlock> blocks = group_blocks(e, count);
    std::vector<int> rc(blocks.size(), OC_HIP_OK);
    std::vector<std::string> msg(blocks.size());
    auto work = [&](size_t g) {
        oc_hip_engine* m = blocks[g].e;
        if (blocks[g].n == 0) return;
        if (hipSetDevice(m->device) != hipSuccess) {
            rc[g] = OC_HIP_ERR_HIP;
            msg[g] = "hipSetDevice failed";
            return;
        }
        rc[g] = compute_host(m, pois + blocks[g].first * stride_bytes, offsets ? offsets + 2 * blocks[g].first : nullptr, blocks[g].n,
                             stride_bytes);
        if (rc[g] != OC_HIP_OK) msg[g] = g_last_error;  // thread-local in the worker
    };
    std::vector<std::thread> threads;
    for (size_t g = 1; g < blocks.size(); g++) threads.emplace_back([&, g] {
        std::lock_guard<std::mutex> lock(blocks[g].e->mu);
        work(g);
    });
    work(0);
    for (std::thread& t : threads) t.join();
    OC_HIP_TRY(hipSetDevice(e->device));
    for (size_t g = 0; g < blocks.size(); g++)
        if (rc[g] != OC_HIP_OK) return fail(rc[g], "group member %zu (device %d): %s", g, blocks[g].e->device, msg[g].c_str());
    return OC_HIP_OK;
}

}  // namespace

extern "C" {

static int compute_impl(oc_hip_engine* e, void* pois, const float* offsets, size_t count, size_t stride_bytes, int memory) {
    OC_ACTIVATE(e);
    if (count == 0) return OC_HIP_OK;
    if (!pois) return fail(OC_HIP_ERR_INVALID, "null POI buffer");
    if (stride_bytes < e->poi_bytes() || (stride_bytes & 3))
        return fail(OC_HIP_ERR_INVALID, "bad POI stride %zu (record is %zu bytes, stride must be a multiple of 4)",
                    stride_bytes, e->poi_bytes());
    if (offsets && e->kind != OC_HIP_ICGN2D1 && e->kind != OC_HIP_ICGN2D2)
        return fail(OC_HIP_ERR_INVALID, "center offsets are an ICGN2D1/ICGN2D2 feature (src/oc_icgn.h:75-76,130-131)");
    std::lock_guard<std::mutex> lock(e->mu);
    TailGuard tail(e);  // also the error exits leave the enqueued work covered by the tail event
    // a queue of a few POIs is not worth waking the other devices for; "group_force_rccl" sends a lone engine's DEVICE
    // queue down the group path as well (a group of one, whose all-gather is a one-rank ncclAllGather)
    const bool lone_rccl = e->replicas.empty() && e->group_allgather && e->group_force_rccl && memory == OC_HIP_DEVICE;
    const bool grouped = (!e->replicas.empty() && count >= 64 * (e->replicas.size() + 1)) || lone_rccl;
    if (memory == OC_HIP_DEVICE) {
        OC_TRY(order_after_default_stream(e));
        if (grouped) OC_TRY(compute_group_device(e, static_cast<char*>(pois), offsets, count, stride_bytes));
        else OC_TRY(run_compute_device(e, static_cast<float*>(pois), (int)(stride_bytes / 4), count, offsets));
        return finish_device_call(e);
    }
    if (grouped) return compute_group_host(e, static_cast<char*>(pois), offsets, count, stride_bytes);
    return compute_host(e, static_cast<char*>(pois), offsets, count, stride_bytes);
}

int oc_hip_compute(oc_hip_engine* e, void* pois, size_t count, size_t stride_bytes, int memory) {
    OC_TRY(check_engine(e));
    return compute_impl(e, pois, nullptr, count, stride_bytes, memory);
}

int oc_hip_compute_with_offsets(oc_hip_engine* e, void* pois, const float* center_offsets, size_t count,
                                size_t stride_bytes, int memory) {
    OC_TRY(check_engine(e));
    if (!center_offsets) return fail(OC_HIP_ERR_INVALID, "null center-offset buffer");
    return compute_impl(e, pois, center_offsets, count, stride_bytes, memory);
}

// Several engines over ONE queue, in order (FFTCC2D then ICGN2D1: examples/test_2d_dic_fftcc_icgn1.cpp:80-99 calls them
// back to back on the same vector).  HOST queues make one round trip over PCIe instead of one per engine: per chunk one
// copy in, every engine's kernels, one copy out.  All engines must live on one device and take the same record type;
// they run on the FIRST engine's stream for the duration of the call (each is ordered behind what its own stream still
// holds -- a prepare() in flight -- and handed back afterwards).
int oc_hip_compute_chain(oc_hip_engine* const* engines, int n_engines, void* pois, size_t count, size_t stride_bytes, int memory) {
    if (!engines || n_engines < 1) return fail(OC_HIP_ERR_INVALID, "compute_chain: need at least one engine");
    for (int i = 0; i < n_engines; i++) OC_TRY(check_engine(engines[i]));
    if (n_engines == 1) return compute_impl(engines[0], pois, nullptr, count, stride_bytes, memory);
    oc_hip_engine* lead = engines[0];
    for (int i = 1; i < n_engines; i++) {
        oc_hip_engine* m = engines[i];
        for (int j = 0; j < i; j++)
            if (engines[j] == m) return fail(OC_HIP_ERR_INVALID, "compute_chain: engine %d is named twice", i);
        if (m->device != lead->device) return fail(OC_HIP_ERR_INVALID, "compute_chain: engines live on different devices (%d, %d)", lead->device, m->device);
        if (m->poi_bytes() != lead->poi_bytes()) return fail(OC_HIP_ERR_INVALID, "compute_chain: 2D and 3D engines cannot share a queue");
        if (!m->replicas.empty() || !lead->replicas.empty() || m->is_replica || lead->is_replica)
            return fail(OC_HIP_ERR_UNSUPPORTED, "compute_chain: device groups run their engines one by one (use oc_hip_compute per engine)");
        if (m->kind == OC_HIP_STRAIN || m->kind == OC_HIP_REGION_FIT || lead->kind == OC_HIP_STRAIN || lead->kind == OC_HIP_REGION_FIT)
            return fail(OC_HIP_ERR_UNSUPPORTED, "compute_chain: Strain / RegionFit have their own prepare / compute calls");
    }
    OC_ACTIVATE(lead);
    if (count == 0) return OC_HIP_OK;
    if (!pois) return fail(OC_HIP_ERR_INVALID, "null POI buffer");
    if (stride_bytes < lead->poi_bytes() || (stride_bytes & 3))
        return fail(OC_HIP_ERR_INVALID, "bad POI stride %zu (record is %zu bytes, stride must be a multiple of 4)", stride_bytes, lead->poi_bytes());
    // every engine's mutex, in address order (two chains over the same engines in different order must not deadlock)
    std::vector<oc_hip_engine*> order(engines, engines + n_engines);
    std::sort(order.begin(), order.end());
    std::vector<std::unique_lock<std::mutex>> locks;
    for (oc_hip_engine* m : order) locks.emplace_back(m->mu);
    // the followers join the lead's stream (device-ordered behind their own pending work) ...
    std::vector<hipStream_t> home(n_engines, nullptr);
    int rc = OC_HIP_OK;
    int joined = 0;
    for (int i = 1; i < n_engines && rc == OC_HIP_OK; i++) {
        home[i] = engines[i]->stream;
        rc = switch_stream(engines[i], lead->stream, false);
        if (rc == OC_HIP_OK) joined = i;
    }
    if (rc == OC_HIP_OK) {
        if (memory == OC_HIP_DEVICE) {
            rc = order_after_default_stream(lead);
            for (int i = 0; i < n_engines && rc == OC_HIP_OK; i++)
                rc = run_compute_device(engines[i], static_cast<float*>(pois), (int)(stride_bytes / 4), count, nullptr);
            if (rc == OC_HIP_OK) rc = finish_device_call(lead);
        } else {
            rc = compute_host(lead, static_cast<char*>(pois), nullptr, count, stride_bytes, engines + 1, n_engines - 1);
        }
    }
    // ... and go home again, ordered behind what the chain enqueued
    const std::string why = g_last_error;
    for (int i = 1; i <= joined; i++) {
        oc_hip_engine* m = engines[i];
        // the end of the chain's work on the lead's stream is this engine's tail there: its home stream waits for it
        if (!m->switch_ev && hipEventCreateWithFlags(&m->switch_ev, hipEventDisableTiming) != hipSuccess) m->switch_ev = nullptr;
        m->tail_marked = m->switch_ev && hipEventRecord(m->switch_ev, lead->stream) == hipSuccess;
        if (!m->tail_marked) (void)hipStreamSynchronize(lead->stream);
        if (switch_stream(m, home[i], true) != OC_HIP_OK) m->stream = home[i];
        (void)mark_tail(m);  // a caller-owned home stream: the chain's work is (transitively) this engine's tail there
    }
    (void)hipGetLastError();
    if (rc != OC_HIP_OK) return fail(rc, "%s", why.c_str());
    return OC_HIP_OK;
}

// compute(POI*) is a batch of ONE: a kernel launch and two PCIe copies per POI, ~1000 times slower per POI than the batch
// path.  Right for callers that genuinely have one POI at a time; a caller that loops over a queue with it gets one hint on
// stderr (per process) at its 4096th call on an engine.  OC_HIP_QUIET=1 silences it.
static void hint_single_poi_loop(oc_hip_engine* e) {
    if (e->single_calls.fetch_add(1, std::memory_order_relaxed) + 1 != 4096) return;
    static std::atomic<bool> said{false};
    const char* q = getenv("OC_HIP_QUIET");
    if ((q && *q && *q != '0') || said.exchange(true)) return;
    fprintf(stderr, "opencorr_hip: compute(POI*) was called 4096 times on one engine -- every call is a GPU launch plus two PCIe copies. "
                    "Hand the POIs over as one queue (compute(std::vector<POI>&), or computeBestOf() for trial positions per POI) for "
                    "~1000x the throughput.  OC_HIP_QUIET=1 silences this hint.\n");
}

int oc_hip_compute_one(oc_hip_engine* e, void* poi) {
    OC_TRY(check_engine(e));
    hint_single_poi_loop(e);
    return compute_impl(e, poi, nullptr, 1, e->poi_bytes(), OC_HIP_HOST);
}

int oc_hip_compute_one_with_offset(oc_hip_engine* e, void* poi, const float* center_offset) {
    OC_TRY(check_engine(e));
    hint_single_poi_loop(e);
    if (!center_offset) return fail(OC_HIP_ERR_INVALID, "null center offset");
    return compute_impl(e, poi, center_offset, 1, e->poi_bytes(), OC_HIP_HOST);
}

int oc_hip_select_best(oc_hip_engine* e, const void* candidates, size_t n_candidates, size_t candidate_stride_bytes,
                       const unsigned* segment_starts, size_t n_segments, void* pois, size_t stride_bytes, int memory) {
    OC_ACTIVATE(e);
    if (n_segments == 0) return OC_HIP_OK;
    if (!candidates || !segment_starts || !pois) return fail(OC_HIP_ERR_INVALID, "select_best: null argument");
    if (candidate_stride_bytes < OC_HIP_POI2D_BYTES || (candidate_stride_bytes & 3) || stride_bytes < OC_HIP_POI2D_BYTES ||
        (stride_bytes & 3))
        return fail(OC_HIP_ERR_INVALID, "select_best: POI2D records need a stride >= %d bytes, multiple of 4", OC_HIP_POI2D_BYTES);
    std::lock_guard<std::mutex> lock(e->mu);
    TailGuard tail(e);  // also the error exits leave the enqueued work covered by the tail event
    OC_TRY(order_after_default_stream(e));
    const float* d_cand = static_cast<const float*>(candidates);
    const unsigned* d_seg = segment_starts;
    float* d_pois = static_cast<float*>(pois);
    if (memory == OC_HIP_HOST) {
        if (segment_starts[n_segments] > n_candidates)
            return fail(OC_HIP_ERR_INVALID, "select_best: the last segment ends at %u, the queue has %zu candidates",
                        segment_starts[n_segments], n_candidates);
        const size_t cb = n_candidates * candidate_stride_bytes, sb = (n_segments + 1) * sizeof(unsigned), pb = n_segments * stride_bytes;
        const size_t off_s = (cb + 255) & ~(size_t)255, off_p = (off_s + sb + 255) & ~(size_t)255;
        OC_TRY(e->poi_stage.reserve(off_p + pb));
        char* base = e->poi_stage.as<char>();
        OC_HIP_TRY(hipMemcpyAsync(base, candidates, cb, hipMemcpyHostToDevice, e->stream));
        OC_HIP_TRY(hipMemcpyAsync(base + off_s, segment_starts, sb, hipMemcpyHostToDevice, e->stream));
        OC_HIP_TRY(hipMemcpyAsync(base + off_p, pois, pb, hipMemcpyHostToDevice, e->stream));
        d_cand = reinterpret_cast<const float*>(base);
        d_seg = reinterpret_cast<const unsigned*>(base + off_s);
        d_pois = reinterpret_cast<float*>(base + off_p);
    }
    OC_HIP_TRY(ochip::launch_poi2d_best_of_segments(d_cand, (int)(candidate_stride_bytes / 4), d_seg, n_segments, d_pois,
                                                    (int)(stride_bytes / 4), e->stream));
    if (memory == OC_HIP_HOST) {
        OC_HIP_TRY(hipMemcpyAsync(pois, d_pois, n_segments * stride_bytes, hipMemcpyDeviceToHost, e->stream));
        OC_HIP_TRY(hipStreamSynchronize(e->stream));
        return OC_HIP_OK;
    }
    return finish_device_call(e);
}

// ---------------------------------------------------------------------------
// reliable / unreliable selection of the RegionFit -> re-ICGN loop (poi_split.hip)
// ---------------------------------------------------------------------------
static int split_params(int ndim, size_t stride_bytes, float low, float high, float conv, int mode, ochip::PoiSplitParams* P) {
    if (ndim != 2 && ndim != 3) return fail(OC_HIP_ERR_INVALID, "ndim must be 2 (POI2D) or 3 (POI3D), got %d", ndim);
    const size_t rec = ndim == 2 ? OC_HIP_POI2D_BYTES : OC_HIP_POI3D_BYTES;
    if (stride_bytes < rec || (stride_bytes & 3)) return fail(OC_HIP_ERR_INVALID, "bad POI stride %zu (record is %zu bytes, stride must be a multiple of 4)", stride_bytes, rec);
    P->mode = mode;
    P->rec_floats = (int)(rec / 4);
    P->zncc_at = ndim == 2 ? 16 : 18;  // result.zncc / result.convergence, src/oc_poi.h:102-136, 187-222
    P->conv_at = ndim == 2 ? 18 : 20;
    P->zncc_low = low;
    P->zncc_high = high;
    P->conv = conv;
    return OC_HIP_OK;
}

// totals[0], totals[1]: records of class 0 / 1; totals[2]: a main-queue index was out of range (merge_recovered)
static int read_split_totals(oc_hip_engine* e, size_t count, size_t totals[3]) {
    unsigned host[3] = {0, 0, 0};
    const unsigned* d = e->split_scratch.as<unsigned>() + ochip::poi_split_scratch_words(count) - 3;
    OC_HIP_TRY(hipMemcpyAsync(host, d, sizeof(host), hipMemcpyDeviceToHost, e->stream));
    OC_HIP_TRY(hipStreamSynchronize(e->stream));
    totals[0] = host[0];
    totals[1] = host[1];
    totals[2] = host[2];
    return OC_HIP_OK;
}

// device records (stride_bytes apart) -> the caller's host queue: only the record's own bytes travel, so whatever the
// caller keeps between records (stride_bytes > record size) stays as it was -- like on the DEVICE path, whose scatter
// writes rec_floats per record
static hipError_t copy_records_to_host(void* dst, const void* src, size_t n, size_t stride_bytes, size_t rec_bytes, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    if (stride_bytes == rec_bytes) return hipMemcpyAsync(dst, src, n * stride_bytes, hipMemcpyDeviceToHost, stream);
    return hipMemcpy2DAsync(dst, stride_bytes, src, stride_bytes, rec_bytes, n, hipMemcpyDeviceToHost, stream);
}

int oc_hip_split_reliable(oc_hip_engine* e, const void* pois, size_t count, size_t stride_bytes, int ndim, float zncc_threshold_low,
                          float zncc_threshold_high, float conv_criterion, void* reliable, size_t reliable_offset, void* unreliable,
                          unsigned* unreliable_index, size_t* n_reliable, size_t* n_unreliable, int memory) {
    OC_ACTIVATE(e);
    if (!n_reliable || !n_unreliable) return fail(OC_HIP_ERR_INVALID, "split_reliable: null count pointer");
    *n_reliable = *n_unreliable = 0;
    if (count == 0) return OC_HIP_OK;
    if (!pois || !reliable || !unreliable || !unreliable_index) return fail(OC_HIP_ERR_INVALID, "split_reliable: null buffer");
    ochip::PoiSplitParams P;
    OC_TRY(split_params(ndim, stride_bytes, zncc_threshold_low, zncc_threshold_high, conv_criterion, 0, &P));
    std::lock_guard<std::mutex> lock(e->mu);
    TailGuard tail(e);  // also the error exits leave the enqueued work covered by the tail event
    OC_TRY(order_after_default_stream(e));
    OC_TRY(e->split_scratch.reserve(ochip::poi_split_scratch_words(count) * sizeof(unsigned)));
    const int stride_f = (int)(stride_bytes / 4);
    const size_t rec_bytes = (size_t)P.rec_floats * 4;
    size_t totals[3];
    if (memory == OC_HIP_DEVICE) {
        OC_HIP_TRY(ochip::launch_poi_split(static_cast<const float*>(pois), stride_f, count, P, nullptr, static_cast<float*>(reliable),
                                           reliable_offset, nullptr, static_cast<float*>(unreliable), unreliable_index, nullptr, 0,
                                           e->split_scratch.as<unsigned>(), e->stream));
        OC_TRY(read_split_totals(e, count, totals));
    } else {
        const size_t qb = count * stride_bytes;
        OC_TRY(e->poi_stage.reserve(3 * qb + count * sizeof(unsigned)));
        char* base = e->poi_stage.as<char>();
        float* d_in = reinterpret_cast<float*>(base);
        float* d_rel = reinterpret_cast<float*>(base + qb);
        float* d_unr = reinterpret_cast<float*>(base + 2 * qb);
        unsigned* d_idx = reinterpret_cast<unsigned*>(base + 3 * qb);
        OC_HIP_TRY(hipMemcpyAsync(d_in, pois, qb, hipMemcpyHostToDevice, e->stream));
        OC_HIP_TRY(ochip::launch_poi_split(d_in, stride_f, count, P, nullptr, d_rel, 0, nullptr, d_unr, d_idx, nullptr, 0,
                                           e->split_scratch.as<unsigned>(), e->stream));
        OC_TRY(read_split_totals(e, count, totals));
        OC_HIP_TRY(copy_records_to_host(static_cast<char*>(reliable) + reliable_offset * stride_bytes, d_rel, totals[0], stride_bytes, rec_bytes, e->stream));
        if (totals[1]) {
            OC_HIP_TRY(copy_records_to_host(unreliable, d_unr, totals[1], stride_bytes, rec_bytes, e->stream));
            OC_HIP_TRY(hipMemcpyAsync(unreliable_index, d_idx, totals[1] * sizeof(unsigned), hipMemcpyDeviceToHost, e->stream));
        }
        OC_HIP_TRY(hipStreamSynchronize(e->stream));
    }
    *n_reliable = totals[0];
    *n_unreliable = totals[1];
    return OC_HIP_OK;
}

int oc_hip_merge_recovered(oc_hip_engine* e, void* pois, size_t count, size_t stride_bytes, int ndim, void* unreliable,
                           unsigned* unreliable_index, size_t n_unreliable, float zncc_threshold_high, float conv_criterion, void* reliable,
                           size_t reliable_offset, size_t* n_recovered, size_t* n_remaining, int memory) {
    OC_ACTIVATE(e);
    if (!n_recovered || !n_remaining) return fail(OC_HIP_ERR_INVALID, "merge_recovered: null count pointer");
    *n_recovered = 0;
    *n_remaining = 0;
    if (n_unreliable == 0) return OC_HIP_OK;
    if (!pois || !reliable || !unreliable || !unreliable_index) return fail(OC_HIP_ERR_INVALID, "merge_recovered: null buffer");
    ochip::PoiSplitParams P;
    OC_TRY(split_params(ndim, stride_bytes, 0.f, zncc_threshold_high, conv_criterion, 1, &P));
    std::lock_guard<std::mutex> lock(e->mu);
    TailGuard tail(e);  // also the error exits leave the enqueued work covered by the tail event
    OC_TRY(order_after_default_stream(e));
    OC_TRY(e->split_scratch.reserve(ochip::poi_split_scratch_words(n_unreliable) * sizeof(unsigned)));
    const int stride_f = (int)(stride_bytes / 4);
    const size_t qb = n_unreliable * stride_bytes, ib = n_unreliable * sizeof(unsigned);
    const size_t rec_bytes = (size_t)P.rec_floats * 4;
    size_t totals[3];
    if (memory == OC_HIP_DEVICE) {
        // the POIs that stay unreliable are compacted into a scratch copy first (an in-place compaction would overwrite
        // records other threads still have to read), then moved back to the front of the caller's arrays
        OC_TRY(e->split_tmp.reserve(qb + ib));
        float* t_rec = e->split_tmp.as<float>();
        unsigned* t_idx = reinterpret_cast<unsigned*>(e->split_tmp.as<char>() + qb);
        // the index list is checked on the device BEFORE the scatter may write anything of the caller's: a list
        // with an entry outside the main queue is refused with `reliable`, `pois`, `unreliable` and the list itself untouched
        {
            unsigned* flag = e->split_scratch.as<unsigned>() + ochip::poi_split_scratch_words(n_unreliable) - 1;
            OC_HIP_TRY(ochip::launch_poi_index_range(unreliable_index, n_unreliable, count, flag, e->stream));
            unsigned bad = 0;
            OC_HIP_TRY(hipMemcpyAsync(&bad, flag, sizeof(bad), hipMemcpyDeviceToHost, e->stream));
            OC_HIP_TRY(hipStreamSynchronize(e->stream));
            if (bad) return fail(OC_HIP_ERR_INVALID, "merge_recovered: an unreliable_index entry is >= the main queue's %zu records (nothing was changed)", count);
        }
        OC_HIP_TRY(ochip::launch_poi_split(static_cast<const float*>(unreliable), stride_f, n_unreliable, P, unreliable_index,
                                           static_cast<float*>(reliable), reliable_offset, nullptr, t_rec, t_idx, static_cast<float*>(pois),
                                           count, e->split_scratch.as<unsigned>(), e->stream));
        OC_TRY(read_split_totals(e, n_unreliable, totals));
        // an index outside the main queue: the kernel wrote nothing through it; the caller's lists are left as they were
        if (totals[2]) return fail(OC_HIP_ERR_INVALID, "merge_recovered: an unreliable_index entry is >= the main queue's %zu records", count);
        if (totals[1]) {
            OC_HIP_TRY(hipMemcpyAsync(unreliable, t_rec, totals[1] * stride_bytes, hipMemcpyDeviceToDevice, e->stream));
            OC_HIP_TRY(hipMemcpyAsync(unreliable_index, t_idx, totals[1] * sizeof(unsigned), hipMemcpyDeviceToDevice, e->stream));
        }
        *n_recovered = totals[0];
        *n_remaining = totals[1];
        return finish_device_call(e);
    }
    // host queues: every index is checked BEFORE anything is enqueued or touched (the host knows the list)
    for (size_t j = 0; j < n_unreliable; j++)
        if (unreliable_index[j] >= count)
            return fail(OC_HIP_ERR_INVALID, "merge_recovered: unreliable_index[%zu] = %u is >= the main queue's %zu records", j, unreliable_index[j], count);
    // the classification and both compactions run on the device; the host only moves the recovered records to where the
    // device's index list says they go
    OC_TRY(e->poi_stage.reserve(3 * qb + 3 * ib));
    char* base = e->poi_stage.as<char>();
    float* d_in = reinterpret_cast<float*>(base);
    float* d_rec = reinterpret_cast<float*>(base + qb);
    float* d_rem = reinterpret_cast<float*>(base + 2 * qb);
    unsigned* d_idx_in = reinterpret_cast<unsigned*>(base + 3 * qb);
    unsigned* d_idx_rec = d_idx_in + n_unreliable;
    unsigned* d_idx_rem = d_idx_rec + n_unreliable;
    OC_HIP_TRY(hipMemcpyAsync(d_in, unreliable, qb, hipMemcpyHostToDevice, e->stream));
    OC_HIP_TRY(hipMemcpyAsync(d_idx_in, unreliable_index, ib, hipMemcpyHostToDevice, e->stream));
    OC_HIP_TRY(ochip::launch_poi_split(d_in, stride_f, n_unreliable, P, d_idx_in, d_rec, 0, d_idx_rec, d_rem, d_idx_rem, nullptr, 0,
                                       e->split_scratch.as<unsigned>(), e->stream));
    OC_TRY(read_split_totals(e, n_unreliable, totals));
    std::vector<unsigned> rec_idx(totals[0]);
    char* rel_dst = static_cast<char*>(reliable) + reliable_offset * stride_bytes;
    if (totals[0]) {
        OC_HIP_TRY(copy_records_to_host(rel_dst, d_rec, totals[0], stride_bytes, rec_bytes, e->stream));
        OC_HIP_TRY(hipMemcpyAsync(rec_idx.data(), d_idx_rec, totals[0] * sizeof(unsigned), hipMemcpyDeviceToHost, e->stream));
    }
    if (totals[1]) {
        OC_HIP_TRY(copy_records_to_host(unreliable, d_rem, totals[1], stride_bytes, rec_bytes, e->stream));
        OC_HIP_TRY(hipMemcpyAsync(unreliable_index, d_idx_rem, totals[1] * sizeof(unsigned), hipMemcpyDeviceToHost, e->stream));
    }
    OC_HIP_TRY(hipStreamSynchronize(e->stream));
    for (size_t j = 0; j < totals[0]; j++)
        std::memcpy(static_cast<char*>(pois) + (size_t)rec_idx[j] * stride_bytes, rel_dst + j * stride_bytes, rec_bytes);
    *n_recovered = totals[0];
    *n_remaining = totals[1];
    return OC_HIP_OK;
}

int oc_hip_set_self_adaptive(oc_hip_engine* e, int enable) {
    OC_TRY(check_engine(e));
    if (!e->is_icgn2d())
        return fail(OC_HIP_ERR_UNSUPPORTED, "self-adaptive subsets are implemented for ICGN2D1/2D2 and ICLM2D1/2D2 (NR2D1 has none in the reference)");
    std::lock_guard<std::mutex> lock(e->mu);
    e->self_adaptive = enable != 0;
    for (oc_hip_engine* r : e->replicas) OC_TRY(oc_hip_set_self_adaptive(r, enable));
    return OC_HIP_OK;
}

int oc_hip_synchronize(oc_hip_engine* e) {
    OC_ACTIVATE(e);
    OC_HIP_TRY(hipStreamSynchronize(e->stream));
    return OC_HIP_OK;
}

int oc_hip_get_kind(const oc_hip_engine* e, int* kind) {
    OC_TRY(check_engine(e));
    if (!kind) return fail(OC_HIP_ERR_INVALID, "null kind");
    *kind = e->kind;
    return OC_HIP_OK;
}

int oc_hip_get_field(const oc_hip_engine* e, const char* name, const float** ptr, size_t* count) {
    OC_TRY(check_engine(e));
    if (!name || !ptr || !count) return fail(OC_HIP_ERR_INVALID, "null argument");
    *ptr = nullptr;
    *count = 0;
    if (!e->img) return fail(OC_HIP_ERR_INVALID, "get_field(%s): no images set", name);
    const size_t n = e->img->count();
    const std::string s(name);
    if (s == "ref") { *ptr = e->img->ref_ptr(); *count = n; }
    else if (s == "tar") { *ptr = e->img->tar_ptr(); *count = n; }
    else if (s == "gx" && e->ref_ready) { *ptr = e->gx.as<float>(); *count = n; }
    else if (s == "gy" && e->ref_ready) { *ptr = e->gy.as<float>(); *count = n; }
    else if (s == "gz" && e->ref_ready && e->is3d()) { *ptr = e->gz.as<float>(); *count = n; }
    else if (s == "lut" && e->tar_ready && !e->is3d()) { *ptr = e->coef.as<float>(); *count = n * 16; }
    else if (s == "lut_gx" && e->tar_ready && e->kind == OC_HIP_NR2D1) { *ptr = e->coef_gx.as<float>(); *count = n * 16; }
    else if (s == "lut_gy" && e->tar_ready && e->kind == OC_HIP_NR2D1) { *ptr = e->coef_gy.as<float>(); *count = n * 16; }
    else if (s == "coef" && e->tar_ready && e->is3d()) { *ptr = e->coef.as<float>(); *count = n; }
    else return fail(OC_HIP_ERR_INVALID, "get_field: '%s' is unknown or not built yet", name);
    return OC_HIP_OK;
}

int oc_hip_read_field(oc_hip_engine* e, const char* name, float* host_dst, size_t count) {
    OC_ACTIVATE(e);
    const float* p = nullptr;
    size_t n = 0;
    OC_TRY(oc_hip_get_field(e, name, &p, &n));
    if (!host_dst || count != n) return fail(OC_HIP_ERR_INVALID, "read_field(%s): expected %zu floats, got %zu", name, n, count);
    OC_HIP_TRY(hipMemcpyAsync(host_dst, p, n * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    OC_HIP_TRY(hipStreamSynchronize(e->stream));
    return OC_HIP_OK;
}

int oc_hip_profile_enable(oc_hip_engine* e, int enable) {
    OC_TRY(check_engine(e));
    std::lock_guard<std::mutex> lock(e->mu);
    e->prof = enable != 0;
    return OC_HIP_OK;
}

int oc_hip_profile_reset(oc_hip_engine* e) {
    OC_ACTIVATE(e);
    std::lock_guard<std::mutex> lock(e->mu);
    OC_HIP_TRY(hipStreamSynchronize(e->stream));
    clear_events(e);
    return OC_HIP_OK;
}

int oc_hip_profile_read(oc_hip_engine* e, double* total_ms, long* launches) {
    OC_ACTIVATE(e);
    if (!total_ms || !launches) return fail(OC_HIP_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lock(e->mu);
    OC_HIP_TRY(hipStreamSynchronize(e->stream));
    double total = 0.0;
    for (auto& ev : e->events) {
        float ms = 0.f;
        OC_HIP_TRY(hipEventElapsedTime(&ms, ev.first, ev.second));
        total += ms;
    }
    *total_ms = total;
    *launches = (long)e->events.size();
    return OC_HIP_OK;
}

}  // extern "C"
