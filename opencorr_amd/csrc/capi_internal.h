// capi_internal.h -- what the translation units of the C-ABI (capi.hip: engines and entry points; capi_host.hip: the chunked
// host-queue pipeline; capi_group.hip: device groups + the RCCL binding; capi_strain.hip: Strain / RegionFit and the reliable /
// unreliable selection) share: the engine object, device buffers, error reporting, the stream / tail-event helpers.
#pragma once
#include "../../include/opencorr_hip.h"

#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
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

namespace ochip_capi {

// the last error message of the calling thread (oc_hip_last_error) and the one way to set it: defined in capi.hip
extern thread_local std::string g_last_error;
int fail(int code, const char* fmt, ...);

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

// grow-only page-locked host buffer (the combining front end of compute(POI*) stages its batches here: copies from / to
// pinned memory are plain asynchronous DMA, pageable ones go through the runtime's own staging with a wait each)
struct PinnedBuf {
    void* p = nullptr;
    size_t bytes = 0;
    ~PinnedBuf() { release(); }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        bytes = 0;
    }
    int reserve(size_t n) {
        if (n <= bytes) return OC_HIP_OK;
        release();
        n = (n + 65535) & ~(size_t)65535;
        hipError_t err = hipHostMalloc(&p, n, hipHostMallocDefault);
        if (err != hipSuccess) {
            p = nullptr;
            return fail(OC_HIP_ERR_NOMEM, "hipHostMalloc(%zu bytes) failed: %s", n, hipGetErrorString(err));
        }
        bytes = n;
        return OC_HIP_OK;
    }
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

void rocfft_init_once();   // capi.hip

}  // namespace ochip_capi
using namespace ochip_capi;

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
    // Combining front end of compute(POI*) (round 6): the reference's single-POI form is called from the CALLER's own OpenMP
    // loops (src/oc_epipolar_search.cpp:184-188, relying on one scratch instance per thread, src/oc_icgn.cpp:61-69,147).
    // Callers that arrive while a launch is in flight are queued; one of them (the leader) hands the whole batch to the
    // engine as ONE queue -- T threads cost one launch per ~T POIs instead of T serialised launches.
    struct SingleRequest {
        void* poi;
        const float* offset;
        int rc = OC_HIP_OK;
        std::string error;
        // The owner spins on `state` (a batch takes ~50 us: cheaper than a futex round trip per served thread), then sleeps on the
        // ENGINE's condition variable.  The leader's last access to a request is the store to `state`: once it is non-zero the
        // owner may return and the request is gone.
        std::atomic<int> state{0};   // 0 = queued, 1 = served, 2 = promoted to leader while still queued
        SingleRequest(void* p, const float* o) : poi(p), offset(o) {}
    };
    std::mutex single_mu;
    std::mutex single_sleep_mu;                  // sleepers of the front end (owners whose spin budget ran out)
    std::condition_variable single_sleep_cv;
    std::atomic<int> single_sleepers{0};
    std::vector<SingleRequest*> single_pending;
    bool single_leader = false;
    int single_combine = 1;                      // "single_combine" tuning key: 0 = every call a launch of its own (the round-5 behaviour)
    std::atomic<unsigned long long> single_batches{0}, single_batched_pois{0}, single_engine_ns{0};   // (engine_ns: time inside the engine calls)
    PinnedBuf single_buf, single_off;            // the leader's contiguous, page-locked copy of a batch's records (and offsets)
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

namespace ochip_capi {

void group_drop_comms(oc_hip_engine* e);  // RCCL communicators of a device group (capi_group.hip)

inline int check_engine(const oc_hip_engine* e) {
    if (!e) return fail(OC_HIP_ERR_INVALID, "null engine handle");
    return OC_HIP_OK;
}

inline int activate(const oc_hip_engine* e) {
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

// The engine's private stream is non-blocking: nothing the caller enqueued is ordered against it.  Inputs that
// live on the device (OC_HIP_DEVICE queues, offsets, images used in place) are usually produced on the legacy
// default stream -- hipMemcpy, plain launches, torch unless told otherwise -- so every entry point that enqueues work
// on the private stream first makes it wait for what the default stream holds at that moment.  Producers on OTHER
// streams must be complete, or be named with oc_hip_set_stream (documented in opencorr_hip.h).
inline int order_after_default_stream(oc_hip_engine* e) {
    if (e->stream != e->own_stream) return OC_HIP_OK;  // caller-chosen stream: stream order is the caller's
    if (!e->order_ev) OC_HIP_TRY(hipEventCreateWithFlags(&e->order_ev, hipEventDisableTiming));
    OC_HIP_TRY(hipEventRecord(e->order_ev, nullptr));
    OC_HIP_TRY(hipStreamWaitEvent(e->own_stream, e->order_ev, 0));
    return OC_HIP_OK;
}

// See oc_hip_engine::switch_ev.  Entry points that return with work enqueued on a caller-owned stream end with this.
inline int mark_tail(oc_hip_engine* e) {
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
inline void drain_engine(oc_hip_engine* e) {
    if (e->stream == e->own_stream) {
        if (e->own_stream) (void)hipStreamSynchronize(e->own_stream);
    } else {
        if (e->tail_marked && e->switch_ev) (void)hipEventSynchronize(e->switch_ev);
        if (e->own_stream) (void)hipStreamSynchronize(e->own_stream);
    }
    (void)hipGetLastError();
}

inline void clear_events(oc_hip_engine* e) {
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

// ---- functions one translation unit defines and another one calls ----------------------------------------------------
// capi.hip
int create_engine(int kind, int rx, int ry, int rz, float conv, float stop, int device, oc_hip_engine** out);
int run_compute_device(oc_hip_engine* e, float* d_pois, int stride_f, size_t count, const float* d_offsets = nullptr);
// capi_host.hip: the chunked host-queue pipeline
int compute_host(oc_hip_engine* e, char* pois, const float* offsets, size_t count, size_t stride_bytes, oc_hip_engine* const* chain = nullptr,
                 int n_chain = 0);
// capi_group.hip: device groups
int compute_group_device(oc_hip_engine* e, char* pois, const float* offsets, size_t count, size_t stride_bytes);
int compute_group_host(oc_hip_engine* e, char* pois, const float* offsets, size_t count, size_t stride_bytes);

// a DEVICE-queue call returns with its work enqueued: on a caller-owned stream the tail event covers it, on the engine's own
// stream the call completes here
inline int finish_device_call(oc_hip_engine* e) {
    if (e->stream == e->own_stream) OC_HIP_TRY(hipStreamSynchronize(e->stream));
    return mark_tail(e);
}

}  // namespace ochip_capi
using namespace ochip_capi;
