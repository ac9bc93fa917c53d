// capi.hip -- implementation of the C-ABI declared in include/opencorr_hip.h.
//
// Host-side engine objects: device-resident image pair, precomputed fields,
// POI staging, rocFFT plans, stream and profiling events.  No CPU compute path
// exists here: every compute call ends in HIP kernel launches or fails.
#include "capi_internal.h"

namespace ochip_capi {

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

static std::once_flag g_rocfft_once;
void rocfft_init_once() {
    std::call_once(g_rocfft_once, [] { rocfft_setup(); });
}

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

}  // namespace ochip_capi

namespace {

// ---------------------------------------------------------------------------
// FFTCC2D pipeline
// ---------------------------------------------------------------------------
size_t fftcc_chunk_limit() {
#if OC_BUILD_AB
    // (experiments, A/B build only: POIs per batch of the rocFFT pipeline)
    const char* s = getenv("OC_HIP_FFTCC_CHUNK");
    if (s && *s) {
        long v = atol(s);
        if (v > 0) return (size_t)v;
    }
#endif
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
    // -- when variant 2's two arrays per wave would leave a CU at most two workgroups (subsets from 26 passes = 41 x 40 up: the
    // measured case, radii 12 ... 20: 4.91 -> 3.98 ms); smaller batches keep variant 2, which is 4 % ahead at equal occupancy
    // (3.49 vs 3.61 ms on a uniform r = 16: one image read per sample fewer).  ADVICE r5.
    if (e->self_adaptive && (e->icgn2d_variant < 0 || ochip::icgn2d_variant_uses_table(variant))) {
        const long long passes_sa = (N + 63) / 64;
        const bool crowded = 3 * (2 * 4 * passes_sa * 256) > (160 * 1024 - 2048);   // three workgroups of variant 2 no longer fit
        variant = crowded ? 7 : (dof == 12 ? 3 : 2);
    }
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
#if OC_BUILD_AB
    const char* s = getenv("OC_HIP_FFTCC3D_CHUNK");
    if (s && *s) {
        long v = atol(s);
        if (v > 0) return (size_t)v;
    }
#endif
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
#if OC_BUILD_AB
        const bool fused32_r5 = fused32 && e->fftcc3d_fused == 2;   // the decomposition of rounds 1 - 5 (fftcc3d_fused_r5.hip)
        if (fused32_r5) OC_TRY(e->flags.reserve(ochip::fftcc3d_fused_flag_bytes(count < kMaxGrid ? count : kMaxGrid)));   // once, before anything is enqueued
#endif
        for (size_t first = 0; first < count; first += kMaxGrid) {
            const size_t n = (count - first) < kMaxGrid ? (count - first) : kMaxGrid;
            float* q = d_pois + first * (size_t)stride_f;
            hipError_t err =
#if OC_BUILD_AB
                             fused32_r5 ? ochip::launch_fftcc3d_fused_r5(P, q, stride_f, n, e->icgn2d_xcd != 0, e->flags.as<unsigned char>(), e->stream) :
#endif
                             fused32 ? ochip::launch_fftcc3d_fused(P, q, stride_f, n, e->icgn2d_xcd != 0, e->stream)
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

}  // namespace

int ochip_capi::run_compute_device(oc_hip_engine* e, float* d_pois, int stride_f, size_t count, const float* d_offsets) {
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
#if !OC_BUILD_AB
        // (the split launch shape, variant 8, exists in the A/B build only: the key would have no effect here -- ADVICE r5)
        if (value != 0)
            return fail(OC_HIP_ERR_UNSUPPORTED, "icgn2d_split_chunks belongs to icgn2d_variant 8, an A/B partner that only the A/B build of the "
                                                "library contains (python -m opencorr_amd.build --ab)");
#endif
        if (value < 0 || value > 256) return fail(OC_HIP_ERR_INVALID, "icgn2d_split_chunks must be 0 (back to back) ... 256");
        e->icgn2d_split_chunks = value;
    } else if (k == "icgn2d_tile_px") {
        if (value < 0 || (value > 0 && value < 16)) return fail(OC_HIP_ERR_INVALID, "icgn2d_tile_px must be 0 (off) or >= 16");
        e->icgn2d_tile_px = value;
    } else if (k == "fftcc2d_fused") {
        e->fftcc2d_fused = value == 2 ? 2 : (value != 0);
    } else if (k == "fftcc3d_fused") {
#if !OC_BUILD_AB
        if (value == 2)
            return fail(OC_HIP_ERR_UNSUPPORTED, "fftcc3d_fused = 2 (the 32^3 kernel of rounds 1 - 5) is an A/B partner that only the A/B build of the "
                                                "library contains (python -m opencorr_amd.build --ab)");
#endif
        e->fftcc3d_fused = value == 2 ? 2 : (value != 0);
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
    } else if (k == "single_combine") {
        e->single_combine = value != 0;
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
    fprintf(stderr, "opencorr_hip: compute(POI*) was called 4096 times on one engine -- calls from concurrent threads are combined into one "
                    "launch per batch, but a sequential loop still pays a GPU launch plus two PCIe copies per POI.  Hand the POIs over as "
                    "one queue (compute(std::vector<POI>&), or computeBestOf() for trial positions per POI) for ~1000x the throughput.  "
                    "OC_HIP_QUIET=1 silences this hint.\n");
}

// One batch of combined single-POI requests through the engine's host-queue path: records gathered into one buffer, ONE
// compute call, records scattered back.  Requests with and without a centre offset cannot share a launch (the engine takes
// an offset queue or none): the leader serves them as two sub-batches.
static void serve_single_batch(oc_hip_engine* e, std::vector<oc_hip_engine::SingleRequest*>& batch) {
    const size_t rec = e->poi_bytes();
    for (int with_off = 0; with_off < 2; with_off++) {
        std::vector<oc_hip_engine::SingleRequest*> part;
        for (auto* r : batch)
            if ((r->offset != nullptr) == (with_off != 0)) part.push_back(r);
        if (part.empty()) continue;
        int rc;
        const auto t_begin = std::chrono::steady_clock::now();
        if (part.size() == 1) {
            rc = compute_impl(e, part[0]->poi, part[0]->offset, 1, rec, OC_HIP_HOST);
        } else {
            rc = e->single_buf.reserve(part.size() * rec);
            char* buf = static_cast<char*>(e->single_buf.p);
            float* off = nullptr;
            if (rc == OC_HIP_OK && with_off) {
                rc = e->single_off.reserve(2 * part.size() * sizeof(float));
                off = static_cast<float*>(e->single_off.p);
            }
            if (rc == OC_HIP_OK) {
                for (size_t i = 0; i < part.size(); i++) {
                    memcpy(buf + i * rec, part[i]->poi, rec);
                    if (off) {
                        off[2 * i] = part[i]->offset[0];
                        off[2 * i + 1] = part[i]->offset[1];
                    }
                }
                rc = compute_impl(e, buf, off, part.size(), rec, OC_HIP_HOST);
            }
            if (rc == OC_HIP_OK)
                for (size_t i = 0; i < part.size(); i++) memcpy(part[i]->poi, buf + i * rec, rec);
        }
        e->single_engine_ns.fetch_add((unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_begin).count(),
                                      std::memory_order_relaxed);
        e->single_batches.fetch_add(1, std::memory_order_relaxed);
        e->single_batched_pois.fetch_add(part.size(), std::memory_order_relaxed);
        for (auto* r : part) {
            r->rc = rc;
            if (rc != OC_HIP_OK) r->error = g_last_error;   // (thread-local in the leader: handed to the request's owner)
        }
    }
}

// compute(POI*) / compute(POI*, center_offset): queue the request.  Whoever finds no leader becomes one: it takes everything
// that is queued as ONE batch, serves it, wakes the owners of the served requests (each on its own condition variable) and
// goes on with what has arrived meanwhile -- back to back, so the GPU never waits for a thread to wake up.  Its own request
// sits in its first batch; after kSingleExtraBatches further batches it promotes the owner of a queued request to leader and
// returns (no caller serves the others for ever).  A POI's result does not depend on the batch it travels in (the per-POI
// solves are independent: tests/test_gpu_parity_2d.py::test_host_pipeline_chunks_change_no_bits).
constexpr int kSingleExtraBatches = 32;

static int compute_single(oc_hip_engine* e, void* poi, const float* offset) {
    if (!poi) return fail(OC_HIP_ERR_INVALID, "null POI");
    if (!e->single_combine) return compute_impl(e, poi, offset, 1, e->poi_bytes(), OC_HIP_HOST);
    oc_hip_engine::SingleRequest req(poi, offset);
    bool lead;
    {
        std::lock_guard<std::mutex> lk(e->single_mu);
        e->single_pending.push_back(&req);
        lead = !e->single_leader;
        if (lead) e->single_leader = true;
    }
    // Publishing a request's new state is the leader's LAST access to it (the owner may return at once).  Owners whose spin
    // budget ran out sleep on the engine's condition variable: they register under its mutex and re-check their state there, the
    // leader takes the same mutex after its stores -- no wake-up is lost, and nobody pays a futex call while everyone spins.
    auto wake_sleepers = [&]() {
        if (e->single_sleepers.load(std::memory_order_acquire) > 0) {
            std::lock_guard<std::mutex> ls(e->single_sleep_mu);
            e->single_sleep_cv.notify_all();
        }
    };
    if (!lead) {
        // a short busy wait (the batch in flight is usually ~50 us from done), then polite polling -- yielding the core between
        // looks, so that 60 waiting threads do not crowd out the leader and the HIP runtime's own threads --, then sleep
        for (int spin = 0; spin < 1500 && req.state.load(std::memory_order_acquire) == 0; spin++) __builtin_ia32_pause();
        if (req.state.load(std::memory_order_acquire) == 0) {
            const auto t0 = std::chrono::steady_clock::now();
            while (req.state.load(std::memory_order_acquire) == 0 && std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(400))
                std::this_thread::yield();
        }
        if (req.state.load(std::memory_order_acquire) == 0) {
            std::unique_lock<std::mutex> ls(e->single_sleep_mu);
            e->single_sleepers.fetch_add(1, std::memory_order_acq_rel);
            e->single_sleep_cv.wait(ls, [&] { return req.state.load(std::memory_order_acquire) != 0; });
            e->single_sleepers.fetch_sub(1, std::memory_order_acq_rel);
        }
        lead = req.state.load(std::memory_order_acquire) == 2;   // promoted while still queued: this thread leads now
    }
    if (lead) {
        std::vector<oc_hip_engine::SingleRequest*> batch;
        bool own_done = false;
        for (int served = 0;; served++) {
            {
                std::lock_guard<std::mutex> lk(e->single_mu);
                if (e->single_pending.empty()) {
                    e->single_leader = false;   // (its own request was queued before this thread became leader: it is done)
                    break;
                }
                if (own_done && served > kSingleExtraBatches) {
                    e->single_pending.front()->state.store(2, std::memory_order_release);   // its owner takes over; the request stays queued
                    wake_sleepers();
                    break;
                }
                batch.clear();
                batch.swap(e->single_pending);
            }
            serve_single_batch(e, batch);
            for (auto* r : batch) {
                if (r == &req) own_done = true;
                else r->state.store(1, std::memory_order_release);
            }
            wake_sleepers();
        }
    }
    if (req.rc != OC_HIP_OK) return fail(req.rc, "%s", req.error.c_str());
    return OC_HIP_OK;
}

int oc_hip_compute_one(oc_hip_engine* e, void* poi) {
    OC_TRY(check_engine(e));
    hint_single_poi_loop(e);
    return compute_single(e, poi, nullptr);
}

int oc_hip_compute_one_with_offset(oc_hip_engine* e, void* poi, const float* center_offset) {
    OC_TRY(check_engine(e));
    hint_single_poi_loop(e);
    if (!center_offset) return fail(OC_HIP_ERR_INVALID, "null center offset");
    if (e->kind != OC_HIP_ICGN2D1 && e->kind != OC_HIP_ICGN2D2)
        return fail(OC_HIP_ERR_INVALID, "center offsets are an ICGN2D1/ICGN2D2 feature (src/oc_icgn.h:75-76,130-131)");
    return compute_single(e, poi, center_offset);
}

// batches / POIs the combining front end has served on this engine (tests, diagnostics)
int oc_hip_single_stats(oc_hip_engine* e, unsigned long long* batches, unsigned long long* pois) {
    OC_TRY(check_engine(e));
    if (batches) *batches = e->single_batches.load();
    if (pois) *pois = e->single_batched_pois.load();
    if (getenv("OC_HIP_SINGLE_DEBUG")) fprintf(stderr, "opencorr_hip: single-POI front end: %llu batches, %llu POIs, %.3f ms inside the engine\n",
                                               e->single_batches.load(), e->single_batched_pois.load(), e->single_engine_ns.load() * 1e-6);
    return OC_HIP_OK;
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
