// nr2d.hip -- NR2D1 (forward-additive Newton-Raphson, first-order shape function) on gfx950.
//
// Replaces NR2D1::compute(POI2D*) (src/oc_nr.cpp:160-322) for a whole POI queue (:324-332);
// SURVEY.md 8(f) row 3.  Unlike IC-GN the steepest-descent images and the Hessian are rebuilt in
// every iteration from the gradients of the WARPED TARGET, which NR2D1::prepare (:119-158)
// makes interpolable: three bicubic LUTs (target, d/dx target, d/dy target), 192 B per sample
// and iteration.
//
// Mapping: one 64-lane wavefront per POI, four consecutive POIs per workgroup, XCD-contiguous
// (as icgn2d.hip).  Sample s = r*W + c is owned by lane s % 64.  Per-sample state in LDS as
// [t][lane] arrays: zero-mean reference subset, warped target subset and its two gradients.
// Reductions use the association of oc_device.h (OC_ORDER_LANES of the oracle): bit-identical.
//
// Per iteration: (1) sweep -- warp, ONE range test / address for the three LUTs, 12 x 16-byte
// gathers, three 16-term polynomials, 21 Hessian sums (packed fp32); (2) mean, norm;
// (3) Hessian reduction + lane-distributed 6x6 LU inverse (Eigen PartialPivLU order);
// (4) error image, ZNSSD, numerator; (5) dp = H^-1 b, p += dp.
#include <atomic>

#include "dic2d_device.h"
#include "oc_kernels.h"

namespace ochip {

struct Nr2dLaunch {
    int stride_f;              // floats between POI records
    int nt;                    // ceil(N / 64)
    int xcd_chunk;             // > 0: workgroup b serves POI group (b % 8) * xcd_chunk + b / 8
    unsigned long long count;  // POIs
};

#ifndef OC_NR_WAVES
#define OC_NR_WAVES 4
#endif
constexpr int kNrWaves = OC_NR_WAVES;  // POIs (waves) per workgroup
// Per-sample LDS arrays of a wave: 4 = zero-mean reference, warped target, its two gradients; 3 = the reference value is
// re-read from the image in the numerator pass and re-centred with the same subtraction (same bits), which lets three
// instead of two workgroups share a CU at r = 16.  More resident waves do NOT help this kernel (config B, NR2D1 launches,
// profiles/r3k_nr2d1_ab_arrays_waves.txt): 4 arrays, 2 x 4 waves per CU 9.01 ms; 3 arrays, 3 x 4 waves 9.60; 3 arrays, 2 x 6
// waves 11.4; 4 x 2 waves 9.13; 2 x 3 waves 10.05 -- the three tables (3.2 GB) already overrun the caches with eight POIs in
// flight per CU.  Default 4.
#ifndef OC_NR_ARRAYS
#define OC_NR_ARRAYS 4
#endif
constexpr int kNrArrays = OC_NR_ARRAYS;
#ifndef OC_NR_MINBLOCKS
#define OC_NR_MINBLOCKS (OC_NR_ARRAYS == 3 ? 3 : 2)
#endif

// passes whose 12 gathers per lane (192 bytes) are issued together.  Measured on config B (FFTCC2D + NR2D1): 9.7 ms with 1,
// 2 or 3 passes per batch at 109 / 174 / 228 VGPRs -- the engine moves three table entries per sample and iteration and
// is bound by that gather throughput (2.6 x ICGN2D1's time for 3 x its gather bytes), not by their latency.
#ifndef OC_NR_BATCH
#define OC_NR_BATCH 1
#endif
constexpr int kNrBatch = OC_NR_BATCH;
// Lockstep passes (round 3, as OC_SWEEP_BARRIER in icgn2d.hip): a workgroup barrier every OC_NR_LOCKSTEP passes would keep the
// four waves of a workgroup -- neighbouring POIs, largely the same lines of the three tables -- close in time.  It does not
// pay here: with two 4-wave workgroups per CU (2 waves per SIMD) a waiting wave leaves its SIMD idle.  Config B, NR2D1
// launches (tools/ab_icgn2d.sh with AB_SRC=nr2d, profiles/r3j_nr2d1_ab_lockstep.txt): off 9.04 ms, every pass 9.55,
// every 2 passes 9.53, every 4 passes 9.32; bit-identical.  Off.
#ifndef OC_NR_LOCKSTEP
#define OC_NR_LOCKSTEP 0
#endif
constexpr int kNrLockstep = OC_NR_LOCKSTEP;

__global__ __launch_bounds__(64 * kNrWaves, OC_NR_MINBLOCKS) void nr2d1_kernel(Nr2dParams P, float* __restrict__ pois, Nr2dLaunch L) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int NT = L.nt;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned long long grp = blockIdx.x;
    if (L.xcd_chunk > 0) grp = (unsigned long long)(blockIdx.x & 7u) * L.xcd_chunk + (blockIdx.x >> 3);
    const unsigned long long slot = grp * kNrWaves + wave;
    if (slot >= L.count) return;
    // the k-th wave of the launch solves POI perm[k] (the locality schedule of icgn2d.hip) or simply POI k
    const unsigned long long idx = P.perm ? (unsigned long long)__builtin_amdgcn_readfirstlane((int)P.perm[slot]) : slot;
    // (with three arrays the reference values pass through the target array while their mean and norm are formed)
    float* __restrict__ l_rs = lds + (size_t)wave * kNrArrays * NT * kWave + lane;
    float* __restrict__ l_ts = kNrArrays == 4 ? l_rs + NT * kWave : l_rs;
    float* __restrict__ l_gx = l_ts + NT * kWave;
    float* __restrict__ l_gy = l_gx + NT * kWave;

    float* poi = pois + idx * (unsigned long long)L.stride_f;
    const float rec = lane < poi2d::FLOATS ? poi[lane] : 0.f;
    const float px = wave_bcast(rec, poi2d::X), py = wave_bcast(rec, poi2d::Y);
    const float u_in = wave_bcast(rec, poi2d::U), ux_in = wave_bcast(rec, poi2d::UX), uy_in = wave_bcast(rec, poi2d::UY);
    const float v_in = wave_bcast(rec, poi2d::V), vx_in = wave_bcast(rec, poi2d::VX), vy_in = wave_bcast(rec, poi2d::VY);
    const float zncc_in = wave_bcast(rec, poi2d::ZNCC);
    const float conv_in = wave_bcast(rec, poi2d::CONV), iter_in = wave_bcast(rec, poi2d::ITER);
    const float u0_in = wave_bcast(rec, poi2d::U0), v0_in = wave_bcast(rec, poi2d::V0);
    const int rx = P.rx, ry = P.ry, height = P.height, width = P.width;

    // guard, src/oc_nr.cpp:165-171: a rejected POI gets -1 (not -3), and the two checks that follow
    // the else-branch (:304-316) still look at it, with the fields it came in with
    if (py - ry < 0 || px - rx < 0 || py + ry > height - 1 || px + rx > width - 1 || fabsf(u_in) >= width ||
        fabsf(v_in) >= height || zncc_in < 0 || isnan(u_in) || isnan(v_in)) {
        if (lane == 0) {
            float zncc = zncc_in < -1 ? zncc_in : -1.f;
            if (conv_in >= P.conv && iter_in >= P.stop) zncc = -4.f;
            if (isnan(zncc) || isnan(u_in) || isnan(v_in)) {
                poi[poi2d::U] = u0_in;
                poi[poi2d::V] = v0_in;
                zncc = -5.f;
            }
            poi[poi2d::ZNCC] = zncc;
        }
        return;
    }
    const int W = 2 * rx + 1, N = W * (2 * ry + 1);
    const float fN = (float)N;
    const int NF = N / kWave;  // passes in which every lane owns a sample; pass NF (if any) is partial
    const int q64 = kWave / W, r64 = kWave - q64 * W;
    const int r0 = lane / W;
    const int c0 = lane - r0 * W;
    const __amdgpu_buffer_rsrc_t r_ref = make_rsrc(P.ref);
    const LutPlanesS r_lut(P.lut, height, width), r_lgx(P.lut_gx, height, width), r_lgy(P.lut_gy, height, width);
    const unsigned w4 = (unsigned)width * 4u;
    auto soff = [&](const SampleWalk& w) { return __umul24((unsigned)w.r, w4) + ((unsigned)w.c << 2); };

    // ---- reference subset, zero-mean + norm (src/oc_nr.cpp:176-179, src/oc_subset.cpp:39-53)
    float ref_norm, ref_mean;
    const unsigned roff = (unsigned)__builtin_amdgcn_readfirstlane(((int)(py - ry) * width + (int)(px - rx)) * 4);
    {
        float acc = 0.f;
        SampleWalk w(lane, r0, c0, W, q64, r64);
#pragma unroll 3
        for (int t = 0; t < NF; t++, w.next()) {
            const float v = buf_f32(r_ref, soff(w), roff);
            acc = acc + v;
            l_rs[t * kWave] = v;
        }
        if (NF < NT) {
            const bool valid = w.s < N;
            const float v = valid ? buf_f32(r_ref, soff(w), roff) : 0.f;
            acc = valid ? acc + v : acc;
            l_rs[NF * kWave] = v;
        }
        const float mean = wave_allreduce_sum(acc) / fN;
        ref_mean = mean;
        acc = 0.f;
#pragma unroll 3
        for (int t = 0; t < NF; t++) {
            const float d = l_rs[t * kWave] - mean;
            l_rs[t * kWave] = d;
            acc = acc + d * d;
        }
        if (NF < NT) {
            const float d = l_rs[NF * kWave] - mean;
            l_rs[NF * kWave] = d;
            acc = (NF * kWave + lane) < N ? acc + d * d : acc;
        }
        ref_norm = uni(sqrtf(wave_allreduce_sum(acc)));
    }

    // ---- Newton-Raphson loop (src/oc_nr.cpp:185-296)
    float cur[6] = {u_in, ux_in, uy_in, v_in, vx_in, vy_in};  // p_current: u ux uy v vx vy (wave-uniform)
    int iter = 0;
    float dp_norm = 0.f, znssd = 0.f;
#pragma nounroll
    do {
        iter++;
        float Wm[9];
        set_warp_2d1(Wm, cur[0], cur[1], cur[2], cur[3], cur[4], cur[5]);
        // (1) warped target subset + its gradients (:195-209) and the Hessian built from them (:213-238).
        // H(i,j) and H(j,i) receive the same products in the same order, so the lower triangle suffices;
        // the 21 sums run as packed-fp32 pairs like in icgn2d.hip: A = (sd1, sd2) = g_x*(x, y), B = g_y*(x, y)
        float acc = 0.f;
        f2 hAA = mk2(0.f, 0.f), hBB = hAA, hAB = hAA, hAs = hAA, hxA = hAA, hyA = hAA, hxB = hAA, hyB = hAA;
        float h00 = 0.f, h33 = 0.f, h30 = 0.f, h21 = 0.f, h54 = 0.f;
        {
            SampleWalk w(lane, r0, c0, W, q64, r64);
            struct NrFetch {
                LutFetch ft, fgx, fgy;
                f2 xy;
            };
            // the gathers of kNrBatch passes (12 x 16 bytes per lane and pass) are in flight before the first is used
            auto fetch = [&](int t, bool valid) {
                if (kNrLockstep > 0 && t % kNrLockstep == 0) __builtin_amdgcn_s_barrier();
                NrFetch q;
                const float xl = (float)(w.c - rx), yl = (float)(w.r - ry);
                // Deformation2D1::warp, src/oc_deformation.cpp:94-105
                const float wx = (Wm[0] * xl + Wm[1] * yl) + Wm[2] * 1.f;
                const float wy = (Wm[3] * xl + Wm[4] * yl) + Wm[5] * 1.f;
                const float x = valid ? px + wx : 1.f, y = valid ? py + wy : 1.f;
                // one range test and one entry offset serve the three tables
                // (BicubicBspline::compute, src/oc_cubic_bspline.cpp:134-181; rule explained at lut_fetch)
                bool out;
                const unsigned e = lut_locate<true>(q.ft, height, width, x, y, out);
                q.fgx.dx = q.fgy.dx = q.ft.dx;
                q.fgx.dy = q.fgy.dy = q.ft.dy;
                r_lut.load(q.ft, e);
                r_lgx.load(q.fgx, e);
                r_lgy.load(q.fgy, e);
                q.xy = mk2(xl, yl);
                w.next();
                return q;
            };
            auto sample = [&](int t, bool valid, const NrFetch& q) {
                const float tv = lut_eval(q.ft), g_x = lut_eval(q.fgx), g_y = lut_eval(q.fgy);
                l_ts[t * kWave] = tv;
                l_gx[t * kWave] = g_x;
                l_gy[t * kWave] = g_y;
                const f2 xy = q.xy;
                const f2 A = g_x * xy, B = g_y * xy;
                const f2 nAA = hAA + A * A, nBB = hBB + B * B, nAB = hAB + A * B, nAs = hAs + A * B.yx;
                const f2 nxA = hxA + g_x * A, nyA = hyA + g_y * A, nxB = hxB + g_x * B, nyB = hyB + g_y * B;
                const float n00 = h00 + g_x * g_x, n33 = h33 + g_y * g_y, n30 = h30 + g_y * g_x;
                const float n21 = h21 + A.y * A.x, n54 = h54 + B.y * B.x;
                const float nacc = acc + tv;
                if (valid) {
                    hAA = nAA; hBB = nBB; hAB = nAB; hAs = nAs; hxA = nxA; hyA = nyA; hxB = nxB; hyB = nyB;
                    h00 = n00; h33 = n33; h30 = n30; h21 = n21; h54 = n54;
                    acc = nacc;
                }
            };
            passes_batched<kNrBatch>(NF, NT, (NF * kWave + lane) < N, fetch, sample);
        }
        // (2) zeroMeanNorm of the target subset (:210)
        const float tmean = wave_allreduce_sum(acc) / fN;
        acc = 0.f;
#pragma unroll 6
        for (int t = 0; t < NF; t++) {
            const float d = l_ts[t * kWave] - tmean;
            acc = acc + d * d;
        }
        if (NF < NT) {
            const float d = l_ts[NF * kWave] - tmean;
            acc = (NF * kWave + lane) < N ? acc + d * d : acc;
        }
        const float tar_norm = uni(sqrtf(wave_allreduce_sum(acc)));
        // (3) Hessian: lane j < 6 assembles column j; inverse (:241)
        float hinv_col[6];
        {
            float h[21] = {h00,   hxA.x, hAA.x, hxA.y, h21,   hAA.y, h30,   hyA.x, hyA.y, h33,  hxB.x,
                           hAB.x, hAs.y, hyB.x, hBB.x, hxB.y, hAs.x, hAB.y, hyB.y, h54,   hBB.y};
            wave_allreduce_sum_multi<21>(h, lane);  // one transposing butterfly for the 21 sums (oc_device.h)
            float col[6];
#pragma unroll
            for (int i = 0; i < 6; i++) col[i] = 0.f;
            int k = 0;
#pragma unroll
            for (int i = 0; i < 6; i++)
#pragma unroll
                for (int j = 0; j <= i; j++) {
                    const float v = h[k++];
                    if (lane == j) col[i] = v;  // H(i,j)
                    if (lane == i) col[j] = v;  // H(j,i)
                }
            lu_inverse_lanes<6>(col, hinv_col, lane);
        }
        // (4) error image, ZNSSD, numerator (:244-262): e = ref * (|tar| / |ref|) - tar
        const float factor = tar_norm / ref_norm;
        f2 nA = mk2(0.f, 0.f), nB = nA;
        float n0 = 0.f, n3 = 0.f, ssd = 0.f;
        {
            SampleWalk w(lane, r0, c0, W, q64, r64);
            auto sample = [&](int t, bool valid) {
                const float g_x = l_gx[t * kWave], g_y = l_gy[t * kWave];
                const float tz = l_ts[t * kWave] - tmean;  // same bits as in the norm pass
                float rsv;
                if constexpr (kNrArrays == 4) rsv = l_rs[t * kWave];
                else rsv = (valid ? buf_f32(r_ref, soff(w), roff) : 0.f) - ref_mean;  // the subtraction of the set-up pass
                const float e = rsv * factor - tz;
                const f2 xy = mk2((float)(w.c - rx), (float)(w.r - ry));
                const f2 A = g_x * xy, B = g_y * xy;
                const f2 mA = nA + A * e, mB = nB + B * e;
                const float m0 = n0 + g_x * e, m3 = n3 + g_y * e, ms = ssd + e * e;
                if (valid) {
                    nA = mA; nB = mB; n0 = m0; n3 = m3; ssd = ms;
                }
            };
#pragma unroll 3
            for (int t = 0; t < NF; t++, w.next()) sample(t, true);
            if (NF < NT) sample(NF, w.s < N);
        }
        float num[7] = {n0, nA.x, nA.y, n3, nB.x, nB.y, ssd};
        wave_allreduce_sum_multi<7>(num, lane);
        znssd = uni(num[6]) / (tar_norm * tar_norm);
        float numj = 0.f;
#pragma unroll
        for (int j = 0; j < 6; j++) numj = lane == j ? num[j] : numj;
        // (5) dp = H^-1 * numerator (:265-272), p += dp (:277-279), convergence norm (:285-293)
        float dp[6];
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const float prod = hinv_col[i] * numj;
            float v = 0.f;
#pragma unroll
            for (int j = 0; j < 6; j++) v += wave_bcast(prod, j);
            dp[i] = v;
        }
#pragma unroll
        for (int i = 0; i < 6; i++) cur[i] = uni(cur[i] + dp[i]);
        const int rx2 = rx * rx, ry2 = ry * ry;
        float d = 0.f;
        d += dp[0] * dp[0];
        d += dp[1] * dp[1] * rx2;
        d += dp[2] * dp[2] * ry2;
        d += dp[3] * dp[3];
        d += dp[4] * dp[4] * rx2;
        d += dp[5] * dp[5] * ry2;
        dp_norm = uni(sqrtf(d));
    } while (iter < P.stop && dp_norm >= P.conv);

    // ---- outputs (src/oc_nr.cpp:296-316); subset_radius is not written by NR2D1
    if (lane == 0) {
        float zncc = 0.5f * (2 - znssd);
        const float fiter = (float)iter;
        if (dp_norm >= P.conv && fiter >= P.stop) zncc = -4.f;
        float out_u = cur[0], out_v = cur[3];
        if (isnan(zncc) || isnan(out_u) || isnan(out_v)) {
            out_u = u_in;
            out_v = v_in;
            zncc = -5.f;
        }
        poi[poi2d::U] = out_u;
        poi[poi2d::UX] = cur[1];
        poi[poi2d::UY] = cur[2];
        poi[poi2d::V] = out_v;
        poi[poi2d::VX] = cur[4];
        poi[poi2d::VY] = cur[5];
        poi[poi2d::U0] = u_in;
        poi[poi2d::V0] = v_in;
        poi[poi2d::ZNCC] = zncc;
        poi[poi2d::ITER] = fiter;
        poi[poi2d::CONV] = dp_norm;
    }
}

constexpr int kNrLdsBudget = 160 * 1024;

int nr2d1_max_samples() { return kNrLdsBudget / (kNrArrays * kNrWaves * (int)sizeof(float) * kWave) * kWave; }

hipError_t launch_nr2d1(const Nr2dParams& p, float* pois, int stride_f, size_t count, hipStream_t stream) {
    if (count == 0) return hipSuccess;
    if ((unsigned long long)p.height * p.width * 64ull > (1ull << 32)) return hipErrorInvalidValue;
    const int N = (2 * p.rx + 1) * (2 * p.ry + 1);
    const int nt = (N + 63) / 64;
    const size_t lds = (size_t)kNrArrays * nt * kWave * sizeof(float) * kNrWaves;
    if (lds > (size_t)kNrLdsBudget) return hipErrorInvalidValue;
    static std::atomic<unsigned long long> attr_devices{0};
    int dev = 0;
    hipError_t derr = hipGetDevice(&dev);
    if (derr != hipSuccess) return derr;
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(attr_devices.load(std::memory_order_acquire) & bit)) {
        hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(nr2d1_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, kNrLdsBudget);
        if (err != hipSuccess) return err;
        attr_devices.fetch_or(bit, std::memory_order_release);
    }
    const size_t groups = (count + kNrWaves - 1) / kNrWaves;
    Nr2dLaunch L;
    L.stride_f = stride_f;
    L.nt = nt;
    L.count = count;
    L.xcd_chunk = (int)((groups + 7) / 8);
    const size_t grid = (size_t)L.xcd_chunk * 8;
    (void)hipGetLastError();  // drop stale errors of earlier, unrelated calls
    hipLaunchKernelGGL(nr2d1_kernel, dim3((unsigned)grid), dim3(64 * kNrWaves), lds, stream, p, pois, L);
    return hipGetLastError();
}

}  // namespace ochip
