// fftcc2d_fusedn_impl.h -- the single-kernel FFTCC2D for an NR x NC transform (shared by fftcc2d_fusedn.hip: square windows,
// and fftcc2d_fusedr.hip: rectangular ones).  See fftcc2d_fusedn.hip for the plan.
#pragma once

#include "dic2d_device.h"
#include "fft_device.h"
#include "oc_kernels.h"

namespace ochip {
namespace fusedn {

using namespace fftdev;

constexpr int kFusedNWaves = 1;  // waves per workgroup

// Round 3: the mapping of fftcc2d_fused32x2_kernel (fftcc2d_fused.hip) for every supported NR x NC.
//   * a lane owns a whole line or column of the transform; windows whose longer side fits 32 lanes share a wave in pairs
//     (half-wave q serves POI q);
//   * the lane gathers COLUMN l of the transform's array straight into its registers (the lanes of a row read
//     consecutive pixels), so the first axis transform needs no LDS;
//   * between passes the array is transposed through an NR x (NC + 1) tile of FLOATS, real parts first, imaginary parts
//     second (half the LDS per POI), in place in the register array (all indices are compile-time constants);
//   * the spectrum product is formed without its two factors 1/2 (exact scaling, folded into the final division).
// The transform is NR x NC: NR = 2 * radius_x lines of NC = 2 * radius_y contiguous elements -- the shape FFTW is planned
// with (fftwf_plan_dft_r2c_2d(width, height), src/oc_fftcc.cpp:40-42) over the window buffer filled [row * width + col]
// (:204-221).  For a square window that is the window itself; for rx != ry it is the window's linear buffer re-cut into
// lines of 2 * radius_y -- the reference's own behaviour, reproduced here as in the rocFFT pipeline.
template <int NR, int NC>
__global__ __launch_bounds__(64 * kFusedNWaves) void fftcc2d_fusedn_kernel(Fftcc2dParams P, float* __restrict__ pois,
                                                                          int stride_f, unsigned long long count,
                                                                          int xcd_chunk) {
    constexpr int NP = NC + 1;  // tile pitch in floats (odd: lines and columns both conflict-free)
    constexpr int M = NR * NC;
    constexpr int NMAX = NR > NC ? NR : NC;
    constexpr int PPW = NMAX <= 32 ? 2 : 1;  // POIs per wave
    constexpr int LANES = kWave / PPW;       // lanes per POI
    __shared__ float lds[kFusedNWaves * PPW * NR * NP];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = PPW == 2 ? lane >> 5 : 0, l = lane & (LANES - 1);
    unsigned long long grp = blockIdx.x;
    if (xcd_chunk > 0) grp = (unsigned long long)(blockIdx.x & 7u) * xcd_chunk + (blockIdx.x >> 3);
    const unsigned long long idx = (grp * kFusedNWaves + wave) * PPW + q;
    if (idx >= count) return;
    float* __restrict__ ft = lds + (wave * PPW + q) * (NR * NP);
    float* poi = pois + idx * (unsigned long long)stride_f;
    const float px = poi[poi2d::X], py = poi[poi2d::Y];
    const float gu = poi[poi2d::U], gv = poi[poi2d::V];
    constexpr int rx = NR / 2, ry = NC / 2;
    const int width = P.width, height = P.height;

    // bounds guard: the reference returns silently and leaves the POI untouched (src/oc_fftcc.cpp:190-196)
    if ((int)px < rx || (int)px >= width - rx || (int)py < ry || (int)py >= height - ry || (int)(px + gu) < rx ||
        (int)(px + gu) >= width - rx || (int)(py + gv) < ry || (int)(py + gv) >= height - ry)
        return;

    // lines (length NC) are owned by lanes < NR, columns (length NR) by lanes < NC; idle lanes shadow line / column 0,
    // contribute nothing to the sums and never write
    const bool act_l = l < NR, act_c = l < NC;
    const int line = act_l ? l : 0, col = act_c ? l : 0;
    auto lanes_sum = [](float x) {
#pragma unroll
        for (int off = 1; off < LANES; off <<= 1) x += __shfl_xor(x, off, kWave);
        return x;
    };

    // ---- window fill, means, zero-mean, sums of squares (src/oc_fftcc.cpp:198-231): element (a, col) of the transform's
    // array is sample s = a * NC + col of the window buffer, i.e. window row s / NR, column s % NR
    c2 v[NMAX];
    float rn, tn;
    {
        const __amdgpu_buffer_rsrc_t r_ref = make_rsrc(P.ref), r_tar = make_rsrc(P.tar);
        static_for<0, NR>([&](auto ac) {
            constexpr int a = decltype(ac)::value;
            const int s = a * NC + col;
            const int r = NR == NC ? a : s / NR, c = NR == NC ? col : s - r * NR;
            const float rxp = px + c - rx, ryp = py + r - ry;
            const float txp = rxp + gu, typ = ryp + gv;
            v[a] = mkc(buf_f32(r_ref, (__umul24((unsigned)(int)ryp, (unsigned)width) + (unsigned)(int)rxp) << 2, 0),
                       buf_f32(r_tar, (__umul24((unsigned)(int)typ, (unsigned)width) + (unsigned)(int)txp) << 2, 0));
        });
        float rsum = 0.f, tsum = 0.f;
#pragma unroll
        for (int a = 0; a < NR; a++) {
            rsum += v[a].x;
            tsum += v[a].y;
        }
        const c2 mean = mkc(lanes_sum(act_c ? rsum : 0.f) / M, lanes_sum(act_c ? tsum : 0.f) / M);
        rn = 0.f;
        tn = 0.f;
#pragma unroll
        for (int a = 0; a < NR; a++) {
            v[a] = v[a] - mean;
            rn += v[a].x * v[a].x;
            tn += v[a].y * v[a].y;
        }
        rn = lanes_sum(act_c ? rn : 0.f);
        tn = lanes_sum(act_c ? tn : 0.f);
        asm volatile("" : "+v"(rn), "+v"(tn));  // formed here, used at the very end
    }
    // ---- forward columns, straight from the registers: Z1(ka, col) in v[fft_pos(NR, ka)]
    fft_mixed_at<false, NR, NR, 0, NMAX>(v);
    // ---- transpose (columns -> lines), in place: real parts, then imaginary parts
#define OC_FUSEDN_PART(PART)                                                           \
    static_for<0, NR>([&](auto kc) {                                                   \
        constexpr int k = decltype(kc)::value, p = fft_pos(NR, k);                     \
        if (act_c) ft[k * NP + col] = v[p].PART;                                       \
    });                                                                                \
    __builtin_amdgcn_wave_barrier();                                                   \
    static_for<0, NC>([&](auto cc) {                                                   \
        constexpr int c = decltype(cc)::value;                                         \
        v[c].PART = ft[line * NP + c];                                                 \
    });                                                                                \
    __builtin_amdgcn_wave_barrier();
    OC_FUSEDN_PART(x)
    OC_FUSEDN_PART(y)
#undef OC_FUSEDN_PART
    // ---- forward lines: Z(line, kc) in v[fft_pos(NC, kc)]
    fft_mixed_at<false, NC, NC, 0, NMAX>(v);
    // ---- spectra of the two real windows and their product conj(R) * T (src/oc_fftcc.cpp:236-241): Z(-k) is line
    // (-line) read backwards; 2R = z + conj(zm), 2T = (z - conj(zm)) / i, the factors 1/2 left to the final division
    c2 t[NMAX];
    {
        const int mline = ((NR - line) % NR) * NP;
        float zmx[NC];
        static_for<0, NC>([&](auto kc) {
            constexpr int k = decltype(kc)::value, p = fft_pos(NC, k);
            if (act_l) ft[line * NP + k] = v[p].x;
        });
        __builtin_amdgcn_wave_barrier();
        static_for<0, NC>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            zmx[k] = ft[mline + (NC - k) % NC];
        });
        __builtin_amdgcn_wave_barrier();
        static_for<0, NC>([&](auto kc) {
            constexpr int k = decltype(kc)::value, p = fft_pos(NC, k);
            if (act_l) ft[line * NP + k] = v[p].y;
        });
        __builtin_amdgcn_wave_barrier();
        static_for<0, NC>([&](auto kc) {
            constexpr int k = decltype(kc)::value, p = fft_pos(NC, k);
            const c2 z = v[p];
            const float zmy = ft[mline + (NC - k) % NC];
            const float rr = z.x + zmx[k], ri = z.y - zmy;
            const float tr = z.y + zmy, ti = zmx[k] - z.x;
            t[k] = mkc((rr * tr) + (ri * ti), (rr * ti) - (ri * tr));
        });
        __builtin_amdgcn_wave_barrier();
    }
    // ---- inverse lines (unnormalised, like FFTW's c2r): g(line, c) in t[fft_pos(NC, c)]
    fft_mixed_at<true, NC, NC, 0, NMAX>(t);
    // ---- transpose (lines -> columns)
#define OC_FUSEDN_PART(PART)                                                           \
    static_for<0, NC>([&](auto cc) {                                                   \
        constexpr int c = decltype(cc)::value, p = fft_pos(NC, c);                     \
        if (act_l) ft[line * NP + c] = t[p].PART;                                      \
    });                                                                                \
    __builtin_amdgcn_wave_barrier();                                                   \
    static_for<0, NR>([&](auto ac) {                                                   \
        constexpr int a = decltype(ac)::value;                                         \
        v[a].PART = ft[a * NP + col];                                                  \
    });                                                                                \
    __builtin_amdgcn_wave_barrier();
    OC_FUSEDN_PART(x)
    OC_FUSEDN_PART(y)
#undef OC_FUSEDN_PART
    // ---- inverse columns: the correlation surface (a, col) in v[fft_pos(NR, a)]
    fft_mixed_at<true, NR, NR, 0, NMAX>(v);

    // ---- arg-max with "strict >, scanning from index 0" (src/oc_fftcc.cpp:246-255): the lane's NR surface values sit
    // at linear indices a * NC + col, ascending in a; then the POI's lanes, the lower index winning a tie
    float best = -2.f;
    int bidx = 0x7fffffff;
    if (act_c) {
        static_for<0, NR>([&](auto ac) {
            constexpr int a = decltype(ac)::value, p = fft_pos(NR, a);
            const float val = v[p].x;
            if (val > best) {
                best = val;
                bidx = a * NC + col;
            }
        });
        if (bidx == 0x7fffffff) bidx = col;  // nothing above -2 (NaN surface): the reference keeps index 0 semantics
    }
#pragma unroll
    for (int off = 1; off < LANES; off <<= 1) {
        const float ov = __shfl_xor(best, off, kWave);
        const int oi = __shfl_xor(bidx, off, kWave);
        if (ov > best || (ov == best && oi < bidx)) {
            best = ov;
            bidx = oi;
        }
    }
    if (l == 0) {
        if (bidx == 0x7fffffff) bidx = 0;
        // the peak is decoded with the WINDOW's width (src/oc_fftcc.cpp:257-266), whatever shape the transform had
        int du = bidx % NR, dv = bidx / NR;
        if (du > rx) du -= NR;
        if (dv > ry) dv -= NC;
        poi[poi2d::U] = (float)du + gu;
        poi[poi2d::V] = (float)dv + gv;
        poi[poi2d::U0] = gu;
        poi[poi2d::V0] = gv;
        poi[poi2d::ZNCC] = (0.25f * best) / (sqrtf(rn * tn) * M);
    }
}

template <int NR, int NC>
hipError_t launch_n(const Fftcc2dParams& p, float* pois, int stride_f, size_t count, bool xcd, hipStream_t stream) {
    constexpr int per_group = kFusedNWaves * ((NR > NC ? NR : NC) <= 32 ? 2 : 1);  // POIs per workgroup
    const size_t groups = (count + per_group - 1) / per_group;
    const int chunk = xcd ? (int)((groups + 7) / 8) : 0;
    const size_t grid = xcd ? (size_t)chunk * 8 : groups;
    (void)hipGetLastError();  // drop stale errors of earlier, unrelated calls
    hipLaunchKernelGGL((fftcc2d_fusedn_kernel<NR, NC>), dim3((unsigned)grid), dim3(64 * kFusedNWaves), 0, stream, p, pois, stride_f,
                       (unsigned long long)count, chunk);
    return hipGetLastError();
}

}  // namespace fusedn
}  // namespace ochip
