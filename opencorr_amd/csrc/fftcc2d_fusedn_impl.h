// fftcc2d_fusedn_impl.h -- the single-kernel FFTCC2D for an NR x NC transform (shared by fftcc2d_fusedn.hip: square windows,
// and fftcc2d_fusedr.hip: rectangular ones).  See fftcc2d_fusedn.hip for the plan.
#pragma once

#include "dic2d_device.h"
#include "fft_device.h"
#include "oc_kernels.h"

namespace ochip {
namespace fusedn {

using namespace fftdev;

constexpr int kFusedNWaves = 4;  // POIs (waves) per workgroup

template <int NR, int NC>
__global__ __launch_bounds__(64 * kFusedNWaves) void fftcc2d_fusedn_kernel(Fftcc2dParams P, float* __restrict__ pois,
                                                                          int stride_f, unsigned long long count,
                                                                          int xcd_chunk) {
    // The transform is NR x NC: NR = 2 * radius_x lines of NC = 2 * radius_y contiguous elements -- the shape FFTW is planned
    // with (fftwf_plan_dft_r2c_2d(width, height), src/oc_fftcc.cpp:40-42) over the window buffer filled [row * width + col]
    // (:204-221).  For a square window that is the window itself; for rx != ry it is the window's linear buffer re-cut into
    // lines of 2 * radius_y -- the reference's own behaviour, reproduced here as in the rocFFT pipeline.
    constexpr int NP = NC + 1;  // LDS line pitch in complex elements
    constexpr int M = NR * NC;
    constexpr int NMAX = NR > NC ? NR : NC;
    constexpr int K = (M + kWave - 1) / kWave;  // samples per lane in the gather
    __shared__ c2 lds[kFusedNWaves * NR * NP];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned long long grp = blockIdx.x;
    if (xcd_chunk > 0) grp = (unsigned long long)(blockIdx.x & 7u) * xcd_chunk + (blockIdx.x >> 3);
    const unsigned long long idx = grp * kFusedNWaves + wave;
    if (idx >= count) return;
    c2* buf = lds + wave * (NR * NP);
    float* poi = pois + idx * (unsigned long long)stride_f;
    const float px = poi[poi2d::X], py = poi[poi2d::Y];
    const float gu = poi[poi2d::U], gv = poi[poi2d::V];
    constexpr int rx = NR / 2, ry = NC / 2;
    const int width = P.width, height = P.height;

    // bounds guard: the reference returns silently and leaves the POI untouched (src/oc_fftcc.cpp:190-196)
    if ((int)px < rx || (int)px >= width - rx || (int)py < ry || (int)py >= height - ry || (int)(px + gu) < rx ||
        (int)(px + gu) >= width - rx || (int)(py + gv) < ry || (int)(py + gv) >= height - ry)
        return;

    // ---- window fill, means, zero-mean, sums of squares (src/oc_fftcc.cpp:198-231); sample s = r*N + c is owned by
    // lane (s mod 64), exactly like fftcc2d_gather_kernel
    float rn, tn;
    {
        const __amdgpu_buffer_rsrc_t r_ref = make_rsrc(P.ref), r_tar = make_rsrc(P.tar);
        float a[K], b[K];
        float rsum = 0.f, tsum = 0.f;
#pragma unroll
        for (int k = 0; k < K; k++) {
            const int s = lane + kWave * k;
            a[k] = 0.f;
            b[k] = 0.f;
            if (s < M) {
                const int r = s / NR, c = s - r * NR;  // window row / column: the window is 2 * ry rows of 2 * rx pixels
                const float rxp = px + c - rx, ryp = py + r - ry;
                a[k] = buf_f32(r_ref, (__umul24((unsigned)(int)ryp, (unsigned)width) + (unsigned)(int)rxp) << 2, 0);
                const float txp = rxp + gu, typ = ryp + gv;
                b[k] = buf_f32(r_tar, (__umul24((unsigned)(int)typ, (unsigned)width) + (unsigned)(int)txp) << 2, 0);
            }
        }
#pragma unroll
        for (int k = 0; k < K; k++) {
            if (lane + kWave * k < M) {
                rsum += a[k];
                tsum += b[k];
            }
        }
        const float rmean = wave_allreduce_sum(rsum) / M;
        const float tmean = wave_allreduce_sum(tsum) / M;
        rn = 0.f;
        tn = 0.f;
#pragma unroll
        for (int k = 0; k < K; k++) {
            const int s = lane + kWave * k;
            if (s < M) {
                const float x = a[k] - rmean, y = b[k] - tmean;
                rn += x * x;
                tn += y * y;
                const int a = s / NC, b2 = s - a * NC;  // line / element of the transform's array
                buf[a * NP + b2] = mkc(x, y);
            }
        }
        rn = wave_allreduce_sum(rn);
        tn = wave_allreduce_sum(tn);
    }
    __builtin_amdgcn_wave_barrier();

    // lines (length NC) are owned by lanes < NR, columns (length NR) by lanes < NC; idle lanes shadow line / column 0 and
    // never write
    const bool act_l = lane < NR, act_c = lane < NC;
    const int line = act_l ? lane : 0, col = act_c ? lane : 0;
    // ---- forward lines: lane y -> Z1[y][k]
    {
        c2 v[NC];
#pragma unroll
        for (int c = 0; c < NC; c++) v[c] = buf[line * NP + c];
        fft_mixed<false, NC>(v);
        __builtin_amdgcn_wave_barrier();
        if (act_l) {
            static_for<0, NC>([&](auto kc) {
                constexpr int k = decltype(kc)::value, p = fft_pos(NC, k);
                buf[line * NP + k] = v[p];
            });
        }
    }
    __builtin_amdgcn_wave_barrier();
    // ---- forward columns: lane x -> Z[k][x]
    c2 v[NMAX];
#pragma unroll
    for (int r = 0; r < NR; r++) v[r] = buf[r * NP + col];
    {
        c2(&vr)[NR] = reinterpret_cast<c2(&)[NR]>(v);
        fft_mixed<false, NR>(vr);
    }
    __builtin_amdgcn_wave_barrier();
    if (act_c) {
        static_for<0, NR>([&](auto kc) {
            constexpr int k = decltype(kc)::value, p = fft_pos(NR, k);
            buf[k * NP + col] = v[p];
        });
    }
    __builtin_amdgcn_wave_barrier();
    // ---- spectra of the two real windows and their product conj(R) * T (src/oc_fftcc.cpp:236-241), column `col`
    c2 t[NR];
    {
        const int mx = (NC - col) % NC;
        static_for<0, NR>([&](auto kc) {
            constexpr int k = decltype(kc)::value, p = fft_pos(NR, k);
            const c2 zm = buf[((NR - k) % NR) * NP + mx];
            const c2 z = v[p];
            const float rr = 0.5f * (z.x + zm.x), ri = 0.5f * (z.y - zm.y);
            const float tr = 0.5f * (z.y + zm.y), ti = -0.5f * (z.x - zm.x);
            t[k] = mkc((rr * tr) + (ri * ti), (rr * ti) - (ri * tr));
        });
    }
    __builtin_amdgcn_wave_barrier();
    // ---- inverse columns, inverse lines (unnormalised)
    fft_mixed<true, NR>(t);
    if (act_c) {
        static_for<0, NR>([&](auto rc) {
            constexpr int r = decltype(rc)::value, p = fft_pos(NR, r);
            buf[r * NP + col] = t[p];
        });
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < NC; k++) v[k] = buf[line * NP + k];
    {
        c2(&vl)[NC] = reinterpret_cast<c2(&)[NC]>(v);
        fft_mixed<true, NC>(vl);
    }

    // ---- arg-max with "strict >, scanning from index 0" (src/oc_fftcc.cpp:246-255): the lane's NC surface values sit
    // at linear indices line*NC + x, ascending in x
    float best = -2.f;
    int bidx = 0x7fffffff;
    if (act_l) {
        static_for<0, NC>([&](auto xc) {
            constexpr int x = decltype(xc)::value, p = fft_pos(NC, x);
            const float val = v[p].x;
            if (val > best) {
                best = val;
                bidx = line * NC + x;
            }
        });
        if (bidx == 0x7fffffff) bidx = line * NC;  // nothing above -2 (NaN surface): the reference keeps index 0 semantics
    }
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const float ov = __shfl_xor(best, off, kWave);
        const int oi = __shfl_xor(bidx, off, kWave);
        if (ov > best || (ov == best && oi < bidx)) {
            best = ov;
            bidx = oi;
        }
    }
    if (lane == 0) {
        if (bidx == 0x7fffffff) bidx = 0;
        // the peak is decoded with the WINDOW's width (src/oc_fftcc.cpp:257-266), whatever shape the transform had
        int du = bidx % NR, dv = bidx / NR;
        if (du > rx) du -= NR;
        if (dv > ry) dv -= NC;
        poi[poi2d::U] = (float)du + gu;
        poi[poi2d::V] = (float)dv + gv;
        poi[poi2d::U0] = gu;
        poi[poi2d::V0] = gv;
        poi[poi2d::ZNCC] = best / (sqrtf(rn * tn) * M);
    }
}

template <int NR, int NC>
hipError_t launch_n(const Fftcc2dParams& p, float* pois, int stride_f, size_t count, bool xcd, hipStream_t stream) {
    const size_t groups = (count + kFusedNWaves - 1) / kFusedNWaves;
    const int chunk = xcd ? (int)((groups + 7) / 8) : 0;
    const size_t grid = xcd ? (size_t)chunk * 8 : groups;
    (void)hipGetLastError();  // drop stale errors of earlier, unrelated calls
    hipLaunchKernelGGL((fftcc2d_fusedn_kernel<NR, NC>), dim3((unsigned)grid), dim3(64 * kFusedNWaves), 0, stream, p, pois, stride_f,
                       (unsigned long long)count, chunk);
    return hipGetLastError();
}

}  // namespace fusedn
}  // namespace ochip
