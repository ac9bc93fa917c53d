// fftcc2d_fusedn.hip -- single-kernel FFTCC2D for square windows whose side N = 2 * radius is not 32
// (N = 20, 24, 30, 36, 40, 48: radii 10, 12, 15, 18, 20, 24; N = 32 has its own kernel in fftcc2d_fused.hip).
//
// Same plan as the 32 x 32 kernel: one wavefront per POI keeps the whole FFTCC2D::compute(POI2D*)
// (src/oc_fftcc.cpp:177-275) on chip -- gather with the arithmetic of fftcc2d_gather_kernel (bit-identical means and
// norms), z = ref + i*tar, ONE complex N x N FFT, R(k) = (Z(k) + conj Z(-k))/2, T(k) = (Z(k) - conj Z(-k))/(2i),
// C = conj(R) T, inverse FFT (unnormalised like FFTW's c2r), arg-max with the first-max rule.
// Here a lane owns a whole line: lane l < N transforms row l (then column l) with an N-point mixed-radix FFT held in
// registers (fft_device.h: factors 2, 3, 4, 5, twiddles from a generated table); the N x (N+1) complex tile in LDS
// (odd pitch: rows and columns both conflict-free) carries the data between the row and the column pass.
// The rocFFT pipeline this replaces spends more time on these windows than the ICGN refinement that follows it
// (config C, r = 20: 4.9 ms of FFTCC against 4.2 ms of ICGN2D2 for 99 856 POIs).
#include "dic2d_device.h"
#include "fft_device.h"
#include "oc_kernels.h"

namespace ochip {

namespace {

using namespace fftdev;

constexpr int kFusedNWaves = 4;  // POIs (waves) per workgroup

template <int N>
__global__ __launch_bounds__(64 * kFusedNWaves) void fftcc2d_fusedn_kernel(Fftcc2dParams P, float* __restrict__ pois,
                                                                          int stride_f, unsigned long long count,
                                                                          int xcd_chunk) {
    constexpr int NP = N + 1;  // LDS row pitch in complex elements (odd)
    constexpr int M = N * N;
    constexpr int K = (M + kWave - 1) / kWave;  // samples per lane in the gather
    __shared__ c2 lds[kFusedNWaves * N * NP];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned long long grp = blockIdx.x;
    if (xcd_chunk > 0) grp = (unsigned long long)(blockIdx.x & 7u) * xcd_chunk + (blockIdx.x >> 3);
    const unsigned long long idx = grp * kFusedNWaves + wave;
    if (idx >= count) return;
    c2* buf = lds + wave * (N * NP);
    float* poi = pois + idx * (unsigned long long)stride_f;
    const float px = poi[poi2d::X], py = poi[poi2d::Y];
    const float gu = poi[poi2d::U], gv = poi[poi2d::V];
    constexpr int rx = N / 2, ry = N / 2;
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
                const int r = s / N, c = s - r * N;
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
                const int r = s / N, c = s - r * N;
                buf[r * NP + c] = mkc(x, y);
            }
        }
        rn = wave_allreduce_sum(rn);
        tn = wave_allreduce_sum(tn);
    }
    __builtin_amdgcn_wave_barrier();

    const bool active = lane < N;
    const int line = active ? lane : 0;  // idle lanes shadow line 0 and never write
    c2 v[N];
    // ---- forward rows: lane y -> Z1[y][k]
#pragma unroll
    for (int c = 0; c < N; c++) v[c] = buf[line * NP + c];
    fft_mixed<false, N>(v);
    __builtin_amdgcn_wave_barrier();
    if (active) {
        static_for<0, N>([&](auto kc) {
            constexpr int k = decltype(kc)::value, p = fft_pos(N, k);
            buf[line * NP + k] = v[p];
        });
    }
    __builtin_amdgcn_wave_barrier();
    // ---- forward columns: lane x -> Z[k][x]
#pragma unroll
    for (int r = 0; r < N; r++) v[r] = buf[r * NP + line];
    fft_mixed<false, N>(v);
    __builtin_amdgcn_wave_barrier();
    if (active) {
        static_for<0, N>([&](auto kc) {
            constexpr int k = decltype(kc)::value, p = fft_pos(N, k);
            buf[k * NP + line] = v[p];
        });
    }
    __builtin_amdgcn_wave_barrier();
    // ---- spectra of the two real windows and their product conj(R) * T (src/oc_fftcc.cpp:236-241), column `line`
    c2 t[N];
    {
        const int mx = (N - line) % N;
        static_for<0, N>([&](auto kc) {
            constexpr int k = decltype(kc)::value, p = fft_pos(N, k);
            const c2 zm = buf[((N - k) % N) * NP + mx];
            const c2 z = v[p];
            const float rr = 0.5f * (z.x + zm.x), ri = 0.5f * (z.y - zm.y);
            const float tr = 0.5f * (z.y + zm.y), ti = -0.5f * (z.x - zm.x);
            t[k] = mkc((rr * tr) + (ri * ti), (rr * ti) - (ri * tr));
        });
    }
    __builtin_amdgcn_wave_barrier();
    // ---- inverse columns, inverse rows (unnormalised)
    fft_mixed<true, N>(t);
    if (active) {
        static_for<0, N>([&](auto rc) {
            constexpr int r = decltype(rc)::value, p = fft_pos(N, r);
            buf[r * NP + line] = t[p];
        });
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < N; k++) v[k] = buf[line * NP + k];
    fft_mixed<true, N>(v);

    // ---- arg-max with "strict >, scanning from index 0" (src/oc_fftcc.cpp:246-255): the lane's N surface values sit
    // at linear indices line*N + x, ascending in x
    float best = -2.f;
    int bidx = 0x7fffffff;
    if (active) {
        static_for<0, N>([&](auto xc) {
            constexpr int x = decltype(xc)::value, p = fft_pos(N, x);
            const float val = v[p].x;
            if (val > best) {
                best = val;
                bidx = line * N + x;
            }
        });
        if (bidx == 0x7fffffff) bidx = line * N;  // nothing above -2 (NaN surface): the reference keeps index 0 semantics
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
        int du = bidx % N, dv = bidx / N;
        if (du > rx) du -= N;
        if (dv > ry) dv -= N;
        poi[poi2d::U] = (float)du + gu;
        poi[poi2d::V] = (float)dv + gv;
        poi[poi2d::U0] = gu;
        poi[poi2d::V0] = gv;
        poi[poi2d::ZNCC] = best / (sqrtf(rn * tn) * M);
    }
}

template <int N>
hipError_t launch_n(const Fftcc2dParams& p, float* pois, int stride_f, size_t count, bool xcd, hipStream_t stream) {
    const size_t groups = (count + kFusedNWaves - 1) / kFusedNWaves;
    const int chunk = xcd ? (int)((groups + 7) / 8) : 0;
    const size_t grid = xcd ? (size_t)chunk * 8 : groups;
    (void)hipGetLastError();  // drop stale errors of earlier, unrelated calls
    hipLaunchKernelGGL(fftcc2d_fusedn_kernel<N>, dim3((unsigned)grid), dim3(64 * kFusedNWaves), 0, stream, p, pois, stride_f,
                       (unsigned long long)count, chunk);
    return hipGetLastError();
}

}  // namespace

bool fftcc2d_fusedn_supported(int rx, int ry) {
    return rx == ry && (rx == 8 || rx == 9 || rx == 10 || rx == 12 || rx == 15 || rx == 18 || rx == 20 || rx == 24 || rx == 25 || rx == 30 ||
                        rx == 32);
}

hipError_t launch_fftcc2d_fusedn(const Fftcc2dParams& p, float* pois, int stride_f, size_t count, bool xcd,
                                 hipStream_t stream) {
    if (count == 0) return hipSuccess;
    if (!fftcc2d_fusedn_supported(p.rx, p.ry)) return hipErrorInvalidValue;
    switch (2 * p.rx) {
        case 16: return launch_n<16>(p, pois, stride_f, count, xcd, stream);
        case 18: return launch_n<18>(p, pois, stride_f, count, xcd, stream);
        case 50: return launch_n<50>(p, pois, stride_f, count, xcd, stream);
        case 60: return launch_n<60>(p, pois, stride_f, count, xcd, stream);
        case 64: return launch_n<64>(p, pois, stride_f, count, xcd, stream);
        case 20: return launch_n<20>(p, pois, stride_f, count, xcd, stream);
        case 24: return launch_n<24>(p, pois, stride_f, count, xcd, stream);
        case 36: return launch_n<36>(p, pois, stride_f, count, xcd, stream);
        case 30: return launch_n<30>(p, pois, stride_f, count, xcd, stream);
        case 40: return launch_n<40>(p, pois, stride_f, count, xcd, stream);
        default: return launch_n<48>(p, pois, stride_f, count, xcd, stream);
    }
}

}  // namespace ochip
