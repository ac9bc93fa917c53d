// fftcc3d_fusedn.hip -- FFTCC3D in ONE kernel for cubic windows whose complex volume fits the LDS: side N = 2 * radius
// from 8 to 26, i.e. radii 4 ... 13 (N = 32 has the register-resident kernel of fftcc3d_fused.hip; larger cubes the
// plane-wise kernel of fftcc3d_planes.hip).
//
// FFTCC3D::compute(POI3D*) (src/oc_fftcc.cpp:327-427) per POI: two N^3 real windows, zero-mean, two 3-D real FFTs, the
// spectrum product conj(R) T, one inverse FFT, arg-max with the first-max rule.  The five-kernel rocFFT pipeline of
// fftcc3d.hip moves every one of those volumes through HBM.  Here one workgroup of N x N threads keeps the POI on chip
// with the plan of the 32^3 kernel -- z = ref + i * tar, ONE complex N^3 FFT, R(k) = (Z(k) + conj Z(-k)) / 2,
// T(k) = (Z(k) - conj Z(-k)) / (2i), C = conj(R) T, inverse FFT, arg-max -- except that the whole complex volume
// (N^2 (N + 1) x 8 bytes: 35 KB at N = 16, 115 KB at N = 24, 146 KB at N = 26) lives in LDS between the passes:
//   thread (a, b) owns one line per pass -- (z, x) -> its y-line (gathered: lanes side by side in x), (z, y) -> its x-line, (y, x) -> its
//   z-line -- loads it into
//   registers, transforms it there with the mixed-radix FFT of fft_device.h and writes it back in place.
// Row pitch N + 1 complex elements (odd): a wave's lanes are adjacent in the line index b, which is the x coordinate in
// the y and z passes (8-byte stride: conflict-free) and the y coordinate in the x passes (stride (N + 1) x 8 bytes, an odd
// number of 8-byte words: conflict-free as well).
// Integer outputs (u, v, w) are the reference's; the float ZNCC differs from FFTW's in the last bits like any other FFT
// (tests: identical integers against the oracle and the rocFFT pipeline, ZNCC within 1e-4 / 5e-6).
#include "oc_device.h"
#include "fft_device.h"
#include "oc_kernels.h"

namespace ochip {

namespace {

using namespace fftdev;


__device__ __forceinline__ int clampi3n(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

template <int N>
struct Cube {
    static constexpr int NP = N + 1;                       // row pitch in complex elements
    static constexpr int T = N * N;                        // lines per pass = active threads
    static constexpr int BLOCK = (T + kWave - 1) / kWave * kWave;
    static constexpr int WAVES = BLOCK / kWave;
    static constexpr int VOL = N * N * NP;                 // complex elements in LDS
};

// two block-wide sums at once (inactive threads contribute zeros); every thread returns the same values
template <int WAVES>
__device__ __forceinline__ void block_sum2n(float& x, float& y, float* red, int lane, int wave) {
    x = wave_allreduce_sum(x);
    y = wave_allreduce_sum(y);
    __syncthreads();
    if (lane == 0) {
        red[wave] = x;
        red[WAVES + wave] = y;
    }
    __syncthreads();
    float sx = 0.f, sy = 0.f;
#pragma unroll
    for (int i = 0; i < WAVES; i++) {
        sx += red[i];
        sy += red[WAVES + i];
    }
    x = sx;
    y = sy;
}

template <int N>
__global__ __launch_bounds__(Cube<N>::BLOCK) void fftcc3d_fusedn_kernel(Fftcc3dParams P, float* __restrict__ pois, int stride_f,
                                                                        unsigned long long count, int xcd_chunk) {
    using C = Cube<N>;
    constexpr int NP = C::NP, T = C::T, WAVES = C::WAVES;
    constexpr int R = N / 2;
    constexpr int M = N * N * N;
    __shared__ c2 vol[C::VOL];
    __shared__ int tab[6][N];  // voxel index of window coordinate k: ref x, y, z, tar x, y, z
    __shared__ float red[2 * WAVES];
    __shared__ int redi[WAVES];
    const int tid = threadIdx.x;
    const bool active = tid < T;
    const int lt = active ? tid : 0;  // idle threads of the last wave shadow line 0 and never write
    const int a = lt / N, b = lt - a * N;
    const int lane = tid & (kWave - 1), wave = tid >> 6;
    unsigned long long idx = blockIdx.x;
    if (xcd_chunk > 0) idx = (unsigned long long)(blockIdx.x & 7u) * xcd_chunk + (blockIdx.x >> 3);
    if (idx >= count) return;
    if (P.perm) idx = P.perm[idx];
    float* poi = pois + idx * (unsigned long long)stride_f;

    // ---- window coordinates -> voxel indices (src/oc_fftcc.cpp:349-358: Point3D(poi->x + k - rx, ...) truncated, the
    // target window displaced by the initial guess); separable: one table per axis and window.  The reference has no
    // bounds guard in 3D; indices are clamped like in fftcc3d_gather_kernel.
    for (int e = tid; e < 6 * N; e += C::BLOCK) {
        const int axis = e / N, k = e - axis * N, which = axis % 3;
        const float p = poi[which == 0 ? poi3d::X : which == 1 ? poi3d::Y : poi3d::Z];
        const float g = poi[which == 0 ? poi3d::U : which == 1 ? poi3d::V : poi3d::W];
        const int D = which == 0 ? P.dx : which == 1 ? P.dy : P.dz;
        float c = p + k - R;
        if (axis >= 3) c = c + g;
        tab[axis][k] = clampi3n((int)c, 0, D - 1);
    }
    __syncthreads();

    // ---- gather: thread (z = a, x = b) reads its Y-line of both windows; z = ref + i * tar.  Lanes are adjacent in b, so a load
    // instruction of a wave covers 64 / N whole rows (round 6; until then thread (z, y) read its x-line with 16-byte loads, one
    // row per LANE, and the texture path handles about one cache line per cycle whatever the lanes take from it: what that cost the
    // 32^3 kernel is in fftcc3d_fused.hip).  The y-pass therefore comes first.  Every lane reads the voxel its own clamped x index
    // names: no special case for windows clamped at a border.
    c2 v[N];
    {
        const float* __restrict__ rp = P.ref + (size_t)tab[2][a] * P.dy * P.dx + tab[0][b];
        const float* __restrict__ tp = P.tar + (size_t)tab[5][a] * P.dy * P.dx + tab[3][b];
        static_for<0, N>([&](auto yc) {
            constexpr int y = decltype(yc)::value;
            v[y] = mkc(rp[(size_t)tab[1][y] * P.dx], tp[(size_t)tab[4][y] * P.dx]);
        });
    }
    // means, zero-mean, sums of squares (src/oc_fftcc.cpp:360-376)
    float rn, tn;
    {
        float rs = 0.f, ts = 0.f;
#pragma unroll
        for (int k = 0; k < N; k++) {
            rs += v[k].x;
            ts += v[k].y;
        }
        rs = active ? rs : 0.f;
        ts = active ? ts : 0.f;
        block_sum2n<WAVES>(rs, ts, red, lane, wave);
        const c2 mean = mkc(rs / M, ts / M);
        rn = 0.f;
        tn = 0.f;
#pragma unroll
        for (int k = 0; k < N; k++) {
            v[k] = v[k] - mean;
            rn += v[k].x * v[k].x;
            tn += v[k].y * v[k].y;
        }
        rn = active ? rn : 0.f;
        tn = active ? tn : 0.f;
        block_sum2n<WAVES>(rn, tn, red, lane, wave);
        asm volatile("" : "+v"(rn), "+v"(tn));  // formed here, used at the very end (see fftcc3d_fused.hip)
    }

    // ---- forward y: thread (z = a, x = b); the spectrum line goes to vol[z][ky][x]
    fft_mixed<false, N>(v);
    if (active) {
        c2* __restrict__ colp = vol + a * N * NP + b;
        static_for<0, N>([&](auto kc) {
            constexpr int k = decltype(kc)::value, p = fft_pos(N, k);  // (constexpr: a run-time fft_pos() sends v[] to scratch)
            colp[k * NP] = v[p];
        });
    }
    __syncthreads();
    // ---- forward x: thread (z = a, ky = b), in place
    {
        c2* __restrict__ row = vol + (a * N + b) * NP;
        static_for<0, N>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            v[j] = row[j];
        });
        fft_mixed<false, N>(v);
        if (active) {
            static_for<0, N>([&](auto kc) {
                constexpr int k = decltype(kc)::value, p = fft_pos(N, k);
                row[k] = v[p];
            });
        }
    }
    __syncthreads();
    // ---- forward z: thread (ky = a, kx = b); Z(kz, ky, kx) goes back in place so that the owner of line (-ky, -kx) can read it
    c2* __restrict__ zp = vol + a * NP + b;  // element kz at zp[kz * N * NP]
    {
        static_for<0, N>([&](auto zc) {
            constexpr int z = decltype(zc)::value;
            v[z] = zp[z * N * NP];
        });
        fft_mixed<false, N>(v);
        if (active) {
            static_for<0, N>([&](auto kc) {
                constexpr int k = decltype(kc)::value, p = fft_pos(N, k);
                zp[k * N * NP] = v[p];
            });
        }
    }
    __syncthreads();
    // ---- spectra of the two real windows and their product conj(R) * T (src/oc_fftcc.cpp:378-386):
    // Z(-k) = line (-ky, -kx) read backwards in kz
    c2 t[N];
    {
        const int my = (N - a) % N, mx = (N - b) % N;
        const c2* __restrict__ mp = vol + my * NP + mx;
        static_for<0, N>([&](auto zc) {
            constexpr int z = decltype(zc)::value, p = fft_pos(N, z);
            const c2 zm = mp[((N - z) % N) * N * NP];
            const c2 zk = v[p];
            const float rr = 0.5f * (zk.x + zm.x), ri = 0.5f * (zk.y - zm.y);
            const float tr = 0.5f * (zk.y + zm.y), ti = -0.5f * (zk.x - zm.x);
            t[z] = mkc((rr * tr) + (ri * ti), (rr * ti) - (ri * tr));
        });
    }
    __syncthreads();  // every mirror line has been read before anybody overwrites its own
    // ---- inverse z (unnormalised, like FFTW's c2r), in place
    fft_mixed<true, N>(t);
    if (active) {
        static_for<0, N>([&](auto zc) {
            constexpr int z = decltype(zc)::value, p = fft_pos(N, z);
            zp[z * N * NP] = t[p];
        });
    }
    __syncthreads();
    // ---- inverse y: thread (z = a, x = b), in place
    {
        c2* __restrict__ colp = vol + a * N * NP + b;
        static_for<0, N>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            t[k] = colp[k * NP];
        });
        fft_mixed<true, N>(t);
        if (active) {
            static_for<0, N>([&](auto jc) {
                constexpr int j = decltype(jc)::value, p = fft_pos(N, j);
                colp[j * NP] = t[p];
            });
        }
    }
    __syncthreads();
    // ---- inverse x: thread (z = a, y = b); the correlation volume's real part stays in registers
    {
        const c2* __restrict__ row = vol + (a * N + b) * NP;
        static_for<0, N>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            t[k] = row[k];
        });
        fft_mixed<true, N>(t);
    }
    // ---- arg-max with "strict >, scanning from index 0" (src/oc_fftcc.cpp:391-400): the thread's N values sit at linear
    // indices (a * N + b) * N + x, ascending in x
    float best = -2.f;
    int bidx = 0x7fffffff;
    if (active) {
        bidx = (a * N + b) * N;
        static_for<0, N>([&](auto xc) {
            constexpr int x = decltype(xc)::value, p = fft_pos(N, x);
            const float val = t[p].x;
            if (val > best) {
                best = val;
                bidx = (a * N + b) * N + x;
            }
        });
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
        red[wave] = best;
        redi[wave] = bidx;
    }
    __syncthreads();
    if (tid == 0) {
        for (int i = 1; i < WAVES; i++)
            if (red[i] > best || (red[i] == best && redi[i] < bidx)) {
                best = red[i];
                bidx = redi[i];
            }
        int du = bidx % N, dv = (bidx / N) % N, dw = bidx / (N * N);  // src/oc_fftcc.cpp:401-403
        if (du > R) du -= N;
        if (dv > R) dv -= N;
        if (dw > R) dw -= N;
        const float gu = poi[poi3d::U], gv = poi[poi3d::V], gw = poi[poi3d::W];
        poi[poi3d::U] = (float)du + gu;
        poi[poi3d::V] = (float)dv + gv;
        poi[poi3d::W] = (float)dw + gw;
        poi[poi3d::U0] = gu;
        poi[poi3d::V0] = gv;
        poi[poi3d::W0] = gw;
        poi[poi3d::ZNCC] = best / (sqrtf(rn * tn) * M);
    }
}

template <int N>
hipError_t launch_cube(const Fftcc3dParams& p, float* pois, int stride_f, size_t count, bool xcd, hipStream_t stream) {
    const int chunk = xcd ? (int)((count + 7) / 8) : 0;
    const size_t grid = xcd ? (size_t)chunk * 8 : count;
    (void)hipGetLastError();  // drop stale errors of earlier, unrelated calls
    hipLaunchKernelGGL(fftcc3d_fusedn_kernel<N>, dim3((unsigned)grid), dim3(Cube<N>::BLOCK), 0, stream, p, pois, stride_f,
                       (unsigned long long)count, chunk);
    return hipGetLastError();
}

}  // namespace

// cubic windows of side 8 ... 26 (radius 4 ... 13)
bool fftcc3d_fusedn_supported(int rx, int ry, int rz) { return rx == ry && ry == rz && rx >= 4 && rx <= 13; }

hipError_t launch_fftcc3d_fusedn(const Fftcc3dParams& p, float* pois, int stride_f, size_t count, bool xcd, hipStream_t stream) {
    if (count == 0) return hipSuccess;
    if (!fftcc3d_fusedn_supported(p.rx, p.ry, p.rz)) return hipErrorInvalidValue;
    switch (2 * p.rx) {
        case 8: return launch_cube<8>(p, pois, stride_f, count, xcd, stream);
        case 10: return launch_cube<10>(p, pois, stride_f, count, xcd, stream);
        case 12: return launch_cube<12>(p, pois, stride_f, count, xcd, stream);
        case 14: return launch_cube<14>(p, pois, stride_f, count, xcd, stream);
        case 16: return launch_cube<16>(p, pois, stride_f, count, xcd, stream);
        case 18: return launch_cube<18>(p, pois, stride_f, count, xcd, stream);
        case 20: return launch_cube<20>(p, pois, stride_f, count, xcd, stream);
        case 22: return launch_cube<22>(p, pois, stride_f, count, xcd, stream);
        case 24: return launch_cube<24>(p, pois, stride_f, count, xcd, stream);
        case 26: return launch_cube<26>(p, pois, stride_f, count, xcd, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace ochip
