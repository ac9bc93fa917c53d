// fftcc3d_planes_impl.h -- (kernel template; instantiated in fftcc3d_planes.hip and fftcc3d_planesb.hip)
// FFTCC3D in ONE kernel for cubic windows too large for the chip: side N = 2 * radius from 28 to 64
// (N = 32 keeps its register-resident kernel), first of all the 60^3 windows of the reference's own DVC example
// (subset radius 30: examples/test_dvc_fftcc_icgn1.cpp:45-47,87-95; examples/test_dvc_gpu_icgn.cpp).
//
// FFTCC3D::compute(POI3D*) (src/oc_fftcc.cpp:327-427) per POI: two N^3 real windows, zero-mean, two 3-D real FFTs, the
// spectrum product conj(R) T, one inverse FFT, arg-max with the first-max rule.  z = ref + i * tar is 1.73 MB at N = 60 --
// 6.6 x what the 32^3 kernel keeps in a workgroup's registers, 11 x the LDS -- so the volume has to pass through memory
// between the axis passes.  The five-kernel rocFFT pipeline (fftcc3d.hip) moves ~10 MB per POI through HBM in 13 plans'
// worth of launches and costs 7.5 us per POI (3.8 ms for the example's 512-POI test queue, round 3).  Here ONE persistent
// 512-thread workgroup per CU owns a POI at a time and a private scratch volume S of N^3 complex elements:
//   A  forward planes.  Wave w takes the z-planes w, w + 8, ...: lane x gathers COLUMN x of both windows (z = ref + i * tar; the
//      lanes side by side in x: round 6 -- inside the kernel the plane is transposed, which only the arg-max has to know; the
//      description below keeps the old names: read "x" as the register index and "y" as the lane), transforms it in registers
//      (mixed-radix FFT of fft_device.h), the wave transposes the plane through
//      its LDS tile (N x (N + 1) floats, real parts then imaginary parts), lane kx transforms along y and writes
//      S[z][ky][kx] -- N consecutive complex numbers per row: coalesced.
//   B  z pass.  Thread (row slot, kx) loads its z-line from S (lanes adjacent in kx: coalesced), transforms it, and needs
//      Z(-k) for the spectrum product: the workgroup's lines of one round go through LDS (real parts, then imaginary parts),
//      where every thread reads the line (-ky, -kx) backwards.  Rows are dealt to the rounds in mirror pairs
//      (0, N/2, 1, N-1, 2, N-2, ...), so a round is closed under k -> -k and the product can go back IN PLACE; then the
//      inverse z transform, and the line returns to S.
//   C  inverse planes.  Wave w takes plane z again: lane kx loads its ky-line, inverse y, transpose, lane y inverse x: a row
//      of the correlation volume, scanned for the maximum on the spot (strict >, ascending linear index).
// Per POI that is the two windows once (mostly L2 hits: neighbouring POIs overlap) and 4 passes over S = 4 x 8 N^3 bytes
// (6.9 MB at N = 60) instead of ~10 MB through five kernels, with the x and y transforms never leaving the chip.
//
// Zero-mean without a second look at the windows: the windows are gathered once, as v' = v - c0 with c0 the window's centre
// voxel (one constant per window); mean and norm follow from the running sums (sum v', sum v'^2), and since the transform
// is linear the only spectral bin that differs from the zero-mean window's is k = 0 -- where R(0) = T(0) = 0 for zero-mean
// windows -- so the product's DC bin is set to zero instead.
// Integer outputs (u, v, w) are the reference's; the float ZNCC differs from FFTW's in the last bits like any other FFT
// (tests: identical integers against the oracle and the rocFFT pipeline, ZNCC within 1e-4 / 1e-5).
#pragma once

#include "oc_device.h"
#include "fft_device.h"
#include "oc_kernels.h"

namespace ochip {

namespace planes {

using namespace fftdev;


constexpr int kPlThreads = 512;
constexpr int kPlWaves = kPlThreads / kWave;  // 8

__device__ __forceinline__ int clampi3p(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

template <int N>
struct Planes {
    static constexpr int NP = N + 1;                           // tile pitch in floats (odd)
    static constexpr int RR = (kPlThreads / N) & ~1;           // rows of z-lines per round of phase B (even: mirror pairs)
    static constexpr int ROUNDS = (N + RR - 1) / RR;
    static constexpr int TILE = N * NP;                        // floats of one wave's plane tile
    static constexpr int XCH = RR * N * NP;                    // floats of phase B's exchange buffer
    static constexpr int LDSF = kPlWaves * TILE > XCH ? kPlWaves * TILE : XCH;
    static_assert(N % 2 == 0 && N >= 8 && N <= 64, "window side");
    static_assert(RR >= 2, "phase B needs at least one mirror pair of rows per round");
};

// row slot -> row of the (ky, kx) plane: 0, N/2, 1, N-1, 2, N-2, ...; slots 2q, 2q+1 (q >= 1) hold the mirror pair (q, N-q),
// slots 0 and 1 the two self-mirrored rows
template <int N>
__device__ __forceinline__ int slot_row(int s) {
    const int q = s >> 1;
    return s == 0 ? 0 : (s == 1 ? N / 2 : ((s & 1) ? N - q : q));
}
__device__ __forceinline__ int mirror_slot(int s) { return s < 2 ? s : (s ^ 1); }

template <int N>
__global__ __launch_bounds__(kPlThreads) void fftcc3d_planes_kernel(Fftcc3dParams P, float* __restrict__ pois, int stride_f,
                                                                   unsigned long long count, c2* __restrict__ scratch) {
    using PL = Planes<N>;
    constexpr int NP = PL::NP, RR = PL::RR;
    constexpr int R = N / 2;
    constexpr int M = N * N * N;
    __shared__ float ldsf[PL::LDSF];
    __shared__ int tab[6][N];  // voxel index of window coordinate k: ref x, y, z, tar x, y, z
    __shared__ float red[4 * kPlWaves];
    __shared__ int redi[kPlWaves];
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool act = lane < N;          // lanes that own a row / column of a plane
    const int l = act ? lane : 0;       // idle lanes shadow lane 0 and never write
    float* __restrict__ ft = ldsf + wave * PL::TILE;
    c2* __restrict__ S = scratch + (size_t)blockIdx.x * M;

    // Workgroups are dealt round-robin to the 8 XCDs: every XCD walks a contiguous eighth of the queue (neighbouring POIs
    // share most of their voxels behind one L2)
    const unsigned long long xcd_chunk = (count + 7) / 8, xcd_lo = (blockIdx.x & 7u) * xcd_chunk;
    const unsigned long long xcd_hi = min(count, xcd_lo + xcd_chunk);
    for (unsigned long long idx = xcd_lo + (blockIdx.x >> 3); idx < xcd_hi; idx += gridDim.x >> 3) {
        float* poi = pois + (P.perm ? (unsigned long long)P.perm[idx] : idx) * (unsigned long long)stride_f;
        __syncthreads();  // the previous POI is done with the tables and the tiles

        // ---- window coordinates -> voxel indices (src/oc_fftcc.cpp:349-358: Point3D(poi->x + k - rx, ...) truncated, the
        // target window displaced by the initial guess); separable: one table per axis and window.  The reference has no
        // bounds guard in 3D; indices are clamped like in fftcc3d_gather_kernel.
        for (int e = tid; e < 6 * N; e += kPlThreads) {
            const int axis = e / N, k = e - axis * N, which = axis % 3;
            const float p = poi[which == 0 ? poi3d::X : which == 1 ? poi3d::Y : poi3d::Z];
            const float g = poi[which == 0 ? poi3d::U : which == 1 ? poi3d::V : poi3d::W];
            const int D = which == 0 ? P.dx : which == 1 ? P.dy : P.dz;
            float c = p + k - R;
            if (axis >= 3) c = c + g;
            tab[axis][k] = clampi3p((int)c, 0, D - 1);
        }
        __syncthreads();
        // the constants the windows are shifted by: their centre voxels (any constant would do; one near the mean keeps the
        // running sums small)
        const float c0r = P.ref[((size_t)tab[2][R] * P.dy + tab[1][R]) * P.dx + tab[0][R]];
        const float c0t = P.tar[((size_t)tab[5][R] * P.dy + tab[4][R]) * P.dx + tab[3][R]];

        // ================= A: forward planes =================
        float s1r = 0.f, s2r = 0.f, s1t = 0.f, s2t = 0.f;
#pragma unroll 1
        for (int z = wave; z < N; z += kPlWaves) {
            c2 v[N];
            {
                // lane x gathers COLUMN x of the plane (round 6): the lanes of the wave sit side by side in x, so a load instruction
                // covers one row of the window.  (Until then lane y read ROW y with 16-byte loads -- one row per lane, and the texture
                // path handles about one cache line per cycle whatever the lanes take from it: fftcc3d_fused.hip.)  Inside the kernel
                // the plane is therefore TRANSPOSED -- lane / row index = x, register / column index = y -- down to the arg-max,
                // which is where x and y are told apart again.
                const float* __restrict__ rp = P.ref + (size_t)tab[2][z] * P.dy * P.dx + tab[0][l];
                const float* __restrict__ tp = P.tar + (size_t)tab[5][z] * P.dy * P.dx + tab[3][l];
                static_for<0, N>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    v[k] = mkc(rp[(size_t)tab[1][k] * P.dx], tp[(size_t)tab[4][k] * P.dx]);
                });
            }
            {
                const c2 c0 = mkc(c0r, c0t);
                float a1r = 0.f, a2r = 0.f, a1t = 0.f, a2t = 0.f;
#pragma unroll
                for (int k = 0; k < N; k++) {
                    v[k] = v[k] - c0;
                    a1r += v[k].x;
                    a1t += v[k].y;
                    a2r += v[k].x * v[k].x;
                    a2t += v[k].y * v[k].y;
                }
                if (act) {
                    s1r += a1r;
                    s2r += a2r;
                    s1t += a1t;
                    s2t += a2t;
                }
            }
            // x transform: lane y holds X(kx) of its row in v[fft_pos(N, kx)]
            fft_mixed<false, N>(v);
            // transpose (rows -> columns) through the wave's tile, in place in the register array
#define OC_PLANES_T1(PART)                                                             \
    static_for<0, N>([&](auto kc) {                                                    \
        constexpr int k = decltype(kc)::value, p = fft_pos(N, k);                      \
        if (act) ft[k * NP + l] = v[p].PART;                                           \
    });                                                                                \
    __builtin_amdgcn_wave_barrier();                                                   \
    static_for<0, N>([&](auto yc) {                                                    \
        constexpr int y = decltype(yc)::value;                                         \
        v[y].PART = ft[l * NP + y];                                                    \
    });                                                                                \
    __builtin_amdgcn_wave_barrier();
            OC_PLANES_T1(x)
            OC_PLANES_T1(y)
#undef OC_PLANES_T1
            // y transform: lane kx holds the plane's spectrum Zp(ky, kx) in v[fft_pos(N, ky)]; out to S[z][ky][kx]
            fft_mixed<false, N>(v);
            if (act) {
                c2* __restrict__ dst = S + (size_t)z * N * N + l;
                static_for<0, N>([&](auto kc) {
                    constexpr int k = decltype(kc)::value, p = fft_pos(N, k);
                    dst[k * N] = v[p];
                });
            }
        }
        // ---- means and norms from the running sums (src/oc_fftcc.cpp:360-376 in exact arithmetic)
        float rn, tn;
        {
            s1r = wave_allreduce_sum(s1r);
            s2r = wave_allreduce_sum(s2r);
            s1t = wave_allreduce_sum(s1t);
            s2t = wave_allreduce_sum(s2t);
            if (lane == 0) {
                red[wave] = s1r;
                red[kPlWaves + wave] = s2r;
                red[2 * kPlWaves + wave] = s1t;
                red[3 * kPlWaves + wave] = s2t;
            }
            __syncthreads();  // also: every plane of S is written before phase B reads it
            double t1r = 0.0, t2r = 0.0, t1t = 0.0, t2t = 0.0;
#pragma unroll
            for (int i = 0; i < kPlWaves; i++) {
                t1r += (double)red[i];
                t2r += (double)red[kPlWaves + i];
                t1t += (double)red[2 * kPlWaves + i];
                t2t += (double)red[3 * kPlWaves + i];
            }
            rn = (float)(t2r - t1r * t1r / (double)M);  // sum (v - mean)^2 = sum v'^2 - (sum v')^2 / M
            tn = (float)(t2t - t1t * t1t / (double)M);
            asm volatile("" : "+v"(rn), "+v"(tn));  // formed here, used at the very end
        }

        // ================= B: z pass, spectrum product, inverse z pass =================
        {
            const int rs = tid / N, kx = tid - rs * N;   // row slot inside the round, column
            const bool busy = rs < RR;
#pragma unroll 1
            for (int round = 0; round < PL::ROUNDS; round++) {
                const int slot = round * RR + rs;
                const bool on = busy && slot < N;
                const int ky = on ? slot_row<N>(slot) : 0;
                const int kxx = on ? kx : 0;
                c2* __restrict__ line = S + (size_t)ky * N + kxx;  // element kz at line[kz * N * N]
                c2 w[N];
                static_for<0, N>([&](auto zc) {
                    constexpr int z = decltype(zc)::value;
                    w[z] = line[(size_t)z * N * N];
                });
                fft_mixed<false, N>(w);  // Z(kz; ky, kx) in w[fft_pos(N, kz)]
                // the mirror line (-ky, -kx) sits in this round as well: slot ^ 1 (or the slot itself for rows 0 and N/2)
                const int mrs = mirror_slot(on ? slot : 0) - round * RR;
                const int mcol = (N - kxx) % N;
                float* __restrict__ mine = ldsf + ((on ? rs : 0) * N + kxx) * NP;
                const float* __restrict__ theirs = ldsf + ((on ? mrs : 0) * N + mcol) * NP;
                float zmx[N];
                __syncthreads();  // the previous round has finished with the exchange buffer
                if (on) {
                    static_for<0, N>([&](auto kc) {
                        constexpr int k = decltype(kc)::value, p = fft_pos(N, k);
                        mine[k] = w[p].x;
                    });
                }
                __syncthreads();
                static_for<0, N>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    zmx[k] = theirs[(N - k) % N];
                });
                __syncthreads();
                if (on) {
                    static_for<0, N>([&](auto kc) {
                        constexpr int k = decltype(kc)::value, p = fft_pos(N, k);
                        mine[k] = w[p].y;
                    });
                }
                __syncthreads();
                // spectra of the two real windows and their product conj(R) * T (src/oc_fftcc.cpp:378-386); the result takes
                // the register of the bin it was formed from, and the inverse transform reads it in natural order (a renaming)
                c2 t[N];
                static_for<0, N>([&](auto kc) {
                    constexpr int k = decltype(kc)::value, p = fft_pos(N, k);
                    const c2 zk = w[p];
                    const float zmy = theirs[(N - k) % N];
                    const float rr = 0.5f * (zk.x + zmx[k]), ri = 0.5f * (zk.y - zmy);
                    const float tr = 0.5f * (zk.y + zmy), ti = -0.5f * (zk.x - zmx[k]);
                    t[k] = mkc((rr * tr) + (ri * ti), (rr * ti) - (ri * tr));
                });
                // the zero-mean windows' spectra vanish at k = 0 (see the header): so does the product
                if (ky == 0 && kxx == 0) t[0] = mkc(0.f, 0.f);
                fft_mixed<true, N>(t);  // unnormalised, like FFTW's c2r
                if (on) {
                    static_for<0, N>([&](auto zc) {
                        constexpr int z = decltype(zc)::value, p = fft_pos(N, z);
                        line[(size_t)z * N * N] = t[p];
                    });
                }
            }
        }
        __syncthreads();  // every line of S is back before phase C reads planes; the exchange buffer becomes the tiles again

        // ================= C: inverse planes + arg-max =================
        float best = -2.f;
        int bidx = 0x7fffffff;
#pragma unroll 1
        for (int z = wave; z < N; z += kPlWaves) {
            c2 v[N];
            {
                const c2* __restrict__ src = S + (size_t)z * N * N + l;
                static_for<0, N>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    v[k] = src[k * N];
                });
            }
            fft_mixed<true, N>(v);  // lane kx: g(y; kx) in v[fft_pos(N, y)]
#define OC_PLANES_T2(PART)                                                             \
    static_for<0, N>([&](auto yc) {                                                    \
        constexpr int y = decltype(yc)::value, p = fft_pos(N, y);                      \
        if (act) ft[l * NP + y] = v[p].PART;                                           \
    });                                                                                \
    __builtin_amdgcn_wave_barrier();                                                   \
    static_for<0, N>([&](auto kc) {                                                    \
        constexpr int k = decltype(kc)::value;                                         \
        v[k].PART = ft[k * NP + l];                                                    \
    });                                                                                \
    __builtin_amdgcn_wave_barrier();
            OC_PLANES_T2(x)
            OC_PLANES_T2(y)
#undef OC_PLANES_T2
            fft_mixed<true, N>(v);  // lane x (the plane is transposed, see the gather): the correlation column (z, y, x = l) in v[fft_pos(N, y)].x
            // arg-max with "strict >, scanning from index 0" (src/oc_fftcc.cpp:391-400): this lane's planes come in
            // ascending z, its column's values in ascending y -- ascending linear index (z * N + y) * N + x
            if (act) {
                static_for<0, N>([&](auto yc) {
                    constexpr int y = decltype(yc)::value, p = fft_pos(N, y);
                    const float val = v[p].x;
                    if (val > best) {
                        best = val;
                        bidx = (z * N + y) * N + l;
                    }
                });
            }
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
            for (int i = 1; i < kPlWaves; i++)
                if (red[i] > best || (red[i] == best && redi[i] < bidx)) {
                    best = red[i];
                    bidx = redi[i];
                }
            if (bidx == 0x7fffffff) bidx = 0;  // nothing above -2 (a NaN volume): the reference keeps index 0
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
}

template <int N>
hipError_t launch_planes(const Fftcc3dParams& p, float* pois, int stride_f, size_t count, void* scratch, int blocks,
                         hipStream_t stream) {
    unsigned grid = (unsigned)(count < (size_t)blocks ? count : (size_t)blocks);
    grid = (grid + 7) / 8 * 8;  // whole XCD rounds (idle workgroups leave at once); never more than `blocks` scratch slots
    (void)hipGetLastError();    // drop stale errors of earlier, unrelated calls
    hipLaunchKernelGGL(fftcc3d_planes_kernel<N>, dim3(grid), dim3(kPlThreads), 0, stream, p, pois, stride_f,
                       (unsigned long long)count, static_cast<c2*>(scratch));
    return hipGetLastError();
}

}  // namespace planes
}  // namespace ochip
