// fftcc3d_fused_r5.hip -- the fused 32^3 FFTCC3D kernel as it stood at the end of round 5: the A/B partner of fftcc3d_fused.hip
// (A/B build only, tuning "fftcc3d_fused" = 2; tests/ab/).  Round 6 replaced its decomposition (x-lines gathered one row per lane,
// five full-volume exchanges, 22 workgroup barriers: 8.27 ms on config E's queue) by the one in fftcc3d_fused.hip (5.7 ms, same
// integers, ZNCC within 4e-7).  Kept verbatim except for the names.
//
// FFTCC3D for 32 x 32 x 32 windows (subset radius 16) in ONE kernel.
//
// FFTCC3D::compute(POI3D*) (src/oc_fftcc.cpp:327-427) needs, per POI, two 3-D real FFTs, a spectrum product and one
// inverse FFT of a 128 KB window.  The rocFFT pipeline of fftcc3d.hip moves ~1.6 MB per POI through HBM between its
// kernels (windows, two spectra, product, correlation volume); here one 1024-thread workgroup keeps the POI on chip:
//   gather  ->  z = ref + i*tar (zero-mean)  ->  ONE complex 32^3 FFT  ->
//   R(k) = (Z(k) + conj Z(-k))/2, T(k) = (Z(k) - conj Z(-k))/(2i), C = conj(R) T  ->
//   inverse complex FFT (unnormalised, like FFTW's c2r)  ->  arg-max with the first-max rule, wrap, ZNCC.
// The 32^3 complex volume (256 KB) lives in REGISTERS, one 32-point line per thread (64 VGPRs): every axis pass is a
// 32-point FFT in registers (fft_device.h), and between passes the volume is re-distributed through LDS, half of it
// (16 planes x 32 x 33 complex = 132 KB) at a time:
//   LX: thread (z, y) holds the x-line   --[z-halves]-->   LY: thread (z, x) holds the y-line
//   LY                                   --[y-halves]-->   LZ: thread (y, x) holds the z-line
// The spectrum product needs Z(-k): the z-lines are exchanged through LDS in two halves of equal y-parity (k -> -k
// preserves the parity of every index, so each half is closed under the mirror).  Row pitch 33 complex keeps the
// line-wise accesses of both sides of every exchange conflict-free.
// HBM traffic: the two windows once (256 KB, mostly L2 hits between neighbouring POIs) and 28 bytes of results.
// Integer outputs (u, v, w) are what the reference computes; the float ZNCC differs from FFTW's in the last bits
// like any other FFT implementation (tested to 1e-5 against the oracle's double-precision DFT).
#include "oc_device.h"
#include "fft_device.h"
#include "oc_kernels.h"

// OC_FUSED32_WAVE_XY: 1 = the x <-> y exchanges synchronise inside the half-wave that owns a z-plane (a wave-level fence) and
// keep only the workgroup barrier between the two z-halves; 0 = two workgroup barriers per half (rounds 1 - 3).  Config E:
// 9.08 against 9.32 ms, bit-identical (profiles/r4s_fftcc3d_fused32_ab_wave_local_xy.txt).
#ifndef OC_FUSED32_WAVE_XY
#define OC_FUSED32_WAVE_XY 1
#endif

namespace ochip {

namespace {

using namespace fftdev;

constexpr int TN = 32;                    // window side (2 * radius)
constexpr int TP = TN + 1;                // LDS row pitch in complex elements
constexpr int kThreads3 = TN * TN;        // one line per thread
constexpr int kWaves = kThreads3 / kWave;  // 16
constexpr int kHalf = 16 * TN * TP;       // complex elements of half a volume in LDS

typedef float float4u __attribute__((ext_vector_type(4), aligned(4)));  // 4-byte aligned 16-byte load

__device__ __forceinline__ int clampi3(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// orders this wave's earlier LDS writes before its later LDS reads (data exchanged between lanes of ONE wave: the LDS executes
// a wave's instructions in issue order, so all that is needed is that the compiler keeps the order and waits for the writes)
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// two block-wide sums at once; every thread returns the same values.  ONE barrier: `red` must not be in use by an earlier
// call (each call site has its own 2 * kWaves floats)
__device__ __forceinline__ void block_sum2(float& x, float& y, float* red, int lane, int wave) {
    x = wave_allreduce_sum(x);
    y = wave_allreduce_sum(y);
    if (lane == 0) {
        red[wave] = x;
        red[kWaves + wave] = y;
    }
    __syncthreads();
    float sx = 0.f, sy = 0.f;
#pragma unroll
    for (int i = 0; i < kWaves; i++) {
        sx += red[i];
        sy += red[kWaves + i];
    }
    x = sx;
    y = sy;
}

// One POI by one workgroup.  CLAMPED = false (the launch that does the work): windows whose x indices are contiguous in both
// volumes -- every window that is not clamped at a volume border -- gathered with 16-byte loads; a workgroup that finds its
// window clamped only raises needs_clamped[idx] and leaves.  CLAMPED = true (a second, small launch whose workgroups scan the
// flags): the flagged POIs, gathered element by element through the index tables.  Two instantiations because the scalar
// gather's 64 addresses raise the register pressure of the WHOLE kernel when both paths live in one: 104 B of scratch per
// thread against 76 B for the contiguous-only instantiation -- and at 1 024 threads x 50 000 POIs every scratch byte is
// 50 MB written to memory and read back (round 5, profiles/r5e_fftcc3d_block_schedule_ab.json: WRITE_SIZE 4.98 GB per launch
// for 28 B of results per POI).
template <bool CLAMPED>
__device__ __forceinline__ void fftcc3d_fused32_r5_poi(const Fftcc3dParams& P, float* __restrict__ pois, int stride_f, unsigned long long idx,
                                                    unsigned char* __restrict__ needs_clamped, unsigned* __restrict__ any_clamped) {
    __shared__ c2 lds[kHalf];
    __shared__ int tab[6][TN];  // voxel index of window coordinate k: ref x, y, z, tar x, y, z
    __shared__ float red[2 * kWaves], red2[2 * kWaves];
    __shared__ int redi[kWaves];
    __shared__ float norms[2];  // sums of squares of the two windows: formed early, needed by thread 0 at the very end
    const int tid = threadIdx.x;
    const int a = tid >> 5, b = tid & 31;
    const int lane = tid & (kWave - 1), wave = tid >> 6;
    float* poi = pois + idx * (unsigned long long)stride_f;
    constexpr int R = TN / 2;
    constexpr int M = TN * TN * TN;

    // ---- window coordinates -> voxel indices (src/oc_fftcc.cpp:349-358: Point3D(poi->x + k - rx, ...) truncated, the
    // target window displaced by the initial guess).  Separable, so one table per axis and window.  The reference has
    // no bounds guard in 3D; indices are clamped like in fftcc3d_gather_kernel.
    if (tid < 6 * TN) {
        const int axis = tid >> 5, k = tid & 31, which = axis % 3;
        const float p = poi[which == 0 ? poi3d::X : which == 1 ? poi3d::Y : poi3d::Z];
        const float g = poi[which == 0 ? poi3d::U : which == 1 ? poi3d::V : poi3d::W];
        const int D = which == 0 ? P.dx : which == 1 ? P.dy : P.dz;
        float c = p + k - R;
        if (axis >= 3) c = c + g;
        tab[axis][k] = clampi3((int)c, 0, D - 1);
    }
    // both windows' x indices contiguous (true unless a window is clamped at the border): 16-byte loads.  Every thread forms
    // the four table entries the test needs itself, so that the vote's barrier is also the one that publishes `tab`
    bool mine_contig;
    {
        const float p = poi[poi3d::X], g = poi[poi3d::U];
        const float c0 = p + 0 - R, cb = p + b - R;
        const int r0 = clampi3((int)c0, 0, P.dx - 1), rb = clampi3((int)cb, 0, P.dx - 1);
        const int t0 = clampi3((int)(c0 + g), 0, P.dx - 1), tb = clampi3((int)(cb + g), 0, P.dx - 1);
        mine_contig = rb == r0 + b && tb == t0 + b;
    }
    const bool contig = __syncthreads_and(mine_contig) != 0;
    if constexpr (!CLAMPED) {
        if (tid == 0) {
            needs_clamped[idx] = contig ? 0 : 1;
            if (!contig) atomicOr(any_clamped, 1u);
        }
        if (!contig) return;
    }

    // ---- gather: thread (z = a, y = b) reads its x-line of both windows; z = ref + i*tar
    c2 v[TN];
    {
        const float* __restrict__ rrow = P.ref + ((size_t)tab[2][a] * P.dy + tab[1][b]) * P.dx;
        const float* __restrict__ trow = P.tar + ((size_t)tab[5][a] * P.dy + tab[4][b]) * P.dx;
        if (!CLAMPED) {
            const float* __restrict__ rp = rrow + tab[0][0];
            const float* __restrict__ tp = trow + tab[3][0];
#pragma unroll
            for (int q = 0; q < TN / 4; q++) {
                const float4u r4 = *reinterpret_cast<const float4u*>(rp + 4 * q);
                const float4u t4 = *reinterpret_cast<const float4u*>(tp + 4 * q);
                v[4 * q + 0] = mkc(r4.x, t4.x);
                v[4 * q + 1] = mkc(r4.y, t4.y);
                v[4 * q + 2] = mkc(r4.z, t4.z);
                v[4 * q + 3] = mkc(r4.w, t4.w);
            }
        } else {
#pragma unroll
            for (int k = 0; k < TN; k++) v[k] = mkc(rrow[tab[0][k]], trow[tab[3][k]]);
        }
    }
    // means, zero-mean, sums of squares (src/oc_fftcc.cpp:360-376)
    {
        float rn, tn;
        float rs = 0.f, ts = 0.f;
#pragma unroll
        for (int k = 0; k < TN; k++) {
            rs += v[k].x;
            ts += v[k].y;
        }
        block_sum2(rs, ts, red, lane, wave);
        const c2 mean = mkc(rs / M, ts / M);
        rn = 0.f;
        tn = 0.f;
#pragma unroll
        for (int k = 0; k < TN; k++) {
            v[k] = v[k] - mean;
            rn += v[k].x * v[k].x;
            tn += v[k].y * v[k].y;
        }
        block_sum2(rn, tn, red2, lane, wave);
        // needed only at the very end, by thread 0: parked in LDS instead of two registers of every thread
        if (tid == 0) {
            norms[0] = rn;
            norms[1] = tn;
        }
    }

    const int zz = a & 15, half = a >> 4;
    // ---- forward x, then LX -> LY through LDS [z & 15][y][x], one z-half at a time
    fft32<false>(v);
    // (a z-plane is written and read by the 32 threads of ONE half-wave: inside the plane a wave-level fence orders its
    // LDS writes before its reads -- the LDS serves a wave's instructions in order -- and the workgroup barrier is only
    // needed where the two z-halves hand the 16 plane slots over)
#pragma unroll
    for (int h = 0; h < 2; h++) {
        if (half == h) {
#pragma unroll
            for (int k = 0; k < TN; k++) lds[(zz * TN + b) * TP + k] = v[bitrev5(k)];
#if OC_FUSED32_WAVE_XY
            wave_lds_fence();
#else
        }
        __syncthreads();
        if (half == h) {
#endif
#pragma unroll
            for (int j = 0; j < TN; j++) v[j] = lds[(zz * TN + j) * TP + b];
        }
        __syncthreads();
    }
    // ---- forward y (thread (z = a, x = b)), then LY -> LZ.  The (y, z) plane of every x is cut into four 16 x 16 blocks
    // (y-half, z-half); the tile holds two of them: [z-half][y & 15][z & 15][x].  Round 0 moves the DIAGONAL blocks, round 1 the
    // off-diagonal ones: a thread (z-half = its own `half` as a writer, y-half = `half` as a reader) hands over 16 values and
    // receives 16 values per round, so it never holds more than one line's worth of data (64 registers).  Round 5: until
    // then round g moved y-half g -- every thread wrote 16 values, half of the threads read 32 -- and a thread that had
    // received its whole z-line in round 0 still held the 16 values it owed round 1: 96 live data registers of the 128 the
    // workgroup size leaves, 16 of them in scratch (64 of the kernel's 76 B per thread).  `half` is uniform over a wave, so
    // the two register-index patterns are two branches, not selects.
    fft32<false>(v);
    c2 w[TN];
#pragma unroll
    for (int d = 0; d < 2; d++) {
        {   // writer (z = a): block (y-half = half ^ d, z-half = half) -> slot `half`
            c2* __restrict__ dst = lds + ((half * 16) * 16 + zz) * TP + b;
            if ((half ^ d) == 0) {
#pragma unroll
                for (int yy = 0; yy < 16; yy++) dst[yy * 16 * TP] = v[bitrev5(yy)];
            } else {
#pragma unroll
                for (int yy = 0; yy < 16; yy++) dst[yy * 16 * TP] = v[bitrev5(yy + 16)];
            }
        }
        __syncthreads();
        {   // reader (y = a): block (y-half = half, z-half = half ^ d) -> slot `half ^ d`
            const c2* __restrict__ src = lds + (((half ^ d) * 16 + zz) * 16) * TP + b;
            if ((half ^ d) == 0) {
#pragma unroll
                for (int z = 0; z < 16; z++) w[z] = src[z * TP];
            } else {
#pragma unroll
                for (int z = 0; z < 16; z++) w[z + 16] = src[z * TP];
            }
        }
        __syncthreads();
    }
    // ---- forward z (thread (y = a, x = b)): Z(kz, ky = a, kx = b) in w[bitrev5(kz)]
    fft32<false>(w);
    // ---- spectra of the two real windows and their product conj(R) * T (src/oc_fftcc.cpp:378-386); Z(-k) comes
    // from the thread that owns line (-ky, -kx), through LDS [ky >> 1][kx][kz], one ky-parity at a time
#pragma unroll
    for (int p = 0; p < 2; p++) {
        if ((a & 1) == p) {
#pragma unroll
            for (int z = 0; z < TN; z++) lds[((a >> 1) * TN + b) * TP + z] = w[bitrev5(z)];
        }
        __syncthreads();
        if ((a & 1) == p) {
            const int my = (TN - a) & (TN - 1), mx = (TN - b) & (TN - 1);
            const c2* __restrict__ mline = lds + ((my >> 1) * TN + mx) * TP;
#pragma unroll
            for (int z = 0; z < TN; z++) {
                const c2 zm = mline[(TN - z) & (TN - 1)];
                const c2 zk = w[bitrev5(z)];
                const float rr = 0.5f * (zk.x + zm.x), ri = 0.5f * (zk.y - zm.y);
                const float tr = 0.5f * (zk.y + zm.y), ti = -0.5f * (zk.x - zm.x);
                w[bitrev5(z)] = mkc((rr * tr) + (ri * ti), (rr * ti) - (ri * tr));
            }
        }
        __syncthreads();
    }
    // ---- inverse z (natural-order input is a renaming of registers), LZ -> LY
    c2 t[TN];
#pragma unroll
    for (int k = 0; k < TN; k++) t[k] = w[bitrev5(k)];
    fft32<true>(t);
    c2 u[TN];
    // (the same two rounds of 16 x 16 blocks the other way round: writer y = a hands over its z-half `half ^ d`, reader z = a
    // receives its y-half `half ^ d`)
#pragma unroll
    for (int d = 0; d < 2; d++) {
        {   // writer (y = a): block (y-half = half, z-half = half ^ d) -> slot `half ^ d`
            c2* __restrict__ dst = lds + (((half ^ d) * 16 + zz) * 16) * TP + b;
            if ((half ^ d) == 0) {
#pragma unroll
                for (int z = 0; z < 16; z++) dst[z * TP] = t[bitrev5(z)];
            } else {
#pragma unroll
                for (int z = 0; z < 16; z++) dst[z * TP] = t[bitrev5(z + 16)];
            }
        }
        __syncthreads();
        {   // reader (z = a): block (y-half = half ^ d, z-half = half) -> slot `half`
            const c2* __restrict__ src = lds + ((half * 16) * 16 + zz) * TP + b;
            if ((half ^ d) == 0) {
#pragma unroll
                for (int yy = 0; yy < 16; yy++) u[yy] = src[yy * 16 * TP];
            } else {
#pragma unroll
                for (int yy = 0; yy < 16; yy++) u[yy + 16] = src[yy * 16 * TP];
            }
        }
        __syncthreads();
    }
    // ---- inverse y (thread (z = a, x = b)), LY -> LX
    fft32<true>(u);
    c2 q[TN];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        if (half == h) {
#pragma unroll
            for (int j = 0; j < TN; j++) lds[(zz * TN + j) * TP + b] = u[bitrev5(j)];
#if OC_FUSED32_WAVE_XY
            wave_lds_fence();
#else
        }
        __syncthreads();
        if (half == h) {
#endif
#pragma unroll
            for (int k = 0; k < TN; k++) q[k] = lds[(zz * TN + b) * TP + k];
        }
        __syncthreads();
    }
    // ---- inverse x (thread (z = a, y = b)): the correlation volume, real part
    fft32<true>(q);

    // ---- arg-max with "strict >, scanning from index 0" (src/oc_fftcc.cpp:391-400): the thread's 32 values sit at
    // linear indices (a*32 + b)*32 + x, ascending in x
    float best = -2.f;
    int bidx = 0;
#pragma unroll
    for (int x = 0; x < TN; x++) {
        const float val = q[bitrev5(x)].x;
        if (val > best) {
            best = val;
            bidx = (a * TN + b) * TN + x;
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
        for (int i = 1; i < kWaves; i++)
            if (red[i] > best || (red[i] == best && redi[i] < bidx)) {
                best = red[i];
                bidx = redi[i];
            }
        int du = bidx % TN, dv = (bidx / TN) % TN, dw = bidx / (TN * TN);  // src/oc_fftcc.cpp:401-403
        if (du > R) du -= TN;
        if (dv > R) dv -= TN;
        if (dw > R) dw -= TN;
        const float gu = poi[poi3d::U], gv = poi[poi3d::V], gw = poi[poi3d::W];
        poi[poi3d::U] = (float)du + gu;
        poi[poi3d::V] = (float)dv + gv;
        poi[poi3d::W] = (float)dw + gw;
        poi[poi3d::U0] = gu;
        poi[poi3d::V0] = gv;
        poi[poi3d::W0] = gw;
        poi[poi3d::ZNCC] = best / (sqrtf(norms[0] * norms[1]) * M);
    }
}

// the launch that does the work: workgroup -> POI (XCD-contiguous ranges of the visiting order)
__global__ __launch_bounds__(kThreads3, 4) void fftcc3d_fused32_r5_kernel(Fftcc3dParams P, float* __restrict__ pois, int stride_f,
                                                                      unsigned long long count, int xcd_chunk,
                                                                      unsigned char* __restrict__ needs_clamped) {
    unsigned long long idx = blockIdx.x;
    if (xcd_chunk > 0) idx = (unsigned long long)(blockIdx.x & 7u) * xcd_chunk + (blockIdx.x >> 3);
    if (idx >= count) return;
    if (P.perm) idx = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)P.perm[idx]);  // wave-uniform: the record's address stays in SGPRs
    fftcc3d_fused32_r5_poi<false>(P, pois, stride_f, idx, needs_clamped, reinterpret_cast<unsigned*>(needs_clamped + ((count + 3) & ~3ull)));
}

// the windows clamped at a volume border: a few persistent workgroups scan the flags the first launch left (normally none is set)
__global__ __launch_bounds__(kThreads3, 4) void fftcc3d_fused32_r5_clamped_kernel(Fftcc3dParams P, float* __restrict__ pois, int stride_f,
                                                                              unsigned long long count,
                                                                              unsigned char* __restrict__ needs_clamped) {
    // (the first launch raises the word behind the flags when ANY window was clamped -- normally none: nothing to scan)
    if (*reinterpret_cast<const unsigned*>(needs_clamped + ((count + 3) & ~3ull)) == 0u) return;
    for (unsigned long long idx = blockIdx.x; idx < count; idx += gridDim.x) {
        if (needs_clamped[idx]) {   // uniform over the workgroup
            fftcc3d_fused32_r5_poi<true>(P, pois, stride_f, idx, needs_clamped, nullptr);
            __syncthreads();        // the next POI reuses the tables and the tile
        }
    }
}

}  // namespace

// one flag per POI of the queue + one "any window clamped" word
size_t fftcc3d_fused_flag_bytes(size_t count) { return ((count + 3) & ~(size_t)3) + 4; }

hipError_t launch_fftcc3d_fused_r5(const Fftcc3dParams& p, float* pois, int stride_f, size_t count, bool xcd, unsigned char* needs_clamped,
                                hipStream_t stream) {
    if (count == 0) return hipSuccess;
    if (!fftcc3d_fused_supported(p.rx, p.ry, p.rz) || !needs_clamped) return hipErrorInvalidValue;
    const int chunk = xcd ? (int)((count + 7) / 8) : 0;
    const size_t grid = xcd ? (size_t)chunk * 8 : count;
    (void)hipGetLastError();  // drop stale errors of earlier, unrelated calls
    hipError_t err = hipMemsetAsync(needs_clamped + ((count + 3) & ~(size_t)3), 0, 4, stream);
    if (err != hipSuccess) return err;
    hipLaunchKernelGGL(fftcc3d_fused32_r5_kernel, dim3((unsigned)grid), dim3(kThreads3), 0, stream, p, pois, stride_f,
                       (unsigned long long)count, chunk, needs_clamped);
    err = hipGetLastError();
    if (err != hipSuccess) return err;   // (nothing to scan behind a launch that failed)
    const unsigned scan = (unsigned)(count < 256 ? count : 256);
    hipLaunchKernelGGL(fftcc3d_fused32_r5_clamped_kernel, dim3(scan), dim3(kThreads3), 0, stream, p, pois, stride_f,
                       (unsigned long long)count, needs_clamped);
    return hipGetLastError();
}

}  // namespace ochip
