// fftcc3d_box.hip -- FFTCC3D in ONE kernel for NON-CUBIC windows: every even side 8 ... 32 per axis (radii 4 ... 16, chosen
// per axis) whose complex volume fits the LDS.
//
// FFTCC3D's constructor takes three radii (src/oc_fftcc.cpp:48-74 plans any 2rx x 2ry x 2rz window).  The cubic windows have
// kernels of their own (fftcc3d_fusedn.hip, fftcc3d_fused.hip, fftcc3d_planes.hip: window side = template argument, lines in
// registers from pass to pass); until round 5 every other shape went through the five-kernel rocFFT pipeline of fftcc3d.hip,
// which moves both windows, both spectra, the product and the correlation volume through HBM.  13^3 shapes are too many
// instantiations of a kernel template, so this ONE kernel takes the three sides at run time and switches, per axis pass, to
// the line transform of that length (fft_device.h fft_mixed<N>, N = 8, 10, ... 32: 26 bodies, forward and inverse):
//   gather -> z = ref + i * tar (zero-mean) in LDS [n0][n1][n2 + 1] ->
//   line passes along n2, n1, n0 in place (one line per thread and pass) ->
//   R(k) = (Z(k) + conj Z(-k)) / 2, T(k) = (Z(k) - conj Z(-k)) / (2i), C = conj(R) T, formed ONCE per mirror pair by the
//   thread that owns the smaller linear index and stored at k and (conjugated: C(-k) = conj C(k) in every bit, see below)
//   at -k -> inverse passes along n0, n1, n2 -> arg-max with the first-max rule, wrap, ZNCC.
// WHICH transform: the reference fills its buffers [(i * 2ry + j) * 2rx + k] (x fastest) but plans
// fftwf_plan_dft_r2c_3d(2rx, 2ry, 2rz) (src/oc_fftcc.cpp:68-70, 349-360) -- FFTW's LAST length is the fastest one -- so what it
// transforms is the same M floats READ AS an array [n0 = 2rx][n1 = 2ry][n2 = 2rz].  For a cube that is the window itself; for
// any other shape it is a reshaped window, the peak is the peak of THAT correlation, and its buffer position is decoded as a
// window position again (:401-403).  The drop-in reproduces this (like the oracle and the rocFFT pipeline, capi.hip
// ensure_fft): thread (a, b) gathers buffer positions (a * n1 + b) * n2 ... + n2 - 1, whatever window voxels they are.
// Every pass is LDS -> registers -> LDS (the cubic kernels keep the line in registers between gather / x pass and between
// z pass / product / inverse z: two LDS round trips fewer); that is the price of one kernel for all shapes.
// C(-k) = conj C(k) exactly: with zk = Z(k), zm = Z(-k) the four half-sums of -k are (rr, -ri, tr, -ti) of k's -- a + b is
// commutative and negation exact -- so (rr tr + ri ti, -(rr ti - ri tr)) are the very products and sums of k, one sign flipped.
// Integer outputs (u, v, w) are the reference's; the float ZNCC differs from FFTW's in the last bits like any other FFT
// (tests: identical integers against the oracle and the rocFFT pipeline, ZNCC within 1e-4 / 2e-5).
#include "oc_device.h"
#include "fft_device.h"
#include "oc_kernels.h"

#include <atomic>

namespace ochip {

namespace {

using namespace fftdev;

constexpr int kBoxMaxSide = 32;
constexpr int kBoxMaxThreads = kBoxMaxSide * kBoxMaxSide;
constexpr int kBoxMaxWaves = kBoxMaxThreads / kWave;
constexpr size_t kBoxLdsLimit = 160 * 1024;

typedef float float4u __attribute__((ext_vector_type(4), aligned(4)));  // 4-byte aligned 16-byte load

__device__ __forceinline__ int clampi3b(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// the N elements base[0], base[stride], ... transformed in place (natural order in, natural order out)
template <bool INV, int N>
__device__ __forceinline__ void line_fft(c2* __restrict__ base, int stride) {
    c2 v[N];
    static_for<0, N>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        v[k] = base[k * stride];
    });
    fft_mixed<INV, N>(v);
    static_for<0, N>([&](auto kc) {
        constexpr int k = decltype(kc)::value, p = fft_pos(N, k);  // (constexpr: a run-time fft_pos() sends v[] to scratch)
        base[k * stride] = v[p];
    });
}

// n is uniform over the workgroup: one branch per pass
template <bool INV>
__device__ __forceinline__ void line_fft_any(int n, c2* __restrict__ base, int stride) {
    switch (n) {
        case 8: line_fft<INV, 8>(base, stride); break;
        case 10: line_fft<INV, 10>(base, stride); break;
        case 12: line_fft<INV, 12>(base, stride); break;
        case 14: line_fft<INV, 14>(base, stride); break;
        case 16: line_fft<INV, 16>(base, stride); break;
        case 18: line_fft<INV, 18>(base, stride); break;
        case 20: line_fft<INV, 20>(base, stride); break;
        case 22: line_fft<INV, 22>(base, stride); break;
        case 24: line_fft<INV, 24>(base, stride); break;
        case 26: line_fft<INV, 26>(base, stride); break;
        case 28: line_fft<INV, 28>(base, stride); break;
        case 30: line_fft<INV, 30>(base, stride); break;
        case 32: line_fft<INV, 32>(base, stride); break;
        default: break;
    }
}

// two block-wide sums at once (threads without a line contribute zeros); every thread returns the same values
__device__ __forceinline__ void block_sum2b(float& x, float& y, float* red, int lane, int wave, int waves) {
    x = wave_allreduce_sum(x);
    y = wave_allreduce_sum(y);
    __syncthreads();
    if (lane == 0) {
        red[wave] = x;
        red[kBoxMaxWaves + wave] = y;
    }
    __syncthreads();
    float sx = 0.f, sy = 0.f;
    for (int i = 0; i < waves; i++) {
        sx += red[i];
        sy += red[kBoxMaxWaves + i];
    }
    x = sx;
    y = sy;
}

// one axis pass: line l = (a, b) with a = l / bdim, b = l - a * bdim starts at vol[a * amul + b * bmul]
template <bool INV>
__device__ __forceinline__ void axis_pass(c2* __restrict__ vol, int tid, int lines, int bdim, int amul, int bmul, int n, int stride) {
    if (tid < lines) {
        const int a = tid / bdim, b = tid - a * bdim;
        line_fft_any<INV>(n, vol + a * amul + b * bmul, stride);
    }
}

__global__ __launch_bounds__(kBoxMaxThreads) void fftcc3d_box_kernel(Fftcc3dParams P, float* __restrict__ pois, int stride_f,
                                                                    unsigned long long count, int xcd_chunk) {
    extern __shared__ c2 box_lds[];
    const int nx = 2 * P.rx, ny = 2 * P.ry, nz = 2 * P.rz;   // the WINDOW: filled [(i * ny + j) * nx + k], x fastest
    const int fx = nz, fy = ny, fz = nx;                     // the TRANSFORM: n2 = 2rz fastest, n0 = 2rx slowest (see the header)
    const int NP = fx + 1;                 // row pitch in complex elements (odd: the line-wise accesses of every pass spread over the banks)
    const int plane = fy * NP;
    c2* __restrict__ vol = box_lds;        // [fz][fy][NP]
    int* tab = reinterpret_cast<int*>(vol + (size_t)fz * plane);   // [6][kBoxMaxSide]: voxel index of window coordinate k (ref x, y, z, tar x, y, z)
    float* red = reinterpret_cast<float*>(tab + 6 * kBoxMaxSide);  // [2 * kBoxMaxWaves]
    int* redi = reinterpret_cast<int*>(red + 2 * kBoxMaxWaves);    // [kBoxMaxWaves]
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1), wave = tid >> 6, waves = blockDim.x >> 6;
    const int M = nx * ny * nz;
    unsigned long long idx = blockIdx.x;
    if (xcd_chunk > 0) idx = (unsigned long long)(blockIdx.x & 7u) * xcd_chunk + (blockIdx.x >> 3);
    if (idx >= count) return;
    if (P.perm) idx = P.perm[idx];
    float* poi = pois + idx * (unsigned long long)stride_f;

    // ---- window coordinates -> voxel indices (src/oc_fftcc.cpp:349-358: Point3D(poi->x + k - rx, ...) truncated, the
    // target window displaced by the initial guess); separable: one table per axis and window.  The reference has no
    // bounds guard in 3D; indices are clamped like in fftcc3d_gather_kernel.
    for (int e = tid; e < 6 * kBoxMaxSide; e += blockDim.x) {
        const int axis = e / kBoxMaxSide, k = e - axis * kBoxMaxSide, which = axis % 3;
        const int n = which == 0 ? nx : which == 1 ? ny : nz;
        if (k < n) {
            const float p = poi[which == 0 ? poi3d::X : which == 1 ? poi3d::Y : poi3d::Z];
            const float g = poi[which == 0 ? poi3d::U : which == 1 ? poi3d::V : poi3d::W];
            const int D = which == 0 ? P.dx : which == 1 ? P.dy : P.dz;
            float c = p + k - (n >> 1);
            if (axis >= 3) c = c + g;
            tab[e] = clampi3b((int)c, 0, D - 1);
        }
    }
    __syncthreads();

    // ---- gather (round 6): buffer position s holds window voxel (k, j, i) = (s % nx, (s / nx) % ny, s / (nx * ny)) and is element
    // (c, b, a) = (s % fx, (s / fx) % fy, s / (fx * fy)) of the transform's array; z = ref + i * tar.  Thread t takes the positions
    // t, t + T, t + 2T, ...: neighbouring lanes read neighbouring voxels of a window row, so a load instruction covers whole rows.
    // (Until then thread (a, b) filled ITS line, fx consecutive positions: a wave's loads were fx floats apart, a cache line per
    // lane -- what that costs is measured in fftcc3d_fused.hip.)  The divisions are multiplications by reciprocals that are exact
    // for s < 2^16 (the volume fits the LDS: M <= 20 480).
    const int xlines = fz * fy;
    const bool has_xline = tid < xlines;
    const int za = has_xline ? tid / fy : 0, yb = has_xline ? tid - za * fy : 0;
    c2* __restrict__ xrow = vol + za * plane + yb * NP;   // the thread's line of the fast axis (first and last pass)
    auto recip = [](int d) { return (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); };   // ceil(2^32 / d)
    const unsigned inv_nx = recip(nx), inv_ny = recip(ny), inv_fx = recip(fx), inv_fy = recip(fy);
    auto divmod = [](unsigned s, unsigned inv, int d, int& q, int& r) {
        q = (int)__umulhi(s, inv);
        r = (int)s - q * d;
    };
    float rs = 0.f, ts = 0.f;
    {
        const size_t pitch = (size_t)P.dx;
        for (int s0 = tid; s0 < M; s0 += (int)blockDim.x) {
            int k, rem, j, i, c, ab, bb, aa;
            divmod((unsigned)s0, inv_nx, nx, rem, k);
            divmod((unsigned)rem, inv_ny, ny, i, j);
            divmod((unsigned)s0, inv_fx, fx, ab, c);
            divmod((unsigned)ab, inv_fy, fy, aa, bb);
            const float r = P.ref[((size_t)tab[2 * kBoxMaxSide + i] * P.dy + tab[1 * kBoxMaxSide + j]) * pitch + tab[k]];
            const float t = P.tar[((size_t)tab[5 * kBoxMaxSide + i] * P.dy + tab[4 * kBoxMaxSide + j]) * pitch + tab[3 * kBoxMaxSide + k]];
            vol[aa * plane + bb * NP + c] = mkc(r, t);
            rs += r;
            ts += t;
        }
    }
    // means, zero-mean, sums of squares (src/oc_fftcc.cpp:360-376); every thread re-reads the elements it wrote itself
    block_sum2b(rs, ts, red, lane, wave, waves);
    float rn = 0.f, tn = 0.f;
    {
        const c2 mean = mkc(rs / M, ts / M);
        for (int s0 = tid; s0 < M; s0 += (int)blockDim.x) {
            int c, ab, bb, aa;
            divmod((unsigned)s0, inv_fx, fx, ab, c);
            divmod((unsigned)ab, inv_fy, fy, aa, bb);
            c2* e = vol + aa * plane + bb * NP + c;
            const c2 d = *e - mean;
            *e = d;
            rn += d.x * d.x;
            tn += d.y * d.y;
        }
    }
    block_sum2b(rn, tn, red, lane, wave, waves);   // (its barriers also publish the volume to the line owners of the first pass)

    // ---- forward passes along the fast axis (thread (a, b): the line it gathered), the middle axis (thread (a, c)) and the slow
    // axis (thread (b, c)), each in place
    axis_pass<false>(vol, tid, xlines, fy, plane, NP, fx, 1);
    __syncthreads();
    axis_pass<false>(vol, tid, fz * fx, fx, plane, 1, fy, NP);
    __syncthreads();
    axis_pass<false>(vol, tid, fy * fx, fx, NP, 1, fz, plane);
    __syncthreads();
    // ---- spectra of the two real windows and their product conj(R) * T (src/oc_fftcc.cpp:378-386), once per mirror pair
    if (tid < fy * fx) {
        const int ky = tid / fx, kx = tid - ky * fx;
        const int my = ky ? fy - ky : 0, mx = kx ? fx - kx : 0;
        c2* own = vol + ky * NP + kx;   // (the two lines coincide for self-mirrored (ky, kx): no __restrict__)
        c2* mir = vol + my * NP + mx;
        const int lin = ky * fx + kx, mlin = my * fx + mx;
        for (int kz = 0; kz < fz; kz++) {
            const int mz = kz ? fz - kz : 0;
            // the pair's owner: the smaller linear index (kz, ky, kx); a self-mirrored bin is its own pair
            if (kz < mz || (kz == mz && lin <= mlin)) {
                const c2 zk = own[kz * plane], zm = mir[mz * plane];
                const float rr = 0.5f * (zk.x + zm.x), ri = 0.5f * (zk.y - zm.y);
                const float tr = 0.5f * (zk.y + zm.y), ti = -0.5f * (zk.x - zm.x);
                const float cr = (rr * tr) + (ri * ti), ci = (rr * ti) - (ri * tr);
                own[kz * plane] = mkc(cr, ci);
                if (kz != mz || lin != mlin) mir[mz * plane] = mkc(cr, -ci);
            }
        }
    }
    __syncthreads();
    // ---- inverse passes, slow to fast (unnormalised, like FFTW's c2r)
    axis_pass<true>(vol, tid, fy * fx, fx, NP, 1, fz, plane);
    __syncthreads();
    axis_pass<true>(vol, tid, fz * fx, fx, plane, 1, fy, NP);
    __syncthreads();
    axis_pass<true>(vol, tid, xlines, fy, plane, NP, fx, 1);
    // ---- arg-max with "strict >, scanning from index 0" (src/oc_fftcc.cpp:391-400): thread (a, b) scans the fast line it has
    // just transformed, buffer positions tid * fx + c ascending in c
    float best = -2.f;
    int bidx = 0x7fffffff;
    if (has_xline) {
        bidx = tid * fx;
        for (int c = 0; c < fx; c++) {
            const float val = xrow[c].x;
            if (val > best) {
                best = val;
                bidx = tid * fx + c;
            }
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
        for (int i = 1; i < waves; i++)
            if (red[i] > best || (red[i] == best && redi[i] < bidx)) {
                best = red[i];
                bidx = redi[i];
            }
        int du = bidx % nx, dv = (bidx / nx) % ny, dw = bidx / (nx * ny);  // src/oc_fftcc.cpp:401-403: decoded as a WINDOW position
        if (du > P.rx) du -= nx;
        if (dv > P.ry) dv -= ny;
        if (dw > P.rz) dw -= nz;
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

size_t box_lds_bytes(int rx, int ry, int rz) {
    const size_t nx = 2 * (size_t)rx, ny = 2 * (size_t)ry, nz = 2 * (size_t)rz;
    return nx * ny * (nz + 1) * sizeof(c2) +   // [n0 = 2rx][n1 = 2ry][2rz + 1]
           6 * kBoxMaxSide * sizeof(int) + 2 * kBoxMaxWaves * sizeof(float) + kBoxMaxWaves * sizeof(int);
}

}  // namespace

// non-cubic windows with every radius in 4 ... 16 whose complex volume fits the LDS (the cubes have kernels of their own)
bool fftcc3d_box_supported(int rx, int ry, int rz) {
    const auto ok = [](int r) { return r >= 4 && r <= kBoxMaxSide / 2; };
    if (!ok(rx) || !ok(ry) || !ok(rz)) return false;
    if (rx == ry && ry == rz) return false;
    return box_lds_bytes(rx, ry, rz) <= kBoxLdsLimit;
}

hipError_t launch_fftcc3d_box(const Fftcc3dParams& p, float* pois, int stride_f, size_t count, bool xcd, hipStream_t stream) {
    if (count == 0) return hipSuccess;
    if (!fftcc3d_box_supported(p.rx, p.ry, p.rz)) return hipErrorInvalidValue;
    const int nx = 2 * p.rx, ny = 2 * p.ry, nz = 2 * p.rz;
    int lines = nz * ny;   // (lines per pass: the products of two sides, whichever axis is the fast one)
    if (nz * nx > lines) lines = nz * nx;
    if (ny * nx > lines) lines = ny * nx;
    const int block = (lines + kWave - 1) / kWave * kWave;
    const size_t lds = box_lds_bytes(p.rx, p.ry, p.rz);
    // more than 64 KB of dynamic LDS has to be asked for, once per device (benign race: every caller sets the same value)
    static std::atomic<unsigned long long> attr_set{0};
    int dev = 0;
    if (hipError_t err = hipGetDevice(&dev); err != hipSuccess) return err;
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(attr_set.load(std::memory_order_relaxed) & bit)) {
        hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(fftcc3d_box_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)kBoxLdsLimit);
        if (err != hipSuccess) return err;
        attr_set.fetch_or(bit, std::memory_order_relaxed);
    }
    const int chunk = xcd ? (int)((count + 7) / 8) : 0;
    const size_t grid = xcd ? (size_t)chunk * 8 : count;
    (void)hipGetLastError();  // drop stale errors of earlier, unrelated calls
    hipLaunchKernelGGL(fftcc3d_box_kernel, dim3((unsigned)grid), dim3(block), lds, stream, p, pois, stride_f, (unsigned long long)count, chunk);
    return hipGetLastError();
}

}  // namespace ochip
