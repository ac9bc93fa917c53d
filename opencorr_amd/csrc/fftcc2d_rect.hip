// fftcc2d_rect.hip -- FFTCC2D in ONE kernel for the RECTANGULAR windows (rx != ry, both radii 4 ... 32) that have no
// instantiation of the register-FFT template: everything outside the 42 pairs of fftcc2d_fusedr.hip.
//
// FFTCC2D's constructor takes two radii (src/oc_fftcc.cpp:36-43 plans any 2rx x 2ry window); 29 x 28 side pairs are too many
// instantiations of fftcc2d_fusedn_impl.h, so -- like fftcc3d_box.hip in 3D -- this kernel takes the two sides at run time and
// switches, per axis pass, to the in-register line transform of that length (fft_device.h fft_mixed<N>).  Three instantiations,
// because a kernel's register allocation is that of its longest line: both sides <= 32 (13 lengths, 121 VGPRs, two POIs per wave,
// one half-wave each), <= 48 (21 lengths, 197 VGPRs) and <= 64 (29 lengths, capped at 256 VGPRs = two waves per SIMD).
//   gather -> z = ref + i * tar (zero-mean) in an NR x (NC + 1) complex tile in LDS -> column pass (lane = column), line pass
//   (lane = line) -> R(k) = (Z(k) + conj Z(-k)) / 2, T(k) = (Z(k) - conj Z(-k)) / (2i), C = conj(R) T once per mirror pair,
//   stored at k and, conjugated, at -k (C(-k) = conj C(k) in every bit: fftcc3d_box.hip says why) -> inverse line pass,
//   inverse column pass -> arg-max with the first-max rule, wrap, ZNCC.
// A POI lives in ONE wave (or half of one): the passes are ordered by wave-level LDS fences, no workgroup barrier.
// The transform is NR x NC = 2rx lines of 2ry contiguous elements over the window buffer filled [row * 2rx + col] -- the shape
// the reference plans FFTW with (fftwf_plan_dft_r2c_2d(width, height), src/oc_fftcc.cpp:40-42, 204-221): for rx != ry the
// window's buffer re-cut into lines of 2ry, reproduced here as in fftcc2d_fusedn_impl.h and the rocFFT pipeline.
// Integer outputs (u, v) are the reference's; the float ZNCC differs from FFTW's in the last bits like any other FFT.
#include "oc_device.h"
#include "fft_device.h"
#include "oc_kernels.h"

namespace ochip {

bool fftcc2d_fusedr_supported(int rx, int ry);  // fftcc2d_fusedr.hip: the 42 pairs with a kernel of their own

namespace {

using namespace fftdev;

// the N elements base[0], base[stride], ... transformed in place (natural order in, natural order out)
template <bool INV, int N>
__device__ __forceinline__ void rect_line_fft(c2* __restrict__ base, int stride) {
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

// n is uniform over the wave: one branch per pass; lengths above MAXN are not compiled into the small instantiation
template <bool INV, int MAXN>
__device__ __forceinline__ void rect_line_any(int n, c2* __restrict__ base, int stride) {
#define OC_RECT_CASE(N)                                                    \
    case N:                                                                \
        if constexpr (N <= MAXN) rect_line_fft<INV, N>(base, stride);      \
        break;
    switch (n) {
        OC_RECT_CASE(8) OC_RECT_CASE(10) OC_RECT_CASE(12) OC_RECT_CASE(14) OC_RECT_CASE(16) OC_RECT_CASE(18) OC_RECT_CASE(20)
        OC_RECT_CASE(22) OC_RECT_CASE(24) OC_RECT_CASE(26) OC_RECT_CASE(28) OC_RECT_CASE(30) OC_RECT_CASE(32) OC_RECT_CASE(34)
        OC_RECT_CASE(36) OC_RECT_CASE(38) OC_RECT_CASE(40) OC_RECT_CASE(42) OC_RECT_CASE(44) OC_RECT_CASE(46) OC_RECT_CASE(48)
        OC_RECT_CASE(50) OC_RECT_CASE(52) OC_RECT_CASE(54) OC_RECT_CASE(56) OC_RECT_CASE(58) OC_RECT_CASE(60) OC_RECT_CASE(62)
        OC_RECT_CASE(64)
        default: break;
    }
#undef OC_RECT_CASE
}

// orders this wave's earlier LDS writes before its later LDS reads (data exchanged between lanes of ONE wave: the LDS executes
// a wave's instructions in issue order, so all that is needed is that the compiler keeps the order and waits for the writes)
__device__ __forceinline__ void rect_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int MAXN>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(2))) void fftcc2d_rect_kernel(Fftcc2dParams P, float* __restrict__ pois, int stride_f,
                                                             unsigned long long count, int xcd_chunk) {
    constexpr int PPW = MAXN <= 32 ? 2 : 1;  // POIs per wave
    constexpr int LANES = kWave / PPW;       // lanes per POI
    extern __shared__ c2 rect_lds[];
    const int NR = 2 * P.rx, NC = 2 * P.ry;  // the transform: NR lines of NC elements (see the header)
    const int NP = NC + 1;                   // tile pitch in complex elements (odd)
    const int M = NR * NC;
    const int lane = threadIdx.x & (kWave - 1);
    const int q = PPW == 2 ? lane >> 5 : 0, l = lane & (LANES - 1);
    unsigned long long grp = blockIdx.x;
    if (xcd_chunk > 0) grp = (unsigned long long)(blockIdx.x & 7u) * xcd_chunk + (blockIdx.x >> 3);
    const unsigned long long idx = grp * PPW + q;
    if (idx >= count) return;
    c2* __restrict__ tile = rect_lds + q * (NR * NP);
    float* poi = pois + idx * (unsigned long long)stride_f;
    const float px = poi[poi2d::X], py = poi[poi2d::Y];
    const float gu = poi[poi2d::U], gv = poi[poi2d::V];
    const int rx = P.rx, ry = P.ry;
    const int width = P.width, height = P.height;

    // bounds guard: the reference returns silently and leaves the POI untouched (src/oc_fftcc.cpp:190-196)
    if ((int)px < rx || (int)px >= width - rx || (int)py < ry || (int)py >= height - ry || (int)(px + gu) < rx ||
        (int)(px + gu) >= width - rx || (int)(py + gv) < ry || (int)(py + gv) >= height - ry)
        return;

    // lines (length NC) are owned by lanes < NR, columns (length NR) by lanes < NC
    const bool act_l = l < NR, act_c = l < NC;
    auto lanes_sum = [](float x) {
#pragma unroll
        for (int off = 1; off < LANES; off <<= 1) x += __shfl_xor(x, off, kWave);
        return x;
    };

    // ---- window fill, means, zero-mean, sums of squares (src/oc_fftcc.cpp:198-231): element (a, col) of the transform's
    // array is sample s = a * NC + col of the window buffer, i.e. window row s / NR, column s % NR
    float rs = 0.f, ts = 0.f;
    if (act_c) {
        int r = l / NR, c = l - r * NR;
        const int qstep = NC / NR, mstep = NC - qstep * NR;
        for (int a = 0; a < NR; a++) {
            const float rxp = px + c - rx, ryp = py + r - ry;
            const float txp = rxp + gu, typ = ryp + gv;
            const float rv = P.ref[(size_t)(int)ryp * width + (int)rxp], tv = P.tar[(size_t)(int)typ * width + (int)txp];
            tile[a * NP + l] = mkc(rv, tv);
            rs += rv;
            ts += tv;
            c += mstep;   // s += NC
            r += qstep;
            if (c >= NR) {
                c -= NR;
                r++;
            }
        }
    }
    rs = lanes_sum(rs);
    ts = lanes_sum(ts);
    float rn = 0.f, tn = 0.f;
    if (act_c) {
        const c2 mean = mkc(rs / M, ts / M);
        for (int a = 0; a < NR; a++) {
            const c2 d = tile[a * NP + l] - mean;   // (the element this lane wrote itself)
            tile[a * NP + l] = d;
            rn += d.x * d.x;
            tn += d.y * d.y;
        }
    }
    rn = lanes_sum(rn);
    tn = lanes_sum(tn);

    // ---- forward: columns (lane = column, the elements it filled), then lines (lane = line).  ONE loop body, so that the
    // switch over the line lengths exists once per direction
#pragma unroll 1
    for (int pass = 0; pass < 2; pass++) {
        const bool cols = pass == 0;
        if (cols ? act_c : act_l) rect_line_any<false, MAXN>(cols ? NR : NC, cols ? tile + l : tile + l * NP, cols ? NP : 1);
        rect_lds_fence();
    }
    // ---- spectra of the two real windows and their product conj(R) * T (src/oc_fftcc.cpp:236-241), once per mirror pair
    if (act_l) {
        const int ml = l ? NR - l : 0;
        c2* own = tile + l * NP;   // (the two lines coincide for l = 0 and l = NR / 2: no __restrict__)
        c2* mir = tile + ml * NP;
        for (int k = 0; k < NC; k++) {
            const int mk = k ? NC - k : 0;
            // the pair's owner: the smaller linear index (line, k); a self-mirrored bin is its own pair
            if (l < ml || (l == ml && k <= mk)) {
                const c2 zk = own[k], zm = mir[mk];
                const float rr = 0.5f * (zk.x + zm.x), ri = 0.5f * (zk.y - zm.y);
                const float tr = 0.5f * (zk.y + zm.y), ti = -0.5f * (zk.x - zm.x);
                const float cr = (rr * tr) + (ri * ti), ci = (rr * ti) - (ri * tr);
                own[k] = mkc(cr, ci);
                if (l != ml || k != mk) mir[mk] = mkc(cr, -ci);
            }
        }
    }
    rect_lds_fence();
    // ---- inverse (unnormalised, like FFTW's c2r): lines, then columns
#pragma unroll 1
    for (int pass = 0; pass < 2; pass++) {
        const bool cols = pass == 1;
        if (cols ? act_c : act_l) rect_line_any<true, MAXN>(cols ? NR : NC, cols ? tile + l : tile + l * NP, cols ? NP : 1);
        rect_lds_fence();
    }

    // ---- arg-max with "strict >, scanning from index 0" (src/oc_fftcc.cpp:246-255): the lane's NR surface values sit at
    // linear indices a * NC + col, ascending in a; then the POI's lanes, the lower index winning a tie
    float best = -2.f;
    int bidx = 0x7fffffff;
    if (act_c) {
        for (int a = 0; a < NR; a++) {
            const float val = tile[a * NP + l].x;
            if (val > best) {
                best = val;
                bidx = a * NC + l;
            }
        }
        if (bidx == 0x7fffffff) bidx = l;  // nothing above -2 (NaN surface): the reference keeps index 0 semantics
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
        poi[poi2d::ZNCC] = best / (sqrtf(rn * tn) * M);
    }
}

template <int MAXN>
hipError_t launch_rect(const Fftcc2dParams& p, float* pois, int stride_f, size_t count, bool xcd, hipStream_t stream) {
    constexpr int PPW = MAXN <= 32 ? 2 : 1;
    const size_t groups = (count + PPW - 1) / PPW;
    const int chunk = xcd ? (int)((groups + 7) / 8) : 0;
    const size_t grid = xcd ? (size_t)chunk * 8 : groups;
    const size_t lds = (size_t)PPW * (2 * p.rx) * (2 * p.ry + 1) * sizeof(c2);   // <= 33 KB
    (void)hipGetLastError();  // drop stale errors of earlier, unrelated calls
    hipLaunchKernelGGL((fftcc2d_rect_kernel<MAXN>), dim3((unsigned)grid), dim3(kWave), lds, stream, p, pois, stride_f,
                       (unsigned long long)count, chunk);
    return hipGetLastError();
}

}  // namespace

// rectangular windows with both radii in 4 ... 32 that fftcc2d_fusedr.hip has no instantiation for
bool fftcc2d_rect_supported(int rx, int ry) {
    return rx != ry && rx >= 4 && rx <= 32 && ry >= 4 && ry <= 32 && !fftcc2d_fusedr_supported(rx, ry);
}

hipError_t launch_fftcc2d_rect(const Fftcc2dParams& p, float* pois, int stride_f, size_t count, bool xcd, hipStream_t stream) {
    if (count == 0) return hipSuccess;
    if (!fftcc2d_rect_supported(p.rx, p.ry)) return hipErrorInvalidValue;
    if (p.rx <= 16 && p.ry <= 16) return launch_rect<32>(p, pois, stride_f, count, xcd, stream);
    if (p.rx <= 24 && p.ry <= 24) return launch_rect<48>(p, pois, stride_f, count, xcd, stream);
    return launch_rect<64>(p, pois, stride_f, count, xcd, stream);
}

}  // namespace ochip
