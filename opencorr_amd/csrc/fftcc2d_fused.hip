// fftcc2d_fused.hip -- FFTCC2D for 32 x 32 windows (subset radius 16) in ONE kernel.
//
// FFTCC2D::compute(POI2D*) (src/oc_fftcc.cpp:177-275) needs, per POI, two 2-D real FFTs, a
// spectrum product and one inverse FFT of a 4 KB window.  The rocFFT pipeline of fftcc2d.hip
// moves ~50 KB per POI through HBM between five kernels; here the whole POI stays on chip:
//   gather  ->  z = ref + i*tar  ->  ONE complex 32x32 FFT in LDS/registers  ->
//   R(k) = (Z(k) + conj Z(-k))/2, T(k) = (Z(k) - conj Z(-k))/(2i), C = conj(R) T  ->
//   inverse complex FFT (unnormalised, like FFTW's c2r)  ->  arg-max with the first-max
//   rule, wrap, ZNCC.
// Two kernels live here:
//   * fftcc2d_fused32x2_kernel (round 3, what the engine launches): two POIs per wave, one lane per window line -- see
//     the comment above it;
//   * fftcc2d_fused32_kernel (rounds 1 - 2, kept as the A/B partner behind -DOC_FFTCC2D_X2=0): one wave per POI; a
//     length-32 transform is one decimation-in-frequency radix-2 step (done while reading the row / column from LDS, each
//     of the two lanes that share a row taking the even or the odd outputs) followed by a 16-point FFT held in
//     registers; gather with the sample -> lane ownership of fftcc2d_gather_kernel (means and norms bit-identical to the
//     rocFFT pipeline's); LDS: 32 rows x 33 complex (pitch 33 keeps both the row and the column accesses conflict-free),
//     8448 B per wave.
// Integer outputs (u, v) are what the reference computes; the float ZNCC differs from
// FFTW's in the last bits like any other FFT implementation (tested to 1e-5 against the
// oracle's double-precision DFT).
#include "dic2d_device.h"
#include "fft_device.h"
#include "oc_kernels.h"

namespace ochip {

namespace {

using namespace fftdev;

constexpr int FN = 32;               // window side (2 * radius)
constexpr int FP = FN + 1;           // LDS row pitch in complex elements
constexpr int FWAVE_LDS = FN * FP;   // float2 elements per wave
// OC_FFTCC2D_X2 = 1 (the default since round 3) launches the two-POIs-per-wave kernel further down instead of this one.
// The round-1 kernel is an A/B partner: it is COMPILED only with -DOC_FFTCC2D_X2=0 (tools/ab_build.py), the library that
// ships does not contain it (round 5).
#ifndef OC_FFTCC2D_X2
#define OC_FFTCC2D_X2 1
#endif
#if !OC_FFTCC2D_X2
#ifndef OC_FFTCC2D_WAVES
#define OC_FFTCC2D_WAVES 4
#endif
constexpr int kFusedWaves = OC_FFTCC2D_WAVES;  // POIs (waves) per workgroup

// One length-32 transform along a line of the LDS tile.  The lane reads elements n and n+16
// (n = 0..15) at `line + n*step`, forms its half of the first radix-2 stage (h = 0: sums ->
// even outputs, h = 1: twiddled differences -> odd outputs) and finishes with fft16; output
// k of the lane is X[2k + h], left in v[bitrev4(k)].
template <bool INV>
__device__ __forceinline__ void fft32_line(const c2* line, int step, int h, c2 (&v)[16]) {
#pragma clang fp contract(fast)
    const float sgn = h ? -1.f : 1.f;
#pragma unroll
    for (int n = 0; n < 16; n++) {
        const c2 x0 = line[n * step], x1 = line[(n + 16) * step];
        const c2 d = x0 + sgn * x1;
        const float c = h ? kCos32[n] : 1.f, s = h ? kSin32[n] : 0.f;
        v[n] = (n == 0) ? d : cmul_tw<INV>(d, c, s);
    }
    fft16<INV>(v);
}
#endif  // !OC_FFTCC2D_X2

}  // namespace

#if !OC_FFTCC2D_X2
__global__ __launch_bounds__(64 * kFusedWaves) void fftcc2d_fused32_kernel(Fftcc2dParams P, float* __restrict__ pois,
                                                                           int stride_f, unsigned long long count,
                                                                           int xcd_chunk) {
    __shared__ c2 lds[kFusedWaves * FWAVE_LDS];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned long long grp = blockIdx.x;
    if (xcd_chunk > 0) grp = (unsigned long long)(blockIdx.x & 7u) * xcd_chunk + (blockIdx.x >> 3);
    const unsigned long long idx = grp * kFusedWaves + wave;
    if (idx >= count) return;
    c2* buf = lds + wave * FWAVE_LDS;
    float* poi = pois + idx * (unsigned long long)stride_f;
    const float px = poi[poi2d::X], py = poi[poi2d::Y];
    const float gu = poi[poi2d::U], gv = poi[poi2d::V];
    const int rx = FN / 2, ry = FN / 2, width = P.width, height = P.height;
    constexpr int M = FN * FN;

    // bounds guard: the reference returns silently and leaves the POI untouched (src/oc_fftcc.cpp:190-196)
    if ((int)px < rx || (int)px >= width - rx || (int)py < ry || (int)py >= height - ry || (int)(px + gu) < rx ||
        (int)(px + gu) >= width - rx || (int)(py + gv) < ry || (int)(py + gv) >= height - ry)
        return;

    // ---- window fill, means, zero-mean, sums of squares (src/oc_fftcc.cpp:198-231); sample
    // s = r*32 + c is owned by lane (s mod 64), exactly like fftcc2d_gather_kernel
    float rn, tn;
    {
        // buffer-resource loads: 32-bit offsets (the guard above keeps both windows inside the images)
        const __amdgpu_buffer_rsrc_t r_ref = make_rsrc(P.ref), r_tar = make_rsrc(P.tar);
        float a[16], b[16];
        float rsum = 0.f, tsum = 0.f;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int s = lane + kWave * k;
            const int r = s >> 5, c = s & 31;
            const float rxp = px + c - rx, ryp = py + r - ry;
            a[k] = buf_f32(r_ref, (__umul24((unsigned)(int)ryp, (unsigned)width) + (unsigned)(int)rxp) << 2, 0);
            const float txp = rxp + gu, typ = ryp + gv;
            b[k] = buf_f32(r_tar, (__umul24((unsigned)(int)typ, (unsigned)width) + (unsigned)(int)txp) << 2, 0);
        }
#pragma unroll
        for (int k = 0; k < 16; k++) {
            rsum += a[k];
            tsum += b[k];
        }
        const float rmean = wave_allreduce_sum(rsum) / M;
        const float tmean = wave_allreduce_sum(tsum) / M;
        rn = 0.f;
        tn = 0.f;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int s = lane + kWave * k;
            const float x = a[k] - rmean, y = b[k] - tmean;
            rn += x * x;
            tn += y * y;
            buf[(s >> 5) * FP + (s & 31)] = mkc(x, y);
        }
        rn = wave_allreduce_sum(rn);
        tn = wave_allreduce_sum(tn);
    }
    __builtin_amdgcn_wave_barrier();

    const int line = lane & 31, h = lane >> 5;
    c2 v[16];
    // ---- forward rows: lane (y, h) -> Z1[y][2k + h]
    fft32_line<false>(buf + line * FP, 1, h, v);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 16; k++) buf[line * FP + 2 * k + h] = v[bitrev4(k)];
    __builtin_amdgcn_wave_barrier();
    // ---- forward columns: lane (x, h) -> Z[2k + h][x]
    fft32_line<false>(buf + line, FP, h, v);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 16; k++) buf[(2 * k + h) * FP + line] = v[bitrev4(k)];
    __builtin_amdgcn_wave_barrier();
    // ---- spectra of the two real windows and their product conj(R) * T (src/oc_fftcc.cpp:236-241)
    {
        const int mx = (FN - line) & (FN - 1);
        c2 zm[16];
#pragma unroll
        for (int k = 0; k < 16; k++) zm[k] = buf[((FN - (2 * k + h)) & (FN - 1)) * FP + mx];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const c2 z = v[bitrev4(k)];
            const float rr = 0.5f * (z.x + zm[k].x), ri = 0.5f * (z.y - zm[k].y);
            const float tr = 0.5f * (z.y + zm[k].y), ti = -0.5f * (z.x - zm[k].x);
            buf[(2 * k + h) * FP + line] = mkc((rr * tr) + (ri * ti), (rr * ti) - (ri * tr));
        }
    }
    __builtin_amdgcn_wave_barrier();
    // ---- inverse rows, inverse columns (unnormalised)
    fft32_line<true>(buf + line * FP, 1, h, v);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 16; k++) buf[line * FP + 2 * k + h] = v[bitrev4(k)];
    __builtin_amdgcn_wave_barrier();
    fft32_line<true>(buf + line, FP, h, v);

    // ---- arg-max with "strict >, scanning from index 0" (src/oc_fftcc.cpp:246-255): the lane's
    // 16 surface values sit at linear indices (2k + h)*32 + x, ascending in k
    float best = -2.f;
    int bidx = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const float val = v[bitrev4(k)].x;
        if (val > best) { best = val; bidx = (2 * k + h) * FN + line; }
    }
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const float ov = __shfl_xor(best, off, kWave);
        const int oi = __shfl_xor(bidx, off, kWave);
        if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
    }
    if (lane == 0) {
        int du = bidx % FN, dv = bidx / FN;
        if (du > rx) du -= FN;
        if (dv > ry) dv -= FN;
        poi[poi2d::U] = (float)du + gu;
        poi[poi2d::V] = (float)dv + gv;
        poi[poi2d::U0] = gu;
        poi[poi2d::V0] = gv;
        poi[poi2d::ZNCC] = best / (sqrtf(rn * tn) * M);
    }
}
#endif  // !OC_FFTCC2D_X2

// ---------------------------------------------------------------------------------------------------------------
// Second mapping (round 3): TWO POIs per wave, one lane per LINE.  Lane (q, l) of half-wave q gathers column l of its POI's
// two windows -- row k of the window is one coalesced 128-byte read of the half-wave -- so z = ref + i*tar arrives in the
// registers already in column ownership and the first axis transform needs no LDS at all; every later pass is a whole
// 32-point FFT in a lane's registers (fft32, shared with the 3D kernel) with ONE transposition through the LDS tile
// between passes.  Per POI: 3 x 32 x 32 element writes and as many reads, against 5 x 32 x 32 writes and 9 x 32 x 32
// reads of the (line, half) mapping above, whose two lanes per line both read the whole line; the radix-2 stage is done
// once per line instead of once per half.  The price: 64 data registers per lane and two tiles per wave.
// Means and norms are column sums followed by a half-wave butterfly (the first mapping: 16 strided samples per lane,
// wave butterfly): the float ZNCC moves in its last bits, the integer peak does not.
// Config B, 250 000 POIs, FFTCC2D launches (tools/ab_icgn2d.sh with TIME_FFTCC=1, profiles/r3l_fftcc2d_ab_two_pois_per_wave.txt):
// (line, half) mapping 0.70 - 0.72 ms (16 / 19 / 32 resident waves per CU alike); this mapping with c2 tiles (9 waves per CU)
// 0.60 ms; with the tile split into a real and an imaginary round (OC_FFTCC2D_X2_SPLIT: half the LDS per POI, 16 waves per
// CU, 120 VGPRs) 0.54 ms.  1 140 VALU (458 of them packed) and 110 LDS wave-instructions per POI against 1 412 / 140.
#ifndef OC_FFTCC2D_X2_WAVES
#define OC_FFTCC2D_X2_WAVES 1
#endif
#ifndef OC_FFTCC2D_X2_SPLIT
#define OC_FFTCC2D_X2_SPLIT 1
#endif
#ifndef OC_FFTCC2D_X2_OCC
#define OC_FFTCC2D_X2_OCC (OC_FFTCC2D_X2_SPLIT ? 4 : 2)   // waves per SIMD the register allocation must allow
#endif
constexpr int kX2Waves = OC_FFTCC2D_X2_WAVES;  // waves per workgroup, two POIs each

__device__ __forceinline__ float half_wave_sum(float v) {
#pragma unroll
    for (int off = 1; off < 32; off <<= 1) v += __shfl_xor(v, off, kWave);
    return v;
}

__global__ __launch_bounds__(64 * kX2Waves, OC_FFTCC2D_X2_OCC) void fftcc2d_fused32x2_kernel(Fftcc2dParams P, float* __restrict__ pois, int stride_f,
                                                                         unsigned long long count, int xcd_chunk) {
    __shared__ c2 lds[kX2Waves * 2 * FWAVE_LDS / (OC_FFTCC2D_X2_SPLIT ? 2 : 1)];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = lane >> 5, l = lane & 31;
    unsigned long long grp = blockIdx.x;
    if (xcd_chunk > 0) grp = (unsigned long long)(blockIdx.x & 7u) * xcd_chunk + (blockIdx.x >> 3);
    const unsigned long long idx = (grp * kX2Waves + wave) * 2 + q;
    if (idx >= count) return;
    c2* tile = lds + (wave * 2 + q) * (FWAVE_LDS / (OC_FFTCC2D_X2_SPLIT ? 2 : 1));
    float* poi = pois + idx * (unsigned long long)stride_f;
    const float px = poi[poi2d::X], py = poi[poi2d::Y];
    const float gu = poi[poi2d::U], gv = poi[poi2d::V];
    const int rx = FN / 2, ry = FN / 2, width = P.width, height = P.height;
    constexpr int M = FN * FN;
    // bounds guard: the reference returns silently and leaves the POI untouched (src/oc_fftcc.cpp:190-196)
    if ((int)px < rx || (int)px >= width - rx || (int)py < ry || (int)py >= height - ry || (int)(px + gu) < rx ||
        (int)(px + gu) >= width - rx || (int)(py + gv) < ry || (int)(py + gv) >= height - ry)
        return;

    // ---- window fill (src/oc_fftcc.cpp:198-231): lane l reads column l, row k = 0 .. 31
    c2 v[FN];
    float rn, tn;
    {
        const __amdgpu_buffer_rsrc_t r_ref = make_rsrc(P.ref), r_tar = make_rsrc(P.tar);
        const float rxp = px + l - rx, txp = rxp + gu;
#pragma unroll
        for (int k = 0; k < FN; k++) {
            const float ryp = py + k - ry, typ = ryp + gv;
            const float a = buf_f32(r_ref, (__umul24((unsigned)(int)ryp, (unsigned)width) + (unsigned)(int)rxp) << 2, 0);
            const float b = buf_f32(r_tar, (__umul24((unsigned)(int)typ, (unsigned)width) + (unsigned)(int)txp) << 2, 0);
            v[k] = mkc(a, b);
        }
        float rsum = 0.f, tsum = 0.f;
#pragma unroll
        for (int k = 0; k < FN; k++) {
            rsum += v[k].x;
            tsum += v[k].y;
        }
        const c2 mean = mkc(half_wave_sum(rsum) / M, half_wave_sum(tsum) / M);
        rn = 0.f;
        tn = 0.f;
#pragma unroll
        for (int k = 0; k < FN; k++) {
            v[k] = v[k] - mean;
            rn += v[k].x * v[k].x;
            tn += v[k].y * v[k].y;
        }
        rn = half_wave_sum(rn);
        tn = half_wave_sum(tn);
        asm volatile("" : "+v"(rn), "+v"(tn));  // formed here, used at the very end
    }
    // ---- forward along the rows' index (the lane's column), straight from the registers: v[bitrev5(kr)] = Z1(kr, c = l)
    fft32<false>(v);
    // Transpositions through the tile happen IN PLACE in the register array: once element W(k) has been written its
    // register is free to receive element k of the new line (all indices are compile-time constants, so "v[bitrev5(k)]" and
    // "v[k]" are just register names).
#if OC_FFTCC2D_X2_SPLIT
    // the tile holds ONE float per element: real parts travel first, imaginary parts second -- half the LDS per POI,
    // i.e. twice the waves per CU, for twice the (4-byte) LDS instructions
    float* __restrict__ ft = reinterpret_cast<float*>(tile);
#define OC_X2_TRANSPOSE(WRITE_IDX, READ_IDX)                                                   \
    _Pragma("unroll") for (int k = 0; k < FN; k++) ft[WRITE_IDX] = v[bitrev5(k)].x;            \
    __builtin_amdgcn_wave_barrier();                                                           \
    _Pragma("unroll") for (int k = 0; k < FN; k++) v[k].x = ft[READ_IDX];                      \
    __builtin_amdgcn_wave_barrier();                                                           \
    _Pragma("unroll") for (int k = 0; k < FN; k++) ft[WRITE_IDX] = v[bitrev5(k)].y;            \
    __builtin_amdgcn_wave_barrier();                                                           \
    _Pragma("unroll") for (int k = 0; k < FN; k++) v[k].y = ft[READ_IDX];                      \
    __builtin_amdgcn_wave_barrier();
#else
#define OC_X2_TRANSPOSE(WRITE_IDX, READ_IDX)                                                   \
    _Pragma("unroll") for (int k = 0; k < FN; k++) tile[WRITE_IDX] = v[bitrev5(k)];            \
    __builtin_amdgcn_wave_barrier();                                                           \
    _Pragma("unroll") for (int k = 0; k < FN; k++) v[k] = tile[READ_IDX];                      \
    __builtin_amdgcn_wave_barrier();
#endif
    // ---- transpose; forward along the columns' index: lane l owns row kr = l
    OC_X2_TRANSPOSE(k * FP + l, l * FP + k)
    fft32<false>(v);  // v[bitrev5(kc)] = Z(kr = l, kc)
    // ---- spectra of the two real windows and their product conj(R) * T (src/oc_fftcc.cpp:236-241); Z(-k) is row
    // (-l) read backwards.  The product replaces Z in its register; the inverse transform takes them in natural order.
    {
        const int mrow = ((FN - l) & (FN - 1)) * FP;
        // 2R = z + conj(zm), 2T = (z - conj(zm)) / i; the product is formed WITHOUT the two factors of one half: scaling by
        // a power of two commutes with every rounding of the inverse transform, so the peak comes out exactly four times too
        // large and the final division takes the 0.25 (same ZNCC bits as with the factors applied here).  Scalar on purpose: the
        // packed form needs 3.5 v_mov per element to build its operand pairs and costs the same cycles.
        auto product = [](c2 z, c2 zm) {
            const float rr = z.x + zm.x, ri = z.y - zm.y;
            const float tr = z.y + zm.y, ti = zm.x - z.x;
            return mkc((rr * tr) + (ri * ti), (rr * ti) - (ri * tr));
        };
#if OC_FFTCC2D_X2_SPLIT
        float zmx[FN];
#pragma unroll
        for (int k = 0; k < FN; k++) ft[l * FP + k] = v[bitrev5(k)].x;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < FN; k++) zmx[k] = ft[mrow + ((FN - k) & (FN - 1))];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < FN; k++) ft[l * FP + k] = v[bitrev5(k)].y;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < FN; k++) v[bitrev5(k)] = product(v[bitrev5(k)], mkc(zmx[k], ft[mrow + ((FN - k) & (FN - 1))]));
        __builtin_amdgcn_wave_barrier();
#else
#pragma unroll
        for (int k = 0; k < FN; k++) tile[l * FP + k] = v[bitrev5(k)];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < FN; k++) v[bitrev5(k)] = product(v[bitrev5(k)], tile[mrow + ((FN - k) & (FN - 1))]);
        __builtin_amdgcn_wave_barrier();
#endif
    }
    // ---- inverse along kc (row kr = l), transpose, inverse along kr (column c = l); unnormalised like FFTW's c2r
    c2 u[FN];
#pragma unroll
    for (int k = 0; k < FN; k++) u[k] = v[bitrev5(k)];  // natural-order input: a renaming of registers
    fft32<true>(u);
#pragma unroll
    for (int k = 0; k < FN; k++) v[k] = u[k];
    OC_X2_TRANSPOSE(l * FP + k, k * FP + l)
#undef OC_X2_TRANSPOSE
#pragma unroll
    for (int k = 0; k < FN; k++) u[k] = v[k];
    fft32<true>(u);  // u[bitrev5(r)] = correlation surface (row r, column l)

    // ---- arg-max with "strict >, scanning from index 0" (src/oc_fftcc.cpp:246-255): the lane's 32 values sit at linear
    // indices r * 32 + l, ascending in r; then the half-wave's 32 columns, the lower index winning a tie
    float best = -2.f;
    int bidx = 0;
#pragma unroll
    for (int r = 0; r < FN; r++) {
        const float val = u[bitrev5(r)].x;
        if (val > best) { best = val; bidx = r * FN + l; }
    }
#pragma unroll
    for (int off = 1; off < 32; off <<= 1) {
        const float ov = __shfl_xor(best, off, kWave);
        const int oi = __shfl_xor(bidx, off, kWave);
        if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
    }
    if (l == 0) {
        int du = bidx % FN, dv = bidx / FN;
        if (du > rx) du -= FN;
        if (dv > ry) dv -= FN;
        poi[poi2d::U] = (float)du + gu;
        poi[poi2d::V] = (float)dv + gv;
        poi[poi2d::U0] = gu;
        poi[poi2d::V0] = gv;
        poi[poi2d::ZNCC] = (0.25f * best) / (sqrtf(rn * tn) * M);
    }
}

bool fftcc2d_fused_supported(int rx, int ry) { return rx == FN / 2 && ry == FN / 2; }

hipError_t launch_fftcc2d_fused(const Fftcc2dParams& p, float* pois, int stride_f, size_t count, bool xcd,
                                hipStream_t stream) {
    if (count == 0) return hipSuccess;
    if (!fftcc2d_fused_supported(p.rx, p.ry)) return hipErrorInvalidValue;
    (void)hipGetLastError();  // drop stale errors of earlier, unrelated calls
#if OC_FFTCC2D_X2
    {
        const size_t groups = (count + 2 * kX2Waves - 1) / (2 * kX2Waves);
        const int chunk = xcd ? (int)((groups + 7) / 8) : 0;
        const size_t grid = xcd ? (size_t)chunk * 8 : groups;
        hipLaunchKernelGGL(fftcc2d_fused32x2_kernel, dim3((unsigned)grid), dim3(64 * kX2Waves), 0, stream, p, pois, stride_f,
                           (unsigned long long)count, chunk);
        return hipGetLastError();
    }
#else
    const size_t groups = (count + kFusedWaves - 1) / kFusedWaves;
    const int chunk = xcd ? (int)((groups + 7) / 8) : 0;
    const size_t grid = xcd ? (size_t)chunk * 8 : groups;
    hipLaunchKernelGGL(fftcc2d_fused32_kernel, dim3((unsigned)grid), dim3(64 * kFusedWaves), 0, stream, p, pois, stride_f,
                       (unsigned long long)count, chunk);
    return hipGetLastError();
#endif
}

}  // namespace ochip
