// dic2d_device.h -- device helpers shared by the 2D correlation kernels (icgn2d.hip, nr2d.hip):
// wave-uniform small dense algebra in the oracle's operation order, the sample walk of a
// one-wave-per-POI kernel, buffer-resource loads and the bicubic LUT fetch / evaluation.
#pragma once

#include "oc_device.h"

namespace ochip {

__device__ __forceinline__ float uni(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

// A wave-uniform value held in a VECTOR register on purpose.  VALU instructions that read an SGPR operand issue markedly
// slower on gfx950 than the same instruction on vector operands (tools/ubench/valu_mix_ubench.hip: the interpolation sweep's
// VALU stream with the six warp coefficients per lane instead of wave-uniform: 186 -> 160 cycles per polynomial and SIMD), so
// the per-sample loops CAN read their uniform factors from VGPRs (OC_UNIFORM_IN_VGPR in icgn2d.hip; the empty asm hides the
// uniformity from the compiler).  Inside the kernel it does not pay -- config B 3.31 - 3.42 against 3.24 ms
// (profiles/r4i_icgn2d1_ab_uniform_in_vgpr.txt): the sweep is not bound by its VALU side -- so the default stays SGPRs.
__device__ __forceinline__ float in_vgpr(float v) {
    asm volatile("" : "+v"(v));
    return v;
}
__device__ __forceinline__ int in_vgpr(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

// ---------------------------------------------------------------------------
// small dense algebra on wave-uniform values (every lane computes the same
// thing).  Same operation order as the oracle (oracle/oc_oracle.cpp lu_inverse,
// inverse3, mat_mul), which restates Eigen's PartialPivLU / cofactor inverse /
// lazy product used at src/oc_icgn.cpp:210,290,759,831.
// ---------------------------------------------------------------------------
// Inverse of an n x n matrix by LU with partial pivoting + solve against the identity,
// distributed over the wave: lane j (j < n) holds COLUMN j of the matrix in col[0..n-1]
// and receives column j of the inverse in inv[0..n-1].  Every scalar operation (pivot
// choice, multipliers f = a_rk / a_kk, eliminations a_rc -= f * a_kc, the two
// triangular solves) is the one the sequential algorithm performs on that element, so
// the result is bit-identical to oracle lu_inverse(); only the element -> lane
// placement differs.  Multipliers and pivots are broadcast lane to lane (`bcast`).
// `bcast(v, k)` hands every lane the value lane k of ITS matrix holds: v_readlane for one matrix per wave
// (WaveBcast), a ds_bpermute inside 8-lane groups when a wave factors eight matrices at once (GroupBcast8; pivots and
// the row permutation are then per-lane values, uniform inside a group).
struct WaveBcast {
    __device__ __forceinline__ float operator()(float v, int k) const { return wave_bcast(v, k); }
};
struct GroupBcast8 {
    int base;  // byte address of the group's lane 0 in ds_bpermute terms
    __device__ __forceinline__ explicit GroupBcast8(int lane) : base((lane & ~7) << 2) {}
    __device__ __forceinline__ float operator()(float v, int k) const {
        return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(base + (k << 2), __builtin_bit_cast(int, v)));
    }
};
template <int n, class Bcast = WaveBcast>
__device__ __forceinline__ void lu_inverse_lanes(float (&col)[n], float (&inv)[n], int lane, const Bcast& bcast = Bcast()) {
    int perm[n];  // wave-uniform row permutation
#pragma unroll
    for (int i = 0; i < n; i++) perm[i] = i;
#pragma unroll
    for (int k = 0; k < n; k++) {
        int piv = k;
        float best = fabsf(bcast(col[k], k));
#pragma unroll
        for (int r = k + 1; r < n; r++) {
            const float v = fabsf(bcast(col[r], k));
            if (v > best) { best = v; piv = r; }
        }
#pragma unroll
        for (int r = k + 1; r < n; r++) {  // swap rows k <-> piv (at most one r matches)
            const bool sw = (piv == r);
            const float a = col[k], b = col[r];
            col[k] = sw ? b : a;
            col[r] = sw ? a : b;
            const int pa = perm[k], pb = perm[r];
            perm[k] = sw ? pb : pa;
            perm[r] = sw ? pa : pb;
        }
        const float d = bcast(col[k], k);
#pragma unroll
        for (int r = k + 1; r < n; r++) {
            const float f = bcast(col[r], k) / d;
            const float upd = col[r] - f * col[k];
            col[r] = lane == k ? f : (lane > k ? upd : col[r]);
        }
    }
    // lane c solves L U x = P e_c
    float y[n];
#pragma unroll
    for (int i = 0; i < n; i++) {
        float v = (perm[i] == lane) ? 1.f : 0.f;
#pragma unroll
        for (int j = 0; j < i; j++) v = v - bcast(col[i], j) * y[j];
        y[i] = v;
    }
#pragma unroll
    for (int i = n - 1; i >= 0; i--) {
        float v = y[i];
#pragma unroll
        for (int j = i + 1; j < n; j++) v = v - bcast(col[i], j) * y[j];
        y[i] = v / bcast(col[i], i);
    }
#pragma unroll
    for (int i = 0; i < n; i++) inv[i] = y[i];
}

// Cooperative inverse of the eight 6 x 6 Hessians of an 8-wave workgroup (icgn2d.hip, ICGN2D1): every wave has filed the
// 21 totals of its lower triangle (row-major, h[i(i+1)/2 + j]) at area[64 w .. 64 w + 20]; ONE wave factors all eight
// matrices in a single instruction stream -- matrix g in lanes 8g .. 8g+5, lane 8g+j holding column j -- and leaves
// H^-1 of matrix g row-major at area[64 g + 24 .. 64 g + 59].  Per matrix the operations are those of lu_inverse_lanes,
// so the bits are the oracle's; the other seven waves do not spend their ~900 VALU instructions (128 v_readlane, 21
// divisions) on a computation that uses six lanes.
__device__ __forceinline__ void coop_inverse6_x8(float* __restrict__ area, int lane) {
    const int g = lane >> 3, j = min(lane & 7, 5);  // lanes 6, 7 of a group shadow lane 5
    float* __restrict__ mine = area + g * 64;
    float col[6], inv[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const int r = max(i, j), c = min(i, j);
        col[i] = mine[(r * (r + 1)) / 2 + c];  // H(i, j) = H(j, i)
    }
    lu_inverse_lanes<6, GroupBcast8>(col, inv, j, GroupBcast8(lane));
    if ((lane & 7) < 6) {
#pragma unroll
        for (int i = 0; i < 6; i++) mine[24 + i * 6 + j] = inv[i];  // H^-1(i, j)
    }
}

template <int n>
__device__ __forceinline__ void mat_mul(const float (&a)[n * n], const float (&b)[n * n], float (&c)[n * n]) {
#pragma unroll
    for (int i = 0; i < n; i++)
#pragma unroll
        for (int j = 0; j < n; j++) {
            float v = a[i * n + 0] * b[0 * n + j];
#pragma unroll
            for (int k = 1; k < n; k++) v = v + a[i * n + k] * b[k * n + j];
            c[i * n + j] = v;
        }
}

__device__ __forceinline__ float cof3(const float (&m)[9], int i, int j) {
    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    return m[i1 * 3 + j1] * m[i2 * 3 + j2] - m[i1 * 3 + j2] * m[i2 * 3 + j1];
}
__device__ __forceinline__ void inverse3(const float (&m)[9], float (&r)[9]) {
    const float c0 = cof3(m, 0, 0), c1 = cof3(m, 1, 0), c2 = cof3(m, 2, 0);
    const float det = (c0 * m[0] + c1 * m[3]) + c2 * m[6];
    const float invdet = 1.f / det;
    r[0] = c0 * invdet; r[1] = c1 * invdet; r[2] = c2 * invdet;
    r[3] = cof3(m, 0, 1) * invdet; r[4] = cof3(m, 1, 1) * invdet; r[5] = cof3(m, 2, 1) * invdet;
    r[6] = cof3(m, 0, 2) * invdet; r[7] = cof3(m, 1, 2) * invdet; r[8] = cof3(m, 2, 2) * invdet;
}

// Deformation2D1::setWarp, src/oc_deformation.cpp:117-128
__device__ __forceinline__ void set_warp_2d1(float (&w)[9], float u, float ux, float uy, float v, float vx, float vy) {
    w[0] = 1.f + ux; w[1] = uy; w[2] = u;
    w[3] = vx; w[4] = 1.f + vy; w[5] = v;
    w[6] = 0.f; w[7] = 0.f; w[8] = 1.f;
}

// walks the samples owned by one lane: s = lane, lane+64, ... as (row r, column c)
struct SampleWalk {
    int r, c, s;
    int W, q64, r64;
    __device__ __forceinline__ SampleWalk(int lane, int r0, int c0, int W_, int q64_, int r64_)
        : r(r0), c(c0), s(lane), W(W_), q64(q64_), r64(r64_) {}
    __device__ __forceinline__ void next() {
        s += kWave;
        c += r64;
        r += q64;
        const bool wrap = c >= W;
        c = wrap ? c - W : c;
        r = wrap ? r + 1 : r;
    }
};

// (f2 / mk2 / mad: oc_device.h)

// SampleWalk for the engines whose local coordinates are integers (no centre offset): the lane's sample as the
// float pair (x_local, y_local) -- small integers, so every step is exact -- plus its byte offset from the subset
// origin in a row-major float image.  One wrap test serves both.
struct FloatWalk {
    f2 xy;
    unsigned off;
    f2 step, wstep;
    unsigned offs, offw;
    float xmax;
    __device__ __forceinline__ FloatWalk(int r0, int c0, int rx, int ry, int W, int q64, int r64, unsigned row_bytes)
        : xy(mk2((float)(c0 - rx), (float)(r0 - ry))),
          off((unsigned)r0 * row_bytes + ((unsigned)c0 << 2)),
          step(mk2((float)r64, (float)q64)),
          wstep(mk2((float)(r64 - W), (float)(q64 + 1))),
          offs((unsigned)q64 * row_bytes + ((unsigned)r64 << 2)),
          offw((unsigned)(q64 + 1) * row_bytes + ((unsigned)r64 << 2) - ((unsigned)W << 2)),
          xmax((float)rx) {}
    __device__ __forceinline__ void next() {
        const f2 a = xy + step, b = xy + wstep;
        const bool wrap = a.x > xmax;  // column index ran past the subset width
        xy = wrap ? b : a;
        off += wrap ? offw : offs;
    }
};

// Passes 0 .. NT-1 of a lane's samples with the global loads of up to B passes in flight: `load(t, valid)` issues the
// loads of pass t (and captures whatever the walk says about it), `use(t, valid, v)` consumes them -- in pass order, so
// every running sum keeps its association.  A timeline of the one-wave-per-POI kernels showed the
// passes of the set-up phase waiting ~2 k cycles each for ONE dependent round trip; with B passes per round trip the
// phase shrinks accordingly.  Passes [0, NF) are full, pass NF (if NF < NT) holds the lanes with `tail_valid`.
template <int B, class Load, class Use>
__device__ __forceinline__ void passes_batched(int NF, int NT, bool tail_valid, Load&& load, Use&& use) {
    int t = 0;
#pragma unroll 1
    for (; t + B <= NF; t += B) {
        decltype(load(0, true)) v[B];
#pragma unroll
        for (int u = 0; u < B; u++) v[u] = load(t + u, true);
#pragma unroll
        for (int u = 0; u < B; u++) use(t + u, true, v[u]);
    }
    // the remaining full passes and the partial one as a last, shorter batch (B - 1 full passes at most + the tail)
    {
        decltype(load(0, true)) v[B];
#pragma unroll
        for (int u = 0; u < B; u++) {
            if (t + u < NF) v[u] = load(t + u, true);
            else if (t + u < NT) v[u] = load(t + u, tail_valid);
        }
#pragma unroll
        for (int u = 0; u < B; u++) {
            if (t + u < NF) use(t + u, true, v[u]);  // (a literal `true` folds the validity selects away)
            else if (t + u < NT) use(t + u, tail_valid, v[u]);
        }
    }
}

// The same walk for bodies that are too register-hungry to be unrolled (the 78 running sums of the 12-DoF Hessian): the
// loads of pass t + 1 are issued before pass t is consumed, so the round trip of one pass hides behind the arithmetic
// of the previous one at the price of one more set of loaded values.
template <class Load, class Use>
__device__ __forceinline__ void passes_prefetched(int NF, int NT, bool tail_valid, Load&& load, Use&& use) {
    if (NT <= 0) return;
    auto cur = NF > 0 ? load(0, true) : load(0, tail_valid);
#pragma unroll 1
    for (int t = 0; t < NF; t++) {
        decltype(cur) nxt = cur;
        if (t + 1 < NF) nxt = load(t + 1, true);
        else if (t + 1 < NT) nxt = load(t + 1, tail_valid);
        use(t, true, cur);
        cur = nxt;
    }
    if (NF < NT) use(NF, tail_valid, cur);
}

// One LUT entry in flight: address generation and the four 16-byte loads are issued for a
// whole group of G samples before any polynomial is evaluated, so each lane keeps G*64 B
// of gathers outstanding (the interpolation sweep is latency/L1-bandwidth bound).  Only the
// fractional offsets travel with the coefficients; dy = -1 marks an out-of-range sample
// (a valid dy lies in [0, 1)).
struct LutFetch {
    float4 c0, c1, c2, c3;
    float dx, dy;
};

// All image-sized arrays are read through buffer resources (V#): a 32-bit per-lane byte offset plus a
// wave-uniform SGPR offset replace the 64-bit per-lane address arithmetic of flat loads -- in these loops
// the address math used to cost as many VALU slots as the arithmetic it fed.  Raw buffer, stride 0,
// num_records = 2^32 - 1 bytes: an image (<= 2^28 pixels = 1 GiB) and one plane of the LUT (<= 4 GiB) both fit.
typedef unsigned u4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, -1, 0x00020000);
}
__device__ __forceinline__ float buf_f32(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
// cache-policy bits of the table gathers (experiments: tools/ab_icgn2d.sh -DOC_LUT_AUX=<bits>; gfx950: 1 = sc0, 2 = nt, 16 = sc1)
#ifndef OC_LUT_AUX
#define OC_LUT_AUX 0
#endif
__device__ __forceinline__ float4 buf_f32x4(__amdgpu_buffer_rsrc_t r, unsigned voff) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, OC_LUT_AUX));
}

// The bicubic coefficient table is stored PLANAR: plane k (the power of dy, k = 0..3) holds for every pixel the
// float4 (coef[k][0], coef[k][1], coef[k][2], coef[k][3]) of src/oc_cubic_bspline.cpp:123-129, pixels in row-major
// order; plane stride = height * width * 16 bytes.  Lanes that own row-consecutive samples of a subset then read 16
// CONSECUTIVE bytes each: one buffer_load_b128 of a wave touches ~10 128-byte lines (two subset rows of ~530 B)
// instead of the ~34 lines it touched -- four times over, once per k -- when an entry was 64 contiguous bytes
// ([y][x][k][l], round 1: TA_TA_BUSY 82 %, 29 tag lookups per load instruction).
//
// LutPlanes4: one descriptor per plane -- a plane may be 4 GiB, i.e. images up to 2^28 pixels.
struct LutPlanes4 {
    __amdgpu_buffer_rsrc_t p0, p1, p2, p3;
    __device__ __forceinline__ LutPlanes4(const float* lut, int height, int width) {
        const size_t plane = (size_t)height * (size_t)width * 4;  // floats
        p0 = make_rsrc(lut);
        p1 = make_rsrc(lut + plane);
        p2 = make_rsrc(lut + 2 * plane);
        p3 = make_rsrc(lut + 3 * plane);
    }
    __device__ __forceinline__ void load(LutFetch& f, unsigned e) const {
        f.c0 = buf_f32x4(p0, e);
        f.c1 = buf_f32x4(p1, e);
        f.c2 = buf_f32x4(p2, e);
        f.c3 = buf_f32x4(p3, e);
    }
};
// LutPlanesS: one descriptor per table, the plane selected by the scalar offset (three SGPRs shared by all tables
// of the same image size): the whole table must stay below 4 GiB, i.e. 2^26 pixels.  NR2D1 reads three tables.
__device__ __forceinline__ float4 buf_f32x4s(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
struct LutPlanesS {
    __amdgpu_buffer_rsrc_t r;
    unsigned plane;  // bytes
    __device__ __forceinline__ LutPlanesS(const float* lut, int height, int width)
        : r(make_rsrc(lut)), plane((unsigned)height * (unsigned)width * 16u) {}
    __device__ __forceinline__ void load(LutFetch& f, unsigned e) const {
        f.c0 = buf_f32x4(r, e);
        f.c1 = buf_f32x4s(r, e, plane);
        f.c2 = buf_f32x4s(r, e, 2u * plane);
        f.c3 = buf_f32x4s(r, e, 3u * plane);
    }
};

// range rule of BicubicBspline::compute (src/oc_cubic_bspline.cpp:137-142): x < 1 || y < 1 ||
// x >= width-2 || y >= height-2 || NaN -> -1.  With xi = (int)floor(x) that is (unsigned)(xi - 1) > width - 4.
// v_cvt_flr_i32_f32 converts with floor in one instruction, saturates +-huge values to INT_MAX / INT_MIN and sends NaN
// to 0 -- all of them "outside" under the unsigned test.  v_fract_f32 returns x - floor(x), which is exact in float,
// i.e. the reference's x - (float)x_integral bit for bit for every in-range x.
__device__ __forceinline__ int floor_to_int(float x) {
    int r;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
// Fills the fractional offsets and returns the byte offset of the sample's pixel inside a plane (16 B per pixel).
// `out` = the sample is outside the interpolatable range; it fetches pixel (0,0), which is always mapped, and the
// caller discards the value.
// MARK: also record "outside" in f.dy (-1; a valid dy lies in [0, 1)) for callers that keep the reference's -1.f
// sentinel as a sample value (lut_eval) instead of abandoning the POI.
template <bool MARK>
__device__ __forceinline__ unsigned lut_locate(LutFetch& f, int height, int width, float x, float y, bool& out) {
    const int xi = floor_to_int(x), yi = floor_to_int(y);
    out = (unsigned)(xi - 1) > (unsigned)(width - 4) || (unsigned)(yi - 1) > (unsigned)(height - 4);
    f.dx = __builtin_amdgcn_fractf(x);
    const float fy = __builtin_amdgcn_fractf(y);
    f.dy = (MARK && out) ? -1.f : fy;
    // 24-bit multiply: full rate, and exact because in-range yi and the width are below 2^24
    return out ? 0u : (__umul24((unsigned)yi, (unsigned)width) + (unsigned)xi) << 4;
}
template <bool MARK, class Planes>
__device__ __forceinline__ void lut_fetch(LutFetch& f, const Planes& lut, int height, int width, float x, float y, bool& out) {
    lut.load(f, lut_locate<MARK>(f, height, width, x, y, out));
}

// explicit 16-term left-to-right polynomial of src/oc_cubic_bspline.cpp:144-177: every product is the one the
// reference forms, in its order ((c * dy^k) * dx^l); the 15 additions are a left-to-right chain.  Plain (unpacked)
// fp32 instructions on purpose: on gfx950 a v_pk_mul_f32 / v_pk_add_f32 occupies a SIMD for 4.3 cycles against 2.4 for
// v_mul_f32 / v_add_f32 (tools/ubench/valu_ubench.hip, profiles/r02b_valu_ubench.json), so a packed pair saves ~10 %
// at best and loses it again to the v_mov_b32 that form the register pairs (the 16-byte loads leave the coefficients
// in the wrong pairing for half of the products).
__device__ __forceinline__ float lut_poly(const LutFetch& f) {
    const float dx = f.dx, dy = f.dy;
    const float dx2 = dx * dx, dy2 = dy * dy;
    const float dx3 = dx2 * dx, dy3 = dy2 * dy;
#if OC_FMA
    // the same 16 terms left to right with every "+ product" fused (oracle bspline2d_eval<true>): 4 + 9 multiplies and
    // 15 v_fma_f32 instead of 28 multiplies and 15 adds
    float w = f.c0.x;
    w = mad(f.c0.y, dx, w);
    w = mad(f.c0.z, dx2, w);
    w = mad(f.c0.w, dx3, w);
    w = mad(f.c1.x, dy, w);
    w = mad(f.c1.y * dy, dx, w);
    w = mad(f.c1.z * dy, dx2, w);
    w = mad(f.c1.w * dy, dx3, w);
    w = mad(f.c2.x, dy2, w);
    w = mad(f.c2.y * dy2, dx, w);
    w = mad(f.c2.z * dy2, dx2, w);
    w = mad(f.c2.w * dy2, dx3, w);
    w = mad(f.c3.x, dy3, w);
    w = mad(f.c3.y * dy3, dx, w);
    w = mad(f.c3.z * dy3, dx2, w);
    w = mad(f.c3.w * dy3, dx3, w);
    return w;
#endif
    float v = f.c0.x;
    v = v + f.c0.y * dx;
    v = v + f.c0.z * dx2;
    v = v + f.c0.w * dx3;
    v = v + f.c1.x * dy;
    v = v + (f.c1.y * dy) * dx;
    v = v + (f.c1.z * dy) * dx2;
    v = v + (f.c1.w * dy) * dx3;
    v = v + f.c2.x * dy2;
    v = v + (f.c2.y * dy2) * dx;
    v = v + (f.c2.z * dy2) * dx2;
    v = v + (f.c2.w * dy2) * dx3;
    v = v + f.c3.x * dy3;
    v = v + (f.c3.y * dy3) * dx;
    v = v + (f.c3.z * dy3) * dx2;
    v = v + (f.c3.w * dy3) * dx3;
    return v;
}
// The same 16 terms with the 24 products formed as 14 packed multiplies on the register pairs the 16-byte loads deliver
// ((x, y) and (z, w) of each float4): (c.x, c.y) * dy^k, then * (1, dx); (c.z, c.w) * dy^k, then * (dx^2, dx^3) -- every
// product is the reference's ((c * dy^k) * dx^l, the factor 1.f is exact), the 15 additions stay the left-to-right chain:
// identical bits, 10 VALU instructions fewer per sample (134 -> 115 per two samples in the sweep, no operand moves).
// MEASURED in round 4 and NOT faster (profiles/r4d_ab_packed_products.txt): ICGN2D1 on config B 3.27 - 3.28 against 3.24 ms,
// ICGN2D2 on config C 3.53 against 3.52 - 3.53, the sweep's VALU side alone 1.12 against 1.04 ms -- a v_pk_mul_f32 holds the
// SIMD twice as long as a v_mul_f32 (round 2: 4.3 against 2.4 cycles), so two products per instruction buy nothing: VALU
// time follows issue cycles, not instruction counts.  Kept behind OC_POLY_PACKED (default 0) as the A/B partner.
#ifndef OC_POLY_PACKED
#define OC_POLY_PACKED 0
#endif
#if OC_POLY_PACKED && OC_FMA
#error "the packed-product polynomial exists in the separately rounded mode only"
#endif
__device__ __forceinline__ float lut_poly_pk(const LutFetch& f) {
    const float dx = f.dx, dy = f.dy;
    const float dx2 = dx * dx, dy2 = dy * dy;
    const float dx3 = dx2 * dx, dy3 = dy2 * dy;
    const f2 m01 = mk2(1.f, dx), m23 = mk2(dx2, dx3);
    const f2 a01 = mk2(f.c0.x, f.c0.y) * m01, a23 = mk2(f.c0.z, f.c0.w) * m23;
    const f2 b01 = (mk2(f.c1.x, f.c1.y) * dy) * m01, b23 = (mk2(f.c1.z, f.c1.w) * dy) * m23;
    const f2 c01 = (mk2(f.c2.x, f.c2.y) * dy2) * m01, c23 = (mk2(f.c2.z, f.c2.w) * dy2) * m23;
    const f2 d01 = (mk2(f.c3.x, f.c3.y) * dy3) * m01, d23 = (mk2(f.c3.z, f.c3.w) * dy3) * m23;
    float v = a01.x;
    v = v + a01.y;
    v = v + a23.x;
    v = v + a23.y;
    v = v + b01.x;
    v = v + b01.y;
    v = v + b23.x;
    v = v + b23.y;
    v = v + c01.x;
    v = v + c01.y;
    v = v + c23.x;
    v = v + c23.y;
    v = v + d01.x;
    v = v + d01.y;
    v = v + d23.x;
    v = v + d23.y;
    return v;
}
__device__ __forceinline__ float lut_value(const LutFetch& f) { return OC_POLY_PACKED ? lut_poly_pk(f) : lut_poly(f); }
__device__ __forceinline__ float lut_eval(const LutFetch& f) {
    const float v = lut_value(f);
    return f.dy < 0.f ? -1.f : v;
}

// Deformation2D2::setWarp, src/oc_deformation.cpp:301-350; q = u ux uy uxx uxy uyy v vx vy vxx vxy vyy
__device__ __forceinline__ void set_warp_2d2(float (&w)[36], const float (&q)[12]) {
    const float u = q[0], ux = q[1], uy = q[2], uxx = q[3], uxy = q[4], uyy = q[5];
    const float v = q[6], vx = q[7], vy = q[8], vxx = q[9], vxy = q[10], vyy = q[11];
    w[0] = 1.f + 2.f * ux + ux * ux + u * uxx;
    w[1] = 2.f * u * uxy + 2.f * (1.f + ux) * uy;
    w[2] = uy * uy + u * uyy;
    w[3] = 2.f * u * (1 + ux);
    w[4] = 2.f * u * uy;
    w[5] = u * u;
    w[6] = 0.5f * (v * uxx + 2.f * (1.f + ux) * vx + u * vxx);
    w[7] = 1.f + uy * vx + ux * vy + v * uxy + u * vxy + vy + ux;
    w[8] = 0.5f * (v * uyy + 2.f * uy * (1.f + vy) + u * vyy);
    w[9] = v + v * ux + u * vx;
    w[10] = u + v * uy + u * vy;
    w[11] = u * v;
    w[12] = vx * vx + v * vxx;
    w[13] = 2.f * v * vxy + 2.f * vx * (1.f + vy);
    w[14] = 1.f + 2.f * vy + vy * vy + v * vyy;
    w[15] = 2.f * v * vx;
    w[16] = 2.f * v * (1.f + vy);
    w[17] = v * v;
    w[18] = 0.5f * uxx; w[19] = uxy; w[20] = 0.5f * uyy; w[21] = 1.f + ux; w[22] = uy; w[23] = u;
    w[24] = 0.5f * vxx; w[25] = vxy; w[26] = 0.5f * vyy; w[27] = vx; w[28] = 1.f + vy; w[29] = v;
    w[30] = 0.f; w[31] = 0.f; w[32] = 0.f; w[33] = 0.f; w[34] = 0.f; w[35] = 1.f;
}

// First damping value of IC-LM: powf(damping.lambda, znssd / znssd0) (src/oc_iclm.cpp:253, :628) as a fixed sequence
// of IEEE double operations -- exp(q * ln(lambda)), ln(lambda) taken once on the host, argument reduction by ln 2
// (hi/lo split), degree-14 Taylor polynomial, one rounding to float.  libm powf implementations differ in the
// last bit between platforms (glibc's is off by one ulp from the correctly rounded value in ~0.05% of cases); this
// form is reproducible, and the oracle restates the identical sequence.
__device__ __forceinline__ float pow_lambda(double log_lambda, float q) {
    const double t = (double)q * log_lambda;
    if (!(t == t)) return __builtin_nanf("");
    if (t > 700.0) return __builtin_inff();
    if (t < -700.0) return 0.f;
    const double kf = floor(t * 1.44269504088896338700e+00 + 0.5);
    const double r = (t - kf * 6.93147180369123816490e-01) - kf * 1.90821492927058770002e-10;
    double e = 1.0 / 87178291200.0;
    e = e * r + 1.0 / 6227020800.0;
    e = e * r + 1.0 / 479001600.0;
    e = e * r + 1.0 / 39916800.0;
    e = e * r + 1.0 / 3628800.0;
    e = e * r + 1.0 / 362880.0;
    e = e * r + 1.0 / 40320.0;
    e = e * r + 1.0 / 5040.0;
    e = e * r + 1.0 / 720.0;
    e = e * r + 1.0 / 120.0;
    e = e * r + 1.0 / 24.0;
    e = e * r + 1.0 / 6.0;
    e = e * r + 0.5;
    e = e * r + 1.0;
    e = e * r + 1.0;
    return (float)ldexp(e, (int)kf);
}


// steepest-descent row of one sample (src/oc_icgn.cpp:191-196; 2D2: 725-745; center-offset
// overloads :390-398 / :953-972).  The local coordinates arrive as floats: without a centre
// offset they are small integers, so the reference's integer products (xl*xl, :733-738) are
// exact in float as well.
template <int DOF>
__device__ __forceinline__ void sd_row(float g_x, float g_y, float fxl, float fyl, float (&sd)[DOF]) {
    if constexpr (DOF == 6) {
        sd[0] = g_x; sd[1] = g_x * fxl; sd[2] = g_x * fyl;
        sd[3] = g_y; sd[4] = g_y * fxl; sd[5] = g_y * fyl;
    } else {
        const float xx = (fxl * fxl) * 0.5f, xy = fxl * fyl, yy = (fyl * fyl) * 0.5f;
        sd[0] = g_x; sd[1] = g_x * fxl; sd[2] = g_x * fyl; sd[3] = g_x * xx; sd[4] = g_x * xy; sd[5] = g_x * yy;
        sd[6] = g_y; sd[7] = g_y * fxl; sd[8] = g_y * fyl; sd[9] = g_y * xx; sd[10] = g_y * xy; sd[11] = g_y * yy;
    }
}

}  // namespace ochip
