// icgn3d_device.h -- device helpers shared by the two ICGN3D1 kernels (icgn3d.hip: sample s owned by thread s mod 512;
// icgn3d_rows.hip: one half-wave per subvolume row): block reductions, the 12 x 12 LU in LDS, the 4 x 4 warp algebra and
// the tricubic B-spline evaluation from global memory or from the LDS-staged coefficient box.
#pragma once

#include "oc_device.h"
#include "oc_kernels.h"

// Phase ablation (tools/ablate_icgn3d.sh): see icgn3d.hip
#ifndef OC_ABLATE
#define OC_ABLATE 0
#endif

namespace ochip {

constexpr int kBlock3d = 512;
constexpr int kWaves3d = kBlock3d / kWave;  // 8
constexpr int kWinCap = 16768;              // floats of LDS for the staged coefficient box (65.5 KB)
constexpr int kBoxSlots = 64;               // passes whose boxes are precomputed together
constexpr int kRedChunk = 13;               // values reduced per LDS round trip

__device__ __forceinline__ float uni3(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

// K simultaneous block-wide sums; on return every thread holds the same K results.
// red: LDS scratch of K * 8 floats.  Two barriers per call.  Association: xor butterfly inside each wave, then a balanced
// tree over the 8 wave sums in wave order.  Lane k (< K) of every wave combines the eight wave sums of value k and the
// K totals are handed round with v_readlane: 8 LDS reads + 7 adds + K broadcasts per thread instead of 8 K reads and
// 7 K adds (which the scheduler hoisted into ~100 live registers and spilled).
template <int K>
__device__ __forceinline__ void block_allreduce(float (&v)[K], float* red, int wave, int lane) {
    static_assert(K <= kWave, "one lane per value");
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = wave_allreduce_sum(v[k]);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < K; k++) red[k * kWaves3d + wave] = v[k];
    }
    __syncthreads();
    float w[kWaves3d];
    const int mine = lane < K ? lane : 0;
#pragma unroll
    for (int i = 0; i < kWaves3d; i++) w[i] = red[mine * kWaves3d + i];
    // xor butterfly over the wave index, ascending offsets == balanced tree in wave order
#pragma unroll
    for (int off = 1; off < kWaves3d; off <<= 1)
#pragma unroll
        for (int i = 0; i < kWaves3d; i += 2 * off) w[i] = w[i] + w[i + off];
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = wave_bcast(w[0], k);
    __syncthreads();
}

// Inverse of the 12 x 12 Hessian by LU with partial (row) pivoting + solve against the identity -- every scalar
// operation (pivot choice with strict >, multipliers f = a_rk / a_kk, eliminations a_rc - f * a_kc, the two triangular
// solves with ascending inner index) is the one oracle lu_inverse() performs (Eigen PartialPivLU, src/oc_icgn.cpp:1339),
// so the result is bit-identical.  ONE wave runs it, the matrix lives in LDS (A: 12 x 12 row-major, perm: 12 ints):
// lane c < 12 owns column c during the elimination and solves for column c of the inverse; everything the lanes share
// (pivot column, multipliers, the triangular factors) is read from LDS with wave-uniform addresses (broadcast reads).
// The register-resident, v_readlane-broadcast form this replaces compiled to 4.9 k instructions with 544 scratch
// accesses inside this kernel and cost 262 k cycles per POI (13 % of the kernel); this one is a few hundred
// instructions in real loops.  LDS operations of one wave execute in order, so no barrier is needed between them.
// out[i * kWave + lane] = H^-1(i, lane) for lane < 12 (0 for the other lanes).
__device__ __forceinline__ void lu_inverse12_lds(float* __restrict__ A, int* __restrict__ perm, float* __restrict__ out, int lane) {
    constexpr int n = 12;
    if (lane < n) perm[lane] = lane;
#pragma unroll 1
    for (int k = 0; k < n; k++) {
        float colk[n];  // column k, the same in every lane
#pragma unroll
        for (int r = 0; r < n; r++) colk[r] = A[r * n + k];
        int piv = k;
        float best = 0.f;
#pragma unroll
        for (int r = 0; r < n; r++) {
            const float v = fabsf(colk[r]);
            const bool take = r == k || (r > k && v > best);
            best = take ? v : best;
            piv = take ? r : piv;
        }
        piv = __builtin_amdgcn_readfirstlane(piv);
        if (piv != k) {  // wave-uniform
            if (lane < n) {
                const float a = A[k * n + lane], b = A[piv * n + lane];
                A[k * n + lane] = b;
                A[piv * n + lane] = a;
            }
            if (lane == 0) {
                const int t = perm[k];
                perm[k] = perm[piv];
                perm[piv] = t;
            }
            // column k after the swap
            float ck = 0.f, cp = 0.f;
#pragma unroll
            for (int r = 0; r < n; r++) {
                ck = r == k ? colk[r] : ck;
                cp = r == piv ? colk[r] : cp;
            }
#pragma unroll
            for (int r = 0; r < n; r++) colk[r] = r == k ? cp : (r == piv ? ck : colk[r]);
        }
        float d = 0.f;
#pragma unroll
        for (int r = 0; r < n; r++) d = r == k ? colk[r] : d;
        const float mine = lane < n ? A[k * n + lane] : 0.f;  // row k, this lane's column
#pragma unroll
        for (int r = 0; r < n; r++) {
            if (r > k) {  // wave-uniform
                const float f = colk[r] / d;
                if (lane == k) A[r * n + k] = f;
                else if (lane > k && lane < n) A[r * n + lane] = A[r * n + lane] - f * mine;
            }
        }
    }
    // lane c solves L U x = P e_c.  The rows are fenced against the instruction scheduler: unfenced it hoists all 144
    // broadcast reads of the two unrolled solves to the top and spills ~250 registers to scratch around them (measured:
    // 244 k cycles per POI for this function, almost all of it scratch traffic).
    float y[n];
#pragma unroll
    for (int i = 0; i < n; i++) {
        float v = (perm[i] == lane) ? 1.f : 0.f;
#pragma unroll
        for (int j = 0; j < i; j++) v = v - A[i * n + j] * y[j];
        y[i] = v;
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = n - 1; i >= 0; i--) {
        float v = y[i];
#pragma unroll
        for (int j = i + 1; j < n; j++) v = v - A[i * n + j] * y[j];
        y[i] = v / A[i * n + i];
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < n; i++) out[i * kWave + lane] = lane < n ? y[i] : 0.f;
}

// 4x4 inverse by cofactor expansion -- same operation order as oracle inverse4()
__device__ __forceinline__ float det3(float a, float b, float c, float d, float e, float f, float g, float h, float i) {
    return (a * (e * i - f * h) - b * (d * i - f * g)) + c * (d * h - e * g);
}
__device__ __forceinline__ void inverse4(const float (&m)[16], float (&r)[16]) {
    float cofm[16];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float s[9];
            int t = 0;
#pragma unroll
            for (int a = 0; a < 4; a++) {
                if (a == i) continue;
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    if (b == j) continue;
                    s[t++] = m[a * 4 + b];
                }
            }
            const float d = det3(s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7], s[8]);
            cofm[i * 4 + j] = ((i + j) & 1) ? -d : d;
        }
    const float det = ((m[0] * cofm[0] + m[1] * cofm[1]) + m[2] * cofm[2]) + m[3] * cofm[3];
    const float invdet = 1.f / det;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) r[i * 4 + j] = cofm[j * 4 + i] * invdet;
}

// Deformation3D1::setWarp, src/oc_deformation.cpp:495-516; q = u ux uy uz v vx vy vz w wx wy wz
__device__ __forceinline__ void set_warp_3d1(float (&w)[16], const float (&q)[12]) {
    w[0] = 1.f + q[1]; w[1] = q[2]; w[2] = q[3]; w[3] = q[0];
    w[4] = q[5]; w[5] = 1.f + q[6]; w[6] = q[7]; w[7] = q[4];
    w[8] = q[9]; w[9] = q[10]; w[10] = 1.f + q[11]; w[11] = q[8];
    w[12] = 0.f; w[13] = 0.f; w[14] = 0.f; w[15] = 1.f;
}

// cubic B-spline basis functions, src/oc_cubic_bspline.cpp:35-53; T = float, or a packed pair of arguments (the same
// IEEE operations, two per issue slot)
// OC_FMA: every Horner step "product + constant" is one fused multiply-add (oracle basis*_fma)
template <class T>
__device__ __forceinline__ T cst(float v);
template <>
__device__ __forceinline__ float cst<float>(float v) { return v; }
template <>
__device__ __forceinline__ f2 cst<f2>(float v) { return splat2(v); }
#if OC_FMA
template <class T>
__device__ __forceinline__ T basis0(T t) { return (1.f / 6.f) * mad(t, mad(t, -t + 3.f, cst<T>(-3.f)), cst<T>(1.f)); }
template <class T>
__device__ __forceinline__ T basis1(T t) { return (1.f / 6.f) * mad(t * t, mad(cst<T>(3.f), t, cst<T>(-6.f)), cst<T>(4.f)); }
template <class T>
__device__ __forceinline__ T basis2(T t) { return (1.f / 6.f) * mad(t, mad(t, mad(cst<T>(-3.f), t, cst<T>(3.f)), cst<T>(3.f)), cst<T>(1.f)); }
#else
template <class T>
__device__ __forceinline__ T basis0(T t) { return (1.f / 6.f) * (t * (t * (-t + 3.f) - 3.f) + 1.f); }
template <class T>
__device__ __forceinline__ T basis1(T t) { return (1.f / 6.f) * (t * t * (3.f * t - 6.f) + 4.f); }
template <class T>
__device__ __forceinline__ T basis2(T t) { return (1.f / 6.f) * (t * (t * (-3.f * t + 3.f) + 3.f) + 1.f); }
#endif
template <class T>
__device__ __forceinline__ T basis3(T t) { return (1.f / 6.f) * (t * t * t); }

typedef float float4u __attribute__((ext_vector_type(4), aligned(4)));  // 4-byte aligned 16-byte load

// ((b0 * r0 + b1 * r1) + b2 * r2) + b3 * r3 -- the weighted sum of four taps as the reference writes it
// (src/oc_cubic_bspline.cpp:390-401): four separately rounded products, three additions left to right.  OC_TAPS_PACKED
// forms the products as two packed multiplies (v_pk_mul_f32 on the pairs (r0, r1), (r2, r3): a ds_read2_b32 / a 16-byte
// load delivers exactly those register pairs); the same IEEE products, so the same bits, 2 instead of 4 multiply
// instructions: with `-mllvm -disable-vector-combine` (or LLVM also packs the additions, one useful lane each, with ~60
// operand moves) the tap block shrinks from 190 to 153 VALU instructions per sample.  MEASURED in round 4 and SLOWER
// (profiles/r4d_ab_packed_products.txt, 256^3 / 8 000 POIs, bit-identical): 12.25 - 12.30 ms against 11.81 -- a packed
// multiply occupies the SIMD about twice as long as a plain one, so VALU time follows issue cycles, not instruction counts
// (round 2's finding, confirmed on the 3D kernel).  Default 0; the macro stays for the A/B.
#ifndef OC_TAPS_PACKED
#define OC_TAPS_PACKED 0
#endif
#if OC_TAPS_PACKED && OC_FMA
#error "the packed tap products exist in the separately rounded mode only"
#endif
__device__ __forceinline__ float taps4(float b0, float b1, float b2, float b3, float r0, float r1, float r2, float r3) {
#if OC_FMA
    // one product, three fused multiply-adds (oracle taps4<true>): the 21 four-tap sums of a sample are 21 v_mul_f32 +
    // 63 v_fma_f32 instead of 84 multiplies + 63 adds
    return mad(b3, r3, mad(b2, r2, mad(b1, r1, b0 * r0)));
#elif OC_TAPS_PACKED
    const f2 p01 = mk2(r0, r1) * mk2(b0, b1), p23 = mk2(r2, r3) * mk2(b2, b3);
    return ((p01.x + p01.y) + p23.x) + p23.y;
#else
    return ((b0 * r0 + b1 * r1) + b2 * r2) + b3 * r3;
#endif
}

// TricubicBspline::compute, src/oc_cubic_bspline.cpp:353-405
__device__ __forceinline__ float bspline3d_eval(const float* __restrict__ coef, int dz, int dy, int dx, float x,
                                                float y, float z) {
    const bool out = (x < 1 || y < 1 || z < 1 || x >= dx - 2 || y >= dy - 2 || z >= dz - 2 || isnan(x) || isnan(y) ||
                      isnan(z));
    const int xi = out ? 1 : (int)floorf(x), yi = out ? 1 : (int)floorf(y), zi = out ? 1 : (int)floorf(z);
    const float fx = x - (float)xi, fy = y - (float)yi, fz = z - (float)zi;
    const float bx0 = basis0(fx), bx1 = basis1(fx), bx2 = basis2(fx), bx3 = basis3(fx);
    const float by[4] = {basis0(fy), basis1(fy), basis2(fy), basis3(fy)};
    const float bz[4] = {basis0(fz), basis1(fz), basis2(fz), basis3(fz)};
    const float* __restrict__ base = coef + ((size_t)(zi - 1) * dy + (yi - 1)) * dx + (xi - 1);
    float sum_y[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        float sum_x[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float4u row = *reinterpret_cast<const float4u*>(base + ((size_t)i * dy + j) * dx);
            sum_x[j] = taps4(bx0, bx1, bx2, bx3, row.x, row.y, row.z, row.w);
        }
        sum_y[i] = taps4(by[0], by[1], by[2], by[3], sum_x[0], sum_x[1], sum_x[2], sum_x[3]);
    }
    const float v = taps4(bz[0], bz[1], bz[2], bz[3], sum_y[0], sum_y[1], sum_y[2], sum_y[3]);
    return out ? -1.f : v;
}

// The same evaluation with the 64 coefficients taken from the staged box: `win` is the box in
// LDS, (ox, oy, oz) its origin in the volume, nx / nxy its row and plane pitches.  Same range
// rule, same weights, same order of operations as bspline3d_eval: identical bits.
// PX > 0: the row pitch is the compile-time constant PX (nx == PX), so the 16 taps of a plane are immediate
// offsets of ONE address instead of 12 more address computations.
template <int PX>
__device__ __forceinline__ float bspline3d_eval_lds(const float* __restrict__ win, int ox, int oy, int oz, int nx_rt, int nxy,
                                                    int dz, int dy, int dx, float x, float y, float z) {
    const int nx = (OC_ABLATE & 32) ? 33 : (PX ? PX : nx_rt);  // ablation 32: a pitch of 33 reads garbage, but free of bank conflicts
    const bool out = (x < 1 || y < 1 || z < 1 || x >= dx - 2 || y >= dy - 2 || z >= dz - 2 || isnan(x) || isnan(y) ||
                      isnan(z));
    const int xi = out ? ox + 1 : (int)floorf(x), yi = out ? oy + 1 : (int)floorf(y), zi = out ? oz + 1 : (int)floorf(z);
    // the x and y weights are evaluated as packed pairs (same operations per component), z scalar
    const f2 fxy = mk2(x, y) - mk2((float)xi, (float)yi);
    const float fz = z - (float)zi;
    const f2 b0 = basis0(fxy), b1 = basis1(fxy), b2 = basis2(fxy), b3 = basis3(fxy);
    const float bx0 = b0.x, bx1 = b1.x, bx2 = b2.x, bx3 = b3.x;
    const float by[4] = {b0.y, b1.y, b2.y, b3.y};
    const float bz[4] = {basis0(fz), basis1(fz), basis2(fz), basis3(fz)};
    const float* __restrict__ base = win + ((zi - 1 - oz) * nxy + (yi - 1 - oy) * nx + (xi - 1 - ox));
    float sum_y[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        float sum_x[4];
        // LDS byte address of the plane; with a compile-time pitch it is made opaque to the optimiser so that the 16
        // taps stay "address + constant" and fold into the offset fields of the ds_read instructions
        typedef const float __attribute__((address_space(3))) * lds_cfp;
        lds_cfp pl = (lds_cfp)(base + i * nxy);
        if constexpr (PX != 0) asm volatile("" : "+v"(pl));
#pragma unroll
        for (int j = 0; j < 4; j++) {
            lds_cfp row = pl + j * nx;
            sum_x[j] = taps4(bx0, bx1, bx2, bx3, row[0], row[1], row[2], row[3]);
        }
        sum_y[i] = taps4(by[0], by[1], by[2], by[3], sum_x[0], sum_x[1], sum_x[2], sum_x[3]);
    }
    const float v = taps4(bz[0], bz[1], bz[2], bz[3], sum_y[0], sum_y[1], sum_y[2], sum_y[3]);
    return out ? -1.f : v;
}

// walks the samples owned by one thread: s = tid, tid+1024, ... as (i = z, j = y, k = x) indices
// `off` = (i*DY + j)*DX + k, the sample's offset (in voxels) from the subvolume's first voxel inside a DZ x DY x DX
// volume, advanced incrementally (DX = DY = 0: not needed).
struct Walk3 {
    int i, j, k, s;
    int SX, SY, di, dj, dk;
    unsigned off;
    int doff, coff_k, coff_j;
    __device__ __forceinline__ Walk3(int tid, int SX_, int SY_, int first_pass = 0, int DX = 0, int DY = 0)
        : SX(SX_), SY(SY_) {
        // The walk's start state depends on the thread index and the subset shape only, so the optimiser computes ALL the
        // walks of the kernel once, ahead of the POI loop, and keeps ~10 values per walk alive across everything -- which
        // is where the kernel's scratch spills came from (78 dwords, round 2).  Hiding the thread index from it makes each
        // sweep set its walk up on the spot (two small divisions) and frees those registers.
        asm volatile("" : "+v"(tid));
        s = tid + first_pass * kBlock3d;
        const int plane = SX_ * SY_;
        i = s / plane;
        int rem = s - i * plane;
        j = rem / SX_;
        k = rem - j * SX_;
        di = kBlock3d / plane;
        rem = kBlock3d - di * plane;
        dj = rem / SX_;
        dk = rem - dj * SX_;
        off = (unsigned)((i * DY + j) * DX + k);
        doff = (di * DY + dj) * DX + dk;
        coff_k = DX - SX_;          // k wrapped: one row further, SX columns back
        coff_j = (DY - SY_) * DX;   // j wrapped: one plane further, SY rows back
    }
    __device__ __forceinline__ void next() {
        s += kBlock3d;
        k += dk;
        off += doff;
        const bool ck = k >= SX;
        k = ck ? k - SX : k;
        off += ck ? coff_k : 0;
        j += dj + (ck ? 1 : 0);
        const bool cj = j >= SY;
        j = cj ? j - SY : j;
        off += cj ? coff_j : 0;
        i += di + (cj ? 1 : 0);
    }
};

// One sample of a thread's walk, recorded so that a batch of samples can be loaded before any of them is used:
// its offset inside the subvolume's box of the volume and its local coordinates (small integers, exact as floats).
struct WalkPoint {
    unsigned off;
    float x, y, z;
};
// The streaming sweeps of the kernel (reference statistics, Hessian, target norm, numerator) read every sample once
// from the volumes and do little arithmetic on it: they are LATENCY bound -- with one load in flight per wave the
// whole chip keeps ~1 MB in flight, i.e. ~1 TB/s at the ~1 us a volume read takes under load (round-2 timeline,
// tools/ablate_icgn3d.sh: 58 % of the kernel's time went here, ~1.7 k cycles per sample and thread).  sweep_batched
// therefore records B walk states first, issues the B samples' loads as independent instructions, and only then
// consumes them -- in sample order, so every per-thread sum keeps its increasing-s association (bit-identical).
// `cnt` = number of samples the thread owns (no per-sample bounds test inside a batch); load(point, s) and
// use(point, loaded, s) receive the sample index s = tid + 512 * m.
template <int B, class Load, class Use>
__device__ __forceinline__ void sweep_batched(Walk3& w, int rx, int ry, int rz, int cnt, Load&& load, Use&& use) {
    int done = 0;
    auto point = [&]() {
        const WalkPoint q = {w.off, (float)(w.k - rx), (float)(w.j - ry), (float)(w.i - rz)};
        w.next();
        return q;
    };
#pragma unroll 1
    for (; done + B <= cnt; done += B) {
        const int s0 = w.s;
        WalkPoint p[B];
#pragma unroll
        for (int u = 0; u < B; u++) p[u] = point();
        decltype(load(p[0], 0)) v[B];
#pragma unroll
        for (int u = 0; u < B; u++) v[u] = load(p[u], s0 + u * kBlock3d);
#pragma unroll
        for (int u = 0; u < B; u++) use(p[u], v[u], s0 + u * kBlock3d);
    }
#pragma unroll 1
    for (; done < cnt; done++) {
        const int s0 = w.s;
        const WalkPoint q = point();
        use(q, load(q, s0), s0);
    }
}

}  // namespace ochip
