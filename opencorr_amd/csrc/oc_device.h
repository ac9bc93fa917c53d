// oc_device.h -- device-side helpers shared by the gfx950 kernels.
//
// Arithmetic contract: IEEE float32, the compiler contracts
// nothing (the library is built with -ffp-contract=off), correctly
// rounded division and sqrt (hipcc default), no fast-math.  Reductions over the
// samples of a subset use ONE fixed association: sample s is owned by lane
// (s % P) which adds its samples in increasing s, and the P partials are
// combined by an xor butterfly with ascending offsets 1, 2, 4, ... P/2.  The CPU
// oracle implements the same association (OC_ORDER_LANES) so results are
// bit-identical.
//
// Two arithmetic modes exist for the ICGN / IC-LM solvers, selected per translation unit by OC_FMA (icgn2d.hip and
// icgn3d.hip are compiled twice; oc_hip_set_tuning("arith_fma") picks the build at run time):
//   OC_FMA = 0 ("sep")  every multiply and every add rounds on its own -- the reference built for baseline x86-64;
//                       oracle orders OC_ORDER_LANES / OC_ORDER_SEQ;
//   OC_FMA = 1 ("fma")  every PER-SAMPLE multiply-add is ONE explicit fused multiply-add (`mad` below -> v_fma_f32 /
//                       v_pk_fma_f32), at exactly the sites oracle/oc_oracle.h lists under "Arithmetic contract" -- the
//                       contraction a compiler with FMA hardware makes of the reference's source expressions; oracle
//                       orders OC_ORDER_LANES_FMA / OC_ORDER_SEQ_FMA.  IEEE fusedMultiplyAdd is defined bit for bit, so
//                       GPU == oracle stays exact; the per-POI dense algebra is NOT fused in either mode.
// Helpers whose bodies depend on OC_FMA are __device__ __forceinline__ only (no symbol is ever emitted for them, and
// every translation unit is its own device code object), and the kernels built from them live in ochip::sep / ochip::fma.
#pragma once

#include <hip/hip_runtime.h>

#ifndef OC_FMA
#define OC_FMA 0
#endif
#if OC_FMA
#define OC_ARITH fma
#else
#define OC_ARITH sep
#endif

namespace ochip {

constexpr int kWave = 64;

// Two floats that travel through the packed-fp32 pipe (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32: two IEEE operations
// per issue slot, each rounded on its own, so results are those of the scalar instructions).
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 mk2(float a, float b) {
    f2 r = {a, b};
    return r;
}
__device__ __forceinline__ f2 splat2(float a) { return mk2(a, a); }

// a * b + c under the translation unit's arithmetic mode: two roundings ("sep") or one ("fma")
__device__ __forceinline__ float mad(float a, float b, float c) {
#if OC_FMA
    return __builtin_fmaf(a, b, c);
#else
    return a * b + c;
#endif
}
__device__ __forceinline__ f2 mad(f2 a, f2 b, f2 c) {
#if OC_FMA
    return __builtin_elementwise_fma(a, b, c);
#else
    return a * b + c;
#endif
}
__device__ __forceinline__ f2 mad(float a, f2 b, f2 c) { return mad(splat2(a), b, c); }
__device__ __forceinline__ f2 mad(f2 a, float b, f2 c) { return mad(a, splat2(b), c); }

// offsets inside a POI2D / POI3D record, in floats (src/oc_poi.h:25-222)
namespace poi2d {
constexpr int X = 0, Y = 1, U = 2, UX = 3, UY = 4, UXX = 5, UXY = 6, UYY = 7, V = 8, VX = 9, VY = 10, VXX = 11,
              VXY = 12, VYY = 13, U0 = 14, V0 = 15, ZNCC = 16, ITER = 17, CONV = 18, FEATURE = 19, SRX = 23, SRY = 24;
constexpr int FLOATS = 25;
}  // namespace poi2d
namespace poi3d {
constexpr int X = 0, Y = 1, Z = 2, P = 3, U = 3, V = 7, W = 11, U0 = 15, V0 = 16, W0 = 17, ZNCC = 18, ITER = 19,
              CONV = 20, SRX = 28, SRY = 29, SRZ = 30;
constexpr int FLOATS = 31;
}  // namespace poi3d

// broadcast lane `src`'s value as a wave-uniform value
__device__ __forceinline__ float wave_bcast(float v, int src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
}

// DPP lane permutation inside a row of 16 lanes (no LDS crossbar, unlike ds_bpermute)
template <int CTRL>
__device__ __forceinline__ float dpp_perm(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

// xor-butterfly all-reduce over the 64 lanes of a wave, ascending offsets 1, 2, 4, ... 32.
// a + b is commutative in IEEE arithmetic, so both partners compute the same bits and
// every lane ends with the same value.  Levels 1 and 2 are quad permutes; after them
// the value is uniform inside each quad, so pairing lane i with 7-i (row_half_mirror)
// adds the same two quad sums as pairing it with i^4, and likewise row_mirror for i^8.
// The four 16-lane row sums are then combined as (r0 + r1) + (r2 + r3), which is what
// offsets 16 and 32 compute.  Bit-identical to the __shfl_xor butterfly, 4 DPP adds +
// 4 v_readlane instead of 6 ds_bpermute round trips.
__device__ __forceinline__ float wave_allreduce_sum(float v) {
    v = v + dpp_perm<0xB1>(v);   // quad_perm [1,0,3,2]  == xor 1
    v = v + dpp_perm<0x4E>(v);   // quad_perm [2,3,0,1]  == xor 2
    v = v + dpp_perm<0x141>(v);  // row_half_mirror       ~= xor 4
    v = v + dpp_perm<0x140>(v);  // row_mirror            ~= xor 8
    const float r0 = wave_bcast(v, 0), r1 = wave_bcast(v, 16), r2 = wave_bcast(v, 32), r3 = wave_bcast(v, 48);
    return (r0 + r1) + (r2 + r3);
}

// K sums at once.  On return every v[k] is the wave-wide sum of the lanes' v[k], wave-uniform, with exactly the
// association of wave_allreduce_sum (xor butterfly, offsets 1, 2, 4, 8, 16, 32: every level adds the same two partial
// sums, and a + b is commutative).  What changes is where the work is done: after the quad levels the values are
// dealt out -- level 4 keeps the first half of the list in the even quads and the second half in the odd quads
// (each lane adds its partner's copy of the half it keeps), level 8 splits again between the low and the high eight
// lanes of a row -- so the row levels (LDS-crossbar permutes) run on a quarter of the registers, and one v_readlane
// per value fetches the total from the lane class that holds it.  ~6 K instead of 11 K instructions.
template <int K>
__device__ __forceinline__ void wave_allreduce_sum_multi(float (&v)[K], int lane) {
    constexpr int H = (K + 1) / 2, H2 = (H + 1) / 2;
#pragma unroll
    for (int k = 0; k < K; k++) {
        v[k] = v[k] + dpp_perm<0xB1>(v[k]);  // xor 1
        v[k] = v[k] + dpp_perm<0x4E>(v[k]);  // xor 2
    }
    const bool odd_quad = (lane & 4) != 0, high8 = (lane & 8) != 0;
    float d[H];
#pragma unroll
    for (int i = 0; i < H; i++) {
        const float a = v[i], b = (i + H < K) ? v[(i + H) % K] : 0.f;
        const float keep = odd_quad ? b : a, give = odd_quad ? a : b;
        d[i] = keep + dpp_perm<0x141>(give);  // row_half_mirror: the neighbouring quad (xor 4 on quad-uniform values)
    }
    float e[H2];
#pragma unroll
    for (int j = 0; j < H2; j++) {
        const float a = d[j], b = (j + H2 < H) ? d[(j + H2) % H] : 0.f;
        const float keep = high8 ? b : a, give = high8 ? a : b;
        e[j] = keep + dpp_perm<0x128>(give);  // row_ror:8 == xor 8: same quad parity, i.e. the same half of the list
    }
#pragma unroll
    for (int j = 0; j < H2; j++) {
        e[j] = e[j] + __shfl_xor(e[j], 16, kWave);
        e[j] = e[j] + __shfl_xor(e[j], 32, kWave);
    }
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int i = k % H, j = i % H2;                            // register that ends up holding value k
        const int holder = (k >= H ? 4 : 0) + (i >= H2 ? 8 : 0);    // odd quad / high eight of row 0
        v[k] = wave_bcast(e[j], holder);
    }
}

// The same reduction, but the K totals are filed into LDS (dst[0..K)) by the four lanes that hold them after the row
// levels instead of being broadcast to the wave: for a consumer in another wave (icgn2d.hip, cooperative inverse).
template <int K>
__device__ __forceinline__ void wave_reduce_sum_multi_to_lds(float (&v)[K], int lane, float* __restrict__ dst) {
    constexpr int H = (K + 1) / 2, H2 = (H + 1) / 2;
#pragma unroll
    for (int k = 0; k < K; k++) {
        v[k] = v[k] + dpp_perm<0xB1>(v[k]);  // xor 1
        v[k] = v[k] + dpp_perm<0x4E>(v[k]);  // xor 2
    }
    const bool odd_quad = (lane & 4) != 0, high8 = (lane & 8) != 0;
    float d[H];
#pragma unroll
    for (int i = 0; i < H; i++) {
        const float a = v[i], b = (i + H < K) ? v[(i + H) % K] : 0.f;
        const float keep = odd_quad ? b : a, give = odd_quad ? a : b;
        d[i] = keep + dpp_perm<0x141>(give);
    }
    float e[H2];
#pragma unroll
    for (int j = 0; j < H2; j++) {
        const float a = d[j], b = (j + H2 < H) ? d[(j + H2) % H] : 0.f;
        const float keep = high8 ? b : a, give = high8 ? a : b;
        e[j] = keep + dpp_perm<0x128>(give);
    }
#pragma unroll
    for (int j = 0; j < H2; j++) {
        e[j] = e[j] + __shfl_xor(e[j], 16, kWave);
        e[j] = e[j] + __shfl_xor(e[j], 32, kWave);
    }
    // lane 0 holds values 0 .. H2-1, lane 8 values H2 .. H-1, lane 4 values H .. H+H2-1, lane 12 the rest
    const int i0 = high8 ? H2 : 0, k0 = (odd_quad ? H : 0) + i0;
    if ((lane & ~12) == 0) {
#pragma unroll
        for (int j = 0; j < H2; j++)
            if (i0 + j < H && k0 + j < K) dst[k0 + j] = e[j];
    }
}

__device__ __forceinline__ bool wave_any(bool p) { return __ballot(p) != 0ull; }

}  // namespace ochip
