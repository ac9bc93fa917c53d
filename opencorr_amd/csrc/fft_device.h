// fft_device.h -- small in-register FFTs shared by the fused FFTCC kernels (fftcc2d_fused.hip, fftcc3d_fused.hip).
//
// Complex numbers are 2-wide vectors (re, im): additions and the twiddle products run on the packed-fp32 pipe
// (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32), one instruction per complex operation.  FMA contraction is allowed
// inside the butterflies only (the FFT is not part of the bit-exact contract: the float ZNCC is compared within 1e-5,
// the integer peak position must be -- and is -- identical).
#pragma once

#include <hip/hip_runtime.h>

namespace ochip {
namespace fftdev {

__device__ constexpr float kCos16[8] = {1.000000000e+00f, 9.238795325e-01f, 7.071067812e-01f, 3.826834324e-01f, 6.123233996e-17f, -3.826834324e-01f, -7.071067812e-01f, -9.238795325e-01f};
__device__ constexpr float kSin16[8] = {0.000000000e+00f, 3.826834324e-01f, 7.071067812e-01f, 9.238795325e-01f, 1.000000000e+00f, 9.238795325e-01f, 7.071067812e-01f, 3.826834324e-01f};
__device__ constexpr float kCos32[16] = {1.000000000e+00f, 9.807852804e-01f, 9.238795325e-01f, 8.314696123e-01f, 7.071067812e-01f, 5.555702330e-01f, 3.826834324e-01f, 1.950903220e-01f, 6.123233996e-17f, -1.950903220e-01f, -3.826834324e-01f, -5.555702330e-01f, -7.071067812e-01f, -8.314696123e-01f, -9.238795325e-01f, -9.807852804e-01f};
__device__ constexpr float kSin32[16] = {0.000000000e+00f, 1.950903220e-01f, 3.826834324e-01f, 5.555702330e-01f, 7.071067812e-01f, 8.314696123e-01f, 9.238795325e-01f, 9.807852804e-01f, 1.000000000e+00f, 9.807852804e-01f, 9.238795325e-01f, 8.314696123e-01f, 7.071067812e-01f, 5.555702330e-01f, 3.826834324e-01f, 1.950903220e-01f};

// complex numbers as 2-wide vectors (re, im): additions and the twiddle products then run on the packed-fp32
// pipe (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32), one instruction per complex operation
typedef float c2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ c2 mkc(float re, float im) {
    c2 r = {re, im};
    return r;
}

// d * exp(-/+ i*angle) with (c, s) = (cos, sin) of the angle; INV selects the + sign:
// forward (d.x c + d.y s, d.y c - d.x s), inverse (d.x c - d.y s, d.y c + d.x s)
template <bool INV>
__device__ __forceinline__ c2 cmul_tw(c2 d, float c, float s) {
#pragma clang fp contract(fast)
    return d * c + d.yx * (INV ? mkc(-s, s) : mkc(s, -s));
}

// 16-point FFT in registers on v[OFF .. OFF+16): radix-2 decimation in frequency, X[k] ends in v[OFF + bitrev4(k)]
template <bool INV, int OFF, int LEN>
__device__ __forceinline__ void fft16_at(c2 (&v)[LEN]) {
#pragma unroll
    for (int span = 8; span >= 1; span >>= 1) {
#pragma unroll
        for (int i0 = 0; i0 < 16; i0++) {
            if (i0 & span) continue;
            const int i1 = i0 + span;
            const int m = (i0 & (span - 1)) * (8 / span);  // twiddle W16^m
            const c2 a = v[OFF + i0], b = v[OFF + i1];
            v[OFF + i0] = a + b;
            const c2 d = a - b;
            if (m == 0) v[OFF + i1] = d;
            else if (m == 4) v[OFF + i1] = INV ? mkc(-d.y, d.x) : mkc(d.y, -d.x);
            else v[OFF + i1] = cmul_tw<INV>(d, kCos16[m], kSin16[m]);
        }
    }
}

__device__ constexpr int bitrev4(int k) { return ((k & 1) << 3) | ((k & 2) << 1) | ((k & 4) >> 1) | ((k & 8) >> 3); }

template <bool INV>
__device__ __forceinline__ void fft16(c2 (&v)[16]) {
    fft16_at<INV, 0, 16>(v);
}

__device__ constexpr int bitrev5(int k) { return ((k & 1) << 4) | bitrev4(k >> 1); }

// 32-point FFT in registers: one radix-2 decimation-in-frequency stage, then two 16-point transforms;
// X[k] ends in v[bitrev5(k)]
template <bool INV>
__device__ __forceinline__ void fft32(c2 (&v)[32]) {
#pragma unroll
    for (int n = 0; n < 16; n++) {
        const c2 a = v[n], b = v[n + 16];
        v[n] = a + b;
        const c2 d = a - b;
        if (n == 0) v[n + 16] = d;
        else if (n == 8) v[n + 16] = INV ? mkc(-d.y, d.x) : mkc(d.y, -d.x);
        else v[n + 16] = cmul_tw<INV>(d, kCos32[n], kSin32[n]);
    }
    fft16_at<INV, 0, 32>(v);
    fft16_at<INV, 16, 32>(v);
}

}  // namespace fftdev
}  // namespace ochip
