// fft_device.h -- small in-register FFTs shared by the fused FFTCC kernels (fftcc2d_fused.hip, fftcc3d_fused.hip).
//
// Complex numbers are 2-wide vectors (re, im): additions and the twiddle products run on the packed-fp32 pipe
// (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32), one instruction per complex operation.  FMA contraction is allowed
// inside the butterflies only (the FFT is not part of the bit-exact contract: the float ZNCC is compared within 1e-5,
// the integer peak position must be -- and is -- identical).
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

// The mixed-radix transforms are plain arithmetic on register arrays: they also compile for the HOST, where
// tests/cpp/fft_host_check.hip runs every supported size against a double-precision DFT (no GPU needed).
#define OC_FFT_FN __host__ __device__ __forceinline__

namespace ochip {
namespace fftdev {

__device__ constexpr float kCos16[8] = {1.000000000e+00f, 9.238795325e-01f, 7.071067812e-01f, 3.826834324e-01f, 6.123233996e-17f, -3.826834324e-01f, -7.071067812e-01f, -9.238795325e-01f};
__device__ constexpr float kSin16[8] = {0.000000000e+00f, 3.826834324e-01f, 7.071067812e-01f, 9.238795325e-01f, 1.000000000e+00f, 9.238795325e-01f, 7.071067812e-01f, 3.826834324e-01f};
__device__ constexpr float kCos32[16] = {1.000000000e+00f, 9.807852804e-01f, 9.238795325e-01f, 8.314696123e-01f, 7.071067812e-01f, 5.555702330e-01f, 3.826834324e-01f, 1.950903220e-01f, 6.123233996e-17f, -1.950903220e-01f, -3.826834324e-01f, -5.555702330e-01f, -7.071067812e-01f, -8.314696123e-01f, -9.238795325e-01f, -9.807852804e-01f};
__device__ constexpr float kSin32[16] = {0.000000000e+00f, 1.950903220e-01f, 3.826834324e-01f, 5.555702330e-01f, 7.071067812e-01f, 8.314696123e-01f, 9.238795325e-01f, 9.807852804e-01f, 1.000000000e+00f, 9.807852804e-01f, 9.238795325e-01f, 8.314696123e-01f, 7.071067812e-01f, 5.555702330e-01f, 3.826834324e-01f, 1.950903220e-01f};

// complex numbers as 2-wide vectors (re, im): additions and the twiddle products then run on the packed-fp32
// pipe (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32), one instruction per complex operation
typedef float c2 __attribute__((ext_vector_type(2)));
OC_FFT_FN c2 mkc(float re, float im) {
    c2 r = {re, im};
    return r;
}

// d * exp(-/+ i*angle) with (c, s) = (cos, sin) of the angle; INV selects the + sign:
// forward (d.x c + d.y s, d.y c - d.x s), inverse (d.x c - d.y s, d.y c + d.x s)
template <bool INV>
OC_FFT_FN c2 cmul_tw(c2 d, float c, float s) {
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


// ---------------------------------------------------------------------------------------------------------------
// Mixed-radix transforms for window sides other than 32 (fftcc2d_fusedn.hip): N = product of 2, 3, 4, 5 and -- round 4 --
// of any odd primes up to 31 (7, 11, 13 for the sides 14, 22, 26, 28, 42, 44, 52, 56; 17 ... 31 for 34, 38, 46, 58, 62).
// One decimation-in-frequency step per factor R (N = R * M): the R elements n2 + M*j are transformed, twiddled by
// W_N^(n2*s) and left in place; the R blocks of M elements are then transformed recursively.  Everything is unrolled
// at compile time on register arrays; X[k] ends in v[fft_pos(N, k)].
// ---------------------------------------------------------------------------------------------------------------
constexpr int fft_smallest_odd_factor(int n) {
    for (int p = 3; p * p <= n; p += 2)
        if (n % p == 0) return p;
    return n;
}
constexpr int fft_radix(int n) { return n % 4 == 0 ? 4 : n % 2 == 0 ? 2 : fft_smallest_odd_factor(n); }
constexpr int kFftMaxPrime = 31;
constexpr int fft_pos(int n, int idx) {
    int pos = 0;
    while (n > 1) {
        const int r = fft_radix(n), m = n / r;
        pos += (idx % r) * m;
        idx /= r;
        n = m;
    }
    return pos;
}

// compile-time loop: f(std::integral_constant<int, B>{}), ..., f(std::integral_constant<int, E-1>{}).  Unlike
// `#pragma unroll`, the loop index is a constant EXPRESSION inside the body, so fft_pos() and the twiddle indices are
// evaluated by the front end (a run-time fft_pos() would turn the register arrays into scratch memory).
template <int B, int E, class F>
OC_FFT_FN void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

// cos / sin of 2*pi*k / N.  The sizes of rounds 1-3 keep their generated tables below (tools/gen_twiddles.py: the bits
// those kernels were validated with); every other size takes the primary template, whose table the front end computes:
// double-precision Taylor series on an argument reduced to [0, pi/4] by the octant symmetries (error < 1e-16, then ONE
// rounding to float -- the same values the generated tables hold).
namespace twdetail {
constexpr double kPi = 3.14159265358979323846264338327950288;
constexpr double sin_small(double x) {  // |x| <= pi/4
    const double x2 = x * x;
    double term = x, sum = x;
    for (int n = 1; n <= 12; n++) {
        term = -term * x2 / ((2.0 * n) * (2.0 * n + 1.0));
        sum += term;
    }
    return sum;
}
constexpr double cos_small(double x) {
    const double x2 = x * x;
    double term = 1.0, sum = 1.0;
    for (int n = 1; n <= 12; n++) {
        term = -term * x2 / ((2.0 * n - 1.0) * (2.0 * n));
        sum += term;
    }
    return sum;
}
// (cos, sin) of 2*pi*k/n with exact symmetries: k is reduced in integers, so cos(pi/2) etc. come out as exact zeros
// only where the generated tables have their 6e-17 -- irrelevant after rounding of the products, but the sign
// structure (which outputs are exact mirror images) is what keeps X(N-k) = conj X(k) tight for real inputs
constexpr double cos2pi(int k, int n) {
    k %= n;
    if (k < 0) k += n;
    if (2 * k > n) k = n - k;          // cos is even about pi
    if (4 * k > n) return -cos2pi(n - 2 * k, 2 * n) ;  // cos(x) = -cos(pi - x): pi - 2 pi k / n = 2 pi (n - 2k) / (2n)
    if (8 * k > n) return sin_small(2.0 * kPi * (n - 4 * k) / (4.0 * n));  // cos(x) = sin(pi/2 - x)
    return cos_small(2.0 * kPi * k / n);
}
constexpr double sin2pi(int k, int n) {
    k %= n;
    if (k < 0) k += n;
    if (2 * k > n) return -sin2pi(n - k, n);
    if (4 * k > n) return sin2pi(n - 2 * k, 2 * n);    // sin(x) = sin(pi - x)
    if (8 * k > n) return cos_small(2.0 * kPi * (n - 4 * k) / (4.0 * n));  // sin(x) = cos(pi/2 - x)
    return sin_small(2.0 * kPi * k / n);
}
template <int N>
struct Table {
    float v[N];
    constexpr float operator[](int i) const { return v[i]; }
};
template <int N>
constexpr Table<N> make_cos() {
    Table<N> t{};
    for (int k = 0; k < N; k++) t.v[k] = (float)cos2pi(k, N);
    return t;
}
template <int N>
constexpr Table<N> make_sin() {
    Table<N> t{};
    for (int k = 0; k < N; k++) t.v[k] = (float)sin2pi(k, N);
    return t;
}
}  // namespace twdetail

template <int N>
struct Twiddle {
    static constexpr twdetail::Table<N> c = twdetail::make_cos<N>();
    static constexpr twdetail::Table<N> s = twdetail::make_sin<N>();
};

template <>
struct Twiddle<20> {
    static constexpr float c[20] = {1.0000000000e+00f, 9.5105651630e-01f, 8.0901699437e-01f, 5.8778525229e-01f, 3.0901699437e-01f, 6.1232339957e-17f, -3.0901699437e-01f, -5.8778525229e-01f, -8.0901699437e-01f, -9.5105651630e-01f, -1.0000000000e+00f, -9.5105651630e-01f, -8.0901699437e-01f, -5.8778525229e-01f, -3.0901699437e-01f, -1.8369701987e-16f, 3.0901699437e-01f, 5.8778525229e-01f, 8.0901699437e-01f, 9.5105651630e-01f};
    static constexpr float s[20] = {0.0000000000e+00f, 3.0901699437e-01f, 5.8778525229e-01f, 8.0901699437e-01f, 9.5105651630e-01f, 1.0000000000e+00f, 9.5105651630e-01f, 8.0901699437e-01f, 5.8778525229e-01f, 3.0901699437e-01f, 1.2246467991e-16f, -3.0901699437e-01f, -5.8778525229e-01f, -8.0901699437e-01f, -9.5105651630e-01f, -1.0000000000e+00f, -9.5105651630e-01f, -8.0901699437e-01f, -5.8778525229e-01f, -3.0901699437e-01f};
};

template <>
struct Twiddle<36> {
    static constexpr float c[36] = {1.0000000000e+00f, 9.8480775301e-01f, 9.3969262079e-01f, 8.6602540378e-01f, 7.6604444312e-01f, 6.4278760969e-01f, 5.0000000000e-01f, 3.4202014333e-01f, 1.7364817767e-01f, 6.1232339957e-17f, -1.7364817767e-01f, -3.4202014333e-01f, -5.0000000000e-01f, -6.4278760969e-01f, -7.6604444312e-01f, -8.6602540378e-01f, -9.3969262079e-01f, -9.8480775301e-01f, -1.0000000000e+00f, -9.8480775301e-01f, -9.3969262079e-01f, -8.6602540378e-01f, -7.6604444312e-01f, -6.4278760969e-01f, -5.0000000000e-01f, -3.4202014333e-01f, -1.7364817767e-01f, -1.8369701987e-16f, 1.7364817767e-01f, 3.4202014333e-01f, 5.0000000000e-01f, 6.4278760969e-01f, 7.6604444312e-01f, 8.6602540378e-01f, 9.3969262079e-01f, 9.8480775301e-01f};
    static constexpr float s[36] = {0.0000000000e+00f, 1.7364817767e-01f, 3.4202014333e-01f, 5.0000000000e-01f, 6.4278760969e-01f, 7.6604444312e-01f, 8.6602540378e-01f, 9.3969262079e-01f, 9.8480775301e-01f, 1.0000000000e+00f, 9.8480775301e-01f, 9.3969262079e-01f, 8.6602540378e-01f, 7.6604444312e-01f, 6.4278760969e-01f, 5.0000000000e-01f, 3.4202014333e-01f, 1.7364817767e-01f, 1.2246467991e-16f, -1.7364817767e-01f, -3.4202014333e-01f, -5.0000000000e-01f, -6.4278760969e-01f, -7.6604444312e-01f, -8.6602540378e-01f, -9.3969262079e-01f, -9.8480775301e-01f, -1.0000000000e+00f, -9.8480775301e-01f, -9.3969262079e-01f, -8.6602540378e-01f, -7.6604444312e-01f, -6.4278760969e-01f, -5.0000000000e-01f, -3.4202014333e-01f, -1.7364817767e-01f};
};

template <>
struct Twiddle<24> {
    static constexpr float c[24] = {1.0000000000e+00f, 9.6592582629e-01f, 8.6602540378e-01f, 7.0710678119e-01f, 5.0000000000e-01f, 2.5881904510e-01f, 6.1232339957e-17f, -2.5881904510e-01f, -5.0000000000e-01f, -7.0710678119e-01f, -8.6602540378e-01f, -9.6592582629e-01f, -1.0000000000e+00f, -9.6592582629e-01f, -8.6602540378e-01f, -7.0710678119e-01f, -5.0000000000e-01f, -2.5881904510e-01f, -1.8369701987e-16f, 2.5881904510e-01f, 5.0000000000e-01f, 7.0710678119e-01f, 8.6602540378e-01f, 9.6592582629e-01f};
    static constexpr float s[24] = {0.0000000000e+00f, 2.5881904510e-01f, 5.0000000000e-01f, 7.0710678119e-01f, 8.6602540378e-01f, 9.6592582629e-01f, 1.0000000000e+00f, 9.6592582629e-01f, 8.6602540378e-01f, 7.0710678119e-01f, 5.0000000000e-01f, 2.5881904510e-01f, 1.2246467991e-16f, -2.5881904510e-01f, -5.0000000000e-01f, -7.0710678119e-01f, -8.6602540378e-01f, -9.6592582629e-01f, -1.0000000000e+00f, -9.6592582629e-01f, -8.6602540378e-01f, -7.0710678119e-01f, -5.0000000000e-01f, -2.5881904510e-01f};
};

template <>
struct Twiddle<30> {
    static constexpr float c[30] = {1.0000000000e+00f, 9.7814760073e-01f, 9.1354545764e-01f, 8.0901699437e-01f, 6.6913060636e-01f, 5.0000000000e-01f, 3.0901699437e-01f, 1.0452846327e-01f, -1.0452846327e-01f, -3.0901699437e-01f, -5.0000000000e-01f, -6.6913060636e-01f, -8.0901699437e-01f, -9.1354545764e-01f, -9.7814760073e-01f, -1.0000000000e+00f, -9.7814760073e-01f, -9.1354545764e-01f, -8.0901699437e-01f, -6.6913060636e-01f, -5.0000000000e-01f, -3.0901699437e-01f, -1.0452846327e-01f, 1.0452846327e-01f, 3.0901699437e-01f, 5.0000000000e-01f, 6.6913060636e-01f, 8.0901699437e-01f, 9.1354545764e-01f, 9.7814760073e-01f};
    static constexpr float s[30] = {0.0000000000e+00f, 2.0791169082e-01f, 4.0673664308e-01f, 5.8778525229e-01f, 7.4314482548e-01f, 8.6602540378e-01f, 9.5105651630e-01f, 9.9452189537e-01f, 9.9452189537e-01f, 9.5105651630e-01f, 8.6602540378e-01f, 7.4314482548e-01f, 5.8778525229e-01f, 4.0673664308e-01f, 2.0791169082e-01f, 5.6655388976e-16f, -2.0791169082e-01f, -4.0673664308e-01f, -5.8778525229e-01f, -7.4314482548e-01f, -8.6602540378e-01f, -9.5105651630e-01f, -9.9452189537e-01f, -9.9452189537e-01f, -9.5105651630e-01f, -8.6602540378e-01f, -7.4314482548e-01f, -5.8778525229e-01f, -4.0673664308e-01f, -2.0791169082e-01f};
};

template <>
struct Twiddle<40> {
    static constexpr float c[40] = {1.0000000000e+00f, 9.8768834060e-01f, 9.5105651630e-01f, 8.9100652419e-01f, 8.0901699437e-01f, 7.0710678119e-01f, 5.8778525229e-01f, 4.5399049974e-01f, 3.0901699437e-01f, 1.5643446504e-01f, 6.1232339957e-17f, -1.5643446504e-01f, -3.0901699437e-01f, -4.5399049974e-01f, -5.8778525229e-01f, -7.0710678119e-01f, -8.0901699437e-01f, -8.9100652419e-01f, -9.5105651630e-01f, -9.8768834060e-01f, -1.0000000000e+00f, -9.8768834060e-01f, -9.5105651630e-01f, -8.9100652419e-01f, -8.0901699437e-01f, -7.0710678119e-01f, -5.8778525229e-01f, -4.5399049974e-01f, -3.0901699437e-01f, -1.5643446504e-01f, -1.8369701987e-16f, 1.5643446504e-01f, 3.0901699437e-01f, 4.5399049974e-01f, 5.8778525229e-01f, 7.0710678119e-01f, 8.0901699437e-01f, 8.9100652419e-01f, 9.5105651630e-01f, 9.8768834060e-01f};
    static constexpr float s[40] = {0.0000000000e+00f, 1.5643446504e-01f, 3.0901699437e-01f, 4.5399049974e-01f, 5.8778525229e-01f, 7.0710678119e-01f, 8.0901699437e-01f, 8.9100652419e-01f, 9.5105651630e-01f, 9.8768834060e-01f, 1.0000000000e+00f, 9.8768834060e-01f, 9.5105651630e-01f, 8.9100652419e-01f, 8.0901699437e-01f, 7.0710678119e-01f, 5.8778525229e-01f, 4.5399049974e-01f, 3.0901699437e-01f, 1.5643446504e-01f, 1.2246467991e-16f, -1.5643446504e-01f, -3.0901699437e-01f, -4.5399049974e-01f, -5.8778525229e-01f, -7.0710678119e-01f, -8.0901699437e-01f, -8.9100652419e-01f, -9.5105651630e-01f, -9.8768834060e-01f, -1.0000000000e+00f, -9.8768834060e-01f, -9.5105651630e-01f, -8.9100652419e-01f, -8.0901699437e-01f, -7.0710678119e-01f, -5.8778525229e-01f, -4.5399049974e-01f, -3.0901699437e-01f, -1.5643446504e-01f};
};

template <>
struct Twiddle<48> {
    static constexpr float c[48] = {1.0000000000e+00f, 9.9144486137e-01f, 9.6592582629e-01f, 9.2387953251e-01f, 8.6602540378e-01f, 7.9335334029e-01f, 7.0710678119e-01f, 6.0876142901e-01f, 5.0000000000e-01f, 3.8268343237e-01f, 2.5881904510e-01f, 1.3052619222e-01f, 6.1232339957e-17f, -1.3052619222e-01f, -2.5881904510e-01f, -3.8268343237e-01f, -5.0000000000e-01f, -6.0876142901e-01f, -7.0710678119e-01f, -7.9335334029e-01f, -8.6602540378e-01f, -9.2387953251e-01f, -9.6592582629e-01f, -9.9144486137e-01f, -1.0000000000e+00f, -9.9144486137e-01f, -9.6592582629e-01f, -9.2387953251e-01f, -8.6602540378e-01f, -7.9335334029e-01f, -7.0710678119e-01f, -6.0876142901e-01f, -5.0000000000e-01f, -3.8268343237e-01f, -2.5881904510e-01f, -1.3052619222e-01f, -1.8369701987e-16f, 1.3052619222e-01f, 2.5881904510e-01f, 3.8268343237e-01f, 5.0000000000e-01f, 6.0876142901e-01f, 7.0710678119e-01f, 7.9335334029e-01f, 8.6602540378e-01f, 9.2387953251e-01f, 9.6592582629e-01f, 9.9144486137e-01f};
    static constexpr float s[48] = {0.0000000000e+00f, 1.3052619222e-01f, 2.5881904510e-01f, 3.8268343237e-01f, 5.0000000000e-01f, 6.0876142901e-01f, 7.0710678119e-01f, 7.9335334029e-01f, 8.6602540378e-01f, 9.2387953251e-01f, 9.6592582629e-01f, 9.9144486137e-01f, 1.0000000000e+00f, 9.9144486137e-01f, 9.6592582629e-01f, 9.2387953251e-01f, 8.6602540378e-01f, 7.9335334029e-01f, 7.0710678119e-01f, 6.0876142901e-01f, 5.0000000000e-01f, 3.8268343237e-01f, 2.5881904510e-01f, 1.3052619222e-01f, 1.2246467991e-16f, -1.3052619222e-01f, -2.5881904510e-01f, -3.8268343237e-01f, -5.0000000000e-01f, -6.0876142901e-01f, -7.0710678119e-01f, -7.9335334029e-01f, -8.6602540378e-01f, -9.2387953251e-01f, -9.6592582629e-01f, -9.9144486137e-01f, -1.0000000000e+00f, -9.9144486137e-01f, -9.6592582629e-01f, -9.2387953251e-01f, -8.6602540378e-01f, -7.9335334029e-01f, -7.0710678119e-01f, -6.0876142901e-01f, -5.0000000000e-01f, -3.8268343237e-01f, -2.5881904510e-01f, -1.3052619222e-01f};
};

template <>
struct Twiddle<16> {
    static constexpr float c[16] = {1.0000000000e+00f, 9.2387953251e-01f, 7.0710678119e-01f, 3.8268343237e-01f, 6.1232339957e-17f, -3.8268343237e-01f, -7.0710678119e-01f, -9.2387953251e-01f, -1.0000000000e+00f, -9.2387953251e-01f, -7.0710678119e-01f, -3.8268343237e-01f, -1.8369701987e-16f, 3.8268343237e-01f, 7.0710678119e-01f, 9.2387953251e-01f};
    static constexpr float s[16] = {0.0000000000e+00f, 3.8268343237e-01f, 7.0710678119e-01f, 9.2387953251e-01f, 1.0000000000e+00f, 9.2387953251e-01f, 7.0710678119e-01f, 3.8268343237e-01f, 1.2246467991e-16f, -3.8268343237e-01f, -7.0710678119e-01f, -9.2387953251e-01f, -1.0000000000e+00f, -9.2387953251e-01f, -7.0710678119e-01f, -3.8268343237e-01f};
};

template <>
struct Twiddle<18> {
    static constexpr float c[18] = {1.0000000000e+00f, 9.3969262079e-01f, 7.6604444312e-01f, 5.0000000000e-01f, 1.7364817767e-01f, -1.7364817767e-01f, -5.0000000000e-01f, -7.6604444312e-01f, -9.3969262079e-01f, -1.0000000000e+00f, -9.3969262079e-01f, -7.6604444312e-01f, -5.0000000000e-01f, -1.7364817767e-01f, 1.7364817767e-01f, 5.0000000000e-01f, 7.6604444312e-01f, 9.3969262079e-01f};
    static constexpr float s[18] = {0.0000000000e+00f, 3.4202014333e-01f, 6.4278760969e-01f, 8.6602540378e-01f, 9.8480775301e-01f, 9.8480775301e-01f, 8.6602540378e-01f, 6.4278760969e-01f, 3.4202014333e-01f, 1.2246467991e-16f, -3.4202014333e-01f, -6.4278760969e-01f, -8.6602540378e-01f, -9.8480775301e-01f, -9.8480775301e-01f, -8.6602540378e-01f, -6.4278760969e-01f, -3.4202014333e-01f};
};

template <>
struct Twiddle<50> {
    static constexpr float c[50] = {1.0000000000e+00f, 9.9211470131e-01f, 9.6858316113e-01f, 9.2977648589e-01f, 8.7630668004e-01f, 8.0901699437e-01f, 7.2896862742e-01f, 6.3742398975e-01f, 5.3582679498e-01f, 4.2577929157e-01f, 3.0901699437e-01f, 1.8738131459e-01f, 6.2790519529e-02f, -6.2790519529e-02f, -1.8738131459e-01f, -3.0901699437e-01f, -4.2577929157e-01f, -5.3582679498e-01f, -6.3742398975e-01f, -7.2896862742e-01f, -8.0901699437e-01f, -8.7630668004e-01f, -9.2977648589e-01f, -9.6858316113e-01f, -9.9211470131e-01f, -1.0000000000e+00f, -9.9211470131e-01f, -9.6858316113e-01f, -9.2977648589e-01f, -8.7630668004e-01f, -8.0901699437e-01f, -7.2896862742e-01f, -6.3742398975e-01f, -5.3582679498e-01f, -4.2577929157e-01f, -3.0901699437e-01f, -1.8738131459e-01f, -6.2790519529e-02f, 6.2790519529e-02f, 1.8738131459e-01f, 3.0901699437e-01f, 4.2577929157e-01f, 5.3582679498e-01f, 6.3742398975e-01f, 7.2896862742e-01f, 8.0901699437e-01f, 8.7630668004e-01f, 9.2977648589e-01f, 9.6858316113e-01f, 9.9211470131e-01f};
    static constexpr float s[50] = {0.0000000000e+00f, 1.2533323356e-01f, 2.4868988716e-01f, 3.6812455268e-01f, 4.8175367410e-01f, 5.8778525229e-01f, 6.8454710593e-01f, 7.7051324278e-01f, 8.4432792550e-01f, 9.0482705247e-01f, 9.5105651630e-01f, 9.8228725073e-01f, 9.9802672843e-01f, 9.9802672843e-01f, 9.8228725073e-01f, 9.5105651630e-01f, 9.0482705247e-01f, 8.4432792550e-01f, 7.7051324278e-01f, 6.8454710593e-01f, 5.8778525229e-01f, 4.8175367410e-01f, 3.6812455268e-01f, 2.4868988716e-01f, 1.2533323356e-01f, 1.2246467991e-16f, -1.2533323356e-01f, -2.4868988716e-01f, -3.6812455268e-01f, -4.8175367410e-01f, -5.8778525229e-01f, -6.8454710593e-01f, -7.7051324278e-01f, -8.4432792550e-01f, -9.0482705247e-01f, -9.5105651630e-01f, -9.8228725073e-01f, -9.9802672843e-01f, -9.9802672843e-01f, -9.8228725073e-01f, -9.5105651630e-01f, -9.0482705247e-01f, -8.4432792550e-01f, -7.7051324278e-01f, -6.8454710593e-01f, -5.8778525229e-01f, -4.8175367410e-01f, -3.6812455268e-01f, -2.4868988716e-01f, -1.2533323356e-01f};
};

template <>
struct Twiddle<60> {
    static constexpr float c[60] = {1.0000000000e+00f, 9.9452189537e-01f, 9.7814760073e-01f, 9.5105651630e-01f, 9.1354545764e-01f, 8.6602540378e-01f, 8.0901699437e-01f, 7.4314482548e-01f, 6.6913060636e-01f, 5.8778525229e-01f, 5.0000000000e-01f, 4.0673664308e-01f, 3.0901699437e-01f, 2.0791169082e-01f, 1.0452846327e-01f, 2.8327694488e-16f, -1.0452846327e-01f, -2.0791169082e-01f, -3.0901699437e-01f, -4.0673664308e-01f, -5.0000000000e-01f, -5.8778525229e-01f, -6.6913060636e-01f, -7.4314482548e-01f, -8.0901699437e-01f, -8.6602540378e-01f, -9.1354545764e-01f, -9.5105651630e-01f, -9.7814760073e-01f, -9.9452189537e-01f, -1.0000000000e+00f, -9.9452189537e-01f, -9.7814760073e-01f, -9.5105651630e-01f, -9.1354545764e-01f, -8.6602540378e-01f, -8.0901699437e-01f, -7.4314482548e-01f, -6.6913060636e-01f, -5.8778525229e-01f, -5.0000000000e-01f, -4.0673664308e-01f, -3.0901699437e-01f, -2.0791169082e-01f, -1.0452846327e-01f, -1.8369701987e-16f, 1.0452846327e-01f, 2.0791169082e-01f, 3.0901699437e-01f, 4.0673664308e-01f, 5.0000000000e-01f, 5.8778525229e-01f, 6.6913060636e-01f, 7.4314482548e-01f, 8.0901699437e-01f, 8.6602540378e-01f, 9.1354545764e-01f, 9.5105651630e-01f, 9.7814760073e-01f, 9.9452189537e-01f};
    static constexpr float s[60] = {0.0000000000e+00f, 1.0452846327e-01f, 2.0791169082e-01f, 3.0901699437e-01f, 4.0673664308e-01f, 5.0000000000e-01f, 5.8778525229e-01f, 6.6913060636e-01f, 7.4314482548e-01f, 8.0901699437e-01f, 8.6602540378e-01f, 9.1354545764e-01f, 9.5105651630e-01f, 9.7814760073e-01f, 9.9452189537e-01f, 1.0000000000e+00f, 9.9452189537e-01f, 9.7814760073e-01f, 9.5105651630e-01f, 9.1354545764e-01f, 8.6602540378e-01f, 8.0901699437e-01f, 7.4314482548e-01f, 6.6913060636e-01f, 5.8778525229e-01f, 5.0000000000e-01f, 4.0673664308e-01f, 3.0901699437e-01f, 2.0791169082e-01f, 1.0452846327e-01f, 5.6655388976e-16f, -1.0452846327e-01f, -2.0791169082e-01f, -3.0901699437e-01f, -4.0673664308e-01f, -5.0000000000e-01f, -5.8778525229e-01f, -6.6913060636e-01f, -7.4314482548e-01f, -8.0901699437e-01f, -8.6602540378e-01f, -9.1354545764e-01f, -9.5105651630e-01f, -9.7814760073e-01f, -9.9452189537e-01f, -1.0000000000e+00f, -9.9452189537e-01f, -9.7814760073e-01f, -9.5105651630e-01f, -9.1354545764e-01f, -8.6602540378e-01f, -8.0901699437e-01f, -7.4314482548e-01f, -6.6913060636e-01f, -5.8778525229e-01f, -5.0000000000e-01f, -4.0673664308e-01f, -3.0901699437e-01f, -2.0791169082e-01f, -1.0452846327e-01f};
};

template <>
struct Twiddle<32> {
    static constexpr float c[32] = {1.0000000000e+00f, 9.8078528040e-01f, 9.2387953251e-01f, 8.3146961230e-01f, 7.0710678119e-01f, 5.5557023302e-01f, 3.8268343237e-01f, 1.9509032202e-01f, 6.1232339957e-17f, -1.9509032202e-01f, -3.8268343237e-01f, -5.5557023302e-01f, -7.0710678119e-01f, -8.3146961230e-01f, -9.2387953251e-01f, -9.8078528040e-01f, -1.0000000000e+00f, -9.8078528040e-01f, -9.2387953251e-01f, -8.3146961230e-01f, -7.0710678119e-01f, -5.5557023302e-01f, -3.8268343237e-01f, -1.9509032202e-01f, -1.8369701987e-16f, 1.9509032202e-01f, 3.8268343237e-01f, 5.5557023302e-01f, 7.0710678119e-01f, 8.3146961230e-01f, 9.2387953251e-01f, 9.8078528040e-01f};
    static constexpr float s[32] = {0.0000000000e+00f, 1.9509032202e-01f, 3.8268343237e-01f, 5.5557023302e-01f, 7.0710678119e-01f, 8.3146961230e-01f, 9.2387953251e-01f, 9.8078528040e-01f, 1.0000000000e+00f, 9.8078528040e-01f, 9.2387953251e-01f, 8.3146961230e-01f, 7.0710678119e-01f, 5.5557023302e-01f, 3.8268343237e-01f, 1.9509032202e-01f, 1.2246467991e-16f, -1.9509032202e-01f, -3.8268343237e-01f, -5.5557023302e-01f, -7.0710678119e-01f, -8.3146961230e-01f, -9.2387953251e-01f, -9.8078528040e-01f, -1.0000000000e+00f, -9.8078528040e-01f, -9.2387953251e-01f, -8.3146961230e-01f, -7.0710678119e-01f, -5.5557023302e-01f, -3.8268343237e-01f, -1.9509032202e-01f};
};

template <>
struct Twiddle<64> {
    static constexpr float c[64] = {1.0000000000e+00f, 9.9518472667e-01f, 9.8078528040e-01f, 9.5694033573e-01f, 9.2387953251e-01f, 8.8192126435e-01f, 8.3146961230e-01f, 7.7301045336e-01f, 7.0710678119e-01f, 6.3439328416e-01f, 5.5557023302e-01f, 4.7139673683e-01f, 3.8268343237e-01f, 2.9028467725e-01f, 1.9509032202e-01f, 9.8017140330e-02f, 6.1232339957e-17f, -9.8017140330e-02f, -1.9509032202e-01f, -2.9028467725e-01f, -3.8268343237e-01f, -4.7139673683e-01f, -5.5557023302e-01f, -6.3439328416e-01f, -7.0710678119e-01f, -7.7301045336e-01f, -8.3146961230e-01f, -8.8192126435e-01f, -9.2387953251e-01f, -9.5694033573e-01f, -9.8078528040e-01f, -9.9518472667e-01f, -1.0000000000e+00f, -9.9518472667e-01f, -9.8078528040e-01f, -9.5694033573e-01f, -9.2387953251e-01f, -8.8192126435e-01f, -8.3146961230e-01f, -7.7301045336e-01f, -7.0710678119e-01f, -6.3439328416e-01f, -5.5557023302e-01f, -4.7139673683e-01f, -3.8268343237e-01f, -2.9028467725e-01f, -1.9509032202e-01f, -9.8017140330e-02f, -1.8369701987e-16f, 9.8017140330e-02f, 1.9509032202e-01f, 2.9028467725e-01f, 3.8268343237e-01f, 4.7139673683e-01f, 5.5557023302e-01f, 6.3439328416e-01f, 7.0710678119e-01f, 7.7301045336e-01f, 8.3146961230e-01f, 8.8192126435e-01f, 9.2387953251e-01f, 9.5694033573e-01f, 9.8078528040e-01f, 9.9518472667e-01f};
    static constexpr float s[64] = {0.0000000000e+00f, 9.8017140330e-02f, 1.9509032202e-01f, 2.9028467725e-01f, 3.8268343237e-01f, 4.7139673683e-01f, 5.5557023302e-01f, 6.3439328416e-01f, 7.0710678119e-01f, 7.7301045336e-01f, 8.3146961230e-01f, 8.8192126435e-01f, 9.2387953251e-01f, 9.5694033573e-01f, 9.8078528040e-01f, 9.9518472667e-01f, 1.0000000000e+00f, 9.9518472667e-01f, 9.8078528040e-01f, 9.5694033573e-01f, 9.2387953251e-01f, 8.8192126435e-01f, 8.3146961230e-01f, 7.7301045336e-01f, 7.0710678119e-01f, 6.3439328416e-01f, 5.5557023302e-01f, 4.7139673683e-01f, 3.8268343237e-01f, 2.9028467725e-01f, 1.9509032202e-01f, 9.8017140330e-02f, 1.2246467991e-16f, -9.8017140330e-02f, -1.9509032202e-01f, -2.9028467725e-01f, -3.8268343237e-01f, -4.7139673683e-01f, -5.5557023302e-01f, -6.3439328416e-01f, -7.0710678119e-01f, -7.7301045336e-01f, -8.3146961230e-01f, -8.8192126435e-01f, -9.2387953251e-01f, -9.5694033573e-01f, -9.8078528040e-01f, -9.9518472667e-01f, -1.0000000000e+00f, -9.9518472667e-01f, -9.8078528040e-01f, -9.5694033573e-01f, -9.2387953251e-01f, -8.8192126435e-01f, -8.3146961230e-01f, -7.7301045336e-01f, -7.0710678119e-01f, -6.3439328416e-01f, -5.5557023302e-01f, -4.7139673683e-01f, -3.8268343237e-01f, -2.9028467725e-01f, -1.9509032202e-01f, -9.8017140330e-02f};
};


// multiplication by -i (forward) / +i (inverse)
template <bool INV>
OC_FFT_FN c2 rot90(c2 d) {
    return INV ? mkc(-d.y, d.x) : mkc(d.y, -d.x);
}

template <bool INV>
OC_FFT_FN void dft2(c2& a, c2& b) {
    const c2 t = a + b;
    b = a - b;
    a = t;
}

template <bool INV>
OC_FFT_FN void dft3(c2& a0, c2& a1, c2& a2) {
#pragma clang fp contract(fast)
    const c2 t = a1 + a2, d = rot90<INV>((a1 - a2) * 8.6602540378e-01f);  // sin(2 pi / 3), times -/+ i
    const c2 m = a0 - t * 0.5f;
    a0 = a0 + t;
    a1 = m + d;
    a2 = m - d;
}

template <bool INV>
OC_FFT_FN void dft4(c2& a0, c2& a1, c2& a2, c2& a3) {
    const c2 s02 = a0 + a2, d02 = a0 - a2, s13 = a1 + a3, d13 = rot90<INV>(a1 - a3);
    a0 = s02 + s13;
    a1 = d02 + d13;
    a2 = s02 - s13;
    a3 = d02 - d13;
}

template <bool INV>
OC_FFT_FN void dft5(c2& a0, c2& a1, c2& a2, c2& a3, c2& a4) {
#pragma clang fp contract(fast)
    constexpr float c1 = 3.0901699437e-01f, c2_ = -8.0901699437e-01f, s1 = 9.5105651630e-01f, s2 = 5.8778525229e-01f;
    const c2 p14 = a1 + a4, m14 = a1 - a4, p23 = a2 + a3, m23 = a2 - a3;
    const c2 r1 = a0 + p14 * c1 + p23 * c2_, r2 = a0 + p14 * c2_ + p23 * c1;
    const c2 i1 = rot90<INV>(m14 * s1 + m23 * s2), i2 = rot90<INV>(m14 * s2 - m23 * s1);
    a0 = a0 + p14 + p23;
    a1 = r1 + i1;
    a4 = r1 - i1;
    a2 = r2 + i2;
    a3 = r2 - i2;
}

// P-point DFT for an odd prime P on the elements v[OFF + j * M], j < P, in place (the symmetric form: with
// p_j = a_j + a_(P-j), m_j = a_j - a_(P-j), j = 1 .. H = (P-1)/2:
//   X_0 = a_0 + sum p_j,   X_k, X_(P-k) = (a_0 + sum_j p_j cos(2 pi j k / P))  -/+  i (sum_j m_j sin(2 pi j k / P))
// -- 2 H^2 multiply-adds instead of the (P-1)^2 complex products of the plain sum).  The inputs are dead once p, m are
// formed, so the outputs land in their registers.
template <bool INV, int P, int OFF, int M, int LEN>
OC_FFT_FN void dft_prime(c2 (&v)[LEN]) {
#pragma clang fp contract(fast)
    constexpr int H = (P - 1) / 2;
    c2 ps[H], ms[H];
    static_for<0, H>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const c2 a = v[OFF + (j + 1) * M], b = v[OFF + (P - 1 - j) * M];
        ps[j] = a + b;
        ms[j] = a - b;
    });
    const c2 a0 = v[OFF];
    c2 total = a0;
    static_for<0, H>([&](auto jc) { total = total + ps[decltype(jc)::value]; });
    static_for<1, H + 1>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        c2 re = a0, im = mkc(0.f, 0.f);
        static_for<0, H>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            constexpr int idx = ((j + 1) * k) % P;
            constexpr float cc = Twiddle<P>::c[idx], ss = Twiddle<P>::s[idx];
            re = re + ps[j] * cc;
            im = im + ms[j] * ss;
        });
        const c2 ri = rot90<INV>(im);  // forward: -i * im
        v[OFF + k * M] = re + ri;
        v[OFF + (P - k) * M] = re - ri;
    });
    v[OFF] = total;
}

template <bool INV, int TOP, int M, int OFF, int LEN, int S, int R>
OC_FFT_FN void fft_mixed_blocks(c2 (&v)[LEN]);

// N-point transform on v[OFF .. OFF+N); TOP is the size whose twiddle table is used (TOP % N == 0)
template <bool INV, int TOP, int N, int OFF, int LEN>
OC_FFT_FN void fft_mixed_at(c2 (&v)[LEN]) {
    if constexpr (N > 1) {
        constexpr int R = fft_radix(N), M = N / R;
        static_assert(R <= kFftMaxPrime, "window side must factor into primes <= 31");
        static_for<0, M>([&](auto n2c) {
            constexpr int n2 = decltype(n2c)::value;
            if constexpr (R == 2) dft2<INV>(v[OFF + n2], v[OFF + n2 + M]);
            if constexpr (R == 3) dft3<INV>(v[OFF + n2], v[OFF + n2 + M], v[OFF + n2 + 2 * M]);
            if constexpr (R == 4) dft4<INV>(v[OFF + n2], v[OFF + n2 + M], v[OFF + n2 + 2 * M], v[OFF + n2 + 3 * M]);
            if constexpr (R == 5)
                dft5<INV>(v[OFF + n2], v[OFF + n2 + M], v[OFF + n2 + 2 * M], v[OFF + n2 + 3 * M], v[OFF + n2 + 4 * M]);
            if constexpr (R > 5) dft_prime<INV, R, OFF + n2, M, LEN>(v);
            if constexpr (n2 != 0) {
                static_for<1, R>([&](auto sc) {
                    constexpr int s = decltype(sc)::value;
                    constexpr int m = (n2 * s) % N * (TOP / N);  // W_N^(n2 s) = W_TOP^m
                    constexpr float tc = Twiddle<TOP>::c[m], ts = Twiddle<TOP>::s[m];
                    v[OFF + n2 + s * M] = cmul_tw<INV>(v[OFF + n2 + s * M], tc, ts);
                });
            }
        });
        fft_mixed_blocks<INV, TOP, M, OFF, LEN, 0, R>(v);
    }
}

// the R sub-transforms of M elements each (compile-time recursion: the block offset is a template argument)
template <bool INV, int TOP, int M, int OFF, int LEN, int S, int R>
OC_FFT_FN void fft_mixed_blocks(c2 (&v)[LEN]) {
    if constexpr (S < R) {
        fft_mixed_at<INV, TOP, M, OFF + S * M, LEN>(v);
        fft_mixed_blocks<INV, TOP, M, OFF, LEN, S + 1, R>(v);
    }
}

template <bool INV, int N>
OC_FFT_FN void fft_mixed(c2 (&v)[N]) {
    fft_mixed_at<INV, N, N, 0, N>(v);
}

}  // namespace fftdev
}  // namespace ochip
