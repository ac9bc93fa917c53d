/*
 * oc_oracle.cpp -- CPU oracle (float32 restatement of OpenCorr's FFTCC -> ICGN path).
 *
 * TEST INFRASTRUCTURE ONLY -- see oc_oracle.h for the contract and for the
 * parity status of each entry point.  Build: oracle/Makefile
 * (g++ -O2 -ffp-contract=off -fopenmp, no third-party dependencies).
 *
 * Every function cites the reference lines it restates (paths relative to the
 * OpenCorr tree).  The arithmetic that the reference delegates to Eigen/FFTW
 * (un-vendored: Eigen 3.4.0, FFTW 3.3.5 per 1_Get_started.md:9-12) is restated
 * from the published algorithms:
 *   - Matrix6f/Matrix12f::inverse()  -> partial-pivot LU, solve against identity
 *   - Matrix3f/4f inverse            -> cofactor / determinant formula
 *   - small fixed products           -> coefficient-wise, inner index ascending
 *   - fftwf r2c/c2r                  -> mixed-radix DFT evaluated in double
 *     precision (the oracle's correlation surface is therefore *more* exact than
 *     either FFTW's or rocFFT's float32 result; only the arg-max and the float
 *     ZNCC derived from it are compared).
 */
#include "oc_oracle.h"

#include <omp.h>

#include <cmath>
#include <complex>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

namespace {

// ---------------------------------------------------------------------------
// reductions
// ---------------------------------------------------------------------------
// The arithmetic contract (oc_oracle.h, "OC_ARITH_FMA"): FMA = false rounds every multiply and every add on its own
// (the reference built for baseline x86-64); FMA = true fuses the PER-SAMPLE multiply-adds -- a*b + c becomes ONE
// correctly rounded fmaf(a, b, c), exactly the contraction a compiler with -ffp-contract=fast / -mfma makes of the
// reference's source expressions (association unchanged, one rounding fewer).  IEEE-754 fusedMultiplyAdd is defined
// bit for bit, so std::fmaf here and v_fma_f32 / v_pk_fma_f32 on gfx950 agree.
template <bool FMA>
static inline float mad(float a, float b, float c) {
    if constexpr (FMA) return __builtin_fmaf(a, b, c);
    else return a * b + c;
}

// K simultaneous sums over samples s = 0..N-1.  SEQ: one running float per sum.
// LANES: P strided partials per sum + ascending xor butterfly (oc_oracle.h).
// mac<FMA>(s, k, x, y): sum k receives the product x * y of sample s -- as a separately rounded product and an add, or
// fused into the running sum.
template <int K>
struct AccSeq {
    float a[K];
    explicit AccSeq(int /*lanes*/, int /*row_length*/ = 0) { for (int k = 0; k < K; k++) a[k] = 0.f; }
    inline void add(int /*s*/, int k, float v) { a[k] += v; }
    template <bool FMA>
    inline void mac(int /*s*/, int k, float x, float y) { a[k] = mad<FMA>(x, y, a[k]); }
    inline void finish() {}
    inline float get(int k) const { return a[k]; }
};

template <int K>
struct AccLanes {
    int P;
    std::vector<float> part;  // [K][P]
    explicit AccLanes(int lanes, int /*row_length*/ = 0) : P(lanes), part((size_t)K * lanes, 0.f) {}
    inline void add(int s, int k, float v) { part[(size_t)k * P + (s & (P - 1))] += v; }
    template <bool FMA>
    inline void mac(int s, int k, float x, float y) {
        float& slot = part[(size_t)k * P + (s & (P - 1))];
        slot = mad<FMA>(x, y, slot);
    }
    inline void finish() {
        std::vector<float> tmp(P);
        for (int k = 0; k < K; k++) {
            float* p = &part[(size_t)k * P];
            for (int off = 1; off < P; off <<= 1) {
                for (int l = 0; l < P; l++) tmp[l] = p[l] + p[l ^ off];
                for (int l = 0; l < P; l++) p[l] = tmp[l];
            }
        }
    }
    inline float get(int k) const { return part[(size_t)k * P]; }
};

// OC_ORDER_ROWS (oc_oracle.h): the association of the ICGN3D1 kernel that gives every subvolume row to one half-wave
// (opencorr_amd/csrc/icgn3d_rows.hip).  Samples arrive in row-major order s = r * SX + k; a body sample goes straight to
// its thread's running sum (a thread's body samples arrive in increasing (r, k)); tail samples are parked and added
// in finish(), in increasing q, AFTER the body -- one running sum per thread, body first.
template <int K>
struct AccRows {
    int P, SX, BW, RT;
    std::vector<float> part;                // [K][P]
    std::vector<float> tail;                // [K][number of tail samples], q-indexed
    long ntail;
    explicit AccRows(int lanes, int row_length) : P(lanes), SX(row_length > 0 ? row_length : 1), part((size_t)K * lanes, 0.f), ntail(0) {
        const int fc = SX / 32, rc = SX % 32;
        const int nch = fc >= 2 ? 2 : fc + (rc >= 28 ? 1 : 0);
        BW = SX < 32 * nch ? SX : 32 * nch;
        RT = SX - BW;
    }
    inline void add(int s, int k, float v) {
        const int r = s / SX, col = s - r * SX;
        if (col < BW) {
            part[(size_t)k * P + (size_t)((r % (P / 32)) * 32 + (col & 31))] += v;
        } else {
            const long q = (long)r * RT + (col - BW);
            if (q >= ntail) {
                ntail = q + 1;
                if (tail.size() < (size_t)K * (size_t)ntail) tail.resize((size_t)K * (size_t)(ntail + 4096), 0.f);
            }
            tail[(size_t)q * K + k] = v;
        }
    }
    // (the row mapping is the A/B partner of the default kernel and exists with separately rounded products only)
    template <bool FMA>
    inline void mac(int s, int k, float x, float y) {
        static_assert(!FMA, "OC_ORDER_ROWS has no fused-arithmetic form");
        add(s, k, x * y);
    }
    inline void finish() {
        for (long q = 0; q < ntail; q++)
            for (int k = 0; k < K; k++) part[(size_t)k * P + (size_t)(q % P)] += tail[(size_t)q * K + k];
        std::vector<float> tmp(P);
        for (int k = 0; k < K; k++) {
            float* p = &part[(size_t)k * P];
            for (int off = 1; off < P; off <<= 1) {
                for (int l = 0; l < P; l++) tmp[l] = p[l] + p[l ^ off];
                for (int l = 0; l < P; l++) p[l] = tmp[l];
            }
        }
    }
    inline float get(int k) const { return part[(size_t)k * P]; }
};

// ---------------------------------------------------------------------------
// small dense algebra (restating what the reference gets from Eigen)
// ---------------------------------------------------------------------------
// inverse of an n x n row-major matrix by LU with partial (row) pivoting and a
// solve against the identity -- Eigen::PartialPivLU::inverse(), which is what
// MatrixBase::inverse() uses for fixed sizes > 4 (src/oc_icgn.cpp:210,759,1339;
// src/oc_icgn.cpp:831 for the 6x6 warp).
static void lu_inverse(const float* A, float* Ainv, int n) {
    float lu[12 * 12];
    int perm[12];
    for (int i = 0; i < n * n; i++) lu[i] = A[i];
    for (int i = 0; i < n; i++) perm[i] = i;
    for (int k = 0; k < n; k++) {
        int piv = k;
        float best = std::fabs(lu[k * n + k]);
        for (int r = k + 1; r < n; r++) {
            float v = std::fabs(lu[r * n + k]);
            if (v > best) { best = v; piv = r; }
        }
        if (piv != k) {
            for (int c = 0; c < n; c++) { float t = lu[k * n + c]; lu[k * n + c] = lu[piv * n + c]; lu[piv * n + c] = t; }
            int t = perm[k]; perm[k] = perm[piv]; perm[piv] = t;
        }
        float d = lu[k * n + k];
        for (int r = k + 1; r < n; r++) {
            float f = lu[r * n + k] / d;
            lu[r * n + k] = f;
            for (int c = k + 1; c < n; c++) lu[r * n + c] = lu[r * n + c] - f * lu[k * n + c];
        }
    }
    // solve L U X = P I, column by column
    for (int col = 0; col < n; col++) {
        float y[12];
        for (int i = 0; i < n; i++) {
            float v = (perm[i] == col) ? 1.f : 0.f;
            for (int j = 0; j < i; j++) v = v - lu[i * n + j] * y[j];
            y[i] = v;
        }
        for (int i = n - 1; i >= 0; i--) {
            float v = y[i];
            for (int j = i + 1; j < n; j++) v = v - lu[i * n + j] * y[j];
            y[i] = v / lu[i * n + i];
        }
        for (int i = 0; i < n; i++) Ainv[i * n + col] = y[i];
    }
}

// c = a * b, n x n row-major, coefficient-wise with ascending inner index
// (Eigen lazy product for small fixed sizes; src/oc_icgn.cpp:290,831,1439).
static void mat_mul(const float* a, const float* b, float* c, int n) {
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            float v = a[i * n + 0] * b[0 * n + j];
            for (int k = 1; k < n; k++) v = v + a[i * n + k] * b[k * n + j];
            c[i * n + j] = v;
        }
}

// 3x3 inverse by cofactors / determinant (Eigen compute_inverse<.,.,3>).
static inline float cof3(const float* m, int i, int j) {
    int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    return m[i1 * 3 + j1] * m[i2 * 3 + j2] - m[i1 * 3 + j2] * m[i2 * 3 + j1];
}
static void inverse3(const float* m, float* r) {
    float c0 = cof3(m, 0, 0), c1 = cof3(m, 1, 0), c2 = cof3(m, 2, 0);
    float det = (c0 * m[0] + c1 * m[3]) + c2 * m[6];
    float invdet = 1.f / det;
    r[0] = c0 * invdet; r[1] = c1 * invdet; r[2] = c2 * invdet;
    r[3] = cof3(m, 0, 1) * invdet; r[4] = cof3(m, 1, 1) * invdet; r[5] = cof3(m, 2, 1) * invdet;
    r[6] = cof3(m, 0, 2) * invdet; r[7] = cof3(m, 1, 2) * invdet; r[8] = cof3(m, 2, 2) * invdet;
}

// 4x4 inverse by cofactor expansion (adjugate / determinant).
static inline float det3(float a, float b, float c, float d, float e, float f, float g, float h, float i) {
    return (a * (e * i - f * h) - b * (d * i - f * g)) + c * (d * h - e * g);
}
static void inverse4(const float* m, float* r) {
    float cofm[16];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            float s[9];
            int t = 0;
            for (int a = 0; a < 4; a++) {
                if (a == i) continue;
                for (int b = 0; b < 4; b++) {
                    if (b == j) continue;
                    s[t++] = m[a * 4 + b];
                }
            }
            float d = det3(s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7], s[8]);
            cofm[i * 4 + j] = ((i + j) & 1) ? -d : d;
        }
    float det = ((m[0] * cofm[0] + m[1] * cofm[1]) + m[2] * cofm[2]) + m[3] * cofm[3];
    float invdet = 1.f / det;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) r[i * 4 + j] = cofm[j * 4 + i] * invdet;
}

// ---------------------------------------------------------------------------
// bicubic B-spline (src/oc_cubic_bspline.h:52-58, src/oc_cubic_bspline.cpp:84-181)
// ---------------------------------------------------------------------------
static const float BC[4][4] = {
    {-144.0f / 336.0f, 384.0f / 336.0f, -384.0f / 336.0f, 144.0f / 336.0f},
    {342.0f / 336.0f, -702.0f / 336.0f, 450.0f / 336.0f, -90.0f / 336.0f},
    {-198.0f / 336.0f, -18.0f / 336.0f, 270.0f / 336.0f, -54.0f / 336.0f},
    {0.0f, 1.0f, 0.0f, 0.0f}};

template <bool FMA = false>
static inline float bspline2d_eval(const float* lut, int height, int width, float x, float y) {
    // src/oc_cubic_bspline.cpp:137-142
    if (x < 1 || y < 1 || x >= width - 2 || y >= height - 2 || std::isnan(x) || std::isnan(y)) return -1.f;
    int xi = (int)std::floor(x);
    int yi = (int)std::floor(y);
    float dx = x - xi, dy = y - yi;
    float dx2 = dx * dx, dy2 = dy * dy;
    float dx3 = dx2 * dx, dy3 = dy2 * dy;
    const float* c = lut + ((size_t)yi * width + xi) * 16;
    if constexpr (FMA) {
        // the same 16 terms, left to right, every "+ product" fused: v + (c * dy^k) * dx^l -> fmaf(c * dy^k, dx^l, v)
        float v = c[0];
        v = mad<true>(c[1], dx, v);
        v = mad<true>(c[2], dx2, v);
        v = mad<true>(c[3], dx3, v);
        v = mad<true>(c[4], dy, v);
        v = mad<true>(c[5] * dy, dx, v);
        v = mad<true>(c[6] * dy, dx2, v);
        v = mad<true>(c[7] * dy, dx3, v);
        v = mad<true>(c[8], dy2, v);
        v = mad<true>(c[9] * dy2, dx, v);
        v = mad<true>(c[10] * dy2, dx2, v);
        v = mad<true>(c[11] * dy2, dx3, v);
        v = mad<true>(c[12], dy3, v);
        v = mad<true>(c[13] * dy3, dx, v);
        v = mad<true>(c[14] * dy3, dx2, v);
        v = mad<true>(c[15] * dy3, dx3, v);
        return v;
    }
    // explicit 16-term left-to-right sum of src/oc_cubic_bspline.cpp:159-177
    float v = c[0];
    v = v + c[1] * dx;
    v = v + c[2] * dx2;
    v = v + c[3] * dx3;
    v = v + c[4] * dy;
    v = v + c[5] * dy * dx;
    v = v + c[6] * dy * dx2;
    v = v + c[7] * dy * dx3;
    v = v + c[8] * dy2;
    v = v + c[9] * dy2 * dx;
    v = v + c[10] * dy2 * dx2;
    v = v + c[11] * dy2 * dx3;
    v = v + c[12] * dy3;
    v = v + c[13] * dy3 * dx;
    v = v + c[14] * dy3 * dx2;
    v = v + c[15] * dy3 * dx3;
    return v;
}

// ---------------------------------------------------------------------------
// tricubic B-spline (src/oc_cubic_bspline.h:80-90, src/oc_cubic_bspline.cpp:35-53,214-405)
// ---------------------------------------------------------------------------
static const float PREF[8] = {1.732176555412860f,  -0.464135309171000f, 0.124364681271139f,  -0.033323415913556f,
                              0.008928982383084f,  -0.002392513618779f, 0.000641072092032f,  -0.000171774749350f};

static inline float basis0(float t) { return (1.f / 6.f) * (t * (t * (-t + 3.f) - 3.f) + 1.f); }
static inline float basis1(float t) { return (1.f / 6.f) * (t * t * (3.f * t - 6.f) + 4.f); }
static inline float basis2(float t) { return (1.f / 6.f) * (t * (t * (-3.f * t + 3.f) + 3.f) + 1.f); }
static inline float basis3(float t) { return (1.f / 6.f) * (t * t * t); }
// the same Horner forms with every "product + constant" fused (src/oc_cubic_bspline.cpp:35-53 under contraction)
static inline float basis0_fma(float t) { return (1.f / 6.f) * mad<true>(t, mad<true>(t, -t + 3.f, -3.f), 1.f); }
static inline float basis1_fma(float t) { return (1.f / 6.f) * mad<true>(t * t, mad<true>(3.f, t, -6.f), 4.f); }
static inline float basis2_fma(float t) { return (1.f / 6.f) * mad<true>(t, mad<true>(t, mad<true>(-3.f, t, 3.f), 3.f), 1.f); }
// ((b0 * r0 + b1 * r1) + b2 * r2) + b3 * r3, src/oc_cubic_bspline.cpp:390-401; fused: one product, three fmaf
template <bool FMA>
static inline float taps4(const float* b, float r0, float r1, float r2, float r3) {
    if constexpr (FMA) return mad<true>(b[3], r3, mad<true>(b[2], r2, mad<true>(b[1], r1, b[0] * r0)));
    else return ((b[0] * r0 + b[1] * r1) + b[2] * r2) + b[3] * r3;
}

template <bool FMA = false>
static inline float bspline3d_eval(const float* coef, int dz, int dy, int dx, float x, float y, float z) {
    if (x < 1 || y < 1 || z < 1 || x >= dx - 2 || y >= dy - 2 || z >= dz - 2 || std::isnan(x) || std::isnan(y) ||
        std::isnan(z))
        return -1.f;
    int xi = (int)std::floor(x), yi = (int)std::floor(y), zi = (int)std::floor(z);
    float fx = x - xi, fy = y - yi, fz = z - zi;
    float bx[4] = {basis0(fx), basis1(fx), basis2(fx), basis3(fx)};
    float by[4] = {basis0(fy), basis1(fy), basis2(fy), basis3(fy)};
    float bz[4] = {basis0(fz), basis1(fz), basis2(fz), basis3(fz)};
    if constexpr (FMA) {
        bx[0] = basis0_fma(fx); bx[1] = basis1_fma(fx); bx[2] = basis2_fma(fx);
        by[0] = basis0_fma(fy); by[1] = basis1_fma(fy); by[2] = basis2_fma(fy);
        bz[0] = basis0_fma(fz); bz[1] = basis1_fma(fz); bz[2] = basis2_fma(fz);
    }
    float sum_y[4];
    for (int i = 0; i < 4; i++) {
        float sum_x[4];
        for (int j = 0; j < 4; j++) {
            const float* row = coef + ((size_t)(zi + i - 1) * dy + (yi + j - 1)) * dx + (xi - 1);
            sum_x[j] = taps4<FMA>(bx, row[0], row[1], row[2], row[3]);
        }
        sum_y[i] = taps4<FMA>(by, sum_x[0], sum_x[1], sum_x[2], sum_x[3]);
    }
    return taps4<FMA>(bz, sum_y[0], sum_y[1], sum_y[2], sum_y[3]);
}

// ---------------------------------------------------------------------------
// double-precision mixed-radix FFT (stands in for FFTW, see file header)
// ---------------------------------------------------------------------------
typedef std::complex<double> cplx;

// Stockham autosort, decimation in frequency, generic radix (4, 2, 3, 5, then any prime).
struct FFT1D {
    int n;
    std::vector<int> factors;
    std::vector<cplx> tw;  // tw[k] = exp(-2 pi i k / n)
    explicit FFT1D(int n_) : n(n_), tw(n_ > 0 ? n_ : 1) {
        int m = n;
        while (m % 4 == 0) { factors.push_back(4); m /= 4; }
        for (int p = 2; p * p <= m; p++)
            while (m % p == 0) { factors.push_back(p); m /= p; }
        if (m > 1) factors.push_back(m);
        const double two_pi = 6.283185307179586476925286766559;
        for (int k = 0; k < n; k++) tw[k] = cplx(std::cos(two_pi * k / n), -std::sin(two_pi * k / n));
    }
    inline cplx w(long idx, int sign) const {
        cplx v = tw[(size_t)(idx % n)];
        return sign > 0 ? std::conj(v) : v;
    }
    // idx known to be < n
    inline cplx wd(int idx, int sign) const {
        cplx v = tw[(size_t)idx];
        return sign > 0 ? std::conj(v) : v;
    }
    // transforms x (length n, contiguous) in place using y as ping-pong scratch
    void run(cplx* x, cplx* y, int sign) const {
        cplx* src = x;
        cplx* dst = y;
        int len = n, s = 1;
        for (int r : factors) {
            const int m = len / r;
            const int tstep = n / len;  // W_len^k = tw[k * tstep]
            const int rstep = n / r;    // W_r^k   = tw[k * rstep]
            for (int p = 0; p < m; p++) {
                for (int q = 0; q < s; q++) {
                    if (r == 2) {
                        cplx a = src[q + s * p], b = src[q + s * (p + m)];
                        dst[q + s * (2 * p)] = a + b;
                        dst[q + s * (2 * p + 1)] = (a - b) * wd(p * tstep, sign);
                    } else if (r == 4) {
                        cplx a0 = src[q + s * p], a1 = src[q + s * (p + m)], a2 = src[q + s * (p + 2 * m)],
                             a3 = src[q + s * (p + 3 * m)];
                        cplx s02 = a0 + a2, d02 = a0 - a2, s13 = a1 + a3, d13 = a1 - a3;
                        // multiply d13 by -i (forward) or +i (inverse)
                        cplx jd = sign > 0 ? cplx(-d13.imag(), d13.real()) : cplx(d13.imag(), -d13.real());
                        dst[q + s * (4 * p)] = s02 + s13;
                        dst[q + s * (4 * p + 1)] = (d02 + jd) * wd(p * tstep, sign);
                        dst[q + s * (4 * p + 2)] = (s02 - s13) * wd(2 * p * tstep, sign);
                        dst[q + s * (4 * p + 3)] = (d02 - jd) * wd(3 * p * tstep, sign);
                    } else {
                        for (int k = 0; k < r; k++) {
                            cplx acc = src[q + s * p];
                            for (int j = 1; j < r; j++) acc += src[q + s * (p + j * m)] * w((long)j * k * rstep, sign);
                            dst[q + s * (r * p + k)] = acc * w((long)p * k * tstep, sign);
                        }
                    }
                }
            }
            len = m;
            s *= r;
            cplx* t = src; src = dst; dst = t;
        }
        if (src != x)
            for (int i = 0; i < n; i++) x[i] = src[i];
    }
};

struct FFTCache {
    std::vector<FFT1D*> plans;
    ~FFTCache() { for (FFT1D* p : plans) delete p; }
    const FFT1D& get(int n) {
        for (FFT1D* p : plans) if (p->n == n) return *p;
        plans.push_back(new FFT1D(n));
        return *plans.back();
    }
};

// in-place N-d complex DFT over a row-major array with dims[0] slowest
static void fft_nd(std::vector<cplx>& a, const std::vector<int>& dims, int sign, FFTCache& cache) {
    size_t total = 1;
    for (int d : dims) total *= d;
    int maxd = 0;
    for (int d : dims) maxd = d > maxd ? d : maxd;
    std::vector<cplx> line(maxd), scratch(maxd);
    size_t stride = total;
    for (size_t ax = 0; ax < dims.size(); ax++) {
        int n = dims[ax];
        stride /= n;
        const FFT1D& f = cache.get(n);
        size_t outer = total / ((size_t)n * stride);
        for (size_t o = 0; o < outer; o++)
            for (size_t i = 0; i < stride; i++) {
                cplx* base = &a[o * n * stride + i];
                if (stride == 1) {
                    f.run(base, scratch.data(), sign);
                } else {
                    for (int k = 0; k < n; k++) line[k] = base[(size_t)k * stride];
                    f.run(line.data(), scratch.data(), sign);
                    for (int k = 0; k < n; k++) base[(size_t)k * stride] = line[k];
                }
            }
    }
}

// circular cross-correlation surface  c = IDFT( conj(DFT(ref)) * DFT(tar) )  (unnormalised,
// like fftwf c2r), both inputs real, via one forward transform of ref + i*tar.
static void xcorr_nd(const float* ref, const float* tar, const std::vector<int>& dims, std::vector<cplx>& buf,
                     std::vector<cplx>& spec, float* surface, FFTCache& cache) {
    size_t total = 1;
    for (int d : dims) total *= d;
    buf.resize(total);
    spec.resize(total);
    for (size_t i = 0; i < total; i++) buf[i] = cplx((double)ref[i], (double)tar[i]);
    fft_nd(buf, dims, -1, cache);
    // index of -k
    std::vector<size_t> strides(dims.size());
    size_t s = 1;
    for (int ax = (int)dims.size() - 1; ax >= 0; ax--) { strides[ax] = s; s *= dims[ax]; }
    for (size_t i = 0; i < total; i++) {
        size_t rem = i, neg = 0;
        for (size_t ax = 0; ax < dims.size(); ax++) {
            size_t k = rem / strides[ax];
            rem -= k * strides[ax];
            neg += ((dims[ax] - k) % dims[ax]) * strides[ax];
        }
        cplx zk = buf[i], znk = std::conj(buf[neg]);
        cplx R = 0.5 * (zk + znk);
        cplx T = cplx(0.0, -0.5) * (zk - znk);
        spec[i] = std::conj(R) * T;
    }
    fft_nd(spec, dims, +1, cache);
    for (size_t i = 0; i < total; i++) surface[i] = (float)spec[i].real();
}

// ---------------------------------------------------------------------------
// ICGN 2D (first and second order) -- src/oc_icgn.cpp:144-341, 685-898
// ---------------------------------------------------------------------------
struct Images2D {
    const float *ref, *gx, *gy, *lut;
    int height, width;
};

// steepest-descent row for one sample (src/oc_icgn.cpp:191-196; 2D2: 725-745)
template <int DOF>
static inline void sd_row(float g_x, float g_y, int xl, int yl, float* sd) {
    if constexpr (DOF == 6) {
        sd[0] = g_x; sd[1] = g_x * xl; sd[2] = g_x * yl;
        sd[3] = g_y; sd[4] = g_y * xl; sd[5] = g_y * yl;
    } else {
        float xx = (xl * xl) * 0.5f, xy = (float)(xl * yl), yy = (yl * yl) * 0.5f;
        sd[0] = g_x; sd[1] = g_x * xl; sd[2] = g_x * yl; sd[3] = g_x * xx; sd[4] = g_x * xy; sd[5] = g_x * yy;
        sd[6] = g_y; sd[7] = g_y * xl; sd[8] = g_y * yl; sd[9] = g_y * xx; sd[10] = g_y * xy; sd[11] = g_y * yy;
    }
}

// steepest-descent row of the center-offset overloads: float local coordinates
// (src/oc_icgn.cpp:390-398; 2D2: 953-972)
template <int DOF>
static inline void sd_row_f(float g_x, float g_y, float xl, float yl, float* sd) {
    if constexpr (DOF == 6) {
        sd[0] = g_x; sd[1] = g_x * xl; sd[2] = g_x * yl;
        sd[3] = g_y; sd[4] = g_y * xl; sd[5] = g_y * yl;
    } else {
        float xx = (xl * xl) * 0.5f, xy = xl * yl, yy = (yl * yl) * 0.5f;
        sd[0] = g_x; sd[1] = g_x * xl; sd[2] = g_x * yl; sd[3] = g_x * xx; sd[4] = g_x * xy; sd[5] = g_x * yy;
        sd[6] = g_y; sd[7] = g_y * xl; sd[8] = g_y * yl; sd[9] = g_y * xx; sd[10] = g_y * xy; sd[11] = g_y * yy;
    }
}

// src/oc_deformation.cpp:117-128
static inline void set_warp_2d1(float* w, float u, float ux, float uy, float v, float vx, float vy) {
    w[0] = 1.f + ux; w[1] = uy; w[2] = u;
    w[3] = vx; w[4] = 1.f + vy; w[5] = v;
    w[6] = 0.f; w[7] = 0.f; w[8] = 1.f;
}

// src/oc_deformation.cpp:301-350; q = u ux uy uxx uxy uyy v vx vy vxx vxy vyy
static inline void set_warp_2d2(float* w, const float* q) {
    float u = q[0], ux = q[1], uy = q[2], uxx = q[3], uxy = q[4], uyy = q[5];
    float v = q[6], vx = q[7], vy = q[8], vxx = q[9], vxy = q[10], vyy = q[11];
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

// First-iteration damping of IC-LM: powf(damping.lambda, znssd / znssd0) (src/oc_iclm.cpp:253, :628).  libm's powf
// is not reproducible across platforms to the last bit, so the restatement fixes the arithmetic: exp(q * ln(lambda))
// in double with ln(lambda) taken once on the host, exp by argument reduction and a degree-14 Taylor polynomial
// evaluated with plain IEEE double operations, rounded once to float.  Relative error ~2e-16 before the rounding, so
// the result is the correctly rounded float power except within ~1e-8 of a rounding tie
// (tests/test_oracle_golden.py compares it with libm's powf over a dense sample).  The HIP engine restates the same
// sequence (opencorr_amd/csrc/dic2d_device.h).
static inline float pow_lambda(double log_lambda, float q) {
    const double t = (double)q * log_lambda;
    if (!(t == t)) return std::numeric_limits<float>::quiet_NaN();
    if (t > 700.0) return std::numeric_limits<float>::infinity();
    if (t < -700.0) return 0.f;
    const double kf = std::floor(t * 1.44269504088896338700e+00 + 0.5);
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
    return (float)std::ldexp(e, (int)kf);
}

// IC-LM damping (struct DampingParameter, src/oc_iclm.h:33-38) with ln(lambda) precomputed
struct LmDamping {
    double log_lambda;
    float alpha, beta;
};

// DOF = 6 -> ICGN2D1, DOF = 12 -> ICGN2D2; with lm != nullptr the Levenberg-Marquardt variants ICLM2D1 / ICLM2D2
// (src/oc_iclm.cpp:150-358, 505-731): same subset, Hessian and numerator code, plus the damped inverse every
// iteration, the accept/reject step, NO abort on out-of-range samples (they enter the subset as -1.f), and for
// 2D2 float (not truncated-int) weights in the convergence norm.
// off: centre offset {x, y} of the compute(POI2D*, Point2D&) overloads (src/oc_icgn.cpp:353-547,
// 910-1126), or nullptr for the plain compute(POI2D*).  self_adaptive: DIC::setSelfAdaptive, the
// subset radius comes from poi->subset_radius (:152-158).
template <int DOF, template <int> class Acc, bool FMA = false>
static void icgn2d_poi(const Images2D& im, int rx, int ry, float conv, float stop, float* poi, int lanes,
                       std::vector<float>& scratch, const float* off = nullptr, int self_adaptive = 0,
                       const LmDamping* lm = nullptr) {
    if (self_adaptive) {
        rx = (int)poi[23];  // Point2D (float) passed to int parameters of ICGN2D1_::update
        ry = (int)poi[24];
        if (rx < 0 || ry < 0) {  // the reference would allocate a negative-sized subset; rejected like the GPU engine does
            poi[16] = poi[16] >= 0 ? -3.f : poi[16];
            return;
        }
    }
    const float px = poi[0], py = poi[1];
    float* p = poi + 2;        // deformation.p[12]: u ux uy uxx uxy uyy v vx vy vxx vxy vyy
    float* res = poi + 14;     // u0 v0 zncc iteration convergence feature
    float* srad = poi + 23;    // subset_radius x,y
    const int height = im.height, width = im.width;

    // guard, src/oc_icgn.cpp:160-167 (2D2: 705-712)
    const float u_in = p[0], v_in = p[6];
    if (py - ry < 0 || px - rx < 0 || py + ry > height - 1 || px + rx > width - 1 || std::fabs(u_in) >= width ||
        std::fabs(v_in) >= height || res[2] < 0 || std::isnan(u_in) || std::isnan(v_in)) {
        res[2] = res[2] >= 0 ? -3.f : res[2];
        return;
    }
    const int W = 2 * rx + 1, H = 2 * ry + 1, N = W * H;
    scratch.resize((size_t)N * 4);
    float* rs = scratch.data();  // zero-mean reference subset
    float* sgx = rs + N;         // reference gradients over the subset
    float* sgy = sgx + N;
    float* ts = sgy + N;         // target subset

    // reference subset + zeroMeanNorm, src/oc_subset.cpp:39-53
    const int x0 = (int)(px - rx), y0 = (int)(py - ry);
    float ref_norm;
    {
        Acc<1> a(lanes);
        for (int r = 0; r < H; r++)
            for (int c = 0; c < W; c++) {
                int s = r * W + c;
                float v = im.ref[(size_t)(y0 + r) * width + (x0 + c)];
                rs[s] = v;
                a.add(s, 0, v);
            }
        a.finish();
        float mean = a.get(0) / (float)N;
        Acc<1> b(lanes);
        for (int s = 0; s < N; s++) {
            rs[s] = rs[s] - mean;
            b.template mac<FMA>(s, 0, rs[s], rs[s]);
        }
        b.finish();
        ref_norm = std::sqrt(b.get(0));
    }

    // steepest-descent image and Hessian, src/oc_icgn.cpp:179-207 (2D2: 716-756)
    constexpr int NH = DOF * (DOF + 1) / 2;
    float hess[DOF * DOF];
    {
        Acc<NH> a(lanes);
        const int gx0 = (int)px, gy0 = (int)py;
        for (int r = 0; r < H; r++)
            for (int c = 0; c < W; c++) {
                int s = r * W + c;
                int xl = c - rx, yl = r - ry;
                float g_x = im.gx[(size_t)(gy0 + yl) * width + (gx0 + xl)];
                float g_y = im.gy[(size_t)(gy0 + yl) * width + (gx0 + xl)];
                sgx[s] = g_x;
                sgy[s] = g_y;
                float sd[DOF];
                if (off)
                    sd_row_f<DOF>(g_x, g_y, xl - off[0], yl - off[1], sd);
                else
                    sd_row<DOF>(g_x, g_y, xl, yl, sd);
                int t = 0;
                for (int i = 0; i < DOF; i++)
                    for (int j = 0; j <= i; j++) a.template mac<FMA>(s, t++, sd[i], sd[j]);
            }
        a.finish();
        int t = 0;
        for (int i = 0; i < DOF; i++)
            for (int j = 0; j <= i; j++) {
                hess[i * DOF + j] = a.get(t);
                hess[j * DOF + i] = a.get(t);
                t++;
            }
    }
    float hinv[DOF * DOF];
    if (!lm) lu_inverse(hess, hinv, DOF);

    // initial guess: first-order terms only (src/oc_icgn.cpp:216, 2D2: 765-770 + src/oc_deformation.cpp:249-266)
    const float u0 = p[0], ux0 = p[1], uy0 = p[2], v0 = p[6], vx0 = p[7], vy0 = p[8];
    constexpr int WN = (DOF == 6) ? 3 : 6;
    float Wm[WN * WN];
    if constexpr (DOF == 6) {
        set_warp_2d1(Wm, u0, ux0, uy0, v0, vx0, vy0);
    } else {
        float q[12] = {u0, ux0, uy0, 0.f, 0.f, 0.f, v0, vx0, vy0, 0.f, 0.f, 0.f};
        set_warp_2d2(Wm, q);
    }

    int iter = 0;
    float dp_norm = 0.f, znssd = 0.f;
    float cur[12] = {0.f};
    float znssd0 = 4.f, lambda = 0.f;  // IC-LM state (src/oc_iclm.cpp:226)
    if (lm) {
        // p_current.setDeformation(p_initial) (src/oc_iclm.cpp:223 / :598): the parameters themselves are copied
        cur[0] = u0; cur[1] = ux0; cur[2] = uy0;
        cur[6] = v0; cur[7] = vx0; cur[8] = vy0;
    }
    do {
        iter++;
        // warped target subset, src/oc_icgn.cpp:230-242 (2D2: 784-796)
        bool negative = false;
        Acc<1> am(lanes);
        for (int r = 0; r < H; r++)
            for (int c = 0; c < W; c++) {
                int s = r * W + c;
                float xl = (float)(c - rx), yl = (float)(r - ry);
                if (off) {  // local_coor = x_local - center_offset (src/oc_icgn.cpp:447-450)
                    xl = xl - off[0];
                    yl = yl - off[1];
                }
                float wx, wy;
                if constexpr (DOF == 6) {
                    // src/oc_deformation.cpp:94-105
                    // (fused: the second product joins the first sum; "* 1.f" is exact, so nothing is left to fuse there)
                    wx = mad<FMA>(Wm[1], yl, Wm[0] * xl) + Wm[2] * 1.f;
                    wy = mad<FMA>(Wm[4], yl, Wm[3] * xl) + Wm[5] * 1.f;
                } else {
                    // src/oc_deformation.cpp:268-282: rows 3 and 4 of W * [x^2 xy y^2 x y 1]
                    float pv[6] = {xl * xl, xl * yl, yl * yl, xl, yl, 1.f};
                    const float* r3 = Wm + 3 * WN;
                    const float* r4 = Wm + 4 * WN;
                    wx = r3[0] * pv[0];
                    wy = r4[0] * pv[0];
                    for (int k = 1; k < 6; k++) {
                        wx = mad<FMA>(r3[k], pv[k], wx);
                        wy = mad<FMA>(r4[k], pv[k], wy);
                    }
                }
                // tar_subset->center = POI (+ center_offset, src/oc_icgn.cpp:425-426), then + warped_coor
                float cx = off ? px + off[0] : px, cy = off ? py + off[1] : py;
                float v = bspline2d_eval<FMA>(im.lut, height, width, cx + wx, cy + wy);
                if (v < 0.f) negative = true;
                ts[s] = v;
                am.add(s, 0, v);
            }
        // src/oc_icgn.cpp:251-255 (the IC-LM classes have no such check)
        if (negative && !lm) {
            res[2] = -3.f;
            return;
        }
        am.finish();
        float tmean = am.get(0) / (float)N;
        Acc<1> an(lanes);
        for (int s = 0; s < N; s++) {
            ts[s] = ts[s] - tmean;
            an.template mac<FMA>(s, 0, ts[s], ts[s]);
        }
        an.finish();
        float tar_norm = std::sqrt(an.get(0));
        // error image, ZNSSD, numerator: src/oc_icgn.cpp:260-276
        float factor = ref_norm / tar_norm;
        Acc<DOF + 1> ae(lanes);
        for (int r = 0; r < H; r++)
            for (int c = 0; c < W; c++) {
                int s = r * W + c;
                float e = mad<FMA>(ts[s], factor, -rs[s]);
                ae.template mac<FMA>(s, DOF, e, e);
                float sd[DOF];
                if (off)
                    sd_row_f<DOF>(sgx[s], sgy[s], (float)(c - rx) - off[0], (float)(r - ry) - off[1], sd);
                else
                    sd_row<DOF>(sgx[s], sgy[s], c - rx, r - ry, sd);
                for (int i = 0; i < DOF; i++) ae.template mac<FMA>(s, i, sd[i], e);
            }
        ae.finish();
        znssd = ae.get(DOF) / (ref_norm * ref_norm);
        if (lm) {
            // src/oc_iclm.cpp:250-257: lambda from the first ZNSSD, then (H + lambda * I)^-1 every iteration
            if (iter == 1) lambda = pow_lambda(lm->log_lambda, znssd / znssd0) - 1.f;
            float hl[DOF * DOF];
            for (int i = 0; i < DOF; i++)
                for (int j = 0; j < DOF; j++) hl[i * DOF + j] = hess[i * DOF + j] + lambda * (i == j ? 1.f : 0.f);
            lu_inverse(hl, hinv, DOF);
        }
        float dp[DOF];
        for (int i = 0; i < DOF; i++) {  // src/oc_icgn.cpp:279-286
            float v = 0.f;
            for (int j = 0; j < DOF; j++) v += hinv[i * DOF + j] * ae.get(j);
            dp[i] = v;
        }
        // warp update W <- W * (dW)^-1, then p <- W: src/oc_icgn.cpp:287-293 (2D2: 828-834)
        float dW[WN * WN], dWi[WN * WN], Wn[WN * WN];
        if constexpr (DOF == 6) {
            set_warp_2d1(dW, dp[0], dp[1], dp[2], dp[3], dp[4], dp[5]);
            inverse3(dW, dWi);
        } else {
            set_warp_2d2(dW, dp);
            lu_inverse(dW, dWi, WN);
        }
        // IC-LM accepts the step only when the ZNSSD went down (src/oc_iclm.cpp:283-300)
        const bool accept = !lm || znssd < znssd0;
        if (lm) lambda = lambda * (accept ? lm->alpha : lm->beta);
        if (accept) {
            mat_mul(Wm, dWi, Wn, WN);
            for (int i = 0; i < WN * WN; i++) Wm[i] = Wn[i];
            znssd0 = znssd;
        }
        const int rx2 = rx * rx, ry2 = ry * ry;
        if constexpr (DOF == 6) {
            // src/oc_deformation.cpp:107-115
            if (accept) {
                cur[0] = Wm[2]; cur[1] = Wm[0] - 1.f; cur[2] = Wm[1];
                cur[6] = Wm[5]; cur[7] = Wm[3]; cur[8] = Wm[4] - 1.f;
            }
            // src/oc_icgn.cpp:296-306; dp = u ux uy v vx vy
            float d = dp[0] * dp[0] + dp[1] * dp[1] * rx2 + dp[2] * dp[2] * ry2 + dp[3] * dp[3] + dp[4] * dp[4] * rx2 +
                      dp[5] * dp[5] * ry2;
            dp_norm = std::sqrt(d);
        } else {
            // src/oc_deformation.cpp:284-299
            const float* r3 = Wm + 3 * WN;
            const float* r4 = Wm + 4 * WN;
            if (accept) {
                cur[0] = r3[5]; cur[1] = r3[3] - 1.f; cur[2] = r3[4]; cur[3] = r3[0] * 2.f; cur[4] = r3[1]; cur[5] = r3[2] * 2.f;
                cur[6] = r4[5]; cur[7] = r4[3]; cur[8] = r4[4] - 1.f; cur[9] = r4[0] * 2.f; cur[10] = r4[1]; cur[11] = r4[2] * 2.f;
            }
            const int rxy2 = rx2 * ry2;
            const float* q = dp;  // u ux uy uxx uxy uyy v vx vy vxx vxy vyy
            float d;
            if (!lm) {
                // src/oc_icgn.cpp:837-857 (the integer-truncated weights are reference behaviour)
                const int rx4 = (int)(rx2 * rx2 * 0.25f), ry4 = (int)(ry2 * ry2 * 0.25f);
                d = q[0] * q[0] + q[1] * q[1] * rx2 + q[2] * q[2] * ry2 + q[3] * q[3] * rx4 + q[5] * q[5] * ry4 +
                    q[4] * q[4] * rxy2 + q[6] * q[6] + q[7] * q[7] * rx2 + q[8] * q[8] * ry2 + q[9] * q[9] * rx4 +
                    q[11] * q[11] * ry4 + q[10] * q[10] * rxy2;
            } else {
                // src/oc_iclm.cpp:675-687: p*p * rx2 * rx2 * 0.25f, left to right in float
                d = q[0] * q[0] + q[1] * q[1] * rx2 + q[2] * q[2] * ry2 + q[3] * q[3] * rx2 * rx2 * 0.25f +
                    q[5] * q[5] * ry2 * ry2 * 0.25f + q[4] * q[4] * rxy2 + q[6] * q[6] + q[7] * q[7] * rx2 +
                    q[8] * q[8] * ry2 + q[9] * q[9] * rx2 * rx2 * 0.25f + q[11] * q[11] * ry2 * ry2 * 0.25f +
                    q[10] * q[10] * rxy2;
            }
            dp_norm = std::sqrt(d);
        }
    } while (iter < stop && dp_norm >= conv);

    // outputs, src/oc_icgn.cpp:310-340 (2D2: 860-897)
    if constexpr (DOF == 6) {
        p[0] = cur[0]; p[1] = cur[1]; p[2] = cur[2];
        p[6] = cur[6]; p[7] = cur[7]; p[8] = cur[8];
    } else {
        for (int i = 0; i < 12; i++) p[i] = cur[i];
    }
    res[0] = u0;
    res[1] = v0;
    res[2] = 0.5f * (2 - znssd);
    res[3] = (float)iter;
    res[4] = dp_norm;
    srad[0] = (float)rx;
    srad[1] = (float)ry;
    if (res[4] >= conv && res[3] >= stop) res[2] = -4.f;
    if (std::isnan(res[2]) || std::isnan(p[0]) || std::isnan(p[6])) {
        p[0] = res[0];
        p[6] = res[1];
        res[2] = -5.f;
    }
}

// ---------------------------------------------------------------------------
// NR2D1 -- forward-additive Newton-Raphson, src/oc_nr.cpp:160-322
// ---------------------------------------------------------------------------
// lut / lut_gx / lut_gy: bicubic coefficient tables of the TARGET image and of its two
// gradient images (NR2D1::prepare, src/oc_nr.cpp:119-158).
template <template <int> class Acc>
static void nr2d1_poi(const float* ref, const float* lut, const float* lut_gx, const float* lut_gy, int height,
                      int width, int rx, int ry, float conv, float stop, float* poi, int lanes,
                      std::vector<float>& scratch) {
    const float px = poi[0], py = poi[1];
    float* p = poi + 2;     // u ux uy uxx uxy uyy v vx vy ...
    float* res = poi + 14;  // u0 v0 zncc iteration convergence feature
    // guard, src/oc_nr.cpp:165-171: failure is -1 here (not -3), and the two checks after the
    // else-branch (:304-316) run for guarded POIs as well
    if (py - ry < 0 || px - rx < 0 || py + ry > height - 1 || px + rx > width - 1 || std::fabs(p[0]) >= width ||
        std::fabs(p[6]) >= height || res[2] < 0 || std::isnan(p[0]) || std::isnan(p[6])) {
        res[2] = res[2] < -1 ? res[2] : -1.f;
    } else {
        const int W = 2 * rx + 1, H = 2 * ry + 1, N = W * H;
        scratch.resize((size_t)N * 4);
        float* rs = scratch.data();
        float* ts = rs + N;
        float* tgx = ts + N;
        float* tgy = tgx + N;
        const int x0 = (int)(px - rx), y0 = (int)(py - ry);
        float ref_norm;
        {
            Acc<1> a(lanes);
            for (int r = 0; r < H; r++)
                for (int c = 0; c < W; c++) {
                    int s = r * W + c;
                    rs[s] = ref[(size_t)(y0 + r) * width + (x0 + c)];
                    a.add(s, 0, rs[s]);
                }
            a.finish();
            float mean = a.get(0) / (float)N;
            Acc<1> b(lanes);
            for (int s = 0; s < N; s++) {
                rs[s] = rs[s] - mean;
                b.add(s, 0, rs[s] * rs[s]);
            }
            b.finish();
            ref_norm = std::sqrt(b.get(0));
        }
        const float u0 = p[0], v0 = p[6];
        float cur[6] = {p[0], p[1], p[2], p[6], p[7], p[8]};  // u ux uy v vx vy
        int iter = 0;
        float dp_norm = 0.f, znssd = 0.f;
        do {
            iter++;
            float Wm[9];
            set_warp_2d1(Wm, cur[0], cur[1], cur[2], cur[3], cur[4], cur[5]);
            // warped target subset and its gradients (:195-209), Hessian from the target gradients (:213-238)
            Acc<1> am(lanes);
            Acc<21> ah(lanes);
            for (int r = 0; r < H; r++)
                for (int c = 0; c < W; c++) {
                    int s = r * W + c;
                    float xl = (float)(c - rx), yl = (float)(r - ry);
                    float wx = (Wm[0] * xl + Wm[1] * yl) + Wm[2] * 1.f;
                    float wy = (Wm[3] * xl + Wm[4] * yl) + Wm[5] * 1.f;
                    float gxp = px + wx, gyp = py + wy;
                    ts[s] = bspline2d_eval(lut, height, width, gxp, gyp);
                    tgx[s] = bspline2d_eval(lut_gx, height, width, gxp, gyp);
                    tgy[s] = bspline2d_eval(lut_gy, height, width, gxp, gyp);
                    am.add(s, 0, ts[s]);
                    float sd[6];
                    sd_row<6>(tgx[s], tgy[s], c - rx, r - ry, sd);
                    int t = 0;
                    for (int i = 0; i < 6; i++)
                        for (int j = 0; j <= i; j++) ah.add(s, t++, sd[i] * sd[j]);  // H(i,j) and H(j,i) get the same products
                }
            am.finish();
            ah.finish();
            float tmean = am.get(0) / (float)N;
            Acc<1> an(lanes);
            for (int s = 0; s < N; s++) {
                ts[s] = ts[s] - tmean;
                an.add(s, 0, ts[s] * ts[s]);
            }
            an.finish();
            float tar_norm = std::sqrt(an.get(0));
            float hess[36], hinv[36];
            {
                int t = 0;
                for (int i = 0; i < 6; i++)
                    for (int j = 0; j <= i; j++) {
                        hess[i * 6 + j] = ah.get(t);
                        hess[j * 6 + i] = ah.get(t);
                        t++;
                    }
            }
            lu_inverse(hess, hinv, 6);
            // error image, ZNSSD, numerator (:244-262)
            float factor = tar_norm / ref_norm;
            Acc<7> ae(lanes);
            for (int r = 0; r < H; r++)
                for (int c = 0; c < W; c++) {
                    int s = r * W + c;
                    float e = rs[s] * factor - ts[s];
                    ae.add(s, 6, e * e);
                    float sd[6];
                    sd_row<6>(tgx[s], tgy[s], c - rx, r - ry, sd);
                    for (int i = 0; i < 6; i++) ae.add(s, i, sd[i] * e);
                }
            ae.finish();
            znssd = ae.get(6) / (tar_norm * tar_norm);
            float dp[6];
            for (int i = 0; i < 6; i++) {
                float v = 0.f;
                for (int j = 0; j < 6; j++) v += hinv[i * 6 + j] * ae.get(j);
                dp[i] = v;
            }
            for (int i = 0; i < 6; i++) cur[i] = cur[i] + dp[i];  // :277-279
            const int rx2 = rx * rx, ry2 = ry * ry;
            float d = 0.f;  // :285-291, one += per term
            d += dp[0] * dp[0];
            d += dp[1] * dp[1] * rx2;
            d += dp[2] * dp[2] * ry2;
            d += dp[3] * dp[3];
            d += dp[4] * dp[4] * rx2;
            d += dp[5] * dp[5] * ry2;
            dp_norm = std::sqrt(d);
        } while (iter < stop && dp_norm >= conv);
        p[0] = cur[0]; p[1] = cur[1]; p[2] = cur[2];
        p[6] = cur[3]; p[7] = cur[4]; p[8] = cur[5];
        res[0] = u0;
        res[1] = v0;
        res[2] = 0.5f * (2 - znssd);
        res[3] = (float)iter;
        res[4] = dp_norm;
    }
    if (res[4] >= conv && res[3] >= stop) res[2] = -4.f;
    if (std::isnan(res[2]) || std::isnan(p[0]) || std::isnan(p[6])) {
        p[0] = res[0];
        p[6] = res[1];
        res[2] = -5.f;
    }
}

// ---------------------------------------------------------------------------
// ICGN3D1 -- src/oc_icgn.cpp:1270-1490
// ---------------------------------------------------------------------------
struct Images3D {
    const float *ref, *gx, *gy, *gz, *coef;
    int dz, dy, dx;
};

template <template <int> class Acc, bool FMA = false>
static void icgn3d1_poi(const Images3D& im, int rx, int ry, int rz, float conv, float stop, float* poi, int lanes,
                        std::vector<float>& scratch) {
    const float px = poi[0], py = poi[1], pz = poi[2];
    float* p = poi + 3;      // u ux uy uz v vx vy vz w wx wy wz
    float* res = poi + 15;   // u0 v0 w0 zncc iteration convergence feature
    float* srad = poi + 28;  // subset_radius
    const int DX = im.dx, DY = im.dy, DZ = im.dz;
    // guard, src/oc_icgn.cpp:1279-1286
    if ((px - rx) < 0 || (py - ry) < 0 || (pz - rz) < 0 || (px + rx) > (DX - 1) || (py + ry) > (DY - 1) ||
        (pz + rz) > (DZ - 1) || std::fabs(p[0]) >= DX || std::fabs(p[4]) >= DY || std::fabs(p[8]) >= DZ || res[3] < 0 ||
        std::isnan(p[0]) || std::isnan(p[4]) || std::isnan(p[8])) {
        res[3] = res[3] >= 0 ? -3.f : res[3];
        return;
    }
    const int SX = 2 * rx + 1, SY = 2 * ry + 1, SZ = 2 * rz + 1;
    const int N = SX * SY * SZ;
    scratch.resize((size_t)N * 5);
    float* rs = scratch.data();
    float* sgx = rs + N;
    float* sgy = sgx + N;
    float* sgz = sgy + N;
    float* ts = sgz + N;

    // reference subvolume, src/oc_subset.cpp:89-135
    float ref_norm;
    {
        const float sx = px - rx, sy = py - ry, sz = pz - rz;
        Acc<1> a(lanes, SX);
        int s = 0;
        for (int i = 0; i < SZ; i++)
            for (int j = 0; j < SY; j++)
                for (int k = 0; k < SX; k++, s++) {
                    float v = im.ref[((size_t)(int)(sz + i) * DY + (int)(sy + j)) * DX + (int)(sx + k)];
                    rs[s] = v;
                    a.add(s, 0, v);
                }
        a.finish();
        float mean = a.get(0) / (float)N;
        Acc<1> b(lanes, SX);
        for (s = 0; s < N; s++) {
            rs[s] = rs[s] - mean;
            b.template mac<FMA>(s, 0, rs[s], rs[s]);
        }
        b.finish();
        ref_norm = std::sqrt(b.get(0));
    }

    // SD image + Hessian, src/oc_icgn.cpp:1299-1337
    float hess[144];
    {
        Acc<78> a(lanes, SX);
        const int cx = (int)px, cy = (int)py, cz = (int)pz;
        int s = 0;
        for (int i = 0; i < SZ; i++)
            for (int j = 0; j < SY; j++)
                for (int k = 0; k < SX; k++, s++) {
                    int xl = k - rx, yl = j - ry, zl = i - rz;
                    size_t g = ((size_t)(cz + zl) * DY + (cy + yl)) * DX + (cx + xl);
                    float g_x = im.gx[g], g_y = im.gy[g], g_z = im.gz[g];
                    sgx[s] = g_x; sgy[s] = g_y; sgz[s] = g_z;
                    float sd[12] = {g_x, g_x * xl, g_x * yl, g_x * zl, g_y, g_y * xl, g_y * yl, g_y * zl,
                                    g_z, g_z * xl, g_z * yl, g_z * zl};
                    int t = 0;
                    for (int r = 0; r < 12; r++)
                        for (int c = 0; c <= r; c++) a.template mac<FMA>(s, t++, sd[r], sd[c]);
                }
        a.finish();
        int t = 0;
        for (int r = 0; r < 12; r++)
            for (int c = 0; c <= r; c++) {
                hess[r * 12 + c] = a.get(t);
                hess[c * 12 + r] = a.get(t);
                t++;
            }
    }
    float hinv[144];
    lu_inverse(hess, hinv, 12);

    auto set_warp = [](float* w, const float* q) {
        // src/oc_deformation.cpp:495-516
        w[0] = 1.f + q[1]; w[1] = q[2]; w[2] = q[3]; w[3] = q[0];
        w[4] = q[5]; w[5] = 1.f + q[6]; w[6] = q[7]; w[7] = q[4];
        w[8] = q[9]; w[9] = q[10]; w[10] = 1.f + q[11]; w[11] = q[8];
        w[12] = 0.f; w[13] = 0.f; w[14] = 0.f; w[15] = 1.f;
    };
    float init[12];
    for (int i = 0; i < 12; i++) init[i] = p[i];
    float Wm[16];
    set_warp(Wm, init);
    float cur[12];
    int iter = 0;
    float dp_norm = 0.f, znssd = 0.f;
    do {
        iter++;
        bool out_of_range = false;
        Acc<1> am(lanes, SX);
        int s = 0;
        for (int i = 0; i < SZ; i++)
            for (int j = 0; j < SY; j++)
                for (int k = 0; k < SX; k++, s++) {
                    float xl = (float)(k - rx), yl = (float)(j - ry), zl = (float)(i - rz);
                    // src/oc_deformation.cpp:518-530
                    // (fused: the second and third products join the running sum; "* 1.f" is exact)
                    float wx = mad<FMA>(Wm[2], zl, mad<FMA>(Wm[1], yl, Wm[0] * xl)) + Wm[3] * 1.f;
                    float wy = mad<FMA>(Wm[6], zl, mad<FMA>(Wm[5], yl, Wm[4] * xl)) + Wm[7] * 1.f;
                    float wz = mad<FMA>(Wm[10], zl, mad<FMA>(Wm[9], yl, Wm[8] * xl)) + Wm[11] * 1.f;
                    float v = bspline3d_eval<FMA>(im.coef, DZ, DY, DX, px + wx, py + wy, pz + wz);
                    if (v < 0.f) out_of_range = true;
                    ts[s] = v;
                    am.add(s, 0, v);
                }
        if (out_of_range) { res[3] = -3.f; return; }
        am.finish();
        float tmean = am.get(0) / (float)N;
        Acc<1> an(lanes, SX);
        for (s = 0; s < N; s++) {
            ts[s] = ts[s] - tmean;
            an.template mac<FMA>(s, 0, ts[s], ts[s]);
        }
        an.finish();
        float tar_norm = std::sqrt(an.get(0));
        float factor = ref_norm / tar_norm;
        Acc<13> ae(lanes, SX);
        s = 0;
        for (int i = 0; i < SZ; i++)
            for (int j = 0; j < SY; j++)
                for (int k = 0; k < SX; k++, s++) {
                    int xl = k - rx, yl = j - ry, zl = i - rz;
                    float e = mad<FMA>(factor, ts[s], -rs[s]);
                    ae.template mac<FMA>(s, 12, e, e);
                    float g_x = sgx[s], g_y = sgy[s], g_z = sgz[s];
                    const float sd[12] = {g_x, g_x * xl, g_x * yl, g_x * zl, g_y, g_y * xl, g_y * yl, g_y * zl,
                                          g_z, g_z * xl, g_z * yl, g_z * zl};
                    for (int q = 0; q < 12; q++) ae.template mac<FMA>(s, q, sd[q], e);
                }
        ae.finish();
        znssd = ae.get(12) / (ref_norm * ref_norm);
        float dp[12];
        for (int i = 0; i < 12; i++) {
            float v = 0.f;
            for (int j = 0; j < 12; j++) v += hinv[i * 12 + j] * ae.get(j);
            dp[i] = v;
        }
        float dW[16], dWi[16], Wn[16];
        set_warp(dW, dp);
        inverse4(dW, dWi);
        mat_mul(Wm, dWi, Wn, 4);
        for (int i = 0; i < 16; i++) Wm[i] = Wn[i];
        // src/oc_deformation.cpp:416-432
        cur[0] = Wm[3]; cur[1] = Wm[0] - 1.f; cur[2] = Wm[1]; cur[3] = Wm[2];
        cur[4] = Wm[7]; cur[5] = Wm[4]; cur[6] = Wm[5] - 1.f; cur[7] = Wm[6];
        cur[8] = Wm[11]; cur[9] = Wm[8]; cur[10] = Wm[9]; cur[11] = Wm[10] - 1.f;
        // src/oc_icgn.cpp:1445
        dp_norm = std::sqrt(dp[0] * dp[0] + dp[4] * dp[4] + dp[8] * dp[8]);
    } while (iter < stop && dp_norm >= conv);

    for (int i = 0; i < 12; i++) p[i] = cur[i];
    res[0] = init[0];
    res[1] = init[4];
    res[2] = init[8];
    res[3] = 0.5f * (2 - znssd);
    res[4] = (float)iter;
    res[5] = dp_norm;
    srad[0] = (float)rx; srad[1] = (float)ry; srad[2] = (float)rz;
    if (res[5] >= conv && res[4] >= stop) res[3] = -4.f;
    if (std::isnan(res[3]) || std::isnan(p[0]) || std::isnan(p[4]) || std::isnan(p[8])) {
        p[0] = res[0]; p[4] = res[1]; p[8] = res[2];
        res[3] = -5.f;
    }
}

// ---------------------------------------------------------------------------
// Strain (src/oc_strain.cpp:149-247 for POI2D, :372-473 for POI3D): plane fit of u, v (, w) over the neighbour POIs
// of a subregion, SURVEY 8f row 4.
//
// What the reference fixes: the neighbour set (squared float distance < radius^2 -- nanoflann's L2_Simple metric and
// RadiusResultSet, strict <; query point included), the ZNCC filter, the KNN fallback when fewer than
// neighbor_number_min POIs lie inside the radius (:165-186), the "at least neighbor_number_min accepted POIs" rule
// (:190), the design matrix [1, dx, dy(, dz)] with float differences (:201-209) and the Cauchy / Green formulas
// (:220-234, :446-466).  What it leaves to third-party code that is not in the tree (nanoflann's kd-tree traversal
// order with sorted = false, Eigen's colPivHouseholderQr in float): the ORDER of the rows and the rounding of the
// least-squares solve.  The restatement fixes both: rows are taken cell by cell over a uniform grid (pitch a little
// above the radius, anchored at the queue's minimum coordinates; the 3 x 3 (x 3) block around the POI in row-major
// cell order) and by ascending queue index inside a cell; KNN rows by ascending (distance^2, index); the normal
// equations are accumulated and solved in double (elimination without pivoting, the matrix is SPD) and rounded to
// float once.  The result is the least-squares solution of the float data to ~1e-13, i.e. it differs from Eigen's
// float QR by that QR's own rounding (~1e-6 relative); the reference's golden table pins it at that level
// (tests/test_oracle_strain.py).  A pivot that vanishes (collinear neighbours) zeroes that gradient component.
// ---------------------------------------------------------------------------
struct StrainGrid {
    float x0, y0, z0, inv_pitch;
    int ncx, ncy, ncz;
};

// shared by the oracle and (restated) by the HIP engine: cell of a coordinate triple
static inline int strain_cell_axis(float c, float c0, float inv_pitch, int nc) {
    const float t = (c - c0) * inv_pitch;
    int k = t >= 0.f ? (t < (float)nc ? (int)t : nc - 1) : 0;  // NaN -> 0
    return k;
}

template <int DIM>
static void strain_solve(const double* S, const double* B, int nrhs, double* grad) {
    // S: (DIM+1) x (DIM+1) normal matrix (row-major, symmetric), B: nrhs right-hand sides of DIM+1 entries.
    // grad[r*(DIM+1) + k]: solution k of right-hand side r.  Gaussian elimination in fixed order.
    constexpr int D = DIM + 1;
    double A[D][D + 3];
    for (int i = 0; i < D; i++) {
        for (int j = 0; j < D; j++) A[i][j] = S[i * D + j];
        for (int r = 0; r < nrhs; r++) A[i][D + r] = B[r * D + i];
    }
    bool dead[D];
    for (int k = 0; k < D; k++) {
        const double piv = A[k][k];
        // a pivot that is zero to rounding relative to the original diagonal: no information along this column
        dead[k] = !(piv > 1e-12 * S[k * D + k]);
        if (dead[k]) continue;
        for (int i = k + 1; i < D; i++) {
            const double f = A[i][k] / piv;
            for (int j = k + 1; j < D + nrhs; j++) A[i][j] = A[i][j] - f * A[k][j];
        }
    }
    for (int r = 0; r < nrhs; r++)
        for (int k = D - 1; k >= 0; k--) {
            double v = 0.0;
            if (!dead[k]) {
                v = A[k][D + r];
                for (int j = k + 1; j < D; j++) v = v - A[k][j] * grad[r * D + j];
                v = v / A[k][k];
            }
            grad[r * D + k] = v;
        }
}

// mode 0: Strain -- the queries are the cloud itself (queries == pois, nq == n), POIs below the ZNCC threshold are
//         neither computed nor used, the strain fields are written.
// mode 1: RegionFit2D/3D::compute(poi_queue) (src/oc_region_fit.cpp:94-174, 251-342) -- the cloud is the queue of
//         reliable POIs given to setNeighbor (:75-78), every cloud POI counts (no ZNCC filter), and the fitted plane
//         itself becomes the query POI's deformation (u, ux, uy(, uz), v, ..., :153-159 / :314-327) with
//         result.zncc reset to 0 (:162 / :330).  Same neighbour rule, KNN fallback and row order as Strain.
template <int DIM>
static void strain_queue(float* pois, long n, int stride, float radius, int nmin, float thr, int approximation, int threads,
                         int mode = 0, float* queries = nullptr, long nq = 0, int qstride = 0) {
    constexpr int D = DIM + 1;
    constexpr int ZNCC = DIM == 2 ? 16 : 18;
    constexpr int U = DIM == 2 ? 2 : 3, V = DIM == 2 ? 8 : 7, Wf = 11;
    constexpr int E0 = DIM == 2 ? 20 : 22;
    if (mode == 0) {
        queries = pois;
        nq = n;
        qstride = stride;
    }
    if (n <= 0 || nq <= 0) return;
    // grid
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (long i = 0; i < n; i++)
        for (int a = 0; a < DIM; a++) {
            const float c = pois[i * stride + a];
            if (c < mn[a]) mn[a] = c;
            if (c > mx[a]) mx[a] = c;
        }
    for (int a = 0; a < 3; a++)
        if (!(mn[a] <= mx[a])) mn[a] = mx[a] = 0.f;
    float pitch = radius * 1.001f;
    const float span = std::fmax(std::fmax(mx[0] - mn[0], mx[1] - mn[1]), mx[2] - mn[2]);
    const float cap = DIM == 2 ? 4096.f : 256.f;  // cells per axis
    if (!(pitch > span / cap)) pitch = span / cap;
    if (!(pitch > 0.f)) pitch = 1.f;
    StrainGrid g;
    g.x0 = mn[0]; g.y0 = mn[1]; g.z0 = mn[2];
    g.inv_pitch = 1.f / pitch;
    g.ncx = (int)((mx[0] - mn[0]) * g.inv_pitch) + 1;
    g.ncy = (int)((mx[1] - mn[1]) * g.inv_pitch) + 1;
    g.ncz = DIM == 3 ? (int)((mx[2] - mn[2]) * g.inv_pitch) + 1 : 1;
    const long ncell = (long)g.ncx * g.ncy * g.ncz;
    std::vector<int> cell(n), start(ncell + 1, 0), order(n);
    for (long i = 0; i < n; i++) {
        const float* p = pois + i * stride;
        const int cx = strain_cell_axis(p[0], g.x0, g.inv_pitch, g.ncx), cy = strain_cell_axis(p[1], g.y0, g.inv_pitch, g.ncy);
        const int cz = DIM == 3 ? strain_cell_axis(p[2], g.z0, g.inv_pitch, g.ncz) : 0;
        cell[i] = (cz * g.ncy + cy) * g.ncx + cx;
        start[cell[i] + 1]++;
    }
    for (long c = 0; c < ncell; c++) start[c + 1] += start[c];
    {
        std::vector<int> cur(start.begin(), start.end() - 1);
        for (long i = 0; i < n; i++) order[cur[cell[i]]++] = (int)i;  // ascending index inside a cell
    }
    const float r2 = radius * radius;
    std::vector<float> out((size_t)nq * 12, 0.f);
    std::vector<char> wrote(nq, 0);
    const int nt = threads <= 0 ? omp_get_max_threads() : threads;
#pragma omp parallel for num_threads(nt) schedule(dynamic, 64)
    for (long i = 0; i < nq; i++) {
        const float* pi = queries + i * qstride;
        if (mode == 0 && !(pi[ZNCC] >= thr)) continue;  // src/oc_strain.cpp:241 / :481
        double S[D * D] = {0.0}, B[3 * D] = {0.0};
        int nfit = 0;
        auto add_row = [&](const float* pj) {
            double row[D];
            row[0] = 1.0;
            for (int a = 0; a < DIM; a++) row[a + 1] = (double)(pj[a] - pi[a]);  // float difference, src/oc_strain.cpp:204-205
            for (int a = 0; a < D; a++)
                for (int b = 0; b < D; b++) S[a * D + b] = S[a * D + b] + row[a] * row[b];
            const double rhs[3] = {(double)pj[U], (double)pj[V], DIM == 3 ? (double)pj[Wf] : 0.0};
            for (int r = 0; r < DIM; r++)
                for (int a = 0; a < D; a++) B[r * D + a] = B[r * D + a] + row[a] * rhs[r];
            nfit++;
        };
        auto dist2 = [&](const float* pj) {
            float d = 0.f;
            for (int a = 0; a < DIM; a++) {
                const float df = pi[a] - pj[a];
                d = d + df * df;
            }
            return d;
        };
        // radius search
        int inside = 0;
        const int cx = strain_cell_axis(pi[0], g.x0, g.inv_pitch, g.ncx), cy = strain_cell_axis(pi[1], g.y0, g.inv_pitch, g.ncy);
        const int cz = DIM == 3 ? strain_cell_axis(pi[2], g.z0, g.inv_pitch, g.ncz) : 0;
        for (int dz = (DIM == 3 ? -1 : 0); dz <= (DIM == 3 ? 1 : 0); dz++)
            for (int dy = -1; dy <= 1; dy++)
                for (int dx = -1; dx <= 1; dx++) {
                    const int x = cx + dx, y = cy + dy, z = cz + dz;
                    if (x < 0 || y < 0 || z < 0 || x >= g.ncx || y >= g.ncy || z >= g.ncz) continue;
                    const long c = ((long)z * g.ncy + y) * g.ncx + x;
                    for (int k = start[c]; k < start[c + 1]; k++) {
                        const float* pj = pois + (long)order[k] * stride;
                        if (dist2(pj) < r2) {
                            inside++;
                            if (mode == 1 || pj[ZNCC] >= thr) add_row(pj);
                        }
                    }
                }
        if (inside < nmin) {
            // KNN fallback (src/oc_strain.cpp:177-186): the nmin nearest POIs, ascending (distance^2, index)
            for (int a = 0; a < D * D; a++) S[a] = 0.0;
            for (int a = 0; a < 3 * D; a++) B[a] = 0.0;
            nfit = 0;
            float last_d = -1.f;
            long last_j = -1;
            for (int k = 0; k < nmin; k++) {
                float best_d = INFINITY;
                long best_j = -1;
                for (long j = 0; j < n; j++) {
                    const float d = dist2(pois + j * stride);
                    if (!(d > last_d || (d == last_d && j > last_j))) continue;  // already taken (NaN never enters)
                    if (d < best_d) {
                        best_d = d;
                        best_j = j;
                    }
                }
                if (best_j < 0) break;
                last_d = best_d;
                last_j = best_j;
                const float* pj = pois + best_j * stride;
                if (mode == 1 || pj[ZNCC] >= thr) add_row(pj);
            }
        }
        if (nfit < nmin) continue;  // src/oc_strain.cpp:190
        double grad[3 * D];
        strain_solve<DIM>(S, B, DIM, grad);
        float* e = out.data() + (size_t)i * 12;
        if (mode == 1) {
            for (int r = 0; r < DIM; r++)
                for (int k = 0; k < D; k++) e[r * D + k] = (float)grad[r * D + k];
            wrote[i] = 1;
            continue;
        }
        if (DIM == 2) {
            const float ux = (float)grad[1], uy = (float)grad[2], vx = (float)grad[D + 1], vy = (float)grad[D + 2];
            if (approximation == 1) {
                e[0] = ux; e[1] = vy; e[2] = 0.5f * (uy + vx);
                wrote[i] = 1;
            }
            if (approximation == 2) {
                e[0] = ux + 0.5f * (ux * ux + vx * vx);
                e[1] = vy + 0.5f * (uy * uy + vy * vy);
                e[2] = 0.5f * (uy + vx + uy * ux + vy * vx);
                wrote[i] = 1;
            }
        } else {
            const float ux = (float)grad[1], uy = (float)grad[2], uz = (float)grad[3];
            const float vx = (float)grad[D + 1], vy = (float)grad[D + 2], vz = (float)grad[D + 3];
            const float wx = (float)grad[2 * D + 1], wy = (float)grad[2 * D + 2], wz = (float)grad[2 * D + 3];
            if (approximation == 1) {
                e[0] = ux; e[1] = vy; e[2] = wz;
                e[3] = 0.5f * (uy + vx); e[4] = 0.5f * (vz + wy); e[5] = 0.5f * (wx + uz);
                wrote[i] = 1;
            }
            if (approximation == 2) {
                e[0] = ux + 0.5f * (ux * ux + vx * vx + wx * wx);
                e[1] = vy + 0.5f * (uy * uy + vy * vy + wy * wy);
                e[2] = wz + 0.5f * (uz * uz + vz * vz + wz * wz);
                e[3] = 0.5f * (uy + vx + uy * ux + vy * vx + wy * wx);
                e[4] = 0.5f * (vz + wy + uz * uy + vz * vy + wz * wy);
                e[5] = 0.5f * (wx + uz + ux * uz + vx * vz + wx * wz);
                wrote[i] = 1;
            }
        }
    }
    // the strain fields are the only thing written, and only for POIs that were fitted (everything else untouched)
    for (long i = 0; i < nq; i++) {
        if (!wrote[i]) continue;
        float* q = queries + i * qstride;
        const float* e = out.data() + (size_t)i * 12;
        if (mode == 0) {
            for (int k = 0; k < (DIM == 2 ? 3 : 6); k++) q[E0 + k] = e[k];
        } else if (DIM == 2) {
            q[2] = e[0]; q[3] = e[1]; q[4] = e[2];  // u ux uy
            q[8] = e[3]; q[9] = e[4]; q[10] = e[5];  // v vx vy
            q[ZNCC] = 0.f;
        } else {
            for (int k = 0; k < 12; k++) q[3 + k] = e[k];  // u ux uy uz v vx vy vz w wx wy wz
            q[ZNCC] = 0.f;
        }
    }
}

static int resolve_threads(int threads) {
    if (threads <= 0) return omp_get_max_threads();
    return threads;
}

}  // namespace

// ===========================================================================
// C entry points
// ===========================================================================
// order = OC_ORDER_SEQ | OC_ORDER_LANES, optionally | OC_ARITH_FMA (oc_oracle.h)
template <int DOF>
static void icgn2d_queue(const Images2D& im, int rx, int ry, float conv, float stop, float* pois, long n, int stride_floats,
                         int order, int lanes, int threads, const float* center_offsets, int self_adaptive,
                         const LmDamping* lm) {
    threads = resolve_threads(threads);
    const bool fma = (order & OC_ARITH_FMA) != 0;
    const int assoc = order & ~OC_ARITH_FMA;
#pragma omp parallel num_threads(threads)
    {
        std::vector<float> scratch;
#pragma omp for schedule(static)
        for (long i = 0; i < n; i++) {
            const float* off = center_offsets ? center_offsets + 2 * i : nullptr;
            float* poi = pois + (size_t)i * stride_floats;
            if (assoc == OC_ORDER_SEQ) {
                if (fma) icgn2d_poi<DOF, AccSeq, true>(im, rx, ry, conv, stop, poi, lanes, scratch, off, self_adaptive, lm);
                else icgn2d_poi<DOF, AccSeq, false>(im, rx, ry, conv, stop, poi, lanes, scratch, off, self_adaptive, lm);
            } else {
                if (fma) icgn2d_poi<DOF, AccLanes, true>(im, rx, ry, conv, stop, poi, lanes, scratch, off, self_adaptive, lm);
                else icgn2d_poi<DOF, AccLanes, false>(im, rx, ry, conv, stop, poi, lanes, scratch, off, self_adaptive, lm);
            }
        }
    }
}

extern "C" {

int oc_oracle_max_threads(void) { return omp_get_max_threads(); }

void oc_oracle_gradient2d(const float* img, int height, int width, float* gx, float* gy, int threads) {
    const float first_factor = 1.f / 12.f;   // src/oc_gradient.cpp:21-22
    const float second_factor = 2.f / 3.f;
    std::memset(gx, 0, sizeof(float) * (size_t)height * width);
    std::memset(gy, 0, sizeof(float) * (size_t)height * width);
    threads = resolve_threads(threads);
#pragma omp parallel for num_threads(threads)
    for (int r = 0; r < height; r++) {
        const float* row = img + (size_t)r * width;
        for (int c = 2; c < width - 2; c++) {  // src/oc_gradient.cpp:45-56
            float result = 0.0f;
            result -= row[c + 2] * first_factor;
            result += row[c + 1] * second_factor;
            result -= row[c - 1] * second_factor;
            result += row[c - 2] * first_factor;
            gx[(size_t)r * width + c] = result;
        }
        if (r >= 2 && r < height - 2) {  // src/oc_gradient.cpp:67-78
            for (int c = 0; c < width; c++) {
                float result = 0.0f;
                result -= img[(size_t)(r + 2) * width + c] * first_factor;
                result += img[(size_t)(r + 1) * width + c] * second_factor;
                result -= img[(size_t)(r - 1) * width + c] * second_factor;
                result += img[(size_t)(r - 2) * width + c] * first_factor;
                gy[(size_t)r * width + c] = result;
            }
        }
    }
}

void oc_oracle_bspline2d_lut(const float* img, int height, int width, float* lut, int threads) {
    std::memset(lut, 0, sizeof(float) * (size_t)height * width * 16);
    threads = resolve_threads(threads);
#pragma omp parallel for num_threads(threads)
    for (int r = 1; r < height - 2; r++) {
        for (int c = 1; c < width - 2; c++) {
            float q[4][4];
            for (int i = 0; i < 4; i++)
                for (int j = 0; j < 4; j++) q[i][j] = img[(size_t)(r - 1 + i) * width + (c - 1 + j)];
            float pm[4][4];
            for (int k = 0; k < 4; k++)
                for (int l = 0; l < 4; l++) {
                    float acc = 0.f;  // src/oc_cubic_bspline.cpp:108-120
                    for (int m = 0; m < 4; m++)
                        for (int n = 0; n < 4; n++) acc += BC[l][m] * BC[k][n] * q[n][m];
                    pm[k][l] = acc;
                }
            float* e = lut + ((size_t)r * width + c) * 16;
            for (int k = 0; k < 4; k++)
                for (int l = 0; l < 4; l++) e[4 * k + l] = pm[3 - k][3 - l];  // src/oc_cubic_bspline.cpp:123-129
        }
    }
}

float oc_oracle_bspline2d_eval(const float* lut, int height, int width, float x, float y) {
    return bspline2d_eval(lut, height, width, x, y);
}
float oc_oracle_bspline2d_eval_fma(const float* lut, int height, int width, float x, float y) {
    return bspline2d_eval<true>(lut, height, width, x, y);
}
int oc_oracle_fma_is_hardware(void) {
#ifdef __FMA__
    return 1;
#else
    return 0;
#endif
}

void oc_oracle_fftcc2d(const float* ref, const float* tar, int height, int width, int rx, int ry, float* pois, long n,
                       int threads, float* surface_out) {
    threads = resolve_threads(threads);
    const int sw = 2 * rx, sh = 2 * ry, size = sw * sh;
    // FFTW is planned as (n0 = width, n1 = height) over a buffer filled [r*width + c]
    // (src/oc_fftcc.cpp:40-42, 204-221); restated as-is.
    std::vector<int> dims = {sw, sh};
#pragma omp parallel num_threads(threads)
    {
        std::vector<float> rsub(size), tsub(size), surf(size);
        std::vector<cplx> buf, spec;
        FFTCache cache;
#pragma omp for schedule(static)
        for (long i = 0; i < n; i++) {
            float* poi = pois + i * OC_POI2D_FLOATS;
            float px = poi[0], py = poi[1];
            float gu = poi[2], gv = poi[8];  // deformation.u, deformation.v
            // src/oc_fftcc.cpp:190-196
            if ((int)px < rx || (int)px >= width - rx || (int)py < ry || (int)py >= height - ry ||
                (int)(px + gu) < rx || (int)(px + gu) >= width - rx || (int)(py + gv) < ry ||
                (int)(py + gv) >= height - ry)
                continue;
            float ref_mean = 0.f, tar_mean = 0.f, ref_norm = 0.f, tar_norm = 0.f;
            for (int r = 0; r < sh; r++)
                for (int c = 0; c < sw; c++) {  // src/oc_fftcc.cpp:204-221
                    float rxp = px + c - rx, ryp = py + r - ry;
                    float v = ref[(size_t)(int)ryp * width + (int)rxp];
                    rsub[r * sw + c] = v;
                    ref_mean += v;
                    float txp = rxp + gu, typ = ryp + gv;
                    v = tar[(size_t)(int)typ * width + (int)txp];
                    tsub[r * sw + c] = v;
                    tar_mean += v;
                }
            ref_mean /= size;
            tar_mean /= size;
            for (int k = 0; k < size; k++) {  // src/oc_fftcc.cpp:225-231
                rsub[k] -= ref_mean;
                tsub[k] -= tar_mean;
                ref_norm += rsub[k] * rsub[k];
                tar_norm += tsub[k] * tsub[k];
            }
            xcorr_nd(rsub.data(), tsub.data(), dims, buf, spec, surf.data(), cache);
            if (surface_out && i == 0) std::memcpy(surface_out, surf.data(), sizeof(float) * size);
            float max_zncc = -2.f;  // src/oc_fftcc.cpp:246-255
            int idx = 0;
            for (int k = 0; k < size; k++)
                if (surf[k] > max_zncc) { max_zncc = surf[k]; idx = k; }
            int du = idx % sw, dv = idx / sw;
            if (du > rx) du -= sw;
            if (dv > ry) dv -= sh;
            poi[2] = (float)du + gu;
            poi[8] = (float)dv + gv;
            poi[14] = gu;
            poi[15] = gv;
            poi[16] = max_zncc / (std::sqrt(ref_norm * tar_norm) * size);
        }
    }
}

void oc_oracle_icgn2d1_ex(const float* ref, const float* gx, const float* gy, const float* tar_lut, int height, int width,
                          int rx, int ry, float conv, float stop, float* pois, long n, int order, int lanes, int threads,
                          const float* center_offsets, int self_adaptive) {
    Images2D im = {ref, gx, gy, tar_lut, height, width};
    icgn2d_queue<6>(im, rx, ry, conv, stop, pois, n, OC_POI2D_FLOATS, order, lanes, threads, center_offsets, self_adaptive, nullptr);
}

void oc_oracle_icgn2d1(const float* ref, const float* gx, const float* gy, const float* tar_lut, int height, int width,
                       int rx, int ry, float conv, float stop, float* pois, long n, int order, int lanes, int threads) {
    Images2D im = {ref, gx, gy, tar_lut, height, width};
    icgn2d_queue<6>(im, rx, ry, conv, stop, pois, n, OC_POI2D_FLOATS, order, lanes, threads, nullptr, 0, nullptr);
}

void oc_oracle_icgn2d2_ex(const float* ref, const float* gx, const float* gy, const float* tar_lut, int height, int width,
                          int rx, int ry, float conv, float stop, float* pois, long n, int order, int lanes, int threads,
                          const float* center_offsets, int self_adaptive) {
    Images2D im = {ref, gx, gy, tar_lut, height, width};
    icgn2d_queue<12>(im, rx, ry, conv, stop, pois, n, OC_POI2D_FLOATS, order, lanes, threads, center_offsets, self_adaptive, nullptr);
}

void oc_oracle_icgn2d2(const float* ref, const float* gx, const float* gy, const float* tar_lut, int height, int width,
                       int rx, int ry, float conv, float stop, float* pois, long n, int order, int lanes, int threads) {
    Images2D im = {ref, gx, gy, tar_lut, height, width};
    icgn2d_queue<12>(im, rx, ry, conv, stop, pois, n, OC_POI2D_FLOATS, order, lanes, threads, nullptr, 0, nullptr);
}

// ICLM2D1 / ICLM2D2 (src/oc_iclm.cpp): dof = 6 or 12; damping = {lambda, alpha, beta} (src/oc_iclm.h:33-38)
void oc_oracle_iclm2d(int dof, const float* ref, const float* gx, const float* gy, const float* tar_lut, int height, int width,
                      int rx, int ry, float conv, float stop, const float* damping, int self_adaptive, float* pois, long n,
                      int stride_floats, int order, int lanes, int threads) {
    Images2D im = {ref, gx, gy, tar_lut, height, width};
    const LmDamping lm = {std::log((double)damping[0]), damping[1], damping[2]};
    if (dof == 6) icgn2d_queue<6>(im, rx, ry, conv, stop, pois, n, stride_floats, order, lanes, threads, nullptr, self_adaptive, &lm);
    else icgn2d_queue<12>(im, rx, ry, conv, stop, pois, n, stride_floats, order, lanes, threads, nullptr, self_adaptive, &lm);
}

void oc_oracle_strain2d(float* pois, long n, int stride_floats, float subregion_radius, int neighbor_number_min,
                        float zncc_threshold, int approximation, int threads) {
    strain_queue<2>(pois, n, stride_floats, subregion_radius, neighbor_number_min, zncc_threshold, approximation, threads);
}

void oc_oracle_strain3d(float* pois, long n, int stride_floats, float subregion_radius, int neighbor_number_min,
                        float zncc_threshold, int approximation, int threads) {
    strain_queue<3>(pois, n, stride_floats, subregion_radius, neighbor_number_min, zncc_threshold, approximation, threads);
}

void oc_oracle_region_fit2d(const float* reliable, long n_reliable, int reliable_stride, float* pois, long n, int stride_floats,
                            float neighbor_search_radius, int neighbor_number_min, int threads) {
    strain_queue<2>(const_cast<float*>(reliable), n_reliable, reliable_stride, neighbor_search_radius, neighbor_number_min, 0.f, 1,
                    threads, 1, pois, n, stride_floats);
}

void oc_oracle_region_fit3d(const float* reliable, long n_reliable, int reliable_stride, float* pois, long n, int stride_floats,
                            float neighbor_search_radius, int neighbor_number_min, int threads) {
    strain_queue<3>(const_cast<float*>(reliable), n_reliable, reliable_stride, neighbor_search_radius, neighbor_number_min, 0.f, 1,
                    threads, 1, pois, n, stride_floats);
}

float oc_oracle_pow_lambda(float lambda, float q) { return pow_lambda(std::log((double)lambda), q); }

// the small dense inverses on their own (tests/test_oracle_dense_algebra.py pins them on LAPACK through numpy: the compiled
// reference of oracle/_ref links a stand-in Eigen, so an independent library has to say that these ARE inverses)
int oc_oracle_inverse(const float* a, float* ainv, int n) {
    if (n == 3) inverse3(a, ainv);
    else if (n == 4) inverse4(a, ainv);
    else if (n >= 1 && n <= 12) lu_inverse(a, ainv, n);
    else return -1;
    return 0;
}
void oc_oracle_mat_mul(const float* a, const float* b, float* c, int n) { mat_mul(a, b, c, n); }

void oc_oracle_nr2d1(const float* ref, const float* tar_lut, const float* tar_lut_gx, const float* tar_lut_gy, int height,
                     int width, int rx, int ry, float conv, float stop, float* pois, long n, int order, int lanes,
                     int threads) {
    threads = resolve_threads(threads);
#pragma omp parallel num_threads(threads)
    {
        std::vector<float> scratch;
#pragma omp for schedule(static)
        for (long i = 0; i < n; i++) {
            if (order == OC_ORDER_SEQ)
                nr2d1_poi<AccSeq>(ref, tar_lut, tar_lut_gx, tar_lut_gy, height, width, rx, ry, conv, stop,
                                  pois + i * OC_POI2D_FLOATS, lanes, scratch);
            else
                nr2d1_poi<AccLanes>(ref, tar_lut, tar_lut_gx, tar_lut_gy, height, width, rx, ry, conv, stop,
                                    pois + i * OC_POI2D_FLOATS, lanes, scratch);
        }
    }
}

void oc_oracle_gradient3d(const float* vol, int dz, int dy, int dx, float* gx, float* gy, float* gz, int threads) {
    const float first_factor = 1.f / 12.f;
    const float second_factor = 2.f / 3.f;
    size_t total = (size_t)dz * dy * dx;
    std::memset(gx, 0, sizeof(float) * total);
    std::memset(gy, 0, sizeof(float) * total);
    std::memset(gz, 0, sizeof(float) * total);
    threads = resolve_threads(threads);
    const size_t sy = dx, sz = (size_t)dy * dx;
#pragma omp parallel for num_threads(threads)
    for (int i = 0; i < dz; i++)
        for (int j = 0; j < dy; j++)
            for (int k = 0; k < dx; k++) {
                size_t g = (size_t)i * sz + (size_t)j * sy + k;
                if (k >= 2 && k < dx - 2) {  // src/oc_gradient.cpp:158-170
                    float result = 0.0f;
                    result -= vol[g + 2] * first_factor;
                    result += vol[g + 1] * second_factor;
                    result -= vol[g - 1] * second_factor;
                    result += vol[g - 2] * first_factor;
                    gx[g] = result;
                }
                if (j >= 2 && j < dy - 2) {  // src/oc_gradient.cpp:187-199
                    float result = 0.0f;
                    result -= vol[g + 2 * sy] * first_factor;
                    result += vol[g + sy] * second_factor;
                    result -= vol[g - sy] * second_factor;
                    result += vol[g - 2 * sy] * first_factor;
                    gy[g] = result;
                }
                if (i >= 2 && i < dz - 2) {  // src/oc_gradient.cpp:216-228
                    float result = 0.0f;
                    result -= vol[g + 2 * sz] * first_factor;
                    result += vol[g + sz] * second_factor;
                    result -= vol[g - sz] * second_factor;
                    result += vol[g - 2 * sz] * first_factor;
                    gz[g] = result;
                }
            }
}

// one 15-tap symmetric pass with clamp-to-edge (src/oc_cubic_bspline.cpp:224-347)
static void prefilter_axis(const float* in, float* out, int dz, int dy, int dx, int axis, int threads) {
    const size_t strides[3] = {(size_t)dy * dx, (size_t)dx, 1};  // z, y, x
    const int dims[3] = {dz, dy, dx};
    const int n = dims[axis];
    const size_t st = strides[axis];
#pragma omp parallel for num_threads(threads)
    for (int i = 0; i < dz; i++)
        for (int j = 0; j < dy; j++)
            for (int k = 0; k < dx; k++) {
                size_t g = (size_t)i * strides[0] + (size_t)j * strides[1] + k;
                int pos = axis == 0 ? i : (axis == 1 ? j : k);
                const float* base = in + g - (size_t)pos * st;
                float acc = PREF[0] * base[(size_t)pos * st];
                for (int t = 1; t <= 7; t++) {
                    int lo = pos - t < 0 ? 0 : pos - t;
                    int hi = pos + t > n - 1 ? n - 1 : pos + t;
                    acc = acc + PREF[t] * (base[(size_t)lo * st] + base[(size_t)hi * st]);
                }
                out[g] = acc;
            }
}

void oc_oracle_bspline3d_prefilter(const float* vol, int dz, int dy, int dx, float* coef, int threads) {
    threads = resolve_threads(threads);
    std::vector<float> tmp((size_t)dz * dy * dx);
    prefilter_axis(vol, coef, dz, dy, dx, 2, threads);          // x: image -> coefficient
    prefilter_axis(coef, tmp.data(), dz, dy, dx, 1, threads);   // y: coefficient -> buffer
    prefilter_axis(tmp.data(), coef, dz, dy, dx, 0, threads);   // z: buffer -> coefficient
}

float oc_oracle_bspline3d_eval(const float* coef, int dz, int dy, int dx, float x, float y, float z) {
    return bspline3d_eval(coef, dz, dy, dx, x, y, z);
}
float oc_oracle_bspline3d_eval_fma(const float* coef, int dz, int dy, int dx, float x, float y, float z) {
    return bspline3d_eval<true>(coef, dz, dy, dx, x, y, z);
}

// exact_sums = 0: the reference's own arithmetic -- means and norms as sequential float32 running sums over the window
// (src/oc_fftcc.cpp:340-376).  exact_sums = 1: the same windows, the same first-maximum rule, but means, zero-mean values,
// norms and the final quotient in double, rounded once: the ZNCC of the peak WITHOUT the rounding noise of a 10^4 ... 10^5-term
// float running sum (which reaches 1e-4 at 32^3 and 3e-4 at 60^3 -- more than north_star's tolerance, and it is the
// reference's own noise).  The integer displacements are the same in both modes whenever the peak is not a float-level tie.
void oc_oracle_fftcc3d_ex(const float* ref, const float* tar, int /*dz*/, int dy, int dx, int rx, int ry, int rz, float* pois,
                          long n, int threads, int exact_sums) {
    threads = resolve_threads(threads);
    const int sx = 2 * rx, sy = 2 * ry, sz = 2 * rz;
    const int size = sx * sy * sz;
    // planned as (dim_x, dim_y, dim_z) over a buffer filled [(i*dim_y + j)*dim_x + k]
    // (src/oc_fftcc.cpp:68-70, 349-360); restated as-is.
    std::vector<int> dims = {sx, sy, sz};
#pragma omp parallel num_threads(threads)
    {
        std::vector<float> rsub(size), tsub(size), surf(size);
        std::vector<cplx> buf, spec;
        FFTCache cache;
#pragma omp for schedule(dynamic, 1)
        for (long q = 0; q < n; q++) {
            float* poi = pois + q * OC_POI3D_FLOATS;
            float px = poi[0], py = poi[1], pz = poi[2];
            float gu = poi[3], gv = poi[7], gw = poi[11];
            float ref_mean = 0.f, tar_mean = 0.f, ref_norm = 0.f, tar_norm = 0.f;
            double ref_sum_d = 0.0, tar_sum_d = 0.0;
            for (int i = 0; i < sz; i++)
                for (int j = 0; j < sy; j++)
                    for (int k = 0; k < sx; k++) {
                        float rxp = px + k - rx, ryp = py + j - ry, rzp = pz + i - rz;
                        float v = ref[((size_t)(int)rzp * dy + (int)ryp) * dx + (int)rxp];
                        rsub[(i * sy + j) * sx + k] = v;
                        ref_mean += v;
                        ref_sum_d += v;
                        float txp = rxp + gu, typ = ryp + gv, tzp = rzp + gw;
                        v = tar[((size_t)(int)tzp * dy + (int)typ) * dx + (int)txp];
                        tsub[(i * sy + j) * sx + k] = v;
                        tar_mean += v;
                        tar_sum_d += v;
                    }
            if (exact_sums) {
                const double rm = ref_sum_d / size, tm = tar_sum_d / size;
                double rn = 0.0, tn = 0.0;
                std::vector<cplx>& z = buf;
                z.resize(size);
                for (int k = 0; k < size; k++) {
                    const double a = (double)rsub[k] - rm, b = (double)tsub[k] - tm;
                    rn += a * a;
                    tn += b * b;
                    z[k] = cplx(a, b);
                }
                // the correlation surface of the double zero-mean windows (same transform as xcorr_nd)
                fft_nd(z, dims, -1, cache);
                spec.resize(size);
                std::vector<size_t> strides = {(size_t)sy * sz, (size_t)sz, 1};
                for (size_t q2 = 0; q2 < (size_t)size; q2++) {
                    size_t rem = q2, neg = 0;
                    for (size_t ax = 0; ax < 3; ax++) {
                        const size_t kk = rem / strides[ax];
                        rem -= kk * strides[ax];
                        neg += ((dims[ax] - kk) % dims[ax]) * strides[ax];
                    }
                    const cplx zk = z[q2], znk = std::conj(z[neg]);
                    spec[q2] = std::conj(0.5 * (zk + znk)) * (cplx(0.0, -0.5) * (zk - znk));
                }
                fft_nd(spec, dims, +1, cache);
                double best = -2.0 * std::sqrt(rn * tn) * size;
                int idx = 0;
                for (int k = 0; k < size; k++)
                    if (spec[k].real() > best) { best = spec[k].real(); idx = k; }
                int du = idx % sx, dv = (idx / sx) % sy, dw = idx / (sx * sy);
                if (du > rx) du -= sx;
                if (dv > ry) dv -= sy;
                if (dw > rz) dw -= sz;
                poi[3] = (float)du + gu; poi[7] = (float)dv + gv; poi[11] = (float)dw + gw;
                poi[15] = gu; poi[16] = gv; poi[17] = gw;
                poi[18] = (float)(best / (std::sqrt(rn * tn) * size));
                continue;
            }
            ref_mean /= size;
            tar_mean /= size;
            for (int k = 0; k < size; k++) {
                rsub[k] -= ref_mean;
                tsub[k] -= tar_mean;
                ref_norm += rsub[k] * rsub[k];
                tar_norm += tsub[k] * tsub[k];
            }
            xcorr_nd(rsub.data(), tsub.data(), dims, buf, spec, surf.data(), cache);
            float max_zncc = -2.f;
            int idx = 0;
            for (int k = 0; k < size; k++)
                if (surf[k] > max_zncc) { max_zncc = surf[k]; idx = k; }
            int du = idx % sx, dv = (idx / sx) % sy, dw = idx / (sx * sy);
            if (du > rx) du -= sx;
            if (dv > ry) dv -= sy;
            if (dw > rz) dw -= sz;
            poi[3] = (float)du + gu;
            poi[7] = (float)dv + gv;
            poi[11] = (float)dw + gw;
            poi[15] = gu;
            poi[16] = gv;
            poi[17] = gw;
            poi[18] = max_zncc / (std::sqrt(ref_norm * tar_norm) * size);
        }
    }
}

void oc_oracle_fftcc3d(const float* ref, const float* tar, int dz, int dy, int dx, int rx, int ry, int rz, float* pois,
                       long n, int threads) {
    oc_oracle_fftcc3d_ex(ref, tar, dz, dy, dx, rx, ry, rz, pois, n, threads, 0);
}

void oc_oracle_icgn3d1(const float* ref, const float* gx, const float* gy, const float* gz, const float* tar_coef,
                       int dz, int dy, int dx, int rx, int ry, int rz, float conv, float stop, float* pois, long n,
                       int order, int lanes, int threads) {
    threads = resolve_threads(threads);
    Images3D im = {ref, gx, gy, gz, tar_coef, dz, dy, dx};
#pragma omp parallel num_threads(threads)
    {
        std::vector<float> scratch;
#pragma omp for schedule(dynamic, 1)
        for (long i = 0; i < n; i++) {
            float* poi = pois + i * OC_POI3D_FLOATS;
            const bool fma = (order & OC_ARITH_FMA) != 0;
            const int assoc = order & ~OC_ARITH_FMA;
            if (assoc == OC_ORDER_SEQ) {
                if (fma) icgn3d1_poi<AccSeq, true>(im, rx, ry, rz, conv, stop, poi, lanes, scratch);
                else icgn3d1_poi<AccSeq, false>(im, rx, ry, rz, conv, stop, poi, lanes, scratch);
            } else if (assoc == OC_ORDER_ROWS) {
                icgn3d1_poi<AccRows, false>(im, rx, ry, rz, conv, stop, poi, lanes, scratch);  // (no fused form: oc_oracle.h)
            } else {
                if (fma) icgn3d1_poi<AccLanes, true>(im, rx, ry, rz, conv, stop, poi, lanes, scratch);
                else icgn3d1_poi<AccLanes, false>(im, rx, ry, rz, conv, stop, poi, lanes, scratch);
            }
        }
    }
}

}  // extern "C"
