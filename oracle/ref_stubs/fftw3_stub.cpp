// fftw3_stub.cpp -- the transforms behind oracle/ref_stubs/fftw3.h (TEST INFRASTRUCTURE ONLY).
//
// r2c: X[k] = sum_n x[n] exp(-2 pi i k.n / N), half spectrum along the last dimension; c2r: the unnormalised inverse
// of a Hermitian half spectrum -- FFTW's definitions (fftw.org/doc: "The 1d Discrete Fourier Transform",
// "Multi-dimensional Transforms").  Evaluated axis by axis as plain O(n^2) DFTs in double with exact-argument
// twiddles, one rounding to float at the end.
#include "fftw3.h"

#include <cmath>
#include <complex>
#include <cstdlib>
#include <vector>

namespace {
typedef std::complex<double> cplx;

struct Twiddles {
    int n = 0;
    std::vector<cplx> w;  // exp(-2 pi i k / n)
    void init(int n_) {
        n = n_;
        w.resize(n_);
        const double two_pi = 6.283185307179586476925286766559;
        for (int k = 0; k < n_; k++) w[k] = cplx(std::cos(two_pi * k / n_), -std::sin(two_pi * k / n_));
    }
};

// in-place DFT of every line along `axis` of a row-major dims[0] x dims[1] x dims[2] complex array
void dft_axis(std::vector<cplx>& a, const int dims[3], int axis, int sign, const Twiddles& tw) {
    const int n = dims[axis];
    size_t stride = 1;
    for (int ax = 2; ax > axis; ax--) stride *= dims[ax];
    const size_t total = (size_t)dims[0] * dims[1] * dims[2], outer = total / ((size_t)n * stride);
    std::vector<cplx> line(n), res(n);
    for (size_t o = 0; o < outer; o++)
        for (size_t i = 0; i < stride; i++) {
            cplx* base = &a[o * n * stride + i];
            for (int k = 0; k < n; k++) line[k] = base[(size_t)k * stride];
            for (int k = 0; k < n; k++) {
                cplx acc = 0.0;
                for (int j = 0; j < n; j++) {
                    const cplx w = tw.w[(size_t)(((long long)j * k) % n)];
                    acc += line[j] * (sign < 0 ? w : std::conj(w));
                }
                res[k] = acc;
            }
            for (int k = 0; k < n; k++) base[(size_t)k * stride] = res[k];
        }
}
}  // namespace

struct oc_stub_fftwf_plan_s {
    int dims[3];  // slowest .. fastest; 2D plans have dims[0] = 1
    bool forward;
    float* real;
    fftwf_complex* freq;
    Twiddles tw[3];
};

static fftwf_plan make_plan(int n0, int n1, int n2, bool forward, float* real, fftwf_complex* freq) {
    fftwf_plan p = new oc_stub_fftwf_plan_s;
    p->dims[0] = n0;
    p->dims[1] = n1;
    p->dims[2] = n2;
    p->forward = forward;
    p->real = real;
    p->freq = freq;
    for (int a = 0; a < 3; a++) p->tw[a].init(p->dims[a]);
    return p;
}

extern "C" {

void* fftw_malloc(size_t n) { return std::malloc(n); }
void fftw_free(void* p) { std::free(p); }
void* fftwf_malloc(size_t n) { return std::malloc(n); }
void fftwf_free(void* p) { std::free(p); }

fftwf_plan fftwf_plan_dft_r2c_2d(int n0, int n1, float* in, fftwf_complex* out, unsigned) { return make_plan(1, n0, n1, true, in, out); }
fftwf_plan fftwf_plan_dft_c2r_2d(int n0, int n1, fftwf_complex* in, float* out, unsigned) { return make_plan(1, n0, n1, false, out, in); }
fftwf_plan fftwf_plan_dft_r2c_3d(int n0, int n1, int n2, float* in, fftwf_complex* out, unsigned) { return make_plan(n0, n1, n2, true, in, out); }
fftwf_plan fftwf_plan_dft_c2r_3d(int n0, int n1, int n2, fftwf_complex* in, float* out, unsigned) { return make_plan(n0, n1, n2, false, out, in); }

void fftwf_execute(const fftwf_plan p) {
    const int* d = p->dims;
    const int nh = d[2] / 2 + 1;
    const size_t total = (size_t)d[0] * d[1] * d[2];
    std::vector<cplx> a(total);
    if (p->forward) {
        for (size_t i = 0; i < total; i++) a[i] = cplx((double)p->real[i], 0.0);
        for (int ax = 0; ax < 3; ax++)
            if (d[ax] > 1) dft_axis(a, d, ax, -1, p->tw[ax]);
        for (int i = 0; i < d[0]; i++)
            for (int j = 0; j < d[1]; j++)
                for (int k = 0; k < nh; k++) {
                    const cplx v = a[((size_t)i * d[1] + j) * d[2] + k];
                    fftwf_complex& o = p->freq[((size_t)i * d[1] + j) * nh + k];
                    o[0] = (float)v.real();
                    o[1] = (float)v.imag();
                }
    } else {
        // Hermitian extension X[-k] = conj(X[k]) of the half spectrum, then the inverse transform
        for (int i = 0; i < d[0]; i++)
            for (int j = 0; j < d[1]; j++)
                for (int k = 0; k < d[2]; k++) {
                    cplx v;
                    if (k < nh) {
                        const fftwf_complex& s = p->freq[((size_t)i * d[1] + j) * nh + k];
                        v = cplx((double)s[0], (double)s[1]);
                    } else {
                        const int ni = (d[0] - i) % d[0], nj = (d[1] - j) % d[1], nk = d[2] - k;
                        const fftwf_complex& s = p->freq[((size_t)ni * d[1] + nj) * nh + nk];
                        v = cplx((double)s[0], -(double)s[1]);
                    }
                    a[((size_t)i * d[1] + j) * d[2] + k] = v;
                }
        for (int ax = 0; ax < 3; ax++)
            if (d[ax] > 1) dft_axis(a, d, ax, +1, p->tw[ax]);
        for (size_t i = 0; i < total; i++) p->real[i] = (float)a[i].real();
    }
}

void fftwf_destroy_plan(fftwf_plan p) { delete p; }

}  // extern "C"
