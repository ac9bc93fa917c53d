// fft_host_check.hip -- the register FFTs of opencorr_amd/csrc/fft_device.h, executed on the HOST against a
// double-precision DFT (test infrastructure; tests/test_fft_device_host.py builds and runs it, no GPU needed).
// For every window side the fused FFTCC2D kernels are instantiated for: forward and inverse transform of a random
// complex line, outputs read through fft_pos() exactly like the kernels do; prints one line per size
//     N  max|err| / max|X|  (forward)  (inverse)
// and exits non-zero if any relative error exceeds 2e-6 * log2(N) -- float butterflies, nothing else.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../opencorr_amd/csrc/fft_device.h"

using namespace ochip::fftdev;

template <bool INV, int N>
static double check_one(unsigned seed) {
    c2 v[N];
    std::vector<double> re(N), im(N);
    srand(seed);
    for (int i = 0; i < N; i++) {
        re[i] = (rand() / (double)RAND_MAX) * 2.0 - 1.0;
        im[i] = (rand() / (double)RAND_MAX) * 2.0 - 1.0;
        v[i] = mkc((float)re[i], (float)im[i]);
        re[i] = (double)v[i].x;
        im[i] = (double)v[i].y;
    }
    fft_mixed<INV, N>(v);
    double worst = 0.0, scale = 0.0;
    const double sign = INV ? 1.0 : -1.0;
    for (int k = 0; k < N; k++) {
        double xr = 0.0, xi = 0.0;
        for (int n = 0; n < N; n++) {
            const double a = sign * 2.0 * M_PI * (double)((long long)k * n % N) / N;
            xr += re[n] * cos(a) - im[n] * sin(a);
            xi += re[n] * sin(a) + im[n] * cos(a);
        }
        const c2 got = v[fft_pos(N, k)];
        worst = fmax(worst, fmax(fabs((double)got.x - xr), fabs((double)got.y - xi)));
        scale = fmax(scale, fmax(fabs(xr), fabs(xi)));
    }
    return worst / scale;
}

static int failures = 0;

template <int N>
static void check() {
    const double f = check_one<false, N>(1000u + N), b = check_one<true, N>(2000u + N);
    const double bar = 2e-6 * log2((double)N);
    const bool ok = f <= bar && b <= bar;
    printf("%d %.3e %.3e %s\n", N, f, b, ok ? "ok" : "FAIL");
    if (!ok) failures++;
}

int main() {
    // every even side from 8 to 64 (FFTCC2D windows are 2 * radius wide)
    check<8>(); check<10>(); check<12>(); check<14>(); check<16>(); check<18>(); check<20>(); check<22>(); check<24>();
    check<26>(); check<28>(); check<30>(); check<32>(); check<34>(); check<36>(); check<38>(); check<40>(); check<42>();
    check<44>(); check<46>(); check<48>(); check<50>(); check<52>(); check<54>(); check<56>(); check<58>(); check<60>();
    check<62>(); check<64>();
    // the prime butterflies on their own, and odd composites (FFTCC3D planes may use them)
    check<7>(); check<11>(); check<13>(); check<17>(); check<19>(); check<23>(); check<29>(); check<31>(); check<15>();
    check<21>(); check<25>(); check<27>(); check<33>(); check<35>(); check<45>(); check<49>(); check<55>(); check<63>();
    return failures ? 1 : 0;
}
