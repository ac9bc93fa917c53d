// fftcc2d_fusedn.hip -- single-kernel FFTCC2D for square windows whose side N = 2 * radius is not 32: every even side
// from 8 to 64, i.e. every radius from 4 to 32 (N = 32 has its own kernel in fftcc2d_fused.hip).  The 5-smooth sides
// (8, 10, 12, 16, 18, 20, 24, 30, 36, 40, 48, 50, 54, 60, 64) are instantiated here, the sides with a prime factor of 7 ... 31
// (14, 22, 26, 28, 34, 38, 42, 44, 46, 52, 56, 58, 62; round 4: fft_device.h dft_prime) in fftcc2d_fusedp.hip; rectangular
// windows (rx != ry) in fftcc2d_fusedr.hip -- all from the same template (fftcc2d_fusedn_impl.h).
//
// Same plan as the 32 x 32 kernel (fftcc2d_fused32x2_kernel): the whole FFTCC2D::compute(POI2D*) (src/oc_fftcc.cpp:177-275)
// on chip -- z = ref + i*tar, ONE complex NR x NC FFT, R(k) = (Z(k) + conj Z(-k))/2, T(k) = (Z(k) - conj Z(-k))/(2i),
// C = conj(R) T, inverse FFT (unnormalised like FFTW's c2r), arg-max with the first-max rule.
// A lane owns a whole column, then a whole line: lane l gathers column l of the transform's array into its registers and
// transforms it there with an N-point mixed-radix FFT (fft_device.h: factors 2, 3, 4, 5, twiddles from a generated
// table); an NR x (NC + 1) tile of floats in LDS (odd pitch: lines and columns both conflict-free; real parts travel
// first, imaginary parts second) carries the data between the passes.  Windows whose longer side fits 32 lanes share a
// wave in pairs.  The rocFFT pipeline this replaces spends more time on these windows than the ICGN refinement that
// follows it (config C, r = 20: 4.9 ms of FFTCC against 4.2 ms of ICGN2D2 for 99 856 POIs).
#include "fftcc2d_fusedn_impl.h"

namespace ochip {

using fusedn::launch_n;

bool fftcc2d_fusedr_supported(int rx, int ry);  // fftcc2d_fusedr.hip
hipError_t launch_fftcc2d_fusedr(const Fftcc2dParams& p, float* pois, int stride_f, size_t count, bool xcd, hipStream_t stream);
bool fftcc2d_fusedp_supported(int r);            // fftcc2d_fusedp.hip: sides with a prime factor of 7 ... 31
hipError_t launch_fftcc2d_fusedp(const Fftcc2dParams& p, float* pois, int stride_f, size_t count, bool xcd, hipStream_t stream);

bool fftcc2d_fusedn_supported(int rx, int ry) {
    if (rx != ry) return fftcc2d_fusedr_supported(rx, ry);
    return rx >= 4 && rx <= 32;  // every even window side from 8 to 64
}

hipError_t launch_fftcc2d_fusedn(const Fftcc2dParams& p, float* pois, int stride_f, size_t count, bool xcd,
                                 hipStream_t stream) {
    if (count == 0) return hipSuccess;
    if (!fftcc2d_fusedn_supported(p.rx, p.ry)) return hipErrorInvalidValue;
    if (p.rx != p.ry) return launch_fftcc2d_fusedr(p, pois, stride_f, count, xcd, stream);
    if (fftcc2d_fusedp_supported(p.rx)) return launch_fftcc2d_fusedp(p, pois, stride_f, count, xcd, stream);
    switch (2 * p.rx) {
        case 8: return launch_n<8, 8>(p, pois, stride_f, count, xcd, stream);
        case 10: return launch_n<10, 10>(p, pois, stride_f, count, xcd, stream);
        case 12: return launch_n<12, 12>(p, pois, stride_f, count, xcd, stream);
        case 54: return launch_n<54, 54>(p, pois, stride_f, count, xcd, stream);
        case 16: return launch_n<16, 16>(p, pois, stride_f, count, xcd, stream);
        case 32: return launch_n<32, 32>(p, pois, stride_f, count, xcd, stream);
        case 18: return launch_n<18, 18>(p, pois, stride_f, count, xcd, stream);
        case 50: return launch_n<50, 50>(p, pois, stride_f, count, xcd, stream);
        case 60: return launch_n<60, 60>(p, pois, stride_f, count, xcd, stream);
        case 64: return launch_n<64, 64>(p, pois, stride_f, count, xcd, stream);
        case 20: return launch_n<20, 20>(p, pois, stride_f, count, xcd, stream);
        case 24: return launch_n<24, 24>(p, pois, stride_f, count, xcd, stream);
        case 36: return launch_n<36, 36>(p, pois, stride_f, count, xcd, stream);
        case 30: return launch_n<30, 30>(p, pois, stride_f, count, xcd, stream);
        case 40: return launch_n<40, 40>(p, pois, stride_f, count, xcd, stream);
        case 48: return launch_n<48, 48>(p, pois, stride_f, count, xcd, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace ochip
