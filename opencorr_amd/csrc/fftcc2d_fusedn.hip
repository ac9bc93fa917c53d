// fftcc2d_fusedn.hip -- single-kernel FFTCC2D for square windows whose side N = 2 * radius is not 32
// (N = 16, 18, 20, 24, 30, 36, 40, 48, 50, 60, 64; N = 32 has its own kernel in fftcc2d_fused.hip); rectangular windows
// (rx != ry) are instantiated in fftcc2d_fusedr.hip from the same template (fftcc2d_fusedn_impl.h).
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

bool fftcc2d_fusedn_supported(int rx, int ry) {
    if (rx != ry) return fftcc2d_fusedr_supported(rx, ry);
    return rx == 8 || rx == 9 || rx == 10 || rx == 12 || rx == 15 || rx == 16 || rx == 18 || rx == 20 || rx == 24 || rx == 25 || rx == 30 || rx == 32;
}

hipError_t launch_fftcc2d_fusedn(const Fftcc2dParams& p, float* pois, int stride_f, size_t count, bool xcd,
                                 hipStream_t stream) {
    if (count == 0) return hipSuccess;
    if (!fftcc2d_fusedn_supported(p.rx, p.ry)) return hipErrorInvalidValue;
    if (p.rx != p.ry) return launch_fftcc2d_fusedr(p, pois, stride_f, count, xcd, stream);
    switch (2 * p.rx) {
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
        default: return launch_n<48, 48>(p, pois, stride_f, count, xcd, stream);
    }
}

}  // namespace ochip
