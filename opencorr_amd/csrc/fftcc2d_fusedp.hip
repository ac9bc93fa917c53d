// fftcc2d_fusedp.hip -- the single-kernel FFTCC2D (fftcc2d_fusedn_impl.h) for the square window sides with a prime factor
// above 5: N = 14, 22, 26, 28, 42, 44, 52, 56 (7, 11, 13) and 34, 38, 46, 58, 62 (17, 19, 23, 29, 31), i.e. the radii 7, 11, 13,
// 14, 17, 19, 21, 22, 23, 26, 28, 29, 31 that fell onto the five-kernel rocFFT pipeline until round 3 (3 - 8 x slower per POI,
// costing more than the ICGN refinement behind it).  The prime factors are transformed by fft_device.h dft_prime -- the
// symmetric (p_j, m_j) form of a P-point DFT, (P-1)^2 / 2 real-by-complex multiply-adds, all in registers; checked for
// every size on the host against a double-precision DFT (tests/test_fft_device_host.py) and on the GPU against the oracle
// and the rocFFT pipeline (tests/test_gpu_parity_2d.py::test_fftcc2d_every_fused_shape).
#include "fftcc2d_fusedn_impl.h"

namespace ochip {

using fusedn::launch_n;

bool fftcc2d_fusedp_supported(int r) {
    switch (2 * r) {
        case 14: case 22: case 26: case 28: case 34: case 38: case 42: case 44: case 46: case 52: case 56: case 58: case 62: return true;
        default: return false;
    }
}

hipError_t launch_fftcc2d_fusedp(const Fftcc2dParams& p, float* pois, int stride_f, size_t count, bool xcd, hipStream_t stream) {
    if (count == 0) return hipSuccess;
    if (p.rx != p.ry) return hipErrorInvalidValue;
    switch (2 * p.rx) {
        case 14: return launch_n<14, 14>(p, pois, stride_f, count, xcd, stream);
        case 22: return launch_n<22, 22>(p, pois, stride_f, count, xcd, stream);
        case 26: return launch_n<26, 26>(p, pois, stride_f, count, xcd, stream);
        case 28: return launch_n<28, 28>(p, pois, stride_f, count, xcd, stream);
        case 34: return launch_n<34, 34>(p, pois, stride_f, count, xcd, stream);
        case 38: return launch_n<38, 38>(p, pois, stride_f, count, xcd, stream);
        case 42: return launch_n<42, 42>(p, pois, stride_f, count, xcd, stream);
        case 44: return launch_n<44, 44>(p, pois, stride_f, count, xcd, stream);
        case 46: return launch_n<46, 46>(p, pois, stride_f, count, xcd, stream);
        case 52: return launch_n<52, 52>(p, pois, stride_f, count, xcd, stream);
        case 56: return launch_n<56, 56>(p, pois, stride_f, count, xcd, stream);
        case 58: return launch_n<58, 58>(p, pois, stride_f, count, xcd, stream);
        case 62: return launch_n<62, 62>(p, pois, stride_f, count, xcd, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace ochip
