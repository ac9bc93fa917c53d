// fftcc2d_fusedr.hip -- the single-kernel FFTCC2D (fftcc2d_fusedn_impl.h) for RECTANGULAR windows, rx != ry, with both
// sides out of {16, 20, 24, 32, 40, 48, 64} (radii 8, 10, 12, 16, 20, 24, 32): 42 instances of the register-FFT kernel.
// Other rectangular shapes keep the rocFFT pipeline (fftcc2d.hip).  What is transformed is the reference's own re-cut of
// the window buffer (FFTW planned with (width, height) over data filled [row * width + col], src/oc_fftcc.cpp:40-42,
// 204-221): 2 * rx lines of 2 * ry elements -- see the kernel.
#include "fftcc2d_fusedn_impl.h"

namespace ochip {

using fusedn::launch_n;

static bool side_ok(int n) { return n == 16 || n == 20 || n == 24 || n == 32 || n == 40 || n == 48 || n == 64; }

bool fftcc2d_fusedr_supported(int rx, int ry) { return rx != ry && side_ok(2 * rx) && side_ok(2 * ry); }

hipError_t launch_fftcc2d_fusedr(const Fftcc2dParams& p, float* pois, int stride_f, size_t count, bool xcd, hipStream_t stream) {
    if (count == 0) return hipSuccess;
    if (!fftcc2d_fusedr_supported(p.rx, p.ry)) return hipErrorInvalidValue;
    switch (2 * p.rx * 100 + 2 * p.ry) {
        case 1620: return launch_n<16, 20>(p, pois, stride_f, count, xcd, stream);
        case 1624: return launch_n<16, 24>(p, pois, stride_f, count, xcd, stream);
        case 1632: return launch_n<16, 32>(p, pois, stride_f, count, xcd, stream);
        case 1640: return launch_n<16, 40>(p, pois, stride_f, count, xcd, stream);
        case 1648: return launch_n<16, 48>(p, pois, stride_f, count, xcd, stream);
        case 1664: return launch_n<16, 64>(p, pois, stride_f, count, xcd, stream);
        case 2016: return launch_n<20, 16>(p, pois, stride_f, count, xcd, stream);
        case 2024: return launch_n<20, 24>(p, pois, stride_f, count, xcd, stream);
        case 2032: return launch_n<20, 32>(p, pois, stride_f, count, xcd, stream);
        case 2040: return launch_n<20, 40>(p, pois, stride_f, count, xcd, stream);
        case 2048: return launch_n<20, 48>(p, pois, stride_f, count, xcd, stream);
        case 2064: return launch_n<20, 64>(p, pois, stride_f, count, xcd, stream);
        case 2416: return launch_n<24, 16>(p, pois, stride_f, count, xcd, stream);
        case 2420: return launch_n<24, 20>(p, pois, stride_f, count, xcd, stream);
        case 2432: return launch_n<24, 32>(p, pois, stride_f, count, xcd, stream);
        case 2440: return launch_n<24, 40>(p, pois, stride_f, count, xcd, stream);
        case 2448: return launch_n<24, 48>(p, pois, stride_f, count, xcd, stream);
        case 2464: return launch_n<24, 64>(p, pois, stride_f, count, xcd, stream);
        case 3216: return launch_n<32, 16>(p, pois, stride_f, count, xcd, stream);
        case 3220: return launch_n<32, 20>(p, pois, stride_f, count, xcd, stream);
        case 3224: return launch_n<32, 24>(p, pois, stride_f, count, xcd, stream);
        case 3240: return launch_n<32, 40>(p, pois, stride_f, count, xcd, stream);
        case 3248: return launch_n<32, 48>(p, pois, stride_f, count, xcd, stream);
        case 3264: return launch_n<32, 64>(p, pois, stride_f, count, xcd, stream);
        case 4016: return launch_n<40, 16>(p, pois, stride_f, count, xcd, stream);
        case 4020: return launch_n<40, 20>(p, pois, stride_f, count, xcd, stream);
        case 4024: return launch_n<40, 24>(p, pois, stride_f, count, xcd, stream);
        case 4032: return launch_n<40, 32>(p, pois, stride_f, count, xcd, stream);
        case 4048: return launch_n<40, 48>(p, pois, stride_f, count, xcd, stream);
        case 4064: return launch_n<40, 64>(p, pois, stride_f, count, xcd, stream);
        case 4816: return launch_n<48, 16>(p, pois, stride_f, count, xcd, stream);
        case 4820: return launch_n<48, 20>(p, pois, stride_f, count, xcd, stream);
        case 4824: return launch_n<48, 24>(p, pois, stride_f, count, xcd, stream);
        case 4832: return launch_n<48, 32>(p, pois, stride_f, count, xcd, stream);
        case 4840: return launch_n<48, 40>(p, pois, stride_f, count, xcd, stream);
        case 4864: return launch_n<48, 64>(p, pois, stride_f, count, xcd, stream);
        case 6416: return launch_n<64, 16>(p, pois, stride_f, count, xcd, stream);
        case 6420: return launch_n<64, 20>(p, pois, stride_f, count, xcd, stream);
        case 6424: return launch_n<64, 24>(p, pois, stride_f, count, xcd, stream);
        case 6432: return launch_n<64, 32>(p, pois, stride_f, count, xcd, stream);
        case 6440: return launch_n<64, 40>(p, pois, stride_f, count, xcd, stream);
        case 6448: return launch_n<64, 48>(p, pois, stride_f, count, xcd, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace ochip
