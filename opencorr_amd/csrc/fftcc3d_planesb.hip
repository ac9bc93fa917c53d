// fftcc3d_planesb.hip -- instances of the plane-wise single-kernel FFTCC3D (fftcc3d_planes_impl.h) for the window sides
// 48 ... 64 (radius 24 ... 32), among them 60^3: the subset radius 30 of the reference's own DVC example
// (examples/test_dvc_fftcc_icgn1.cpp:45-47).
#include "fftcc3d_planes_impl.h"

namespace ochip {

using planes::launch_planes;

hipError_t launch_fftcc3d_planes_b(const Fftcc3dParams& p, float* pois, int stride_f, size_t count, void* scratch, int blocks,
                                   hipStream_t stream) {
    switch (2 * p.rx) {
#define OC_PLANES_CASE(NN) \
    case NN: return launch_planes<NN>(p, pois, stride_f, count, scratch, blocks, stream);
        OC_PLANES_CASE(48) OC_PLANES_CASE(50) OC_PLANES_CASE(52) OC_PLANES_CASE(54) OC_PLANES_CASE(56) OC_PLANES_CASE(58)
        OC_PLANES_CASE(60) OC_PLANES_CASE(62) OC_PLANES_CASE(64)
#undef OC_PLANES_CASE
        default: return hipErrorInvalidValue;
    }
}

}  // namespace ochip
