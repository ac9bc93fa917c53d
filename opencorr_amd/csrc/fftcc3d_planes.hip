// fftcc3d_planes.hip -- the plane-wise single-kernel FFTCC3D (fftcc3d_planes_impl.h) for cubic windows of side 28 ... 64
// except 32: the dispatcher and the instances up to side 46; sides 48 ... 64 (among them the 60^3 windows of the reference's
// DVC example) are instantiated in fftcc3d_planesb.hip.
#include "fftcc3d_planes_impl.h"

namespace ochip {

using planes::launch_planes;

hipError_t launch_fftcc3d_planes_b(const Fftcc3dParams& p, float* pois, int stride_f, size_t count, void* scratch, int blocks,
                                   hipStream_t stream);  // fftcc3d_planesb.hip

// cubic windows of side 28 ... 64 except 32 (which has the register-resident kernel): radius 14, 15, 17 ... 32
bool fftcc3d_planes_supported(int rx, int ry, int rz) { return rx == ry && ry == rz && rx >= 14 && rx <= 32 && rx != 16; }

// bytes of scratch for `blocks` persistent workgroups (one complex N^3 volume each); blocks is a multiple of 8
size_t fftcc3d_planes_scratch_bytes(int r, int blocks) {
    const size_t n = (size_t)(2 * r);
    return n * n * n * 8u * (size_t)blocks;
}

hipError_t launch_fftcc3d_planes(const Fftcc3dParams& p, float* pois, int stride_f, size_t count, void* scratch, int blocks,
                                 hipStream_t stream) {
    if (count == 0) return hipSuccess;
    if (!fftcc3d_planes_supported(p.rx, p.ry, p.rz) || !scratch || blocks < 8 || (blocks & 7)) return hipErrorInvalidValue;
    if (2 * p.rx >= 48) return launch_fftcc3d_planes_b(p, pois, stride_f, count, scratch, blocks, stream);
    switch (2 * p.rx) {
#define OC_PLANES_CASE(NN) \
    case NN: return launch_planes<NN>(p, pois, stride_f, count, scratch, blocks, stream);
        OC_PLANES_CASE(28) OC_PLANES_CASE(30) OC_PLANES_CASE(34) OC_PLANES_CASE(36) OC_PLANES_CASE(38) OC_PLANES_CASE(40)
        OC_PLANES_CASE(42) OC_PLANES_CASE(44) OC_PLANES_CASE(46)
#undef OC_PLANES_CASE
        default: return hipErrorInvalidValue;
    }
}

}  // namespace ochip
