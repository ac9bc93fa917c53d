// prepare3d.hip -- volume-level precompute of ICGN3D1::prepare() on gfx950.
//
//   grad3d_kernel            Gradient3D4::getGradientX/Y/Z     (src/oc_gradient.cpp:143-231)
//   bspline3d_prefilter_axis TricubicBspline::prepare          (src/oc_cubic_bspline.cpp:214-351)
//
// Streaming kernels, one thread per voxel, x fastest so every access is coalesced; the 15-tap
// prefilter is run three times (x: volume -> coef, y: coef -> tmp, z: tmp -> coef) exactly like
// the reference's three loops, with its clamp-to-edge rule for the 7 border samples.
#include "oc_device.h"
#include "oc_kernels.h"

namespace ochip {

__global__ __launch_bounds__(256) void grad3d_kernel(const float* __restrict__ vol, int dz, int dy, int dx,
                                                     float* __restrict__ gx, float* __restrict__ gy,
                                                     float* __restrict__ gz) {
    const float first_factor = 1.f / 12.f;   // src/oc_gradient.cpp:21-22
    const float second_factor = 2.f / 3.f;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y, i = blockIdx.z;
    if (k >= dx) return;
    const size_t sy = (size_t)dx, sz = (size_t)dy * dx;
    const size_t g = (size_t)i * sz + (size_t)j * sy + k;
    float vx = 0.f, vy = 0.f, vz = 0.f;
    if (k >= 2 && k < dx - 2) {
        float result = 0.0f;
        result -= vol[g + 2] * first_factor;
        result += vol[g + 1] * second_factor;
        result -= vol[g - 1] * second_factor;
        result += vol[g - 2] * first_factor;
        vx = result;
    }
    if (j >= 2 && j < dy - 2) {
        float result = 0.0f;
        result -= vol[g + 2 * sy] * first_factor;
        result += vol[g + sy] * second_factor;
        result -= vol[g - sy] * second_factor;
        result += vol[g - 2 * sy] * first_factor;
        vy = result;
    }
    if (i >= 2 && i < dz - 2) {
        float result = 0.0f;
        result -= vol[g + 2 * sz] * first_factor;
        result += vol[g + sz] * second_factor;
        result -= vol[g - sz] * second_factor;
        result += vol[g - 2 * sz] * first_factor;
        vz = result;
    }
    gx[g] = vx;
    gy[g] = vy;
    gz[g] = vz;
}

// BSPLINE_PREFILTER of src/oc_cubic_bspline.h:80-90
__device__ constexpr float kPrefilter[8] = {1.732176555412860f,  -0.464135309171000f, 0.124364681271139f,
                                            -0.033323415913556f, 0.008928982383084f,  -0.002392513618779f,
                                            0.000641072092032f,  -0.000171774749350f};

// out = b0*in[p] + b1*(in[p-1] + in[p+1]) + ... + b7*(in[p-7] + in[p+7]) along `axis`
// (0 = z, 1 = y, 2 = x), indices clamped to [0, n-1] (getHigh/getLow, src/oc_cubic_bspline.cpp:21-31).
__global__ __launch_bounds__(256) void bspline3d_prefilter_axis_kernel(const float* __restrict__ in,
                                                                       float* __restrict__ out, int dz, int dy, int dx,
                                                                       int axis) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y, i = blockIdx.z;
    if (k >= dx) return;
    const size_t sy = (size_t)dx, sz = (size_t)dy * dx;
    const size_t g = (size_t)i * sz + (size_t)j * sy + k;
    const int pos = axis == 0 ? i : (axis == 1 ? j : k);
    const int n = axis == 0 ? dz : (axis == 1 ? dy : dx);
    const size_t st = axis == 0 ? sz : (axis == 1 ? sy : 1);
    const float* base = in + (g - (size_t)pos * st);
    float acc = kPrefilter[0] * base[(size_t)pos * st];
#pragma unroll
    for (int t = 1; t <= 7; t++) {
        const int lo = pos - t < 0 ? 0 : pos - t;
        const int hi = pos + t > n - 1 ? n - 1 : pos + t;
        acc = acc + kPrefilter[t] * (base[(size_t)lo * st] + base[(size_t)hi * st]);
    }
    out[g] = acc;
}

hipError_t launch_grad3d(const float* vol, int dz, int dy, int dx, float* gx, float* gy, float* gz,
                         hipStream_t stream) {
    dim3 block(256), grid((dx + 255) / 256, dy, dz);
    (void)hipGetLastError();  // drop stale errors of earlier, unrelated calls
    hipLaunchKernelGGL(grad3d_kernel, grid, block, 0, stream, vol, dz, dy, dx, gx, gy, gz);
    return hipGetLastError();
}

hipError_t launch_bspline3d_prefilter(const float* vol, int dz, int dy, int dx, float* coef, float* tmp,
                                      hipStream_t stream) {
    dim3 block(256), grid((dx + 255) / 256, dy, dz);
    (void)hipGetLastError();  // drop stale errors of earlier, unrelated calls
    hipLaunchKernelGGL(bspline3d_prefilter_axis_kernel, grid, block, 0, stream, vol, coef, dz, dy, dx, 2);
    (void)hipGetLastError();  // drop stale errors of earlier, unrelated calls
    hipLaunchKernelGGL(bspline3d_prefilter_axis_kernel, grid, block, 0, stream, (const float*)coef, tmp, dz, dy, dx, 1);
    (void)hipGetLastError();  // drop stale errors of earlier, unrelated calls
    hipLaunchKernelGGL(bspline3d_prefilter_axis_kernel, grid, block, 0, stream, (const float*)tmp, coef, dz, dy, dx, 0);
    return hipGetLastError();
}

}  // namespace ochip
