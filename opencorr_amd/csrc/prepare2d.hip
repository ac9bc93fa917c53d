// prepare2d.hip -- image-level precompute of ICGN2D1/2D2::prepare() on gfx950.
//
//   grad2d_kernel        Gradient2D4::getGradientX/Y   (src/oc_gradient.cpp:37-79)
//   bspline2d_lut_kernel BicubicBspline::prepare       (src/oc_cubic_bspline.cpp:84-132)
//   colmajor_to_rowmajor Eigen::MatrixXf (column-major, src/oc_image.h:37) -> x-fastest
//
// All three are HBM streaming kernels: one thread per pixel, 4 B read (neighbours
// come from L1/L2), 8 B (gradients) or 64 B (LUT: 4 planes x 16 B) written per pixel.
#include "oc_device.h"
#include "oc_kernels.h"

namespace ochip {

// 4th-order central difference with the reference's operation order:
// result = 0; result -= f[+2]*(1/12); result += f[+1]*(2/3); result -= f[-1]*(2/3); result += f[-2]*(1/12)
// (src/oc_gradient.cpp:21-22, 50-54).  Two-pixel zero border.
__global__ __launch_bounds__(256) void grad2d_kernel(const float* __restrict__ img, int height, int width,
                                                     float* __restrict__ gx, float* __restrict__ gy) {
    const float first_factor = 1.f / 12.f;
    const float second_factor = 2.f / 3.f;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y;
    if (c >= width) return;
    const size_t g = (size_t)r * width + c;
    float vx = 0.f, vy = 0.f;
    if (c >= 2 && c < width - 2) {
        float result = 0.0f;
        result -= img[g + 2] * first_factor;
        result += img[g + 1] * second_factor;
        result -= img[g - 1] * second_factor;
        result += img[g - 2] * first_factor;
        vx = result;
    }
    if (r >= 2 && r < height - 2) {
        float result = 0.0f;
        result -= img[g + 2 * (size_t)width] * first_factor;
        result += img[g + (size_t)width] * second_factor;
        result -= img[g - (size_t)width] * second_factor;
        result += img[g - 2 * (size_t)width] * first_factor;
        vy = result;
    }
    gx[g] = vx;
    gy[g] = vy;
}

// BC = B * C of src/oc_cubic_bspline.h:52-58
__device__ constexpr float kBC[4][4] = {
    {-144.0f / 336.0f, 384.0f / 336.0f, -384.0f / 336.0f, 144.0f / 336.0f},
    {342.0f / 336.0f, -702.0f / 336.0f, 450.0f / 336.0f, -90.0f / 336.0f},
    {-198.0f / 336.0f, -18.0f / 336.0f, 270.0f / 336.0f, -54.0f / 336.0f},
    {0.0f, 1.0f, 0.0f, 0.0f}};

// Per interior pixel (1 <= r < H-2, 1 <= c < W-2): P = BC * Q * BC^T over the 4x4
// neighbourhood accumulated in the reference's k,l,m,n loop order
// (acc += BC[l][m] * BC[k][n] * q[n][m], src/oc_cubic_bspline.cpp:108-120), stored
// flipped coef[k][l] = P[3-k][3-l] (:123-129).  Border entries are zero (calloc in
// the reference, src/oc_array.h:92).  The table is stored PLANAR (dic2d_device.h): plane k holds the float4
// coef[k][0..3] of every pixel, row-major -- a wave's store is 1 KiB of consecutive bytes per plane, and the
// solvers' gathers touch a quarter of the cache lines an interleaved 64-byte entry cost them.
__global__ __launch_bounds__(256) void bspline2d_lut_kernel(const float* __restrict__ img, int height, int width,
                                                            float* __restrict__ lut) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y;
    if (c >= width) return;
    const size_t plane = (size_t)height * width;  // float4 elements per plane
    float4* out = reinterpret_cast<float4*>(lut) + ((size_t)r * width + c);
    if (r < 1 || r >= height - 2 || c < 1 || c >= width - 2) {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        out[0] = z; out[plane] = z; out[2 * plane] = z; out[3 * plane] = z;
        return;
    }
    float q[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) q[i][j] = img[(size_t)(r - 1 + i) * width + (c - 1 + j)];
    float pm[4][4];
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
        for (int l = 0; l < 4; l++) {
            float acc = 0.f;
#pragma unroll
            for (int m = 0; m < 4; m++)
#pragma unroll
                for (int n = 0; n < 4; n++) acc += kBC[l][m] * kBC[k][n] * q[n][m];
            pm[k][l] = acc;
        }
#pragma unroll
    for (int k = 0; k < 4; k++) out[k * plane] = make_float4(pm[3 - k][3], pm[3 - k][2], pm[3 - k][1], pm[3 - k][0]);
}

// dst[r*width + c] = src[c*height + r]; LDS-tiled so both sides stay coalesced.
__global__ __launch_bounds__(256) void colmajor_to_rowmajor_kernel(const float* __restrict__ src, int height,
                                                                   int width, float* __restrict__ dst) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    for (int j = ty; j < 32; j += 8) {
        int c = c0 + j, r = r0 + tx;  // src is contiguous in r
        if (c < width && r < height) tile[j][tx] = src[(size_t)c * height + r];
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        int r = r0 + j, c = c0 + tx;  // dst is contiguous in c
        if (r < height && c < width) dst[(size_t)r * width + c] = tile[tx][j];
    }
}

// The same gradients as a streaming kernel: a thread owns FOUR consecutive pixels of a row (16-byte accesses) and walks
// down kGradRows rows with the centre values of rows r-2 .. r+2 in a rotating register window, so every pixel is fetched
// once for the y gradient (the scalar kernel above re-read it four times through L1) and the x gradient of the four
// pixels needs, beside the window's centre values, one aligned pair to the left and one to the right.  Same expression
// per pixel, same bits.
// Needs width % 4 == 0 and 16-byte aligned rows; everything else takes the scalar kernel.  The 3D twin (prepare3d.hip)
// reaches 3.3 TB/s this way; here: 201 MB (4096^2: 4 B in, 8 B out per pixel) in 78.8 us -> see DESIGN.md section 4.
constexpr int kGradRows = 16;
__global__ __launch_bounds__(256) void grad2d4_kernel(const float* __restrict__ img, int height, int width,
                                                      float* __restrict__ gx, float* __restrict__ gy) {
    const float first_factor = 1.f / 12.f;
    const float second_factor = 2.f / 3.f;
    const int c = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int r0 = blockIdx.y * kGradRows;
    if (c >= width) return;
    const int r1 = min(r0 + kGradRows, height);
    auto row4 = [&](int r) -> float4 {
        if (r < 0 || r >= height) return make_float4(0.f, 0.f, 0.f, 0.f);  // never used: the border rows below write zeros
        return *reinterpret_cast<const float4*>(img + (size_t)r * width + c);
    };
    float4 m2 = row4(r0 - 2), m1 = row4(r0 - 1), c0 = row4(r0), p1 = row4(r0 + 1);
    auto comp = [&](float a2, float a1, float b1, float b2) {
        float result = 0.0f;
        result -= a2 * first_factor;
        result += a1 * second_factor;
        result -= b1 * second_factor;
        result += b2 * first_factor;
        return result;
    };
#pragma unroll 4
    for (int r = r0; r < r1; r++) {
        const float4 p2 = row4(r + 2);
        const size_t g = (size_t)r * width + c;
        // x: f[k] = img[r][c - 2 + k], k = 0 .. 7: the four centre values are in the window already; the two values to the
        // left and the two to the right are 8-byte aligned pairs (c is a multiple of 4).  Where a pair would leave the row
        // (first / last thread of a row) it is not needed: the pixels that would use it lie in the two-pixel zero border.
        // (loaded unconditionally from a clamped address -- no branch, no wait in the middle of the row's loads)
        const float2 lo = *reinterpret_cast<const float2*>(img + g - (c >= 4 ? 2 : 0));
        const float2 hi = *reinterpret_cast<const float2*>(img + g + (c + 8 <= width ? 4 : 0));
        const float f[8] = {lo.x, lo.y, c0.x, c0.y, c0.z, c0.w, hi.x, hi.y};
        float vx[4], vy[4];
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int cc = c + t;
            vx[t] = (cc >= 2 && cc < width - 2) ? comp(f[t + 4], f[t + 3], f[t + 1], f[t]) : 0.f;
        }
        const bool ry = r >= 2 && r < height - 2;
        vy[0] = ry ? comp(p2.x, p1.x, m1.x, m2.x) : 0.f;
        vy[1] = ry ? comp(p2.y, p1.y, m1.y, m2.y) : 0.f;
        vy[2] = ry ? comp(p2.z, p1.z, m1.z, m2.z) : 0.f;
        vy[3] = ry ? comp(p2.w, p1.w, m1.w, m2.w) : 0.f;
        *reinterpret_cast<float4*>(gx + g) = make_float4(vx[0], vx[1], vx[2], vx[3]);
        *reinterpret_cast<float4*>(gy + g) = make_float4(vy[0], vy[1], vy[2], vy[3]);
        m2 = m1; m1 = c0; c0 = p1; p1 = p2;
    }
}

hipError_t launch_grad2d(const float* img, int height, int width, float* gx, float* gy, hipStream_t stream) {
    (void)hipGetLastError();  // drop stale errors of earlier, unrelated calls
    const bool vec = (width & 3) == 0 && ((reinterpret_cast<uintptr_t>(img) | reinterpret_cast<uintptr_t>(gx) | reinterpret_cast<uintptr_t>(gy)) & 15) == 0;
    if (vec) {
        dim3 block(256), grid((width / 4 + 255) / 256, (height + kGradRows - 1) / kGradRows);
        hipLaunchKernelGGL(grad2d4_kernel, grid, block, 0, stream, img, height, width, gx, gy);
    } else {
        dim3 block(256), grid((width + 255) / 256, height);
        hipLaunchKernelGGL(grad2d_kernel, grid, block, 0, stream, img, height, width, gx, gy);
    }
    return hipGetLastError();
}

hipError_t launch_bspline2d_lut(const float* img, int height, int width, float* lut, hipStream_t stream) {
    dim3 block(256), grid((width + 255) / 256, height);
    (void)hipGetLastError();  // drop stale errors of earlier, unrelated calls
    hipLaunchKernelGGL(bspline2d_lut_kernel, grid, block, 0, stream, img, height, width, lut);
    return hipGetLastError();
}

hipError_t launch_colmajor_to_rowmajor(const float* src, int height, int width, float* dst, hipStream_t stream) {
    dim3 block(256), grid((width + 31) / 32, (height + 31) / 32);
    (void)hipGetLastError();  // drop stale errors of earlier, unrelated calls
    hipLaunchKernelGGL(colmajor_to_rowmajor_kernel, grid, block, 0, stream, src, height, width, dst);
    return hipGetLastError();
}

}  // namespace ochip
