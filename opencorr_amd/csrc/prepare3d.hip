// prepare3d.hip -- volume-level precompute of ICGN3D1::prepare() on gfx950.
//
//   grad3d_kernel            Gradient3D4::getGradientX/Y/Z     (src/oc_gradient.cpp:143-231)
//   bspline3d_prefilter_axis TricubicBspline::prepare          (src/oc_cubic_bspline.cpp:214-351)
//
// HBM streaming kernels, x fastest so every access is coalesced.  What matters is that a voxel is fetched from HBM once:
// a 512^3 volume has 1 MB planes, so the +-2 planes of the z gradient and the +-7 planes / rows of the z / y prefilter
// passes do not survive in a 4 MB L2 when every thread handles one voxel (round 1: 25-32 % of the HBM rate).  The
// kernels below WALK along the filtered axis instead: a thread keeps the taps it needs in a rotating register window
// and loads one new value per output (z gradient: 5-deep window; prefilter along y or z: 15-deep), the x pass of the
// prefilter stages a row segment in LDS so that every value is fetched once.  Rows whose length is a multiple of 4 (and
// 16-byte aligned buffers) take the `*4` kernels: a thread owns 4 consecutive x positions and every access is a 16-byte
// one; anything else takes the scalar kernels of the same structure.  Arithmetic and its order are the reference's:
// three prefilter passes (x: volume -> coef, y: coef -> tmp, z: tmp -> coef) like its three loops, with its
// clamp-to-edge rule for the 7 border samples.
#include "oc_device.h"
#include "oc_kernels.h"

namespace ochip {

// One thread per (x, y) column and run of `run` z positions: the five planes z-2..z+2 of the column travel in registers.
__global__ __launch_bounds__(256) void grad3d_kernel(const float* __restrict__ vol, int dz, int dy, int dx, int run,
                                                     float* __restrict__ gx, float* __restrict__ gy,
                                                     float* __restrict__ gz) {
    const float first_factor = 1.f / 12.f;   // src/oc_gradient.cpp:21-22
    const float second_factor = 2.f / 3.f;
    const int k = blockIdx.x * 64 + (threadIdx.x & 63);
    const int j = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (k >= dx || j >= dy) return;
    const size_t sy = (size_t)dx, sz = (size_t)dy * dx;
    const int i0 = blockIdx.z * run, i1 = min(i0 + run, dz);
    const float* __restrict__ col = vol + (size_t)j * sy + k;
    auto plane = [&](int i) { return (i >= 0 && i < dz) ? col[(size_t)i * sz] : 0.f; };
    float m2 = plane(i0 - 2), m1 = plane(i0 - 1), c = plane(i0), p1 = plane(i0 + 1);
    const bool x_in = k >= 2 && k < dx - 2, y_in = j >= 2 && j < dy - 2;
#pragma unroll 4
    for (int i = i0; i < i1; i++) {
        const float p2 = plane(i + 2);
        const size_t g = (size_t)i * sz + (size_t)j * sy + k;
        float vx = 0.f, vy = 0.f, vz = 0.f;
        if (x_in) {
            float result = 0.0f;
            result -= vol[g + 2] * first_factor;
            result += vol[g + 1] * second_factor;
            result -= vol[g - 1] * second_factor;
            result += vol[g - 2] * first_factor;
            vx = result;
        }
        if (y_in) {
            float result = 0.0f;
            result -= vol[g + 2 * sy] * first_factor;
            result += vol[g + sy] * second_factor;
            result -= vol[g - sy] * second_factor;
            result += vol[g - 2 * sy] * first_factor;
            vy = result;
        }
        if (i >= 2 && i < dz - 2) {
            float result = 0.0f;
            result -= p2 * first_factor;
            result += p1 * second_factor;
            result -= m1 * second_factor;
            result += m2 * first_factor;
            vz = result;
        }
        gx[g] = vx;
        gy[g] = vy;
        gz[g] = vz;
        m2 = m1; m1 = c; c = p1; p1 = p2;
    }
}

// The same walk with four x positions per thread (dx % 4 == 0): the z window holds float4s, the x taps come from the
// centre float4 plus two 8-byte halo loads, the y taps are four 16-byte loads of neighbouring rows (L2 hits).
__global__ __launch_bounds__(256) void grad3d4_kernel(const float* __restrict__ vol, int dz, int dy, int dx, int run,
                                                      float* __restrict__ gx, float* __restrict__ gy,
                                                      float* __restrict__ gz) {
    const float first_factor = 1.f / 12.f;
    const float second_factor = 2.f / 3.f;
    const int k = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int j = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (k >= dx || j >= dy) return;
    const size_t sy = (size_t)dx, sz = (size_t)dy * dx;
    const int i0 = blockIdx.z * run, i1 = min(i0 + run, dz);
    const float* __restrict__ col = vol + (size_t)j * sy + k;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    auto plane = [&](int i) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);  // (not `cond ? *p : zero`: that selects between ADDRESSES and parks `zero` in scratch)
        if (i >= 0 && i < dz) v = *reinterpret_cast<const float4*>(col + (size_t)i * sz);
        return v;
    };
    float4 m2 = plane(i0 - 2), m1 = plane(i0 - 1), c = plane(i0), p1 = plane(i0 + 1);
    const bool y_in = j >= 2 && j < dy - 2;
    const bool has_l = k >= 2, has_r = k + 5 < dx;
#pragma unroll 2
    for (int i = i0; i < i1; i++) {
        const float4 p2 = plane(i + 2);
        const size_t g = (size_t)i * sz + (size_t)j * sy + k;
        // x: voxels k-2 .. k+5 are L.x L.y c.x c.y c.z c.w R.x R.y
        float2 L = make_float2(0.f, 0.f), R = make_float2(0.f, 0.f);
        if (has_l) L = *reinterpret_cast<const float2*>(vol + g - 2);
        if (has_r) R = *reinterpret_cast<const float2*>(vol + g + 4);
        auto tapx = [&](float q2, float q1, float r1, float r2, int kk) {
            float result = 0.0f;
            result -= q2 * first_factor;
            result += q1 * second_factor;
            result -= r1 * second_factor;
            result += r2 * first_factor;
            return (kk >= 2 && kk < dx - 2) ? result : 0.f;
        };
        const float4 vx = make_float4(tapx(c.z, c.y, L.y, L.x, k), tapx(c.w, c.z, c.x, L.y, k + 1), tapx(R.x, c.w, c.y, c.x, k + 2),
                                      tapx(R.y, R.x, c.z, c.y, k + 3));
        float4 vy = zero, vz = zero;
        if (y_in) {
            const float4 a2 = *reinterpret_cast<const float4*>(vol + g + 2 * sy), a1 = *reinterpret_cast<const float4*>(vol + g + sy);
            const float4 b1 = *reinterpret_cast<const float4*>(vol + g - sy), b2 = *reinterpret_cast<const float4*>(vol + g - 2 * sy);
            auto tap = [&](float q2, float q1, float r1, float r2) {
                float result = 0.0f;
                result -= q2 * first_factor;
                result += q1 * second_factor;
                result -= r1 * second_factor;
                result += r2 * first_factor;
                return result;
            };
            vy = make_float4(tap(a2.x, a1.x, b1.x, b2.x), tap(a2.y, a1.y, b1.y, b2.y), tap(a2.z, a1.z, b1.z, b2.z),
                             tap(a2.w, a1.w, b1.w, b2.w));
        }
        if (i >= 2 && i < dz - 2) {
            auto tap = [&](float q2, float q1, float r1, float r2) {
                float result = 0.0f;
                result -= q2 * first_factor;
                result += q1 * second_factor;
                result -= r1 * second_factor;
                result += r2 * first_factor;
                return result;
            };
            vz = make_float4(tap(p2.x, p1.x, m1.x, m2.x), tap(p2.y, p1.y, m1.y, m2.y), tap(p2.z, p1.z, m1.z, m2.z),
                             tap(p2.w, p1.w, m1.w, m2.w));
        }
        *reinterpret_cast<float4*>(gx + g) = vx;
        *reinterpret_cast<float4*>(gy + g) = vy;
        *reinterpret_cast<float4*>(gz + g) = vz;
        m2 = m1; m1 = c; c = p1; p1 = p2;
    }
}

// BSPLINE_PREFILTER of src/oc_cubic_bspline.h:80-90
__device__ constexpr float kPrefilter[8] = {1.732176555412860f,  -0.464135309171000f, 0.124364681271139f,
                                            -0.033323415913556f, 0.008928982383084f,  -0.002392513618779f,
                                            0.000641072092032f,  -0.000171774749350f};

// out = b0*in[p] + b1*(in[p-1] + in[p+1]) + ... + b7*(in[p-7] + in[p+7]) along one axis, indices clamped to [0, n-1]
// (getHigh/getLow, src/oc_cubic_bspline.cpp:21-31), the sum formed left to right like the reference's expression.
__device__ __forceinline__ float prefilter15(const float (&w)[15], int first) {
    // w is a rotating window: tap t (0..14, i.e. position p - 7 + t) sits in w[(first + t) % 15]
    float acc = kPrefilter[0] * w[(first + 7) % 15];
#pragma unroll
    for (int t = 1; t <= 7; t++) acc = acc + kPrefilter[t] * (w[(first + 7 - t) % 15] + w[(first + 7 + t) % 15]);
    return acc;
}

// Prefilter along z (AXIS 0) or y (AXIS 1): one thread per column and run of `run` (a multiple of 15) positions; the 15
// taps rotate through registers (the loop is unrolled by 15 so that every window index is a compile-time constant),
// one load per output.
template <int AXIS>
__global__ __launch_bounds__(256) void bspline3d_prefilter_walk_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                                       int dz, int dy, int dx, int run) {
    const int k = blockIdx.x * 64 + (threadIdx.x & 63);
    const int q = blockIdx.y * 4 + (threadIdx.x >> 6);  // the axis that is neither x nor the filtered one
    const int n = AXIS == 0 ? dz : dy, nq = AXIS == 0 ? dy : dz;
    if (k >= dx || q >= nq) return;
    const size_t st = AXIS == 0 ? (size_t)dy * dx : (size_t)dx;
    const size_t origin = AXIS == 0 ? (size_t)q * dx + k : (size_t)q * dy * dx + k;
    const float* __restrict__ col = in + origin;
    float* __restrict__ ocol = out + origin;
    const int p0 = blockIdx.z * run, p1 = min(p0 + run, n);
    auto at = [&](int p) { return col[(size_t)min(max(p, 0), n - 1) * st]; };
    float w[15];
#pragma unroll
    for (int t = 0; t < 14; t++) w[t] = at(p0 - 7 + t);
#pragma unroll 1
    for (int pb = p0; pb < p1; pb += 15) {
#pragma unroll
        for (int r = 0; r < 15; r++) {
            const int p = pb + r;
            if (p < p1) {
                w[(r + 14) % 15] = at(p + 7);
                ocol[(size_t)p * st] = prefilter15(w, r);
            }
        }
    }
}

// Prefilter along x: 8 consecutive outputs per thread from the 24 values x0-8 .. x0+15 (six 16-byte loads when the row is
// 16-byte aligned and inside the volume; the clamped form otherwise).
__global__ __launch_bounds__(256) void bspline3d_prefilter_x_kernel(const float* __restrict__ in, float* __restrict__ out, int dz,
                                                                    int dy, int dx) {
    const int x0 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 8;
    const int j = blockIdx.y * 4 + (threadIdx.x >> 6), i = blockIdx.z;
    if (x0 >= dx || j >= dy) return;
    const float* __restrict__ row = in + ((size_t)i * dy + j) * dx;
    float* __restrict__ orow = out + ((size_t)i * dy + j) * dx;
    float v[24];  // v[t] = row[clamp(x0 - 8 + t)]
    // 16-byte loads only from 16-byte-aligned addresses: this kernel is also the fallback for volumes whose base pointer is
    // merely 4-byte aligned (a view into a larger device allocation), where dx % 4 == 0 says nothing about the rows
    if (x0 >= 8 && x0 + 16 <= dx && (dx & 3) == 0 && (reinterpret_cast<uintptr_t>(row) & 15) == 0) {
#pragma unroll
        for (int t = 0; t < 6; t++) {
            const float4 q = *reinterpret_cast<const float4*>(row + x0 - 8 + 4 * t);
            v[4 * t] = q.x; v[4 * t + 1] = q.y; v[4 * t + 2] = q.z; v[4 * t + 3] = q.w;
        }
    } else {
#pragma unroll
        for (int t = 0; t < 24; t++) v[t] = row[min(max(x0 - 8 + t, 0), dx - 1)];
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
        if (x0 + u < dx) {
            float acc = kPrefilter[0] * v[u + 8];
#pragma unroll
            for (int t = 1; t <= 7; t++) acc = acc + kPrefilter[t] * (v[u + 8 - t] + v[u + 8 + t]);
            orow[x0 + u] = acc;
        }
    }
}

// The walk with four x positions per thread (dx % 4 == 0): the window holds 15 float4s.
__device__ __forceinline__ float4 prefilter15x4(const float4 (&w)[15], int first) {
    auto one = [&](auto pick) {
        float acc = kPrefilter[0] * pick(w[(first + 7) % 15]);
#pragma unroll
        for (int t = 1; t <= 7; t++) acc = acc + kPrefilter[t] * (pick(w[(first + 7 - t) % 15]) + pick(w[(first + 7 + t) % 15]));
        return acc;
    };
    return make_float4(one([](const float4& q) { return q.x; }), one([](const float4& q) { return q.y; }),
                       one([](const float4& q) { return q.z; }), one([](const float4& q) { return q.w; }));
}

template <int AXIS>
__global__ __launch_bounds__(256) void bspline3d_prefilter_walk4_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                                        int dz, int dy, int dx, int run) {
    const int k = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int q = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int n = AXIS == 0 ? dz : dy, nq = AXIS == 0 ? dy : dz;
    if (k >= dx || q >= nq) return;
    const size_t st = AXIS == 0 ? (size_t)dy * dx : (size_t)dx;
    const size_t origin = AXIS == 0 ? (size_t)q * dx + k : (size_t)q * dy * dx + k;
    const float* __restrict__ col = in + origin;
    float* __restrict__ ocol = out + origin;
    const int p0 = blockIdx.z * run, p1 = min(p0 + run, n);
    auto at = [&](int p) { return *reinterpret_cast<const float4*>(col + (size_t)min(max(p, 0), n - 1) * st); };
    float4 w[15];
#pragma unroll
    for (int t = 0; t < 14; t++) w[t] = at(p0 - 7 + t);
#pragma unroll 1
    for (int pb = p0; pb < p1; pb += 15) {
#pragma unroll
        for (int r = 0; r < 15; r++) {
            const int p = pb + r;
            if (p < p1) {
                w[(r + 14) % 15] = at(p + 7);
                *reinterpret_cast<float4*>(ocol + (size_t)p * st) = prefilter15x4(w, r);
            }
        }
    }
}

// Prefilter along x for dx % 4 == 0: every wave stages a 256-value row segment plus 8 halo values per side in LDS (one
// coalesced 16-byte load per lane, 16 clamped scalar loads for the halo) and every lane forms its 4 outputs from five
// 16-byte LDS reads.
__global__ __launch_bounds__(256) void bspline3d_prefilter_x4_kernel(const float* __restrict__ in, float* __restrict__ out, int dz,
                                                                     int dy, int dx) {
    __shared__ __attribute__((aligned(16))) float seg[4][272];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int s0 = blockIdx.x * 256, x0 = s0 + lane * 4;
    const int j = min((int)(blockIdx.y * 4 + wv), dy - 1), i = blockIdx.z;  // surplus waves redo the last row (no divergent barrier)
    const bool live = (int)(blockIdx.y * 4 + wv) < dy;
    const float* __restrict__ row = in + ((size_t)i * dy + j) * dx;
    float* __restrict__ orow = out + ((size_t)i * dy + j) * dx;
    float* __restrict__ s = seg[wv];
    const float last = row[dx - 1];
    *reinterpret_cast<float4*>(s + 8 + lane * 4) = x0 < dx ? *reinterpret_cast<const float4*>(row + x0) : make_float4(last, last, last, last);
    if (lane < 16) {
        const int p = lane < 8 ? s0 - 8 + lane : s0 + 256 + (lane - 8);
        s[lane < 8 ? lane : 256 + lane] = row[min(max(p, 0), dx - 1)];
    }
    __syncthreads();
    if (!live || x0 >= dx) return;
    float v[20];  // v[t] = segment value at x0 - 8 + t
#pragma unroll
    for (int t = 0; t < 5; t++) {
        const float4 q = *reinterpret_cast<const float4*>(s + lane * 4 + 4 * t);
        v[4 * t] = q.x; v[4 * t + 1] = q.y; v[4 * t + 2] = q.z; v[4 * t + 3] = q.w;
    }
    float o[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        float acc = kPrefilter[0] * v[u + 8];
#pragma unroll
        for (int t = 1; t <= 7; t++) acc = acc + kPrefilter[t] * (v[u + 8 - t] + v[u + 8 + t]);
        o[u] = acc;
    }
    *reinterpret_cast<float4*>(orow + x0) = make_float4(o[0], o[1], o[2], o[3]);
}

static bool rows_of_float4(int dx, const void* a, const void* b, const void* c = nullptr, const void* d = nullptr) {
    const uintptr_t bits = (uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d;
    return (dx & 3) == 0 && (bits & 15) == 0;
}

hipError_t launch_grad3d(const float* vol, int dz, int dy, int dx, float* gx, float* gy, float* gz,
                         hipStream_t stream) {
    const int run = 64;  // z positions per thread: 4 halo planes per 64 outputs
    (void)hipGetLastError();  // drop stale errors of earlier, unrelated calls
    if (rows_of_float4(dx, vol, gx, gy, gz))
        hipLaunchKernelGGL(grad3d4_kernel, dim3((dx + 255) / 256, (dy + 3) / 4, (dz + run - 1) / run), dim3(256), 0, stream, vol, dz, dy,
                           dx, run, gx, gy, gz);
    else
        hipLaunchKernelGGL(grad3d_kernel, dim3((dx + 63) / 64, (dy + 3) / 4, (dz + run - 1) / run), dim3(256), 0, stream, vol, dz, dy,
                           dx, run, gx, gy, gz);
    return hipGetLastError();
}

hipError_t launch_bspline3d_prefilter(const float* vol, int dz, int dy, int dx, float* coef, float* tmp,
                                      hipStream_t stream) {
    const int run = 60;  // positions per thread along the walked axis (a multiple of 15): 14 halo loads per 60 outputs
    (void)hipGetLastError();  // drop stale errors of earlier, unrelated calls
    if (rows_of_float4(dx, vol, coef, tmp)) {
        hipLaunchKernelGGL(bspline3d_prefilter_x4_kernel, dim3((dx + 255) / 256, (dy + 3) / 4, dz), dim3(256), 0, stream, vol, coef, dz, dy,
                           dx);
        (void)hipGetLastError();
        hipLaunchKernelGGL(bspline3d_prefilter_walk4_kernel<1>, dim3((dx + 255) / 256, (dz + 3) / 4, (dy + run - 1) / run), dim3(256), 0,
                           stream, (const float*)coef, tmp, dz, dy, dx, run);
        (void)hipGetLastError();
        hipLaunchKernelGGL(bspline3d_prefilter_walk4_kernel<0>, dim3((dx + 255) / 256, (dy + 3) / 4, (dz + run - 1) / run), dim3(256), 0,
                           stream, (const float*)tmp, coef, dz, dy, dx, run);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(bspline3d_prefilter_x_kernel, dim3((dx + 511) / 512, (dy + 3) / 4, dz), dim3(256), 0, stream, vol, coef, dz, dy, dx);
    (void)hipGetLastError();
    hipLaunchKernelGGL(bspline3d_prefilter_walk_kernel<1>, dim3((dx + 63) / 64, (dz + 3) / 4, (dy + run - 1) / run), dim3(256), 0,
                       stream, (const float*)coef, tmp, dz, dy, dx, run);
    (void)hipGetLastError();
    hipLaunchKernelGGL(bspline3d_prefilter_walk_kernel<0>, dim3((dx + 63) / 64, (dy + 3) / 4, (dz + run - 1) / run), dim3(256), 0,
                       stream, (const float*)tmp, coef, dz, dy, dx, run);
    return hipGetLastError();
}

}  // namespace ochip
