// valu_mix_ubench.hip -- what the VALU stream of the ICGN2D1 interpolation sweep costs per instruction, ingredient by ingredient (a
// plain stream of v_mul_f32 / v_add_f32 on register operands runs at 2.2 cycles: valu_chain_ubench.hip).  Starts from the VALU-only build of
// coissue_ubench.hip (the ICGN2D1 interpolation sweep without its gathers: coefficients stay in registers) and strips one
// ingredient at a time (STRIP bit mask):
//   1   no "negative value seen" tracking (v_cmp + s_or per sample)
//   2   no LDS store of the interpolated value
//   4   local coordinates from registers instead of the LDS coordinate table (no ds_read)
//   8   no floor / fraction / address arithmetic (dx, dy straight from the warped coordinates)
//   16  warp coefficients per lane (VGPR operands) instead of wave-uniform (SGPR operands)
//   32  no warp at all (ax, ay from registers)
//   64  every wave-uniform operand of the loop (warp coefficients, POI centre, image size) laundered into VGPRs (in_vgpr)
// Output per variant: time, and cycles per polynomial per SIMD; the VALU instruction counts of each loop come from the ISA
// (tools/ubench/loop_isa.py on the --save-temps assembly), so cycles per instruction are computed off-line.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -I opencorr_amd/csrc tools/ubench/valu_mix_ubench.hip -o valu_mix_ubench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "dic2d_device.h"

using namespace ochip;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int W = 4096, H = 4096, RX = 16, SUB = 33, N = SUB * SUB, NT = (N + 63) / 64, NF = N / 64, ITERS = 3, G = 2;

template <int STRIP, int WPB>
__global__ __launch_bounds__(64 * WPB) void sweep(const float* __restrict__ lut, float* __restrict__ out, int npoi, int grid_side) {
    __shared__ f2 tab_xy[NT * 64];
    __shared__ float ts[WPB][NT * 64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int s = threadIdx.x; s < NT * 64; s += 64 * WPB) {
        const int r = s / SUB, c = s - r * SUB;
        tab_xy[s] = mk2((float)(c - RX), (float)(r - RX));
    }
    __syncthreads();
    const int poi = blockIdx.x * WPB + wave;
    if (poi >= npoi) return;
    const float px = 24.f + (float)(poi % grid_side) * 8.1f, py = 24.f + (float)(poi / grid_side) * 8.1f;
    float Wm[6] = {1.001f, 5e-4f, 2.3f + 1e-3f * (px - 2048.f), -5e-4f, 1.002f, -1.7f + 2e-3f * (py - 2048.f)};
    const LutPlanes4 r_lut(lut, H, W);
    float acc = 0.f;
    bool negative = false;
    LutFetch keep[G];
#pragma unroll
    for (int g = 0; g < G; g++) r_lut.load(keep[g], (unsigned)(lane + 64 * g) * 16u);   // once: the coefficients stay in registers
    unsigned sink = 0;
    float* l_ts = &ts[wave][lane];
    float rx = (float)(lane & 31) - 16.f, ry = (float)(lane >> 5) - 16.f;
#pragma nounroll
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 6; i++) {
            Wm[i] = Wm[i] + (float)it * 1e-6f;
            if (!(STRIP & 16)) Wm[i] = uni(Wm[i]);
            else Wm[i] = Wm[i] + (float)lane * 1e-9f;
        }
        float Wv[6];
#pragma unroll
        for (int i = 0; i < 6; i++) Wv[i] = (STRIP & 64) ? in_vgpr(Wm[i]) : Wm[i];
        const float pxv = (STRIP & 64) ? in_vgpr(px) : px, pyv = (STRIP & 64) ? in_vgpr(py) : py;
        const int Hv = (STRIP & 64) ? in_vgpr(H) : H, Wdv = (STRIP & 64) ? in_vgpr(W) : W;
        int t0 = 0;
#pragma nounroll
        for (int q = 0; q < NF / G; q++, t0 += G) {
            LutFetch f[G];
#pragma unroll
            for (int g = 0; g < G; g++) {
                float xl, yl;
                if (STRIP & 4) {
                    xl = rx;
                    yl = ry;
                    rx = rx + 1e-3f;
                    ry = ry + 2e-3f;
                } else {
                    const f2 lxy = tab_xy[(t0 + g) * 64 + lane];
                    xl = lxy.x;
                    yl = lxy.y;
                }
                float ax, ay;
                if (STRIP & 32) {
                    ax = xl;
                    ay = yl;
                } else {
                    const float wx = (Wv[0] * xl + Wv[1] * yl) + Wv[2];
                    const float wy = (Wv[3] * xl + Wv[4] * yl) + Wv[5];
                    ax = pxv + wx;
                    ay = pyv + wy;
                }
                bool outside = false;
                if (STRIP & 8) {
                    f[g].dx = ax * 1e-3f;
                    f[g].dy = ay * 1e-3f;
                } else {
                    sink ^= lut_locate<false>(f[g], Hv, Wdv, ax, ay, outside);   // the address is formed (and kept alive by one xor)
                }
                keep[g].dx = f[g].dx;
                keep[g].dy = f[g].dy;
                f[g] = keep[g];
                if (!(STRIP & 1)) negative = negative || outside;
            }
#pragma unroll
            for (int g = 0; g < G; g++) {
                const float v = lut_poly(f[g]);
                if (!(STRIP & 1)) negative = negative || v < 0.f;
                acc = acc + v;
                if (!(STRIP & 2)) l_ts[(t0 + g) * 64] = v;
            }
        }
    }
    out[(size_t)poi * 64 + lane] = negative ? -1.f : acc + l_ts[(lane & 7) * 64] + (float)sink;
}

__global__ void fill(float* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = (float)((i * 2654435761u >> 20) & 255) * (1.f / 64.f);
}

template <int STRIP, int WPB>
void run(const float* lut, float* out, int npoi, int side) {
    const int grid = (npoi + WPB - 1) / WPB;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL((sweep<STRIP, WPB>), dim3(grid), dim3(64 * WPB), 0, 0, lut, out, npoi, side);
    CHECK(hipDeviceSynchronize());
    double best = 1e30;
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(a));
        for (int i = 0; i < 3; i++) hipLaunchKernelGGL((sweep<STRIP, WPB>), dim3(grid), dim3(64 * WPB), 0, 0, lut, out, npoi, side);
        CHECK(hipEventRecord(b));
        CHECK(hipDeviceSynchronize());
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, a, b));
        best = ms / 3 < best ? ms / 3 : best;
    }
    const double polys_per_simd = (double)npoi * ITERS * (NF / G) * G / 1024.0;
    printf("{\"strip\": %d, \"waves_per_workgroup\": %d, \"ms\": %.4f, \"cycles_per_polynomial_per_simd\": %.1f}\n", STRIP, WPB, best,
           best * 1e-3 * 2.4e9 / polys_per_simd);
}

int main() {
    const int side = 500, npoi = side * side;
    const size_t lut_floats = (size_t)W * H * 16;
    float *lut, *out;
    CHECK(hipMalloc(&lut, lut_floats * 4));
    CHECK(hipMalloc(&out, (size_t)npoi * 64 * 4));
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, lut, lut_floats);
    CHECK(hipDeviceSynchronize());
    run<0, 8>(lut, out, npoi, side);
    run<0, 4>(lut, out, npoi, side);
    run<1, 4>(lut, out, npoi, side);
    run<2, 4>(lut, out, npoi, side);
    run<3, 4>(lut, out, npoi, side);
    run<7, 4>(lut, out, npoi, side);
    run<15, 4>(lut, out, npoi, side);
    run<23, 4>(lut, out, npoi, side);
    run<31, 4>(lut, out, npoi, side);
    run<47, 4>(lut, out, npoi, side);
    run<8, 4>(lut, out, npoi, side);
    run<16, 4>(lut, out, npoi, side);
    run<4, 4>(lut, out, npoi, side);
    run<64, 4>(lut, out, npoi, side);
    run<64, 8>(lut, out, npoi, side);
    run<67, 4>(lut, out, npoi, side);
    return 0;
}
