// coissue_ubench.hip -- ONE combined ceiling for the interpolation sweep of icgn2d_kernel<6> (the metric kernel).
//
// The sweep issues, per sample and iteration, (i) a 64-byte gather from the planar bicubic table (4 x buffer_load_b128) and
// (ii) ~70 VALU instructions: the warp (10 separately rounded operations), floor / fraction / address (8), the explicit
// 16-term polynomial of src/oc_cubic_bspline.cpp:159-177 (28 multiplies + 15 dependent adds), the running sum and one LDS
// store.  Round 3 measured the two sides separately (compute-free lockstep gather: 1.86 ms for config B's samples; the VALU
// mix at its measured issue costs: 1.1 - 1.7 ms) and the kernel's sweeps at neither.  This benchmark runs BOTH, exactly as
// the kernel issues them -- the kernel's own device functions (dic2d_device.h: lut_fetch, lut_poly, LutPlanes4), 8-wave
// workgroups re-aligned every two pass groups, G = 2 gathers back to back, the per-workgroup coordinate table in LDS, three
// workgroups per CU (6 waves per SIMD, 80 VGPRs) -- and NOTHING ELSE: no set-up passes, no reductions, no solve, no numerator
// pass.  Three builds of one loop:
//   A  gathers + full VALU mix          (what perfect overlap of THIS instruction stream reaches on the hardware)
//   B  gathers, polynomial replaced by 15 adds  (the gather side alone, in this loop structure)
//   C  VALU mix, gathers issued in the first iteration only (coefficients stay in registers afterwards: the VALU side alone)
// max(B, C) is the bound no schedule of this stream can beat, B + C what zero overlap would cost, A what the hardware makes
// of it.  Samples = 250 000 POIs x 18 passes x 64 lanes x ITERS iterations (ITERS = 3; config B's mean is 3.10: the times
// scale by 3.10 / 3).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -I opencorr_amd/csrc tools/ubench/coissue_ubench.hip -o coissue_ubench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "dic2d_device.h"

using namespace ochip;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int W = 4096, H = 4096, RX = 16, SUB = 33, N = SUB * SUB, NT = (N + 63) / 64, NF = N / 64, ITERS = 3, G = 2, WPB = 8;

// MODE 0 = A (both), 1 = B (gathers only), 2 = C (VALU only); 3 / 4 = A / C with the polynomial's products as packed pairs
// (lut_poly_pk: same bits, 10 instructions fewer per sample)
template <int MODE, int LOCK>
__global__ __launch_bounds__(64 * WPB, 6) void sweep(const float* __restrict__ lut, float* __restrict__ out, int npoi, int grid_side) {
    __shared__ f2 tab_xy[NT * 64];
    __shared__ float ts[WPB][NT * 64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int s = threadIdx.x; s < NT * 64; s += 64 * WPB) {
        const int r = s / SUB, c = s - r * SUB;
        tab_xy[s] = mk2((float)(c - RX), (float)(r - RX));
    }
    __syncthreads();
    const int chunk = (npoi / WPB + 7) / 8;
    const int grp = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    const int poi = grp * WPB + wave;
    if (poi >= npoi) return;
    // POI centre and a displacement field like SURVEY 8(d)'s: u = 2.3 + 1e-3 x' + 5e-4 y', v = -1.7 - 5e-4 x' + 2e-3 y'
    const float px = 24.f + (float)(poi % grid_side) * 8.1f, py = 24.f + (float)(poi / grid_side) * 8.1f;
    float Wm[6] = {1.001f, 5e-4f, 2.3f + 1e-3f * (px - 2048.f), -5e-4f, 1.002f, -1.7f + 2e-3f * (py - 2048.f)};
    const LutPlanes4 r_lut(lut, H, W);
    float acc = 0.f;
    bool negative = false;
    LutFetch keep[G] = {};
    float* l_ts = &ts[wave][lane];
#pragma nounroll
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 6; i++) Wm[i] = uni(Wm[i] + (float)it * 1e-6f);  // wave-uniform, changes per iteration like W <- W dW^-1
        int t0 = 0;
#pragma nounroll
        for (int q = 0; q < NF / G; q++, t0 += G) {
            if (LOCK > 0 && q % LOCK == 0) __builtin_amdgcn_s_barrier();
            LutFetch f[G];
#pragma unroll
            for (int g = 0; g < G; g++) {
                const f2 lxy = tab_xy[(t0 + g) * 64 + lane];
                const float xl = lxy.x, yl = lxy.y;
                const float wx = (Wm[0] * xl + Wm[1] * yl) + Wm[2];
                const float wy = (Wm[3] * xl + Wm[4] * yl) + Wm[5];
                const float ax = px + wx, ay = py + wy;
                bool outside;
                if (MODE == 2 || MODE == 4) {
                    const unsigned off = lut_locate<false>(f[g], H, W, ax, ay, outside);
                    if (it == 0) r_lut.load(keep[g], off);
                    keep[g].dx = f[g].dx;
                    keep[g].dy = f[g].dy;
                    f[g] = keep[g];
                } else {
                    lut_fetch<false>(f[g], r_lut, H, W, ax, ay, outside);
                }
                negative = negative || outside;
            }
#pragma unroll
            for (int g = 0; g < G; g++) {
                float v;
                if (MODE == 1)  // every fetched float is used (or the loads would be narrowed): 15 adds instead of 43 operations
                    v = (((f[g].c0.x + f[g].c0.y) + (f[g].c0.z + f[g].c0.w)) + ((f[g].c1.x + f[g].c1.y) + (f[g].c1.z + f[g].c1.w))) +
                        (((f[g].c2.x + f[g].c2.y) + (f[g].c2.z + f[g].c2.w)) + ((f[g].c3.x + f[g].c3.y) + (f[g].c3.z + f[g].c3.w)));
                else if (MODE >= 3) v = lut_poly_pk(f[g]);
                else v = lut_poly(f[g]);
                negative = negative || v < 0.f;
                acc = acc + v;
                l_ts[(t0 + g) * 64] = v;
            }
        }
    }
    out[(size_t)poi * 64 + lane] = negative ? -1.f : acc + l_ts[(lane & 7) * 64];
}

__global__ void fill(float* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = (float)((i * 2654435761u >> 20) & 255) * (1.f / 64.f);
}

template <int MODE, int LOCK>
double run(const float* lut, float* out, int npoi, int side) {
    const int grid = ((npoi / WPB + 7) / 8) * 8;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL((sweep<MODE, LOCK>), dim3(grid), dim3(64 * WPB), 0, 0, lut, out, npoi, side);
    CHECK(hipDeviceSynchronize());
    double best = 1e30;
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(a));
        for (int i = 0; i < 3; i++) hipLaunchKernelGGL((sweep<MODE, LOCK>), dim3(grid), dim3(64 * WPB), 0, 0, lut, out, npoi, side);
        CHECK(hipEventRecord(b));
        CHECK(hipDeviceSynchronize());
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, a, b));
        best = ms / 3 < best ? ms / 3 : best;
    }
    return best;
}

int main() {
    const int side = 500, npoi = side * side;
    const size_t lut_floats = (size_t)W * H * 16;
    float *lut, *out;
    CHECK(hipMalloc(&lut, lut_floats * 4));
    CHECK(hipMalloc(&out, (size_t)npoi * 64 * 4));
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, lut, lut_floats);
    CHECK(hipDeviceSynchronize());
    const double samples = (double)npoi * ITERS * (NF / G) * G * 64, bytes = samples * 64;
    const double a2 = run<0, 2>(lut, out, npoi, side), b2 = run<1, 2>(lut, out, npoi, side), c2 = run<2, 2>(lut, out, npoi, side);
    const double a0 = run<0, 0>(lut, out, npoi, side), b0 = run<1, 0>(lut, out, npoi, side), c0 = run<2, 0>(lut, out, npoi, side);
    const double p2 = run<3, 2>(lut, out, npoi, side), q2 = run<4, 2>(lut, out, npoi, side);
    printf("{\"iters\": %d, \"passes\": %d, \"samples\": %.0f, \"bytes\": %.0f,\n", ITERS, (NF / G) * G, samples, bytes);
    printf(" \"lockstep2\": {\"both_ms\": %.4f, \"gather_only_ms\": %.4f, \"valu_only_ms\": %.4f, \"both_TBps\": %.2f},\n", a2, b2, c2, bytes / a2 / 1e9);
    printf(" \"lockstep2_packed_products\": {\"both_ms\": %.4f, \"valu_only_ms\": %.4f},\n", p2, q2);
    printf(" \"free_running\": {\"both_ms\": %.4f, \"gather_only_ms\": %.4f, \"valu_only_ms\": %.4f, \"both_TBps\": %.2f}}\n", a0, b0, c0, bytes / a0 / 1e9);
    return 0;
}
