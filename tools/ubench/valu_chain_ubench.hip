// valu_chain_ubench.hip -- what a VALU instruction of the solvers' kind costs, and why (round 4).
//
// The three solvers run at ~4.4 cycles per VALU wave-instruction and SIMD (coissue_ubench.hip; PMC: SQ_INSTS_VALU x 4.4 cycles =
// the kernels' time), although a plain v_mul_f32 / v_add_f32 stream retires at 1.5 - 2.4 cycles (valu_ubench / dual_issue_ubench,
// round 2).  This benchmark evaluates the reference's 16-term bicubic polynomial (src/oc_cubic_bspline.cpp:159-177: 28 products, 15
// left-to-right additions; dic2d_device.h lut_poly) on register-resident coefficients -- no memory, no LDS -- and varies what
// could explain the difference:
//   G     independent samples interleaved per wave (1, 2, 4): instruction-level parallelism inside a wave
//   TREE  0 = the reference's left-to-right chain of 15 dependent additions, 1 = four partial sums (other bits: a probe only)
//   W     waves per SIMD (2, 4, 6, 8)
// Output: cycles per polynomial per SIMD (2.4 GHz nominal) and per VALU instruction (47 per polynomial: 4 powers + 28 + 15).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize tools/ubench/valu_chain_ubench.hip -o valu_chain_ubench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Fetch {
    float c[16];
    float dx, dy;
};

template <int TREE>
__device__ __forceinline__ float poly(const Fetch& f) {
    const float dx = f.dx, dy = f.dy;
    const float dx2 = dx * dx, dy2 = dy * dy;
    const float dx3 = dx2 * dx, dy3 = dy2 * dy;
    const float t0 = f.c[0], t1 = f.c[1] * dx, t2 = f.c[2] * dx2, t3 = f.c[3] * dx3;
    const float t4 = f.c[4] * dy, t5 = (f.c[5] * dy) * dx, t6 = (f.c[6] * dy) * dx2, t7 = (f.c[7] * dy) * dx3;
    const float t8 = f.c[8] * dy2, t9 = (f.c[9] * dy2) * dx, t10 = (f.c[10] * dy2) * dx2, t11 = (f.c[11] * dy2) * dx3;
    const float t12 = f.c[12] * dy3, t13 = (f.c[13] * dy3) * dx, t14 = (f.c[14] * dy3) * dx2, t15 = (f.c[15] * dy3) * dx3;
    if (TREE) return (((t0 + t1) + (t2 + t3)) + ((t4 + t5) + (t6 + t7))) + (((t8 + t9) + (t10 + t11)) + ((t12 + t13) + (t14 + t15)));
    float v = t0;
    v = v + t1; v = v + t2; v = v + t3; v = v + t4; v = v + t5; v = v + t6; v = v + t7; v = v + t8;
    v = v + t9; v = v + t10; v = v + t11; v = v + t12; v = v + t13; v = v + t14; v = v + t15;
    return v;
}

template <int G, int TREE>
__global__ __launch_bounds__(256) void k(float* __restrict__ out, int iters, float seed) {
    Fetch f[G];
#pragma unroll
    for (int g = 0; g < G; g++) {
#pragma unroll
        for (int i = 0; i < 16; i++) f[g].c[i] = seed * (float)(i + 1 + g) + (float)threadIdx.x * 1e-3f;
        f[g].dx = 0.25f + 0.01f * g;
        f[g].dy = 0.5f + 0.02f * g;
    }
    float acc = 0.f;
#pragma nounroll
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int g = 0; g < G; g++) {
            acc = acc + poly<TREE>(f[g]);
            f[g].dx = f[g].dx + 1e-6f;  // a new argument every time: nothing can be hoisted
            f[g].dy = f[g].dy - 1e-6f;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int G, int TREE>
void run(float* out, int W) {
    const int iters = 4096 / G, grid = 256 * W;  // 256 CUs, W workgroups of 4 waves each = W waves per SIMD
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<G, TREE>), dim3(grid), dim3(256), 0, 0, out, iters, 1.0f);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL((k<G, TREE>), dim3(grid), dim3(256), 0, 0, out, iters, 1.0f);
        CHECK(hipEventRecord(b));
        CHECK(hipDeviceSynchronize());
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, a, b));
        best = ms < best ? ms : best;
    }
    const double polys_per_simd = (double)iters * G * W;             // each SIMD hosts W waves
    const double cyc = best * 1e-3 * 2.4e9 / polys_per_simd;
    printf("{\"G\": %d, \"tree\": %d, \"waves_per_simd\": %d, \"ms\": %.4f, \"cycles_per_polynomial_per_simd\": %.1f, \"cycles_per_valu_instr\": %.2f}\n", G, TREE, W,
           best, cyc, cyc / 46.0);  // 46 VALU instructions per polynomial in the ISA of the loop (4 powers, 27 products, 15 adds)
}

int main() {
    float* out;
    CHECK(hipMalloc(&out, 256 * 8 * 256 * 4));
    for (int W : {2, 4, 6, 8}) {
        run<1, 0>(out, W);
        run<2, 0>(out, W);
        run<4, 0>(out, W);
        run<1, 1>(out, W);
        run<2, 1>(out, W);
    }
    return 0;
}
