// valu_chain_ubench.hip -- what a VALU instruction of the solvers' kind costs, and why (round 4).
//
// The three solvers spend ~4.5 cycles of SIMD time per VALU wave-instruction they retire (PMC: SQ_INSTS_VALU against the kernels'
// time).  What does an instruction of their kind cost when nothing else is in the way?  This benchmark evaluates the reference's
// 16-term bicubic polynomial (src/oc_cubic_bspline.cpp:159-177: 28 products, 15 left-to-right additions; dic2d_device.h lut_poly) on
// register-resident coefficients -- no memory, no LDS -- plain and as packed instructions on two samples per lane, and varies what
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


// ---- two samples per lane, every operation a packed one (v_pk_mul_f32 / v_pk_add_f32): what a packed instruction costs.
// PAIR 0: the coefficient pairs {A.c[i], B.c[i]} are register-resident as pairs (the ideal); PAIR 1: they are formed from the
// two samples' separately loaded coefficient quads inside the loop (v_pk_mov_b32 / v_mov_b32), as a gather would deliver them.
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f2 poly2(const f2* c, f2 dx, f2 dy) {
    const f2 dx2 = dx * dx, dy2 = dy * dy;
    const f2 dx3 = dx2 * dx, dy3 = dy2 * dy;
    const f2 t0 = c[0], t1 = c[1] * dx, t2 = c[2] * dx2, t3 = c[3] * dx3;
    const f2 t4 = c[4] * dy, t5 = (c[5] * dy) * dx, t6 = (c[6] * dy) * dx2, t7 = (c[7] * dy) * dx3;
    const f2 t8 = c[8] * dy2, t9 = (c[9] * dy2) * dx, t10 = (c[10] * dy2) * dx2, t11 = (c[11] * dy2) * dx3;
    const f2 t12 = c[12] * dy3, t13 = (c[13] * dy3) * dx, t14 = (c[14] * dy3) * dx2, t15 = (c[15] * dy3) * dx3;
    f2 v = t0;
    v = v + t1; v = v + t2; v = v + t3; v = v + t4; v = v + t5; v = v + t6; v = v + t7; v = v + t8;
    v = v + t9; v = v + t10; v = v + t11; v = v + t12; v = v + t13; v = v + t14; v = v + t15;
    return v;
}

template <int G, int PAIR>
__global__ __launch_bounds__(256) void kp(float* __restrict__ out, int iters, float seed) {
    f2 c[G][16];
    float a[G][16], b[G][16];
    f2 dx[G], dy[G];
#pragma unroll
    for (int g = 0; g < G; g++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            a[g][i] = seed * (float)(i + 1 + g) + (float)threadIdx.x * 1e-3f;
            b[g][i] = seed * (float)(i + 3 + g) - (float)threadIdx.x * 1e-3f;
            c[g][i] = f2{a[g][i], b[g][i]};
        }
        dx[g] = f2{0.25f + 0.01f * g, 0.26f + 0.01f * g};
        dy[g] = f2{0.5f + 0.02f * g, 0.51f + 0.02f * g};
    }
    f2 acc = f2{0.f, 0.f};
#pragma nounroll
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int g = 0; g < G; g++) {
            if (PAIR) {
                f2 cc[16];
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    float x = a[g][i], y = b[g][i];
                    asm volatile("" : "+v"(x), "+v"(y));   // the pairing happens here, every time
                    cc[i] = f2{x, y};
                }
                acc = acc + poly2(cc, dx[g], dy[g]);
            } else {
                acc = acc + poly2(c[g], dx[g], dy[g]);
            }
            dx[g] = dx[g] + f2{1e-6f, 1e-6f};
            dy[g] = dy[g] - f2{1e-6f, 1e-6f};
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y;
}

template <int G, int PAIR>
void runp(float* out, int W) {
    const int iters = 2048 / G, grid = 256 * W;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL((kp<G, PAIR>), dim3(grid), dim3(256), 0, 0, out, iters, 1.0f);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL((kp<G, PAIR>), dim3(grid), dim3(256), 0, 0, out, iters, 1.0f);
        CHECK(hipEventRecord(b));
        CHECK(hipDeviceSynchronize());
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, a, b));
        best = ms < best ? ms : best;
    }
    const double polys_per_simd = (double)iters * G * W * 2.0;       // two polynomials per lane and pass
    const double cyc = best * 1e-3 * 2.4e9 / polys_per_simd;
    printf("{\"packed\": 1, \"G\": %d, \"pairing_in_loop\": %d, \"waves_per_simd\": %d, \"ms\": %.4f, \"cycles_per_polynomial_per_simd\": %.1f, "
           "\"cycles_per_packed_instr\": %.2f}\n", G, PAIR, W, best, cyc, 2.0 * cyc / 46.0);
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
        runp<1, 0>(out, W);
        runp<2, 0>(out, W);
        runp<1, 1>(out, W);
        runp<2, 1>(out, W);
    }
    return 0;
}
