// gather_ubench.hip -- how should a wave fetch 64 LUT entries of 64 B each?
// Emulates the ICGN2D interpolation sweep's memory side: every wave walks the 33x33
// neighbourhood of "its" POI in a [H][W][16] float LUT, 64 entries per pass, and sums
// what it reads (so the loads cannot be dropped).  Variants:
//   0 strided   : lane s loads its own entry with 4 x dwordx4 (lane stride 64 B)   [current kernel]
//   1 coop      : 4 lanes share an entry; instruction q covers entries 16q..16q+15 contiguously
//   2 coop+lds  : as 1, then ds_write_b128 x4 / ds_read_b128 x4 so lane s ends with entry s
//   3 coop+dma  : as 1 with global_load_lds_dwordx4 into LDS, then ds_read_b128 x4
//   4 planar    : the table as four planes [k][H][W] of float4; lane s loads 16 B from each plane (lane stride
//                 16 B: a wave's load covers two subset rows of ~530 contiguous bytes)          [round-2 kernel]
//   5 planar, quad-aligned rows (round 3): every subset row takes 36 lane slots (9 quads, 3 idle lanes) so that the four
//                 lanes of a quad read ONE 64-byte-aligned piece -- 19 passes instead of 18.  With XOFF = 1..3 added to the
//                 subset origin the same variant (and variant 4) shows what an arbitrary origin costs.
// Build: hipcc --offload-arch=gfx950 -O3 gather_ubench.hip -o gather_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int W = 4096, H = 4096, SUB = 33, N = SUB * SUB, NT = (N + 63) / 64, ITERS = 3;

__device__ __forceinline__ unsigned entry_of(int s, int x0, int y0) {
    s = s < N ? s : 0;
    const int r = s / SUB, c = s - r * SUB;
    return ((unsigned)(y0 + r) * W + (unsigned)(x0 + c)) << 6;  // byte offset
}

template <int VARIANT, int WPB, int LOCK = 0>
__global__ __launch_bounds__(64 * WPB) void k(const char* __restrict__ lut, float* __restrict__ out, int npoi, int grid_side, int xoff) {
    __shared__ float4 stage[WPB][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunk = (npoi / WPB + 7) / 8;
    const int grp = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    const int poi = grp * WPB + wave;
    if (poi >= npoi) return;
    const int x0 = 8 + (poi % grid_side) * 8 + xoff, y0 = 8 + (poi / grid_side) * 8;
    float acc = 0.f;
    const int swz = ((lane & 3) ^ ((lane >> 4) & 3)) * 16;
    for (int it = 0; it < ITERS; it++) {
        constexpr int NTV = VARIANT == 5 ? (SUB * 36 + 63) / 64 : NT;
        for (int t = 0; t < NTV; t++) {
            // LOCK > 0 (round 3): the workgroup's waves re-align every LOCK passes, so that neighbouring POIs ask for the
            // same table lines at about the same time (what OC_SWEEP_BARRIER does in icgn2d.hip)
            if (LOCK > 0 && t % LOCK == 0) __builtin_amdgcn_s_barrier();
            unsigned e;
            if (VARIANT == 5) {
                // slot q = (row, position): the row's first pixel sits at position (x0 & 3), so quads are 64-byte aligned
                const int q = t * 64 + lane, r = min(q / 36, SUB - 1), p = q - (q / 36) * 36;
                const int c = min(max(p - (x0 & 3), 0), SUB - 1);
                e = ((unsigned)(y0 + it + r) * W + (unsigned)(x0 + c)) << 6;
            } else {
                e = entry_of(t * 64 + lane, x0, y0 + it);
            }
            float4 c0, c1, c2, c3;
            if (VARIANT == 0) {
                const float4* p = reinterpret_cast<const float4*>(lut + e);
                c0 = p[0]; c1 = p[1]; c2 = p[2]; c3 = p[3];
            } else if (VARIANT == 4 || VARIANT == 5) {
                constexpr size_t plane = (size_t)W * H * 16;
                const char* p = lut + (e >> 2);
                c0 = *reinterpret_cast<const float4*>(p);
                c1 = *reinterpret_cast<const float4*>(p + plane);
                c2 = *reinterpret_cast<const float4*>(p + 2 * plane);
                c3 = *reinterpret_cast<const float4*>(p + 3 * plane);
            } else {
                unsigned eq[4];
#pragma unroll
                for (int q = 0; q < 4; q++) eq[q] = __builtin_amdgcn_ds_bpermute(4 * ((lane >> 2) + 16 * q), e);
                if (VARIANT == 1) {
                    c0 = *reinterpret_cast<const float4*>(lut + eq[0] + (lane & 3) * 16);
                    c1 = *reinterpret_cast<const float4*>(lut + eq[1] + (lane & 3) * 16);
                    c2 = *reinterpret_cast<const float4*>(lut + eq[2] + (lane & 3) * 16);
                    c3 = *reinterpret_cast<const float4*>(lut + eq[3] + (lane & 3) * 16);
                } else if (VARIANT == 2) {
                    float4 d[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) d[q] = *reinterpret_cast<const float4*>(lut + eq[q] + swz);
#pragma unroll
                    for (int q = 0; q < 4; q++) stage[wave][q * 64 + lane] = d[q];
                    __builtin_amdgcn_wave_barrier();
                    const int sw = (lane >> 2) & 3;
                    c0 = stage[wave][lane * 4 + (0 ^ sw)];
                    c1 = stage[wave][lane * 4 + (1 ^ sw)];
                    c2 = stage[wave][lane * 4 + (2 ^ sw)];
                    c3 = stage[wave][lane * 4 + (3 ^ sw)];
                    __builtin_amdgcn_wave_barrier();
                } else {
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(lut + eq[q] + swz),
                                                         (__attribute__((address_space(3))) void*)(&stage[wave][q * 64]), 16, 0, 0);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_wave_barrier();
                    const int sw = (lane >> 2) & 3;
                    c0 = stage[wave][lane * 4 + (0 ^ sw)];
                    c1 = stage[wave][lane * 4 + (1 ^ sw)];
                    c2 = stage[wave][lane * 4 + (2 ^ sw)];
                    c3 = stage[wave][lane * 4 + (3 ^ sw)];
                    __builtin_amdgcn_wave_barrier();
                }
            }
            // weights make every float matter and differ per piece, so a wrong transposition shows in the checksum
            acc += (c0.x + 2.f * c0.y + 3.f * c0.z + 4.f * c0.w) + 5.f * (c1.x + 2.f * c1.y + 3.f * c1.z + 4.f * c1.w) +
                   7.f * (c2.x + 2.f * c2.y + 3.f * c2.z + 4.f * c2.w) + 11.f * (c3.x + 2.f * c3.y + 3.f * c3.z + 4.f * c3.w);
        }
    }
    out[(size_t)poi * 64 + lane] = acc;
}

template <int VARIANT, int WPB = 4, int LOCK = 0>
double run(const char* lut, float* out, int npoi, int side, std::vector<float>& host, int xoff = 0) {
    const int groups = npoi / WPB;
    const int grid = ((groups + 7) / 8) * 8;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<VARIANT, WPB, LOCK>), dim3(grid), dim3(64 * WPB), 0, 0, lut, out, npoi, side, xoff);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k<VARIANT, WPB, LOCK>), dim3(grid), dim3(64 * WPB), 0, 0, lut, out, npoi, side, xoff);
    CHECK(hipEventRecord(b));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    CHECK(hipMemcpy(host.data(), out, host.size() * 4, hipMemcpyDeviceToHost));
    return ms / 3;
}

int main() {
    const int side = 500, npoi = side * side;
    const size_t lut_bytes = (size_t)W * H * 64;
    char* lut;
    float* out;
    CHECK(hipMalloc(&lut, lut_bytes));
    CHECK(hipMalloc(&out, (size_t)npoi * 64 * 4));
    std::vector<float> init(1 << 22);
    for (size_t i = 0; i < init.size(); i++) init[i] = (float)((i * 2654435761u >> 20) & 255) * (1.f / 64.f);
    for (size_t off = 0; off < lut_bytes; off += init.size() * 4) CHECK(hipMemcpy(lut + off, init.data(), init.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> h0((size_t)npoi * 64), h((size_t)npoi * 64);
    const double bytes = (double)npoi * ITERS * NT * 64 * 64;
    double ms = run<0>(lut, out, npoi, side, h0);
    printf("variant 0 strided  : %.3f ms  %.2f TB/s (algorithmic)\n", ms, bytes / ms / 1e9);
    ms = run<1>(lut, out, npoi, side, h);
    printf("variant 1 coop     : %.3f ms  %.2f TB/s (sums differ by design)\n", ms, bytes / ms / 1e9);
    ms = run<2>(lut, out, npoi, side, h);
    size_t bad = 0;
    for (size_t i = 0; i < h.size(); i++) bad += h[i] != h0[i];
    printf("variant 2 coop+lds : %.3f ms  %.2f TB/s  mismatches %zu\n", ms, bytes / ms / 1e9, bad);
    ms = run<3>(lut, out, npoi, side, h);
    bad = 0;
    for (size_t i = 0; i < h.size(); i++) bad += h[i] != h0[i];
    printf("variant 3 coop+dma : %.3f ms  %.2f TB/s  mismatches %zu\n", ms, bytes / ms / 1e9, bad);
    ms = run<4>(lut, out, npoi, side, h);
    printf("variant 4 planar   : %.3f ms  %.2f TB/s (sums differ by design: other bytes)\n", ms, bytes / ms / 1e9);
    for (int xoff = 0; xoff < 4; xoff++) {
        const double m4 = run<4>(lut, out, npoi, side, h, xoff), m5 = run<5>(lut, out, npoi, side, h, xoff);
        printf("origin + %d: planar %.3f ms (%.2f TB/s)   planar, quad-aligned rows (19 passes) %.3f ms (%.2f TB/s of the same samples)\n", xoff, m4,
               bytes / m4 / 1e9, m5, bytes / m5 / 1e9);
    }
    // round 3: 8-wave workgroups (the shape of the ICGN2D1 kernel), free-running against lockstep
    {
        const double a0 = run<4, 8, 0>(lut, out, npoi, side, h), a2 = run<4, 8, 2>(lut, out, npoi, side, h), a4 = run<4, 8, 4>(lut, out, npoi, side, h),
                     a1 = run<4, 8, 1>(lut, out, npoi, side, h);
        printf("planar, 8 waves per workgroup: free %.3f ms (%.2f TB/s)  barrier every pass %.3f  every 2 passes %.3f (%.2f TB/s)  every 4 passes %.3f\n", a0,
               bytes / a0 / 1e9, a1, a2, bytes / a2 / 1e9, a4);
        const double b0 = run<5, 8, 0>(lut, out, npoi, side, h, 1), b2 = run<5, 8, 2>(lut, out, npoi, side, h, 1), b4 = run<5, 8, 4>(lut, out, npoi, side, h, 1);
        printf("quad-aligned rows, origin + 1, 8 waves: free %.3f ms  every 2 passes %.3f (%.2f TB/s)  every 4 passes %.3f\n", b0, b2, bytes / b2 / 1e9, b4);
        const double c0 = run<4, 8, 0>(lut, out, npoi, side, h, 1), c2 = run<4, 8, 2>(lut, out, npoi, side, h, 1);
        printf("planar, origin + 1, 8 waves: free %.3f ms  every 2 passes %.3f (%.2f TB/s)\n", c0, c2, bytes / c2 / 1e9);
    }
    return 0;
}
