// icgn2d_band.hip -- ICGN2D1 / ICGN2D2 with the bicubic coefficient BAND of a workgroup staged in LDS (round 6).
//
// Replaces ICGN2D1::compute(POI2D*) (src/oc_icgn.cpp:144-341) / ICGN2D2::compute(POI2D*) (:685-898) for a whole queue, like
// icgn2d.hip, whose arithmetic it repeats operation for operation (same bits: oracle OC_ORDER_LANES / _FMA).  What differs is
// where the 64 bytes per sample of BicubicBspline::compute (src/oc_cubic_bspline.cpp:134-181) come from.  icgn2d.hip gathers
// them per lane from the planar table in global memory: four buffer_load_b128 per sample through the CU's texture path, which
// moves 64 B per clock and is what its interpolation sweeps wait for (gathers alone 1.78 ms, their arithmetic alone 1.04 ms,
// DESIGN.md 4.1).  The LDS moves 256 B per clock (ds_read_b128).  So:
//   * the eight waves of a workgroup -- eight consecutive POIs of the visiting order, normally neighbours in a row of the POI grid
//     -- sweep their subsets in lockstep WINDOWS of two passes (128 samples = ~4 subset rows).  In a window all of them touch ONE
//     band of the table: kBandRows rows x kBandCols pixels x 4 planes x 16 B = 36 KB.  Every iteration each wave bounds, for
//     every window at once (one lane per window corner), the image of the window's index rectangle under its warp (affine: the
//     corners bound it exactly, every float operation being monotone; quadratic: corners + a bound on the square terms); the
//     workgroup agrees on one box origin per window (minimum over its waves) and every wave knows per window whether its samples
//     lie inside that box (FAST) or not (SLOW: a POI of the next grid row in the same workgroup, a large deformation gradient);
//   * the box is loaded ONCE per workgroup and window with LDS-DMA (buffer_load_dwordx4 ... lds: global -> LDS without passing
//     through registers, addresses formed on the scalar unit: no VALU instruction, no VGPR); a FAST wave then reads its four
//     float4 per sample with ds_read_b128 -- two VALU instructions of address arithmetic per sample, like the global gather's --
//     a SLOW wave gathers from global memory as icgn2d.hip does, started before it waits for the band;
//   * the warped target subset lives in REGISTERS (a 16 + k float vector addressed with the VGPR index mode, so the pass loops
//     stay rolled): the LDS of a workgroup is the coordinate table, the band, the boxes and H^-1 -- no per-wave arrays -- and three
//     workgroups per CU stay resident;
//   * all eight waves stay until the workgroup's last POI is done: a finished wave keeps staging its rows of the band (and
//     passes every barrier), it only stops computing.
// Launch shape: 512 threads, one POI per wave, XCD-contiguous groups, tile-ordered queue (poi_order.hip).  Not available for
// self-adaptive radii (one subset size per launch), IC-LM, or subsets above 20 / 28 passes: icgn2d.hip serves those.
#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "dic2d_device.h"
#include "oc_kernels.h"

namespace ochip {
namespace OC_ARITH {

constexpr int kBandWaves = 8;    // POIs per workgroup
constexpr int kBandWP = 2;       // passes per window
constexpr int kBandCols = 96;    // pixels per staged row
constexpr int kBandPitch = 97;   // row pitch of the staged box in float4 (odd multiple of 16 B: consecutive rows fall on different banks)
constexpr int kBandRows = 6;     // staged rows
constexpr int kBandPlane = kBandRows * kBandPitch;   // float4 per plane
constexpr int kBandMaxWin = 16;  // windows per sweep (32 passes)

struct Icgn2dBandLaunch {
    int stride_f;
    int nt;                    // ceil(N / 64)
    int xcd_chunk;
    unsigned long long count;
    int ablate;                // timing experiments (environment OC_BAND_ABLATE; results are NOT valid except for 0, 4, 16): 1 = no window
                               // barriers, 2 = no staging, 4 = every wave takes the global gathers (the launch shape alone),
                               // 8 = no polynomial sweeps, 16 = exactly three iterations for every POI
};

typedef float f16v __attribute__((ext_vector_type(16)));

// the warped target subset of a wave, one float per pass and lane, in registers: 16 + TSB passes, indexed with a wave-uniform
// pass number (s_set_gpr_idx / v_mov: one VALU instruction per access)
template <int TSB>
struct TsRegs {
    typedef float bvec __attribute__((ext_vector_type(TSB)));
    f16v a;
    bvec b;
    __device__ __forceinline__ void set(int t, float v) {
        if (t < 16) a[t] = v;
        else b[t - 16] = v;
    }
    __device__ __forceinline__ float get(int t) const { return t < 16 ? a[t] : b[t - 16]; }
};

__device__ __forceinline__ int dpp_quad_xor1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true); }
__device__ __forceinline__ int dpp_quad_xor2(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true); }

constexpr size_t icgn2d_band_lds_bytes(int nt, int dof) {
    return (size_t)3 * nt * kWave * 4 + (size_t)4 * kBandPlane * 16 + (size_t)kBandWaves * kBandMaxWin * 16 + (size_t)kBandWaves * dof * dof * 4 + 16;
}

// what a pass of the Hessian sweep / the numerator pass fetches for one sample
struct GradSample {
    float gx, gy, ref = 0.f;
};

template <int DOF, int TSB, int OFFS>
__global__ __launch_bounds__(64 * kBandWaves, DOF == 6 ? 6 : 4) void icgn2d_band_kernel(Icgn2dParams P, float* __restrict__ pois,
                                                                                     Icgn2dBandLaunch L) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    typedef __attribute__((address_space(3))) void* lds_vp;
    constexpr int NH = DOF * (DOF + 1) / 2;
    constexpr int WP = kBandWP;
    constexpr bool COOP = DOF == 6;
    constexpr int kSetupBatch = 6, kHessBatch = 6, kNumBatch = DOF == 6 ? 6 : 4;

    const int NT = L.nt;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f2* __restrict__ tab_xy = reinterpret_cast<f2*>(lds);
    unsigned* __restrict__ tab_off = reinterpret_cast<unsigned*>(lds + 2 * NT * kWave);
    float4* __restrict__ band = reinterpret_cast<float4*>(lds + 3 * NT * kWave);
    int4* __restrict__ boxes = reinterpret_cast<int4*>(band + 4 * kBandPlane);   // [wave][window]: xmin, ymin, xmax, ymax
    float* __restrict__ hrow_area = reinterpret_cast<float*>(boxes + kBandWaves * kBandMaxWin);  // H^-1 of the eight POIs, row-major
    volatile int* ctrl = reinterpret_cast<volatile int*>(hrow_area + kBandWaves * DOF * DOF);
    float* __restrict__ coop_area = reinterpret_cast<float*>(band);  // the band is idle until the first sweep

    const int height = P.height, width = P.width;
    const int rx = P.rx, ry = P.ry;
    const int W = 2 * rx + 1, N = W * (2 * ry + 1);
    const float fN = (float)N;
    const int NF = N / kWave;
    const bool tail_valid = (NF * kWave + lane) < N;
    const int nwin = (NT + WP - 1) / WP;
    {
        const unsigned w4t = (unsigned)width * 4u;
        for (int s = threadIdx.x; s < NT * kWave; s += kWave * kBandWaves) {
            const int r = s / W, c = s - r * W;
            tab_xy[s] = mk2((float)(c - rx), (float)(r - ry));
            tab_off[s] = (unsigned)r * w4t + ((unsigned)c << 2);
        }
        if (threadIdx.x == 0) ctrl[0] = 0;
        __syncthreads();
    }
    auto tab_at = [&](int t) { return tab_xy[t * kWave + lane]; };
    auto off_at = [&](int t) { return tab_off[t * kWave + lane]; };

    unsigned long long grp = blockIdx.x;
    if (L.xcd_chunk > 0) grp = (unsigned long long)(blockIdx.x & 7u) * L.xcd_chunk + (blockIdx.x >> 3);
    const unsigned long long slot = grp * kBandWaves + wave;
    bool active = slot < L.count;  // wave-uniform
    const unsigned long long idx =
        active ? (P.perm ? (unsigned long long)__builtin_amdgcn_readfirstlane((int)P.perm[slot]) : slot) : 0ull;
    float* poi = pois + idx * (unsigned long long)L.stride_f;
    const float rec = (active && lane < poi2d::FLOATS) ? poi[lane] : 0.f;
    const float px = wave_bcast(rec, poi2d::X), py = wave_bcast(rec, poi2d::Y);
    const float u_in = wave_bcast(rec, poi2d::U), ux_in = wave_bcast(rec, poi2d::UX), uy_in = wave_bcast(rec, poi2d::UY);
    const float v_in = wave_bcast(rec, poi2d::V), vx_in = wave_bcast(rec, poi2d::VX), vy_in = wave_bcast(rec, poi2d::VY);
    const float zncc_in = wave_bcast(rec, poi2d::ZNCC);
    float offx = 0.f, offy = 0.f;
    if constexpr (OFFS) {
        if (active) {
            offx = uni(P.offsets[2 * idx]);
            offy = uni(P.offsets[2 * idx + 1]);
        }
    }
    // guard, src/oc_icgn.cpp:160-167 (2D2: 705-712)
    if (active && (py - ry < 0 || px - rx < 0 || py + ry > height - 1 || px + rx > width - 1 || fabsf(u_in) >= width ||
                   fabsf(v_in) >= height || zncc_in < 0 || isnan(u_in) || isnan(v_in))) {
        if (lane == 0) poi[poi2d::ZNCC] = zncc_in >= 0 ? -3.f : zncc_in;
        active = false;
    }
    const unsigned goff = (unsigned)__builtin_amdgcn_readfirstlane((((int)py - ry) * width + ((int)px - rx)) * 4);
    const unsigned roff = (unsigned)__builtin_amdgcn_readfirstlane((((int)(py - ry)) * width + ((int)(px - rx))) * 4);
    const __amdgpu_buffer_rsrc_t r_gx = make_rsrc(P.gx), r_gy = make_rsrc(P.gy), r_ref = make_rsrc(P.ref);
    const LutPlanes4 r_lut(P.lut, height, width);
    // staging: wave v owns plane v & 3 and the rows (v >> 2), (v >> 2) + 2, (v >> 2) + 4 of the box: per row one 1 KiB LDS-DMA piece
    // (pixels 0 .. 63) and one half piece (pixels 64 .. 95, lanes 0 .. 31) -- addresses on the scalar unit only
    const int st_plane = wave & 3, st_row0 = wave >> 2;
    const __amdgpu_buffer_rsrc_t r_stage = st_plane == 0 ? r_lut.p0 : (st_plane == 1 ? r_lut.p1 : (st_plane == 2 ? r_lut.p2 : r_lut.p3));
    const bool band_ok = width >= kBandCols + 4 && height >= kBandRows + 4 && !(L.ablate & 4);

    TsRegs<TSB> ts;

    float ref_norm = 0.f, ref_mean = 0.f;
    float h[NH];
#pragma unroll
    for (int i = 0; i < NH; i++) h[i] = 0.f;
    if (active) {
        // ---- reference subset, zero-mean + norm (src/oc_icgn.cpp:174-176, src/oc_subset.cpp:39-53)
        float acc = 0.f;
        passes_batched<kSetupBatch>(
            NF, NT, tail_valid, [&](int t, bool valid) { return valid ? buf_f32(r_ref, off_at(t), roff) : 0.f; },
            [&](int t, bool valid, float v) {
                acc = valid ? acc + v : acc;
                ts.set(t, v);
            });
        const float mean = wave_allreduce_sum(acc) / fN;
        ref_mean = uni(mean);
        acc = 0.f;
#pragma unroll 3
        for (int t = 0; t < NF; t++) {
            const float d = ts.get(t) - mean;
            acc = mad(d, d, acc);
        }
        if (NF < NT) {
            const float d = ts.get(NF) - mean;
            acc = tail_valid ? mad(d, d, acc) : acc;
        }
        ref_norm = uni(sqrtf(wave_allreduce_sum(acc)));

        // ---- steepest-descent image + Hessian (src/oc_icgn.cpp:179-207; 2D2: 716-756)
        if constexpr (DOF == 6) {
            f2 hAA = mk2(0.f, 0.f), hBB = hAA, hAB = hAA, hAs = hAA, hxA = hAA, hyA = hAA, hxB = hAA, hyB = hAA;
            float h00 = 0.f, h33 = 0.f, h30 = 0.f, h21 = 0.f, h54 = 0.f;
            passes_batched<kHessBatch>(
                NF, NT, tail_valid,
                [&](int t, bool valid) {
                    GradSample v;
                    const unsigned off = off_at(t);
                    v.gx = valid ? buf_f32(r_gx, off, goff) : 0.f;
                    v.gy = valid ? buf_f32(r_gy, off, goff) : 0.f;
                    return v;
                },
                [&](int t, bool valid, const GradSample& v) {
                    const float g_x = v.gx, g_y = v.gy;
                    const f2 xy = tab_at(t) - mk2(offx, offy);
                    const f2 A = g_x * xy, B = g_y * xy;
                    const f2 nAA = mad(A, A, hAA), nBB = mad(B, B, hBB), nAB = mad(A, B, hAB), nAs = mad(A, B.yx, hAs);
                    const f2 nxA = mad(g_x, A, hxA), nyA = mad(g_y, A, hyA), nxB = mad(g_x, B, hxB), nyB = mad(g_y, B, hyB);
                    const float n00 = mad(g_x, g_x, h00), n33 = mad(g_y, g_y, h33), n30 = mad(g_y, g_x, h30);
                    const float n21 = mad(A.y, A.x, h21), n54 = mad(B.y, B.x, h54);
                    if (valid) {
                        hAA = nAA; hBB = nBB; hAB = nAB; hAs = nAs; hxA = nxA; hyA = nyA; hxB = nxB; hyB = nyB;
                        h00 = n00; h33 = n33; h30 = n30; h21 = n21; h54 = n54;
                    }
                });
            h[0] = h00;
            h[1] = hxA.x; h[2] = hAA.x;
            h[3] = hxA.y; h[4] = h21; h[5] = hAA.y;
            h[6] = h30; h[7] = hyA.x; h[8] = hyA.y; h[9] = h33;
            h[10] = hxB.x; h[11] = hAB.x; h[12] = hAs.y; h[13] = hyB.x; h[14] = hBB.x;
            h[15] = hxB.y; h[16] = hAs.x; h[17] = hAB.y; h[18] = hyB.y; h[19] = h54; h[20] = hBB.y;
        } else {
            f2 hp[12][6];
            float hd[12];
#pragma unroll
            for (int r = 0; r < 12; r++) {
                hd[r] = 0.f;
#pragma unroll
                for (int q = 0; q < 6; q++) hp[r][q] = mk2(0.f, 0.f);
            }
            auto fetch = [&](int t, bool valid) {
                GradSample v;
                const unsigned off = off_at(t);
                v.gx = valid ? buf_f32(r_gx, off, goff) : 0.f;
                v.gy = valid ? buf_f32(r_gy, off, goff) : 0.f;
                return v;
            };
            auto sample = [&](int t, bool valid, const GradSample& v) {
                const float g_x = v.gx, g_y = v.gy;
                const f2 lxy = tab_at(t) - mk2(offx, offy);
                const float fxl = lxy.x, fyl = lxy.y;
                const float xx = (fxl * fxl) * 0.5f, xy = fxl * fyl, yy = (fyl * fyl) * 0.5f;
                const f2 m01 = mk2(1.f, fxl), m23 = mk2(fyl, xx), m45 = mk2(xy, yy);
                const f2 sdp[6] = {g_x * m01, g_x * m23, g_x * m45, g_y * m01, g_y * m23, g_y * m45};
#pragma unroll
                for (int r = 0; r < 12; r++) {
                    const float sr = (r & 1) ? sdp[r / 2].y : sdp[r / 2].x;
#pragma unroll
                    for (int q = 0; q < (r + 1) / 2; q++) {
                        const f2 nv = mad(sr, sdp[q], hp[r][q]);
                        hp[r][q] = valid ? nv : hp[r][q];
                    }
                    if ((r & 1) == 0) hd[r] = valid ? mad(sr, sr, hd[r]) : hd[r];
                }
            };
            passes_prefetched(NF, NT, tail_valid, fetch, sample);
            int k = 0;
#pragma unroll
            for (int r = 0; r < 12; r++)
#pragma unroll
                for (int c = 0; c <= r; c++, k++)
                    h[k % NH] = (c == r && (r & 1) == 0) ? hd[r] : ((c & 1) ? hp[r][c / 2].y : hp[r][c / 2].x);
        }
    }
    // ---- inverse of the Hessian (:210 / :759), filed row-major in hrow_area
    if constexpr (COOP) {
        if (active) wave_reduce_sum_multi_to_lds<NH>(h, lane, coop_area + wave * 64);
        else if (lane < 24) coop_area[wave * 64 + lane] = 0.f;
        __syncthreads();
        if (wave == 0) coop_inverse6_x8(coop_area, lane);
        __syncthreads();
        // (H^-1 row-major moves out of the band area, which the first sweep overwrites)
        if (lane < DOF * DOF) hrow_area[wave * DOF * DOF + lane] = coop_area[wave * 64 + 24 + lane];
    } else {
        if (active) {
            float col[DOF], hinv_col[DOF];
#pragma unroll
            for (int i = 0; i < DOF; i++) col[i] = 0.f;
            wave_allreduce_sum_multi<NH>(h, lane);
            int k = 0;
#pragma unroll
            for (int i = 0; i < DOF; i++)
#pragma unroll
                for (int j = 0; j <= i; j++) {
                    const float v = h[k++];
                    if (lane == j) col[i] = v;
                    if (lane == i) col[j] = v;
                }
            lu_inverse_lanes<DOF>(col, hinv_col, lane);
            // lane j holds column j of H^-1: element (i, j) goes to row i
            if (lane < DOF) {
#pragma unroll
                for (int i = 0; i < DOF; i++) hrow_area[wave * DOF * DOF + i * DOF + lane] = hinv_col[i];
            }
        }
    }

    // ---- IC-GN loop (src/oc_icgn.cpp:216-307; 2D2: 762-858)
    float Wm[9];    // 2D1
    float Wcol[6];  // 2D2: column j in lane j
    float row3[6], row4[6];
#pragma unroll
    for (int i = 0; i < 9; i++) Wm[i] = 0.f;
#pragma unroll
    for (int i = 0; i < 6; i++) Wcol[i] = row3[i] = row4[i] = 0.f;
    if constexpr (DOF == 6) {
        set_warp_2d1(Wm, u_in, ux_in, uy_in, v_in, vx_in, vy_in);
    } else {
        const float q[12] = {u_in, ux_in, uy_in, 0.f, 0.f, 0.f, v_in, vx_in, vy_in, 0.f, 0.f, 0.f};
        float w36[36];
        set_warp_2d2(w36, q);
#pragma unroll
        for (int i = 0; i < 6; i++) {
            float c = 0.f;
#pragma unroll
            for (int j = 0; j < 6; j++) c = lane == j ? w36[i * 6 + j] : c;
            Wcol[i] = c;
        }
    }
    const float tcx = px + offx, tcy = py + offy;
    // Deformation2D1::warp (src/oc_deformation.cpp:94-105) / Deformation2D2::warp (:268-282) of local coordinates (xl, yl)
    auto warp_point = [&](float xl, float yl, float& ax, float& ay) {
        float wx, wy;
        if constexpr (DOF == 6) {
            wx = mad(Wm[1], yl, Wm[0] * xl) + Wm[2];
            wy = mad(Wm[4], yl, Wm[3] * xl) + Wm[5];
        } else {
            const float pv[6] = {xl * xl, xl * yl, yl * yl, xl, yl, 1.f};
            wx = row3[0] * pv[0];
            wy = row4[0] * pv[0];
#pragma unroll
            for (int k = 1; k < 6; k++) {
                wx = mad(row3[k], pv[k], wx);
                wy = mad(row4[k], pv[k], wy);
            }
        }
        ax = tcx + wx;
        ay = tcy + wy;
    };
    // Boxes of ALL windows of this wave at once, lane 4 w + c = corner c of window w's index rectangle (all columns x the rows
    // its samples lie in; the row of the window's first / last sample comes from the coordinate table, whose y entry is
    // (float)(row - ry)): the corner's target pixel, min / max over the quad.  The box bounds the pixels floor(x), floor(y) of
    // every sample of the window: exactly for the affine warp (monotone float operations), with the square terms' bound
    // 2 (|a_xx| X^2 + |a_yy| Y^2) + 0.01 on either side for the quadratic one.  corner_out = a corner leaves the interpolatable
    // range (2D1: a corner IS a sample, the reference abandons the POI, see icgn2d.hip kCornerTest) -- a wave-uniform flag.
    const int bw_l = lane >> 2;
    auto publish_boxes = [&](bool& corner_out) {
        if constexpr (DOF == 12) {
#pragma unroll
            for (int k = 0; k < 6; k++) {
                row3[k] = wave_bcast(Wcol[3], k);
                row4[k] = wave_bcast(Wcol[4], k);
            }
        }
        const int cs = min((lane & 2) ? kWave * WP * (bw_l + 1) - 1 : kWave * WP * bw_l, N - 1);
        const float cxl = (float)((lane & 1) ? rx : -rx) - offx;
        const float cyl = tab_xy[cs].y - offy;
        float ax, ay;
        warp_point(cxl, cyl, ax, ay);
        float lox = ax, hix = ax, loy = ay, hiy = ay;
        if constexpr (DOF == 12) {
            const float X = (float)rx + fabsf(offx), Y = (float)ry + fabsf(offy);
            const float dx_ = 2.f * (fabsf(row3[0]) * X * X + fabsf(row3[2]) * Y * Y) + 0.01f;
            const float dy_ = 2.f * (fabsf(row4[0]) * X * X + fabsf(row4[2]) * Y * Y) + 0.01f;
            lox -= dx_; hix += dx_; loy -= dy_; hiy += dy_;
        }
        int x0 = floor_to_int(lox), x1 = floor_to_int(hix), y0 = floor_to_int(loy), y1 = floor_to_int(hiy);
        const bool cout = (unsigned)(x0 - 1) > (unsigned)(width - 4) || (unsigned)(y0 - 1) > (unsigned)(height - 4) ||
                          (unsigned)(x1 - 1) > (unsigned)(width - 4) || (unsigned)(y1 - 1) > (unsigned)(height - 4);
        corner_out = __builtin_amdgcn_ballot_w64(cout && bw_l < nwin) != 0;
        x0 = min(x0, dpp_quad_xor1(x0)); x1 = max(x1, dpp_quad_xor1(x1));
        y0 = min(y0, dpp_quad_xor1(y0)); y1 = max(y1, dpp_quad_xor1(y1));
        x0 = min(x0, dpp_quad_xor2(x0)); x1 = max(x1, dpp_quad_xor2(x1));
        y0 = min(y0, dpp_quad_xor2(y0)); y1 = max(y1, dpp_quad_xor2(y1));
        // (a box that leaves the range is not FAST: publish it as "nothing", so that it neither drags the origin nor is read)
        if (x0 < 1 || y0 < 1 || x1 > width - 3 || y1 > height - 3) { x0 = y0 = 0x7fffffff; x1 = y1 = -0x7fffffff; }
        if ((lane & 3) == 0 && bw_l < nwin) boxes[wave * kBandMaxWin + bw_l] = make_int4(x0, y0, x1, y1);
    };
    auto publish_none = [&]() {
        if (lane < nwin) boxes[wave * kBandMaxWin + lane] = make_int4(0x7fffffff, 0x7fffffff, -0x7fffffff, -0x7fffffff);
    };
    bool corner_out = false;
    if (active) {
        publish_boxes(corner_out);
        if (lane == 0) atomicAdd(const_cast<int*>(ctrl), 1);
    } else {
        publish_none();
    }

    int iter = 0;
    float dp_norm = 0.f, znssd = 0.f;
    float cur[12];
#pragma unroll
    for (int i = 0; i < 12; i++) cur[i] = 0.f;

#pragma nounroll
    for (;;) {
        __syncthreads();  // the boxes of this iteration are published, the finished waves have reported, the band is free
        if (ctrl[0] <= 0) return;
        // lane w: origin of window w's staged box = the minimum over the waves' boxes; this wave is FAST in window w when its
        // own box lies inside [X0, X0 + cols) x [Y0, Y0 + rows)
        int ox = 0, oy = 0;
        bool fast_l = false;
        if (lane < nwin) {
            int mx = 0x7fffffff, my = 0x7fffffff;
#pragma unroll
            for (int v = 0; v < kBandWaves; v++) {
                const int4 b = boxes[v * kBandMaxWin + lane];
                mx = min(mx, b.x);
                my = min(my, b.y);
            }
            // (inside the image, so that every row piece of the box is a legal load)
            ox = max(0, min(mx, width - kBandCols));
            oy = max(0, min(my, height - kBandRows));
            const int4 mine = boxes[wave * kBandMaxWin + lane];
            // (an empty box -- a window whose bound leaves the interpolatable range -- is SLOW: its samples are tested one by one)
            fast_l = band_ok && mine.x <= mine.z && mine.x >= ox && mine.z < ox + kBandCols && mine.y >= oy && mine.w < oy + kBandRows;
        }
        const unsigned long long fast_mask = __builtin_amdgcn_ballot_w64(fast_l);
        bool negative = false;
        float acc = 0.f;
        if (active) {
            iter++;
            // src/oc_icgn.cpp:251-255 for the affine warp: a corner sample outside the interpolatable range is a -1.f in the target
            // subset (decided before the sweep from the corners, icgn2d.hip kCornerTest); ICGN2D2 tests every sample
            if (DOF == 6 && corner_out && !(L.ablate & 16)) negative = true;
        }
        const bool compute = active && !(L.ablate & 8) && !(DOF == 6 && corner_out);
#pragma nounroll
        for (int w = 0; w < nwin; w++) {
            const int X0 = __builtin_amdgcn_readlane(ox, w), Y0 = __builtin_amdgcn_readlane(oy, w);
            const bool fast = ((fast_mask >> w) & 1ull) != 0;
            // ---- stage the band of window w: this wave's three rows of its plane
            if (band_ok && !(L.ablate & 2)) {
#pragma unroll
                for (int j = 0; j < kBandRows / 2; j++) {
                    const int row = 2 * j + st_row0;
                    const unsigned so = ((unsigned)(Y0 + row) * (unsigned)width + (unsigned)X0) << 4;
                    float4* dst = band + (st_plane * kBandPlane + row * kBandPitch);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(r_stage, (lds_vp)dst, 16, lane << 4, so, 0, 0);
                    if (lane < kBandCols - kWave)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_stage, (lds_vp)(dst + kWave), 16, lane << 4, so + kWave * 16, 0, 0);
                }
            }
            // a SLOW wave gathers from global memory, like icgn2d.hip -- before it waits for the band it does not read
            auto sweep_window = [&](auto fastc) {
                constexpr bool FAST = decltype(fastc)::value;
                const unsigned sbase = (unsigned)((Y0 * kBandPitch + X0) << 4);
#pragma unroll
                for (int g = 0; g < WP; g++) {
                    const int t = w * WP + g;
                    if (t < NT) {
                        const bool full = t < NF;
                        const bool valid = full || tail_valid;
                        const f2 lxy = tab_at(t);
                        const float xl = lxy.x - offx, yl = lxy.y - offy;
                        float ax, ay;
                        warp_point(xl, yl, ax, ay);
                        if (!full) {
                            ax = valid ? ax : (float)(X0 + 1);
                            ay = valid ? ay : (float)(Y0 + 1);
                        }
                        int xi = floor_to_int(ax), yi = floor_to_int(ay);
                        LutFetch f;
                        f.dx = __builtin_amdgcn_fractf(ax);
                        f.dy = __builtin_amdgcn_fractf(ay);
                        if constexpr (DOF == 12) {
                            const bool out = (unsigned)(xi - 1) > (unsigned)(width - 4) || (unsigned)(yi - 1) > (unsigned)(height - 4);
                            negative = negative || out;
                            xi = out ? X0 + 1 : xi;   // (a pixel inside the box and the image: the value is discarded)
                            yi = out ? Y0 + 1 : yi;
                        }
                        if constexpr (FAST) {
                            const char* __restrict__ q0 = reinterpret_cast<const char*>(band) + (((unsigned)(__umul24((unsigned)yi, kBandPitch) + (unsigned)xi) << 4) - sbase);
                            const float4* __restrict__ q = reinterpret_cast<const float4*>(q0);
                            f.c0 = q[0];
                            f.c1 = q[kBandPlane];
                            f.c2 = q[2 * kBandPlane];
                            f.c3 = q[3 * kBandPlane];
                        } else {
                            r_lut.load(f, (__umul24((unsigned)yi, (unsigned)width) + (unsigned)xi) << 4);
                        }
                        const float v = lut_value(f);
                        if (full) {
                            negative = negative || v < 0.f;
                            acc = acc + v;
                        } else {
                            negative = negative || (valid && v < 0.f);
                            acc = valid ? acc + v : acc;
                        }
                        ts.set(t, v);
                    }
                }
            };
            if (compute && !fast) sweep_window(std::false_type{});
            if (!(L.ablate & 1)) __syncthreads();  // the band of window w has landed (vmcnt) and is visible
            if (compute && fast) sweep_window(std::true_type{});
            if (w + 1 < nwin && !(L.ablate & 1)) __syncthreads();  // every wave has read it: it may be overwritten
        }
        bool done = false;
        if (active) {
            // src/oc_icgn.cpp:251-255
            if (!(L.ablate & 16) && wave_any(negative)) {
                if (lane == 0) poi[poi2d::ZNCC] = -3.f;
                done = true;
            }
        }
        if (active && !done) {
            // zeroMeanNorm of the target subset (src/oc_icgn.cpp:257)
            const float tmean = wave_allreduce_sum(acc) / fN;
            acc = 0.f;
#pragma unroll 6
            for (int t = 0; t < NF; t++) {
                const float d = ts.get(t) - tmean;
                acc = mad(d, d, acc);
            }
            if (NF < NT) {
                const float d = ts.get(NF) - tmean;
                acc = tail_valid ? mad(d, d, acc) : acc;
            }
            const float tar_norm = uni(sqrtf(wave_allreduce_sum(acc)));
            // error image, ZNSSD, numerator (src/oc_icgn.cpp:260-276)
            const float factor = ref_norm / tar_norm;
            float num[DOF];
#pragma unroll
            for (int i = 0; i < DOF; i++) num[i] = 0.f;
            float ssd = 0.f;
            f2 nA = mk2(0.f, 0.f), nB = nA;
            f2 np12[6];
#pragma unroll
            for (int q = 0; q < 6; q++) np12[q] = mk2(0.f, 0.f);
            passes_batched<kNumBatch>(
                NF, NT, tail_valid,
                [&](int t, bool valid) {
                    GradSample v;
                    const unsigned off = off_at(t);
                    v.gx = valid ? buf_f32(r_gx, off, goff) : 0.f;
                    v.gy = valid ? buf_f32(r_gy, off, goff) : 0.f;
                    v.ref = valid ? buf_f32(r_ref, off, roff) : 0.f;
                    return v;
                },
                [&](int t, bool valid, const GradSample& v) {
                    const float g_x = v.gx, g_y = v.gy;
                    const float tz = ts.get(t) - tmean;
                    const float rsv = v.ref - ref_mean;
                    const float e = mad(tz, factor, -rsv);
                    ssd = valid ? mad(e, e, ssd) : ssd;
                    if constexpr (DOF == 6) {
                        const f2 xy = tab_at(t) - mk2(offx, offy);
                        const f2 A = g_x * xy, B = g_y * xy;
                        const f2 mA = mad(A, e, nA), mB = mad(B, e, nB);
                        const float m0 = mad(g_x, e, num[0]), m3 = mad(g_y, e, num[3 % DOF]);
                        if (valid) {
                            nA = mA; nB = mB; num[0] = m0; num[3 % DOF] = m3;
                        }
                    } else {
                        const f2 lxy = tab_at(t) - mk2(offx, offy);
                        const float fxl = lxy.x, fyl = lxy.y;
                        const float xx = (fxl * fxl) * 0.5f, xy = fxl * fyl, yy = (fyl * fyl) * 0.5f;
                        const f2 m01 = mk2(1.f, fxl), m23 = mk2(fyl, xx), m45 = mk2(xy, yy);
                        const f2 sdp[6] = {g_x * m01, g_x * m23, g_x * m45, g_y * m01, g_y * m23, g_y * m45};
#pragma unroll
                        for (int q = 0; q < 6; q++) {
                            const f2 nv = mad(sdp[q], e, np12[q]);
                            np12[q] = valid ? nv : np12[q];
                        }
                    }
                });
            if constexpr (DOF == 6) {
                num[1] = nA.x; num[2] = nA.y; num[4 % DOF] = nB.x; num[5 % DOF] = nB.y;
            } else {
#pragma unroll
                for (int q = 0; q < 6; q++) {
                    num[(2 * q) % DOF] = np12[q].x;
                    num[(2 * q + 1) % DOF] = np12[q].y;
                }
            }
            float red[DOF + 1];
#pragma unroll
            for (int j = 0; j < DOF; j++) red[j] = num[j];
            red[DOF] = ssd;
            wave_allreduce_sum_multi<DOF + 1>(red, lane);
            znssd = uni(red[DOF]) / (ref_norm * ref_norm);
            // dp = H^-1 * numerator (src/oc_icgn.cpp:279-286)
            float dp[DOF];
            {
                // lane i < DOF: dp[i] = sum_j H^-1(i, j) * num[j], ascending j like the reference loop
                const float* __restrict__ hrow = hrow_area + wave * DOF * DOF + min(lane, DOF - 1) * DOF;
                float mine = 0.f;
#pragma unroll
                for (int j = 0; j < DOF; j++) mine += hrow[j] * red[j];
#pragma unroll
                for (int i = 0; i < DOF; i++) dp[i] = wave_bcast(mine, i);
            }
            // W <- W * (dW)^-1 ; p <- W (src/oc_icgn.cpp:287-293 / 828-834)
            const int rx2 = rx * rx, ry2 = ry * ry;
            if constexpr (DOF == 6) {
                float dW[9], dWi[9], Wn[9];
                set_warp_2d1(dW, dp[0], dp[1], dp[2], dp[3 % DOF], dp[4 % DOF], dp[5 % DOF]);
                inverse3(dW, dWi);
                mat_mul<3>(Wm, dWi, Wn);
#pragma unroll
                for (int i = 0; i < 9; i++) Wm[i] = uni(Wn[i]);
                cur[0] = Wm[2]; cur[1] = Wm[0] - 1.f; cur[2] = Wm[1];
                cur[6] = Wm[5]; cur[7] = Wm[3]; cur[8] = Wm[4] - 1.f;
                const float d = dp[0] * dp[0] + dp[1] * dp[1] * rx2 + dp[2] * dp[2] * ry2 + dp[3 % DOF] * dp[3 % DOF] +
                                dp[4 % DOF] * dp[4 % DOF] * rx2 + dp[5 % DOF] * dp[5 % DOF] * ry2;
                dp_norm = uni(sqrtf(d));
            } else {
                float dW[36];
                float dpf[12];
#pragma unroll
                for (int i = 0; i < 12; i++) dpf[i] = dp[i % DOF];
                set_warp_2d2(dW, dpf);
                float dcol[6], dinv[6];
#pragma unroll
                for (int i = 0; i < 6; i++) {
                    float c = 0.f;
#pragma unroll
                    for (int j = 0; j < 6; j++) c = lane == j ? dW[i * 6 + j] : c;
                    dcol[i] = c;
                }
                lu_inverse_lanes<6>(dcol, dinv, lane);
                float ncol[6];
#pragma unroll
                for (int i = 0; i < 6; i++) {
                    float v = wave_bcast(Wcol[i], 0) * dinv[0];
#pragma unroll
                    for (int k = 1; k < 6; k++) v = v + wave_bcast(Wcol[i], k) * dinv[k];
                    ncol[i] = v;
                }
#pragma unroll
                for (int i = 0; i < 6; i++) Wcol[i] = ncol[i];
                const float r30 = wave_bcast(Wcol[3], 0), r31 = wave_bcast(Wcol[3], 1), r32 = wave_bcast(Wcol[3], 2);
                const float r33 = wave_bcast(Wcol[3], 3), r34 = wave_bcast(Wcol[3], 4), r35 = wave_bcast(Wcol[3], 5);
                const float r40 = wave_bcast(Wcol[4], 0), r41 = wave_bcast(Wcol[4], 1), r42 = wave_bcast(Wcol[4], 2);
                const float r43 = wave_bcast(Wcol[4], 3), r44 = wave_bcast(Wcol[4], 4), r45 = wave_bcast(Wcol[4], 5);
                cur[0] = r35; cur[1] = r33 - 1.f; cur[2] = r34; cur[3] = r30 * 2.f; cur[4] = r31; cur[5] = r32 * 2.f;
                cur[6] = r45; cur[7] = r43; cur[8] = r44 - 1.f; cur[9] = r40 * 2.f; cur[10] = r41; cur[11] = r42 * 2.f;
                const int rxy2 = rx2 * ry2;
                constexpr int D = DOF;
                // src/oc_icgn.cpp:837-857 (integer-truncated weights are reference behaviour)
                const int rx4 = (int)(rx2 * rx2 * 0.25f), ry4 = (int)(ry2 * ry2 * 0.25f);
                const float d = dp[0] * dp[0] + dp[1] * dp[1] * rx2 + dp[2] * dp[2] * ry2 + dp[3 % D] * dp[3 % D] * rx4 +
                                dp[5 % D] * dp[5 % D] * ry4 + dp[4 % D] * dp[4 % D] * rxy2 + dp[6 % D] * dp[6 % D] +
                                dp[7 % D] * dp[7 % D] * rx2 + dp[8 % D] * dp[8 % D] * ry2 + dp[9 % D] * dp[9 % D] * rx4 +
                                dp[11 % D] * dp[11 % D] * ry4 + dp[10 % D] * dp[10 % D] * rxy2;
                dp_norm = uni(sqrtf(d));
            }
            if ((L.ablate & 16) ? iter >= 3 : !(iter < P.stop && dp_norm >= P.conv)) {
                // ---- outputs (src/oc_icgn.cpp:310-340; 2D2: 860-897)
                if (lane == 0) {
                    float zncc = 0.5f * (2 - znssd);
                    const float fiter = (float)iter;
                    if (dp_norm >= P.conv && fiter >= P.stop) zncc = -4.f;
                    float out_u = cur[0], out_v = cur[6];
                    if (isnan(zncc) || isnan(out_u) || isnan(out_v)) {
                        out_u = u_in;
                        out_v = v_in;
                        zncc = -5.f;
                    }
                    poi[poi2d::U] = out_u;
                    poi[poi2d::UX] = cur[1];
                    poi[poi2d::UY] = cur[2];
                    poi[poi2d::V] = out_v;
                    poi[poi2d::VX] = cur[7];
                    poi[poi2d::VY] = cur[8];
                    if constexpr (DOF == 12) {
                        poi[poi2d::UXX] = cur[3];
                        poi[poi2d::UXY] = cur[4];
                        poi[poi2d::UYY] = cur[5];
                        poi[poi2d::VXX] = cur[9];
                        poi[poi2d::VXY] = cur[10];
                        poi[poi2d::VYY] = cur[11];
                    }
                    poi[poi2d::U0] = u_in;
                    poi[poi2d::V0] = v_in;
                    poi[poi2d::ZNCC] = zncc;
                    poi[poi2d::ITER] = fiter;
                    poi[poi2d::CONV] = dp_norm;
                    poi[poi2d::SRX] = (float)rx;
                    poi[poi2d::SRY] = (float)ry;
                }
                done = true;
            }
        }
        // the next iteration's boxes (every wave read this iteration's before the window loop, i.e. before its barriers)
        if (active && !done) {
            publish_boxes(corner_out);
        } else {
            if (active) {
                if (lane == 0) atomicSub(const_cast<int*>(ctrl), 1);
                active = false;
            }
            publish_none();
        }
    }
}

template <int DOF, int TSB, int OFFS>
static hipError_t launch_band_t(const Icgn2dParams& p, float* pois, int stride_f, size_t count, int nt, bool xcd, hipStream_t stream) {
    const size_t lds = icgn2d_band_lds_bytes(nt, DOF);
    auto kern = icgn2d_band_kernel<DOF, TSB, OFFS>;
    static std::atomic<unsigned long long> attr_devices{0};
    int dev = 0;
    hipError_t derr = hipGetDevice(&dev);
    if (derr != hipSuccess) return derr;
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(attr_devices.load(std::memory_order_acquire) & bit)) {
        hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (err != hipSuccess) return err;
        attr_devices.fetch_or(bit, std::memory_order_release);
    }
    const size_t groups = (count + kBandWaves - 1) / kBandWaves;
    Icgn2dBandLaunch L;
    L.stride_f = stride_f;
    L.nt = nt;
    L.count = count;
    L.xcd_chunk = xcd ? (int)((groups + 7) / 8) : 0;
    L.ablate = std::getenv("OC_BAND_ABLATE") ? std::atoi(std::getenv("OC_BAND_ABLATE")) : 0;
    const size_t grid = xcd ? (size_t)L.xcd_chunk * 8 : groups;
    (void)hipGetLastError();
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * kBandWaves), lds, stream, p, pois, L);
    return hipGetLastError();
}

template <int DOF, int TSB>
static hipError_t launch_band_o(const Icgn2dParams& p, float* pois, int stride_f, size_t count, int nt, bool xcd, hipStream_t stream) {
    return p.offsets ? launch_band_t<DOF, TSB, 1>(p, pois, stride_f, count, nt, xcd, stream)
                     : launch_band_t<DOF, TSB, 0>(p, pois, stride_f, count, nt, xcd, stream);
}

// passes the register vector holds: 16 + TSB; ICGN2D1 up to 20 (35 x 35), ICGN2D2 up to 28 (42 x 42)
constexpr int kBandNtMax1 = 20, kBandNtMax2 = 28;

hipError_t launch_icgn2d1_band(const Icgn2dParams& p, float* pois, int stride_f, size_t count, bool xcd, hipStream_t stream) {
    if (count == 0) return hipSuccess;
    const int nt = ((2 * p.rx + 1) * (2 * p.ry + 1) + 63) / 64;
    if (p.self_adaptive || nt > kBandNtMax1) return hipErrorInvalidValue;
    return nt <= 18 ? launch_band_o<6, 2>(p, pois, stride_f, count, nt, xcd, stream) : launch_band_o<6, 4>(p, pois, stride_f, count, nt, xcd, stream);
}

hipError_t launch_icgn2d2_band(const Icgn2dParams& p, float* pois, int stride_f, size_t count, bool xcd, hipStream_t stream) {
    if (count == 0) return hipSuccess;
    const int nt = ((2 * p.rx + 1) * (2 * p.ry + 1) + 63) / 64;
    if (p.self_adaptive || nt > kBandNtMax2) return hipErrorInvalidValue;
    return nt <= 24 ? launch_band_o<12, 8>(p, pois, stride_f, count, nt, xcd, stream) : launch_band_o<12, 12>(p, pois, stride_f, count, nt, xcd, stream);
}

}  // namespace OC_ARITH

#if !OC_FMA
bool icgn2d_band_supported(int dof, int rx, int ry) {
    return (2 * rx + 1) * (2 * ry + 1) <= (dof == 6 ? sep::kBandNtMax1 : sep::kBandNtMax2) * kWave;
}
hipError_t launch_icgn2d1_band(const Icgn2dParams& p, float* pois, int stride_f, size_t count, bool xcd, hipStream_t stream) {
    return p.arith_fma ? fma::launch_icgn2d1_band(p, pois, stride_f, count, xcd, stream) : sep::launch_icgn2d1_band(p, pois, stride_f, count, xcd, stream);
}
hipError_t launch_icgn2d2_band(const Icgn2dParams& p, float* pois, int stride_f, size_t count, bool xcd, hipStream_t stream) {
    return p.arith_fma ? fma::launch_icgn2d2_band(p, pois, stride_f, count, xcd, stream) : sep::launch_icgn2d2_band(p, pois, stride_f, count, xcd, stream);
}
#endif

}  // namespace ochip
