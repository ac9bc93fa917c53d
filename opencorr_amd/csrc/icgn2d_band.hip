// icgn2d_band.hip -- ICGN2D1 / ICGN2D2 with the bicubic coefficient BAND of a workgroup staged in LDS (round 6).
//
// Replaces ICGN2D1::compute(POI2D*) (src/oc_icgn.cpp:144-341) / ICGN2D2::compute(POI2D*) (:685-898) for a whole queue, like
// icgn2d.hip, whose arithmetic it repeats operation for operation (same bits: oracle OC_ORDER_LANES / _FMA).  What differs is
// where the 64 bytes per sample of BicubicBspline::compute (src/oc_cubic_bspline.cpp:134-181) come from:
//   * icgn2d.hip gathers them per lane from the planar table in global memory (four buffer_load_b128 through the CU's L1);
//   * here the eight waves of a workgroup -- eight consecutive POIs of the visiting order, normally neighbours in a row of the
//     POI grid -- sweep their subsets in lockstep WINDOWS of kBandWP passes (128 samples = ~4 subset rows).  In a window all of
//     them touch ONE band of the table: a few rows x (7 x pitch + subset width) pixels x 4 planes x 16 B.  The band is bounded
//     per wave from the images of the window's index rectangle under the POI's warp (affine: the four corners bound it exactly,
//     every float operation being monotone), the waves' boxes are united (in wave order, while the union fits the LDS area),
//     the union is loaded ONCE per workgroup with coalesced 1 KB buffer loads, and every sample inside it reads its four
//     float4 with ds_read_b128.  A sample outside the staged box (a POI of the next grid row in the same workgroup, a large
//     deformation gradient, the quadratic warp of ICGN2D2 bulging past its corners) takes the global gather per LANE -- the box
//     is a performance decision, never a correctness one.
//   * the warped target subset lives in REGISTERS (ts[NTMAX], every pass loop unrolled): the LDS of a workgroup is the
//     coordinate table, the band and 1 KB of boxes -- no per-wave arrays -- so three workgroups per CU stay resident.
//   * all eight waves stay until the workgroup's last POI has converged: a wave that is done keeps loading its share of the
//     band (and passes every barrier), it only stops computing.
// Launch shape: 512 threads, one POI per wave, XCD-contiguous groups, tile-ordered queue (poi_order.hip).  Not available for
// self-adaptive radii (one subset size per launch), IC-LM, or subsets above NTMAX passes: icgn2d.hip serves those.
#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "dic2d_device.h"
#include "oc_kernels.h"

namespace ochip {
namespace OC_ARITH {

constexpr int kBandWaves = 8;   // POIs per workgroup
constexpr int kBandWP = 2;      // passes per window
constexpr int kBandCapX = 96;   // pixels per staged row
constexpr int kBandCapR = 6;    // staged rows

// what a pass of the Hessian sweep / the numerator pass fetches for one sample
struct GradSample {
    float gx, gy, ref = 0.f;
};

struct Icgn2dBandLaunch {
    int stride_f;
    int nt;                    // ceil(N / 64)
    int xcd_chunk;
    unsigned long long count;
    int ablate;                // timing experiments (environment OC_BAND_ABLATE; results are NOT valid except for 0, 4, 16): 1 = no window
                               // barriers, 2 = no staging loads, 4 = nothing staged (every sample takes the global gather: the launch
                               // shape alone), 8 = no polynomial sweeps, 16 = exactly three iterations for every POI
};

__device__ __forceinline__ int dpp_quad_xor1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true); }
__device__ __forceinline__ int dpp_quad_xor2(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true); }
__device__ __forceinline__ float4 buf_f32x4so(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}

constexpr size_t icgn2d_band_lds_bytes(int nt, int ntmax) {
    return (size_t)3 * nt * kWave * 4 + (size_t)4 * kBandCapR * kBandCapX * 16 + (size_t)kBandWaves * ((ntmax + kBandWP - 1) / kBandWP) * 16 + 16;
}

template <int DOF, int NTMAX, int OFFS>
__global__ __launch_bounds__(64 * kBandWaves, DOF == 6 ? 6 : 4) void icgn2d_band_kernel(Icgn2dParams P, float* __restrict__ pois,
                                                                                       Icgn2dBandLaunch L) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NH = DOF * (DOF + 1) / 2;
    constexpr int WP = kBandWP, NWMAX = (NTMAX + WP - 1) / WP;
    constexpr int CAPX = kBandCapX, CAPR = kBandCapR, PLANE = CAPR * CAPX;
    constexpr bool COOP = DOF == 6;
    static_assert(NWMAX * 4 <= kWave, "one lane per (window, corner)");
    constexpr int kSetupBatch = 6, kHessBatch = 6, kNumBatch = DOF == 6 ? 6 : 4;

    const int NTA = L.nt;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f2* __restrict__ tab_xy = reinterpret_cast<f2*>(lds);
    unsigned* __restrict__ tab_off = reinterpret_cast<unsigned*>(lds + 2 * NTA * kWave);
    float4* __restrict__ band = reinterpret_cast<float4*>(lds + 3 * NTA * kWave);
    int4* __restrict__ boxes = reinterpret_cast<int4*>(band + 4 * PLANE);
    volatile int* ctrl = reinterpret_cast<volatile int*>(boxes + kBandWaves * NWMAX);
    float* __restrict__ coop_area = reinterpret_cast<float*>(band);  // the band is idle until the first sweep

    const int height = P.height, width = P.width;
    const int rx = P.rx, ry = P.ry;
    const int W = 2 * rx + 1, N = W * (2 * ry + 1);
    const float fN = (float)N;
    const int NT = (N + kWave - 1) / kWave;
    const int NF = N / kWave;
    const bool tail_valid = (NF * kWave + lane) < N;
    const int nwin = (NT + WP - 1) / WP;
    {
        const unsigned w4t = (unsigned)width * 4u;
        for (int s = threadIdx.x; s < NTA * kWave; s += kWave * kBandWaves) {
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
    // the plane this wave stages (constant for its life): waves 0-3 the even rows of a round, 4-7 the odd ones
    const int st_plane = wave & 3, st_row0 = wave >> 2;
    const __amdgpu_buffer_rsrc_t r_stage = st_plane == 0 ? r_lut.p0 : (st_plane == 1 ? r_lut.p1 : (st_plane == 2 ? r_lut.p2 : r_lut.p3));

    // every pass loop is unrolled over NTMAX with wave-uniform guards: ts[] must stay in registers
    float ts[NTMAX];
#pragma unroll
    for (int t = 0; t < NTMAX; t++) ts[t] = 0.f;
    // passes [0, NF) full, pass NF (if NF < NT) partial; loads of up to B passes in flight, uses in pass order
    auto passes = [&](auto bconst, auto&& load, auto&& use) {
        constexpr int B = decltype(bconst)::value;
#pragma unroll
        for (int t0 = 0; t0 < NTMAX; t0 += B) {
            if (t0 < NT) {
                __builtin_amdgcn_sched_barrier(0);  // keep the batches apart: the unrolled loop must not pile their loads up
                decltype(load(0, true)) v[B];
#pragma unroll
                for (int u = 0; u < B; u++) {
                    const int t = t0 + u;
                    if (t < NTMAX) {
                        if (t < NF) v[u] = load(t, true);
                        else if (t < NT) v[u] = load(t, tail_valid);
                    }
                }
#pragma unroll
                for (int u = 0; u < B; u++) {
                    const int t = t0 + u;
                    if (t < NTMAX) {
                        if (t < NF) use(t, true, v[u]);
                        else if (t < NT) use(t, tail_valid, v[u]);
                    }
                }
            }
        }
    };

    float ref_norm = 0.f, ref_mean = 0.f;
    float hinv_col[DOF];
    float hinv_row[DOF];
#pragma unroll
    for (int j = 0; j < DOF; j++) hinv_row[j] = hinv_col[j] = 0.f;
    float h[NH];
#pragma unroll
    for (int i = 0; i < NH; i++) h[i] = 0.f;
    if (active) {
        // ---- reference subset, zero-mean + norm (src/oc_icgn.cpp:174-176, src/oc_subset.cpp:39-53)
        float acc = 0.f;
        passes(std::integral_constant<int, kSetupBatch>{},
               [&](int t, bool valid) { return valid ? buf_f32(r_ref, off_at(t), roff) : 0.f; },
               [&](int t, bool valid, float v) {
                   acc = valid ? acc + v : acc;
                   ts[t] = v;
               });
        const float mean = wave_allreduce_sum(acc) / fN;
        ref_mean = uni(mean);
        acc = 0.f;
#pragma unroll
        for (int t = 0; t < NTMAX; t++) {
            if (t < NF) {
                const float d = ts[t] - mean;
                acc = mad(d, d, acc);
            } else if (t < NT) {
                const float d = ts[t] - mean;
                acc = tail_valid ? mad(d, d, acc) : acc;
            }
        }
        ref_norm = uni(sqrtf(wave_allreduce_sum(acc)));
#pragma unroll
        for (int t = 0; t < NTMAX; t++) ts[t] = 0.f;  // (dead until the first sweep: say so to the register allocator)

        // ---- steepest-descent image + Hessian (src/oc_icgn.cpp:179-207; 2D2: 716-756)
        if constexpr (DOF == 6) {
            f2 hAA = mk2(0.f, 0.f), hBB = hAA, hAB = hAA, hAs = hAA, hxA = hAA, hyA = hAA, hxB = hAA, hyB = hAA;
            float h00 = 0.f, h33 = 0.f, h30 = 0.f, h21 = 0.f, h54 = 0.f;
            passes(std::integral_constant<int, kHessBatch>{},
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
            h[10 % NH] = hxB.x; h[11 % NH] = hAB.x; h[12 % NH] = hAs.y; h[13 % NH] = hyB.x; h[14 % NH] = hBB.x;
            h[15 % NH] = hxB.y; h[16 % NH] = hAs.x; h[17 % NH] = hAB.y; h[18 % NH] = hyB.y; h[19 % NH] = h54; h[20 % NH] = hBB.y;
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
            // (a rolled loop: the 78 running sums leave no room for an unrolled body; the pass index is not a register index here)
            passes_prefetched(NF, NT, tail_valid, fetch, sample);
            int k = 0;
#pragma unroll
            for (int r = 0; r < 12; r++)
#pragma unroll
                for (int c = 0; c <= r; c++, k++)
                    h[k % NH] = (c == r && (r & 1) == 0) ? hd[r] : ((c & 1) ? hp[r][c / 2].y : hp[r][c / 2].x);
        }
    }
    // ---- inverse of the Hessian (:210 / :759)
    if constexpr (COOP) {
        if (active) wave_reduce_sum_multi_to_lds<NH>(h, lane, coop_area + wave * 64);
        else if (lane < 24) coop_area[wave * 64 + lane] = 0.f;
        __syncthreads();
        if (wave == 0) coop_inverse6_x8(coop_area, lane);
        __syncthreads();
        const float* __restrict__ row = coop_area + wave * 64 + 24 + min(lane, DOF - 1) * DOF;
#pragma unroll
        for (int j = 0; j < DOF; j++) hinv_row[j] = lane < DOF ? row[j] : 0.f;
    } else {
        if (active) {
            float col[DOF];
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
#pragma unroll
            for (int i = 0; i < DOF; i++)
#pragma unroll
                for (int j = 0; j < DOF; j++) {
                    const float v = wave_bcast(hinv_col[i], j);
                    hinv_row[j] = lane == i ? v : hinv_row[j];
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
    // lane 4 w + c: corner c of window w's index rectangle (all columns x the rows the window's samples lie in)
    float cxl, cyl;
    const int bw_l = lane >> 2;
    {
        const int s0 = kWave * WP * bw_l, s1 = min(kWave * WP * (bw_l + 1), N) - 1;
        const int ra = s0 / W, rb = max(s1, 0) / W;
        cxl = (float)((lane & 1) ? rx : -rx) - offx;
        cyl = (float)(((lane & 2) ? rb : ra) - ry) - offy;
    }
    if (active && lane == 0) atomicAdd(const_cast<int*>(ctrl), 1);

    int iter = 0;
    float dp_norm = 0.f, znssd = 0.f;
    float cur[12];
#pragma unroll
    for (int i = 0; i < 12; i++) cur[i] = 0.f;

#pragma nounroll
    for (;;) {
        __syncthreads();  // B0: the finished waves of the previous iteration have reported; the band is free
        if (ctrl[0] <= 0) return;
        if (active) {
            iter++;
            if constexpr (DOF == 12) {
#pragma unroll
                for (int k = 0; k < 6; k++) {
                    row3[k] = wave_bcast(Wcol[3], k);
                    row4[k] = wave_bcast(Wcol[4], k);
                }
            }
            // boxes of all windows at once: the corner's target pixel, clamped to the interpolatable range, min / max over the quad
            float ax, ay;
            warp_point(cxl, cyl, ax, ay);
            int xi = floor_to_int(ax), yi = floor_to_int(ay);
            xi = min(max(xi, 1), width - 3);
            yi = min(max(yi, 1), height - 3);
            int x0 = min(xi, dpp_quad_xor1(xi)), x1 = max(xi, dpp_quad_xor1(xi));
            int y0 = min(yi, dpp_quad_xor1(yi)), y1 = max(yi, dpp_quad_xor1(yi));
            x0 = min(x0, dpp_quad_xor2(x0)); x1 = max(x1, dpp_quad_xor2(x1));
            y0 = min(y0, dpp_quad_xor2(y0)); y1 = max(y1, dpp_quad_xor2(y1));
            if ((lane & 3) == 0 && bw_l < nwin) boxes[wave * NWMAX + bw_l] = make_int4(x0, y0, x1, y1);
        } else {
            if (lane < nwin) boxes[wave * NWMAX + lane] = make_int4(1, 1, 0, 0);
        }
        __syncthreads();  // B1
        // lane w: union of the waves' boxes of window w, in wave order, while it fits the staged area
        int ux0 = 0x7fffffff, uy0 = 0x7fffffff, ux1 = -0x7fffffff, uy1 = -0x7fffffff;
        if (lane < nwin) {
#pragma unroll
            for (int v = 0; v < kBandWaves; v++) {
                const int4 b = boxes[v * NWMAX + lane];
                const int nx0 = min(ux0, b.x), ny0 = min(uy0, b.y), nx1 = max(ux1, b.z), ny1 = max(uy1, b.w);
                const bool ok = b.x <= b.z && (nx1 - nx0) < CAPX && (ny1 - ny0) < CAPR;
                ux0 = ok ? nx0 : ux0; uy0 = ok ? ny0 : uy0; ux1 = ok ? nx1 : ux1; uy1 = ok ? ny1 : uy1;
            }
        }
        const bool none = ux1 < ux0 || (L.ablate & 4);
        const int ubw = none ? 0 : ux1 - ux0 + 1, ubh = none ? 0 : uy1 - uy0 + 1;

        bool negative = false;
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < NTMAX; t++) ts[t] = 0.f;  // (the previous iteration's subset is dead)
#pragma unroll
        for (int w = 0; w < NWMAX; w++) {
            if (w < nwin) {
                const int X0 = __builtin_amdgcn_readlane(ux0, w), Y0 = __builtin_amdgcn_readlane(uy0, w);
                const int BW = __builtin_amdgcn_readlane(ubw, w), BH = __builtin_amdgcn_readlane(ubh, w);
                // ---- stage the band: wave (row parity, plane) x lanes along the row
#pragma unroll
                for (int j = 0; j < CAPR / 2; j++) {
                    const int row = 2 * j + st_row0;
                    if (row < BH && !(L.ablate & 2)) {
                        const unsigned so = ((unsigned)(Y0 + row) * (unsigned)width + (unsigned)X0) << 4;
                        float4 sv[2];
#pragma unroll
                        for (int c = 0; c < 2; c++) {
                            const int bx = lane + 64 * c;
                            if (bx < BW) sv[c] = buf_f32x4so(r_stage, (unsigned)bx << 4, so);
                        }
#pragma unroll
                        for (int c = 0; c < 2; c++) {
                            const int bx = lane + 64 * c;
                            if (bx < BW) band[(st_plane * CAPR + row) * CAPX + bx] = sv[c];
                        }
                    }
                }
                if (!(L.ablate & 1)) __syncthreads();  // B2
                if (active && !(L.ablate & 8)) {
#pragma unroll
                    for (int g = 0; g < WP; g++) {
                        const int t = w * WP + g;
                        if (t < NTMAX && t < NT) {
                            LutFetch f;
                            const bool full = t < NF;
                            const bool valid = full || tail_valid;
                            const f2 lxy = tab_at(t);
                            const float xl = lxy.x - offx, yl = lxy.y - offy;
                            float ax, ay;
                            warp_point(xl, yl, ax, ay);
                            if (!full) {
                                ax = valid ? ax : 1.f;
                                ay = valid ? ay : 1.f;
                            }
                            const int xi = floor_to_int(ax), yi = floor_to_int(ay);
                            const bool out = (unsigned)(xi - 1) > (unsigned)(width - 4) || (unsigned)(yi - 1) > (unsigned)(height - 4);
                            f.dx = __builtin_amdgcn_fractf(ax);
                            f.dy = __builtin_amdgcn_fractf(ay);
                            const unsigned bx = (unsigned)(xi - X0), by = (unsigned)(yi - Y0);
                            if (bx < (unsigned)BW && by < (unsigned)BH) {  // (the box lies inside the interpolatable range)
                                const float4* __restrict__ q = band + (by * CAPX + bx);
                                f.c0 = q[0];
                                f.c1 = q[PLANE];
                                f.c2 = q[2 * PLANE];
                                f.c3 = q[3 * PLANE];
                            } else {
                                r_lut.load(f, out ? 0u : (__umul24((unsigned)yi, (unsigned)width) + (unsigned)xi) << 4);
                            }
                            negative = negative || out;
                            const float v = lut_value(f);
                            if (full) {
                                negative = negative || v < 0.f;
                                acc = acc + v;
                            } else {
                                negative = negative || (valid && v < 0.f);
                                acc = valid ? acc + v : acc;
                            }
                            ts[t] = v;
                        }
                    }
                }
                if (w + 1 < nwin && !(L.ablate & 1)) __syncthreads();  // B3: the band may be overwritten
            }
        }
        if (active) {
            // src/oc_icgn.cpp:251-255
            if (!(L.ablate & 16) && wave_any(negative)) {
                if (lane == 0) {
                    poi[poi2d::ZNCC] = -3.f;
                    atomicSub(const_cast<int*>(ctrl), 1);
                }
                active = false;
            }
        }
        if (active) {
            // zeroMeanNorm of the target subset (src/oc_icgn.cpp:257)
            const float tmean = wave_allreduce_sum(acc) / fN;
            acc = 0.f;
#pragma unroll
            for (int t = 0; t < NTMAX; t++) {
                if (t < NF) {
                    const float d = ts[t] - tmean;
                    acc = mad(d, d, acc);
                } else if (t < NT) {
                    const float d = ts[t] - tmean;
                    acc = tail_valid ? mad(d, d, acc) : acc;
                }
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
            passes(std::integral_constant<int, kNumBatch>{},
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
                       const float tz = ts[t] - tmean;
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
                float mine = 0.f;
#pragma unroll
                for (int j = 0; j < DOF; j++) mine += hinv_row[j] * red[j];
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
                    atomicSub(const_cast<int*>(ctrl), 1);
                }
                active = false;
            }
        }
    }
}

template <int DOF, int NTMAX, int OFFS>
static hipError_t launch_band_t(const Icgn2dParams& p, float* pois, int stride_f, size_t count, int nt, bool xcd, hipStream_t stream) {
    const size_t lds = icgn2d_band_lds_bytes(nt, NTMAX);
    auto kern = icgn2d_band_kernel<DOF, NTMAX, OFFS>;
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

// passes the two instantiations hold in registers (33 x 33 = 18 passes, 35 x 34 = 19; 41 x 41 = 27, 42 x 42 = 28)
constexpr int kBandNtMax1 = 19, kBandNtMax2 = 28;

hipError_t launch_icgn2d1_band(const Icgn2dParams& p, float* pois, int stride_f, size_t count, bool xcd, hipStream_t stream) {
    if (count == 0) return hipSuccess;
    const int nt = ((2 * p.rx + 1) * (2 * p.ry + 1) + 63) / 64;
    if (p.self_adaptive || nt > kBandNtMax1) return hipErrorInvalidValue;
    return p.offsets ? launch_band_t<6, kBandNtMax1, 1>(p, pois, stride_f, count, nt, xcd, stream)
                     : launch_band_t<6, kBandNtMax1, 0>(p, pois, stride_f, count, nt, xcd, stream);
}

hipError_t launch_icgn2d2_band(const Icgn2dParams& p, float* pois, int stride_f, size_t count, bool xcd, hipStream_t stream) {
    if (count == 0) return hipSuccess;
    const int nt = ((2 * p.rx + 1) * (2 * p.ry + 1) + 63) / 64;
    if (p.self_adaptive || nt > kBandNtMax2) return hipErrorInvalidValue;
    return p.offsets ? launch_band_t<12, kBandNtMax2, 1>(p, pois, stride_f, count, nt, xcd, stream)
                     : launch_band_t<12, kBandNtMax2, 0>(p, pois, stride_f, count, nt, xcd, stream);
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
