// icgn3d.hip -- ICGN3D1 (inverse-compositional Gauss-Newton for DVC) on gfx950.
//
// Replaces ICGN3D1::compute(POI3D*) (src/oc_icgn.cpp:1270-1490) for a whole POI queue (:1492-1500).
//
// Mapping: ONE 512-thread workgroup (8 wavefronts) per POI, two workgroups per CU, persistent over
// the queue.  Sample s = (i*SY + j)*SX + k of the (2rz+1)(2ry+1)(2rx+1) subvolume is owned by
// thread s % 512.
//
// The 64-tap tricubic gather (src/oc_cubic_bspline.cpp:353-405) is what bounds the kernel.  As
// 16 unaligned 16-byte loads per sample it keeps the texture path busy while neighbouring lanes
// re-fetch three quarters of each other's taps.  Instead the workgroup sweeps the subvolume in
// passes of 512 consecutive samples and, per pass, stages the box of B-spline coefficients those
// samples can touch into LDS with coalesced row loads (the box is the image of the pass's index
// box under the affine warp -- monotone in every index even in floating point, so evaluating the
// 8 corners with the samples' own expression bounds it exactly); the 64 taps are then LDS reads.
// A pass whose box does not fit (large rotations / strains) falls back to global loads.
//
// The warped-target subvolume (the only per-sample state produced inside the Gauss-Newton loop)
// goes to a per-workgroup global scratch slot (written once, read twice per iteration, coalesced;
// L2 resident); the zero-mean reference subvolume and the three gradients are re-read from the
// volumes each iteration (x-contiguous, coalesced).
// Reductions: per-thread partial sums in increasing s, xor butterfly inside each wave
// (offsets 1..32), then a balanced tree over the 8 wave sums in wave order -- the same
// association as the oracle's OC_ORDER_LANES with lanes = 512.
#include <cstdio>
#include <cstdlib>

#include "icgn3d_device.h"

// Phase ablation for tools/ablate_icgn3d.sh (cdna_hip_programming.md: "ablate, then match"): a build with
// -DOC_ABLATE=<mask> leaves a phase out and runs a fixed number of iterations, so that timing differences price the
// phases.  Results of such a build are garbage; the product is always built with the mask at 0.
//   1 = no coefficient staging (global -> LDS)   2 = no tap evaluation   4 = no Hessian sweep   8 = no numerator sweep
//   32 = taps addressed with a row pitch of 33 floats (wrong rows, but no LDS bank conflicts: what a conflict-free layout is worth)
//   64 = nothing left out, only the three forced iterations: the partner of the masks above
//   16 = timeline: results stay valid, and the six strain floats of every POI record receive the shader-clock
//        kilocycles its workgroup spent in  reference stats | Hessian sweep + reduction | LU inverse | warped-subvolume
//        sweeps (boxes, staging, taps) | mean / norm / numerator sweeps + reductions | solve + warp update
#ifndef OC_ABLATE
#define OC_ABLATE 0
#endif

namespace ochip {
// (kernel and launcher live in ochip::sep or ochip::fma: this file is compiled once per arithmetic mode, oc_device.h)
namespace OC_ARITH {

// Hessian rows [R0, R1): sums of sd[r]*sd[c], c <= r, over all samples (src/oc_icgn.cpp:1299-1337),
// block-reduced and filed into the symmetric matrix A (LDS).
template <int R0, int R1>
__device__ __forceinline__ void hessian_rows(const Icgn3dParams& P, int tid, int wave, int lane, int SX, int SY, int N,
                                             int rx, int ry, int rz, int cx, int cy, int cz, int DX, int DY, float* red,
                                             float* __restrict__ A) {
    constexpr int NE = (R1 * (R1 + 1) - R0 * (R0 + 1)) / 2;
    // The running sums H(r,c) += sd[r]*sd[c] as packed-fp32 pairs over adjacent columns (v_pk_mul_f32 /
    // v_pk_add_f32: two IEEE operations per issue slot, each rounded on its own -- every H(r,c) receives the same
    // products in the same order).  sd = g_a * (1, x | y, z) for a = x, y, z: six pairs; row r takes the column
    // pairs (0,1), (2,3), ... below the diagonal and, when r is even, the single diagonal term.
    f2 hp[12][6];
    float hd[12];
#pragma unroll
    for (int r = 0; r < 12; r++) {
        hd[r] = 0.f;
#pragma unroll
        for (int q = 0; q < 6; q++) hp[r][q] = mk2(0.f, 0.f);
    }
    Walk3 w(tid, SX, SY, 0, DX, DY);
    // gradient voxel of sample (i, j, k): (cz - rz + i, cy - ry + j, cx - rx + k) -- integer arithmetic, so the
    // subvolume is one box and the walk's offset addresses it
    const size_t gbase = ((size_t)(cz - rz) * DY + (cy - ry)) * DX + (cx - rx);
    const float* __restrict__ pgx = P.gx + gbase;
    const float* __restrict__ pgy = P.gy + gbase;
    const float* __restrict__ pgz = P.gz + gbase;
    struct G3 {
        float x, y, z;
    };
    const int cnt = N > tid ? (N - tid + kBlock3d - 1) / kBlock3d : 0;
    sweep_batched<4>(
        w, rx, ry, rz, cnt, [&](const WalkPoint& q, int) { return G3{pgx[q.off], pgy[q.off], pgz[q.off]}; },
        [&](const WalkPoint& q, const G3& g, int) {
            const float g_x = g.x, g_y = g.y, g_z = g.z;
            const f2 m01 = mk2(1.f, q.x), m23 = mk2(q.y, q.z);  // g * 1.f is exact
            const f2 sdp[6] = {g_x * m01, g_x * m23, g_y * m01, g_y * m23, g_z * m01, g_z * m23};
#pragma unroll
            for (int r = R0; r < R1; r++) {
                const float sr = (r & 1) ? sdp[r / 2].y : sdp[r / 2].x;
#pragma unroll
                for (int q2 = 0; q2 < (r + 1) / 2; q2++) hp[r][q2] = mad(sr, sdp[q2], hp[r][q2]);
                if ((r & 1) == 0) hd[r] = mad(sr, sr, hd[r]);
            }
        });
    float h[NE];
    {
        int t = 0;
#pragma unroll
        for (int r = R0; r < R1; r++)
#pragma unroll
            for (int c = 0; c <= r; c++, t++)
                h[t] = (c == r && (r & 1) == 0) ? hd[r] : ((c & 1) ? hp[r][c / 2].y : hp[r][c / 2].x);
    }
    constexpr int NCH = (NE + kRedChunk - 1) / kRedChunk;
#pragma unroll
    for (int ch = 0; ch < NCH; ch++) {
        float part[kRedChunk];
#pragma unroll
        for (int q = 0; q < kRedChunk; q++) part[q] = (ch * kRedChunk + q < NE) ? h[(ch * kRedChunk + q) % NE] : 0.f;
        block_allreduce<kRedChunk>(part, red, wave, lane);
#pragma unroll
        for (int q = 0; q < kRedChunk; q++)
            if (ch * kRedChunk + q < NE) h[(ch * kRedChunk + q) % NE] = part[q];
    }
    // every thread holds the totals: one lane files them into the symmetric matrix in LDS (row-major 12 x 12), where the
    // LU of wave 0 picks them up (same wave: LDS operations of a wave execute in order)
    if (tid == 0) {
        int t = 0;
#pragma unroll
        for (int r = R0; r < R1; r++)
#pragma unroll
            for (int c = 0; c <= r; c++, t++) {
                A[r * 12 + c] = h[t];
                A[c * 12 + r] = h[t];
            }
    }
}

// PX: row pitch of the staged coefficient box in LDS (0 = the box's own width, decided per pass)
template <int PX>
__global__ __launch_bounds__(kBlock3d, 4) void icgn3d1_kernel(Icgn3dParams P, float* __restrict__ pois, int stride_f,
                                                           unsigned long long count) {
    __shared__ __attribute__((aligned(16))) float lds[kRedChunk * kWaves3d + 12 * kWave + kWinCap + 6 * kBoxSlots];
    float* red = lds;                               // kRedChunk * 8 floats
    float* lds_hinv = lds + kRedChunk * kWaves3d;   // 12 x 64 floats: column j of H^-1 in lane j (parked between solves)
    float* win = lds_hinv + 12 * kWave;             // staged coefficient box of the current pass
    int* boxes = reinterpret_cast<int*>(win + kWinCap);  // origin + extent of the box of each pass (kBoxSlots x 6)
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1), wave = tid >> 6;
    const int rx = P.rx, ry = P.ry, rz = P.rz, DX = P.dx, DY = P.dy, DZ = P.dz;
    const int SX = 2 * rx + 1, SY = 2 * ry + 1, SZ = 2 * rz + 1;
    const int N = SX * SY * SZ;
    const float fN = (float)N;
    float* __restrict__ ts = P.scratch + (size_t)blockIdx.x * N;
    const int plane = SX * SY;

    // Workgroups are dealt round-robin to the 8 XCDs (each with its own L2): every XCD walks a
    // contiguous eighth of the queue, its workgroups interleaved over it, so that subvolumes in
    // flight behind one L2 overlap (neighbouring POIs share most of their voxels).
    const unsigned long long xcd_chunk = (count + 7) / 8, xcd_lo = (blockIdx.x & 7u) * xcd_chunk;
    const unsigned long long xcd_hi = min(count, xcd_lo + xcd_chunk);
    for (unsigned long long idx = xcd_lo + (blockIdx.x >> 3); idx < xcd_hi; idx += gridDim.x >> 3) {
        // the k-th solve of the launch takes POI perm[k] (a locality schedule, poi_order.hip) or simply POI k
        float* poi = pois + (P.perm ? (unsigned long long)P.perm[idx] : idx) * (unsigned long long)stride_f;
        // every thread reads the same record: keep it in SGPRs (the hot loops need the VGPRs)
        const float px = uni3(poi[poi3d::X]), py = uni3(poi[poi3d::Y]), pz = uni3(poi[poi3d::Z]);
        float init[12];
#pragma unroll
        for (int i = 0; i < 12; i++) init[i] = uni3(poi[poi3d::P + i]);
        const float zncc_in = uni3(poi[poi3d::ZNCC]);
        __syncthreads();  // everyone has read the record before anyone may overwrite it

        // guard, src/oc_icgn.cpp:1279-1286
        if ((px - rx) < 0 || (py - ry) < 0 || (pz - rz) < 0 || (px + rx) > (DX - 1) || (py + ry) > (DY - 1) ||
            (pz + rz) > (DZ - 1) || fabsf(init[0]) >= DX || fabsf(init[4]) >= DY || fabsf(init[8]) >= DZ ||
            zncc_in < 0 || isnan(init[0]) || isnan(init[4]) || isnan(init[8])) {
            if (tid == 0) poi[poi3d::ZNCC] = zncc_in >= 0 ? -3.f : zncc_in;
            continue;
        }

        [[maybe_unused]] unsigned long long tl_mark = __builtin_readcyclecounter();
        [[maybe_unused]] unsigned long long tl[6] = {0, 0, 0, 0, 0, 0};
        auto lap = [&](int slot) {
            if constexpr (OC_ABLATE & 16) {
                __builtin_amdgcn_sched_barrier(0);  // keep the surrounding arithmetic on its side of the stamp
                const unsigned long long now = __builtin_readcyclecounter();
                tl[slot] += now - tl_mark;
                tl_mark = now;
            }
        };
        // ---- reference subvolume mean + norm (src/oc_subset.cpp:89-135)
        const float sxf = px - rx, syf = py - ry, szf = pz - rz;
        // Subset3D::fill reads voxel (int(start.z + i), int(start.y + j), int(start.x + k)) (src/oc_subset.cpp:89-103):
        // float additions, truncated per element.  Unless an addition rounds across an integer (non-integer POI
        // coordinates next to a power of two), that is the box starting at (int)start -- checked per POI, and then the
        // walk's incremental offset replaces three conversions and a 64-bit address per sample.
        bool ref_box = true;
        for (int q = tid; q < SX + SY + SZ; q += kBlock3d) {
            const int ax = q < SX ? 0 : (q < SX + SY ? 1 : 2);
            const int e = ax == 0 ? q : (ax == 1 ? q - SX : q - SX - SY);
            const float st = ax == 0 ? sxf : (ax == 1 ? syf : szf);
            ref_box = ref_box && ((int)(st + e) == (int)st + e);
        }
        ref_box = __syncthreads_and(ref_box ? 1 : 0) != 0;
        const float* __restrict__ pref = P.ref + (((size_t)(int)szf * DY + (int)syf) * DX + (int)sxf);
        float ref_mean, ref_norm;
        const int cnt = N > tid ? (N - tid + kBlock3d - 1) / kBlock3d : 0;  // samples this thread owns
        // reference voxel of a sample: the fast path (the subvolume is one box) is a separate instantiation so that a
        // batch's loads are straight-line code
        auto ref_fast = [&](const WalkPoint& q, int) { return pref[q.off]; };
        auto ref_slow = [&](const WalkPoint& q, int) {
            return P.ref[((size_t)(int)(szf + ((int)q.z + rz)) * DY + (int)(syf + ((int)q.y + ry))) * DX + (int)(sxf + ((int)q.x + rx))];
        };
        auto ref_stats = [&](auto&& ref_of) {
            float acc[1] = {0.f};
            Walk3 w(tid, SX, SY, 0, DX, DY);
            sweep_batched<8>(w, rx, ry, rz, cnt, ref_of, [&](const WalkPoint&, float v, int) { acc[0] += v; });
            block_allreduce<1>(acc, red, wave, lane);
            ref_mean = acc[0] / fN;
            acc[0] = 0.f;
            Walk3 w2(tid, SX, SY, 0, DX, DY);
            sweep_batched<8>(w2, rx, ry, rz, cnt, ref_of, [&](const WalkPoint&, float v, int) {
                const float d = v - ref_mean;
                acc[0] = mad(d, d, acc[0]);
            });
            block_allreduce<1>(acc, red, wave, lane);
            ref_norm = sqrtf(acc[0]);
        };
        if (ref_box) ref_stats(ref_fast);
        else ref_stats(ref_slow);

        lap(0);
        // ---- SD image + Hessian (src/oc_icgn.cpp:1299-1337) and its inverse (:1339)
        const int cx = (int)px, cy = (int)py, cz = (int)pz;
        {
            // The 78 unique sums are accumulated in two sweeps over the samples (rows 0-7: 36 running sums, rows 8-11: 42)
            // so that a batch of four samples' gradients fits the 128-VGPR budget beside the accumulators -- one sweep
            // over all 78 sums spilled 24 registers per sample.  The coefficient window is idle before the first sweep of
            // the Gauss-Newton loop: it hosts the matrix.
            float* A = win;
            if constexpr (OC_ABLATE & 4) {
                if (tid < 144) A[tid] = (tid / 12 == tid % 12) ? 1.0e12f : 0.f;
                __syncthreads();
            } else {
                hessian_rows<0, 8>(P, tid, wave, lane, SX, SY, N, rx, ry, rz, cx, cy, cz, DX, DY, red, A);
                hessian_rows<8, 12>(P, tid, wave, lane, SX, SY, N, rx, ry, rz, cx, cy, cz, DX, DY, red, A);
            }
            lap(1);
            // one wave inverts; the other seven leave their issue slots to the second workgroup of the CU and pick H^-1
            // up from LDS after the next barrier
            if (wave == 0) lu_inverse12_lds(A, reinterpret_cast<int*>(win + 144), lds_hinv, lane);
            lap(2);
            // visible to every wave after the barriers of the first block_allreduce below
        }

        // ---- IC-GN loop (src/oc_icgn.cpp:1344-1447)
        float Wm[16];
        set_warp_3d1(Wm, init);
        int iter = 0;
        float dp_norm = 0.f, znssd = 0.f;
        bool failed = false;
#pragma nounroll
        do {
            iter++;
            bool out_of_range = false;
            float acc[1] = {0.f};
            {
                // Deformation3D1::warp (src/oc_deformation.cpp:518-530) + subvolume centre, as every sample evaluates it
                // (OC_FMA: the second and third products join the running sum; mad is monotone in every argument like the
                // separately rounded form, so the corner argument of the coefficient boxes below holds in both modes)
                auto warp_x = [&](float xl, float yl, float zl) { return px + (mad(Wm[2], zl, mad(Wm[1], yl, Wm[0] * xl)) + Wm[3] * 1.f); };
                auto warp_y = [&](float xl, float yl, float zl) { return py + (mad(Wm[6], zl, mad(Wm[5], yl, Wm[4] * xl)) + Wm[7] * 1.f); };
                auto warp_z = [&](float xl, float yl, float zl) { return pz + (mad(Wm[10], zl, mad(Wm[9], yl, Wm[8] * xl)) + Wm[11] * 1.f); };
                // ---- coefficient boxes of all passes of this sweep, one pass per thread (the box of a pass
                // costs more arithmetic than a sample does, so it is not recomputed by every wave in every pass).
                // A pass = M * 512 consecutive samples: thread tid owns s = tid + 512 * (M * pass + m), m < M.
                const int M = P.samples_per_pass;
                const int pass_len = M * kBlock3d;
                const int npass = (N + pass_len - 1) / pass_len;
                for (int round0 = 0; round0 < npass; round0 += kBoxSlots) {
                    __syncthreads();  // the previous round's boxes are no longer needed
                    for (int pass = round0 + tid; pass < min(npass, round0 + kBoxSlots); pass += kBlock3d) {
                        // index box of the pass's samples [s0, s1]: whole rows and (when several planes are
                        // touched) whole planes -- conservative, still a box
                        const int s0 = pass * pass_len, s1 = min(s0 + pass_len - 1, N - 1);
                        const int i0 = s0 / plane, i1 = s1 / plane;
                        const int ra = (s0 - i0 * plane) / SX, rb = (s1 - i1 * plane) / SX;
                        const int j0 = i0 == i1 ? ra : 0, j1 = i0 == i1 ? rb : SY - 1;
                        const bool one_row = i0 == i1 && j0 == j1;
                        const int k0 = one_row ? s0 - i0 * plane - ra * SX : 0, k1 = one_row ? s1 - i1 * plane - rb * SX : SX - 1;
                        // its image under the warp: every coordinate is monotone in each index (also in floating
                        // point), so the 8 corners bound what any sample of the pass computes
                        float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
#pragma unroll
                        for (int c = 0; c < 8; c++) {
                            const float xl = (float)(((c & 1) ? k1 : k0) - rx), yl = (float)(((c & 2) ? j1 : j0) - ry),
                                        zl = (float)(((c & 4) ? i1 : i0) - rz);
                            const float q[3] = {warp_x(xl, yl, zl), warp_y(xl, yl, zl), warp_z(xl, yl, zl)};
#pragma unroll
                            for (int a = 0; a < 3; a++) {
                                lo[a] = fminf(lo[a], q[a]);
                                hi[a] = fmaxf(hi[a], q[a]);
                            }
                        }
                        // taps of an in-range sample lie in [floor - 1, floor + 2]; in-range means [1, D - 2)
                        const int D[3] = {DX, DY, DZ};
                        int o[3], n[3];
                        bool usable = true;
#pragma unroll
                        for (int a = 0; a < 3; a++) {
                            usable = usable && lo[a] == lo[a] && hi[a] == hi[a] && fabsf(lo[a]) < 1.0e9f && fabsf(hi[a]) < 1.0e9f;
                            const int fl = (int)floorf(fmaxf(lo[a], 1.f)) - 1, fh = (int)floorf(fminf(hi[a], (float)(D[a] - 3))) + 2;
                            o[a] = max(fl, 0);
                            n[a] = min(fh, D[a] - 1) - o[a] + 1;
                        }
                        // n[0] = 0 marks "do not stage": nothing of the pass is in range (every sample evaluates
                        // to -1), or the box does not fit -> global taps
                        const bool stage = usable && n[0] >= 4 && n[1] >= 4 && n[2] >= 4 && (PX == 0 || n[0] <= PX) &&
                                           (long long)(PX ? PX : n[0]) * n[1] * n[2] <= kWinCap;
                        int* slot = boxes + (pass - round0) * 6;
                        slot[0] = o[0]; slot[1] = o[1]; slot[2] = o[2];
                        slot[3] = stage ? n[0] : 0; slot[4] = n[1]; slot[5] = n[2];
                    }
                    __syncthreads();
                    Walk3 w(tid, SX, SY, round0 * M);
                    for (int pass = round0; pass < min(npass, round0 + kBoxSlots); pass++) {
                        const int* slot = boxes + (pass - round0) * 6;
                        int o[3], n[3];
#pragma unroll
                        for (int a = 0; a < 3; a++) {
                            o[a] = __builtin_amdgcn_readfirstlane(slot[a]);
                            n[a] = __builtin_amdgcn_readfirstlane(slot[3 + a]);
                        }
                        const bool staged = n[0] > 0;
                        const int nx = n[0];                 // floats fetched per row
                        const int pitch = PX ? PX : n[0];    // floats between rows in LDS
                        const int nxy = pitch * n[1];
                        if (staged && !(OC_ABLATE & 1)) {
                            __syncthreads();  // the previous pass has finished reading the box
                            // rows of the box, round-robin over the waves; (zr, yr) advance without a division
                            const int rows = n[1] * n[2];
                            int zr = wave / n[1], yr = wave - zr * n[1];
                            const int dzr = kWaves3d / n[1], dyr = kWaves3d - dzr * n[1];
                            if (nx <= kWave) {
                                // a row fits one wave-wide load: fetch kStageRows rows before the first LDS write, so
                                // that their latencies overlap (one row at a time left the wave waiting ~30 times per
                                // pass); (zr, yr) are wave-uniform and advance on the scalar unit
                                constexpr int kStageRows = 16;
                                for (int row0 = wave; row0 < rows; row0 += kStageRows * kWaves3d) {
                                    float v[kStageRows];
#pragma unroll
                                    for (int u = 0; u < kStageRows; u++) {
                                        const int row = row0 + u * kWaves3d;
                                        v[u] = 0.f;
                                        if (row < rows && lane < nx)
                                            v[u] = P.coef[((size_t)(o[2] + zr) * DY + (o[1] + yr)) * DX + o[0] + lane];
                                        yr += dyr;
                                        zr += dzr;
                                        if (yr >= n[1]) {
                                            yr -= n[1];
                                            zr++;
                                        }
                                    }
#pragma unroll
                                    for (int u = 0; u < kStageRows; u++) {
                                        const int row = row0 + u * kWaves3d;
                                        if (row < rows && lane < nx) win[row * pitch + lane] = v[u];
                                    }
                                }
                            } else {
                                for (int row = wave; row < rows; row += kWaves3d) {
                                    const float* __restrict__ src = P.coef + ((size_t)(o[2] + zr) * DY + (o[1] + yr)) * DX + o[0];
                                    for (int x = lane; x < nx; x += kWave) win[row * pitch + x] = src[x];
                                    yr += dyr;
                                    zr += dzr;
                                    if (yr >= n[1]) {
                                        yr -= n[1];
                                        zr++;
                                    }
                                }
                            }
                            __syncthreads();
                        }
                        for (int m = 0; m < M; m++, w.next()) {
                            if (w.s < N) {
                                const float xl = (float)(w.k - rx), yl = (float)(w.j - ry), zl = (float)(w.i - rz);
                                const float x = warp_x(xl, yl, zl), y = warp_y(xl, yl, zl), z = warp_z(xl, yl, zl);
                                float v;
                                if constexpr (OC_ABLATE & 2) v = x + y + z;
                                else
                                    v = staged ? bspline3d_eval_lds<PX>(win, o[0], o[1], o[2], pitch, nxy, DZ, DY, DX, x, y, z)
                                               : bspline3d_eval(P.coef, DZ, DY, DX, x, y, z);
                                out_of_range = out_of_range || (v < 0.f);
                                ts[w.s] = v;
                                acc[0] += v;
                            }
                        }
                    }
                }
            }
            lap(3);
            // src/oc_icgn.cpp:1396-1400
            if (__syncthreads_or(out_of_range && !(OC_ABLATE & 111) ? 1 : 0)) {
                failed = true;
                break;
            }
            block_allreduce<1>(acc, red, wave, lane);
            const float tmean = acc[0] / fN;
            acc[0] = 0.f;
            {
                Walk3 w(tid, SX, SY, 0, DX, DY);
                sweep_batched<8>(w, rx, ry, rz, cnt, [&](const WalkPoint&, int sidx) { return ts[sidx]; },
                                 [&](const WalkPoint&, float v, int) {
                                     const float d = v - tmean;
                                     acc[0] = mad(d, d, acc[0]);
                                 });
            }
            block_allreduce<1>(acc, red, wave, lane);
            const float tar_norm = sqrtf(acc[0]);
            // error image, ZNSSD, numerator (src/oc_icgn.cpp:1403-1433)
            const float factor = ref_norm / tar_norm;
            float num[13];
#pragma unroll
            for (int i = 0; i < 13; i++) num[i] = 0.f;
            {
                Walk3 w(tid, SX, SY, 0, DX, DY);
                const size_t gbase = ((size_t)(cz - rz) * DY + (cy - ry)) * DX + (cx - rx);
                const float* __restrict__ pgx = P.gx + gbase;
                const float* __restrict__ pgy = P.gy + gbase;
                const float* __restrict__ pgz = P.gz + gbase;
                struct S5 {
                    float r, t, x, y, z;
                };
                auto numerator = [&](auto&& ref_of) {
                    sweep_batched<4>(
                        w, rx, ry, rz, (OC_ABLATE & 8) ? 0 : cnt,
                        [&](const WalkPoint& q, int sidx) { return S5{ref_of(q, sidx), ts[sidx], pgx[q.off], pgy[q.off], pgz[q.off]}; },
                        [&](const WalkPoint& q, const S5& v, int) {
                            const float rsv = v.r - ref_mean;
                            const float tz = v.t - tmean;
                            const float e = mad(factor, tz, -rsv);
                            const float g_x = v.x, g_y = v.y, g_z = v.z;
                            const float fx = q.x, fy = q.y, fz = q.z;
                            num[12] = mad(e, e, num[12]);
                            num[0] = mad(g_x, e, num[0]); num[1] = mad(g_x * fx, e, num[1]); num[2] = mad(g_x * fy, e, num[2]); num[3] = mad(g_x * fz, e, num[3]);
                            num[4] = mad(g_y, e, num[4]); num[5] = mad(g_y * fx, e, num[5]); num[6] = mad(g_y * fy, e, num[6]); num[7] = mad(g_y * fz, e, num[7]);
                            num[8] = mad(g_z, e, num[8]); num[9] = mad(g_z * fx, e, num[9]); num[10] = mad(g_z * fy, e, num[10]); num[11] = mad(g_z * fz, e, num[11]);
                        });
                };
                if (ref_box) numerator(ref_fast);
                else numerator(ref_slow);
            }
            block_allreduce<13>(num, red, wave, lane);
            lap(4);
            znssd = num[12] / (ref_norm * ref_norm);
            // dp = H^-1 * numerator (src/oc_icgn.cpp:1435-1443)
            float numj = 0.f;
#pragma unroll
            for (int j = 0; j < 12; j++) numj = lane == j ? num[j] : numj;
            float dp[12];
#pragma unroll
            for (int i = 0; i < 12; i++) {
                const float prod = lds_hinv[i * kWave + lane] * numj;
                float v = 0.f;
#pragma unroll
                for (int j = 0; j < 12; j++) v += wave_bcast(prod, j);
                dp[i] = v;
            }
            float dW[16], dWi[16], Wn[16];
            set_warp_3d1(dW, dp);
            inverse4(dW, dWi);
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    float v = Wm[i * 4 + 0] * dWi[0 * 4 + j];
#pragma unroll
                    for (int k = 1; k < 4; k++) v = v + Wm[i * 4 + k] * dWi[k * 4 + j];
                    Wn[i * 4 + j] = v;
                }
#pragma unroll
            for (int i = 0; i < 16; i++) Wm[i] = uni3(Wn[i]);
            // src/oc_icgn.cpp:1445
            dp_norm = uni3(sqrtf(dp[0] * dp[0] + dp[4] * dp[4] + dp[8] * dp[8]));
            lap(5);
        } while (iter < P.stop && ((OC_ABLATE & 111) ? iter < 3 : dp_norm >= P.conv));

        if (failed) {
            if (tid == 0) poi[poi3d::ZNCC] = -3.f;
            continue;
        }
        // ---- outputs (src/oc_icgn.cpp:1449-1489)
        if (tid == 0) {
            // Deformation3D1::setDeformation(), src/oc_deformation.cpp:416-432: p <- W after the last update
            const float cur[12] = {Wm[3], Wm[0] - 1.f, Wm[1], Wm[2], Wm[7],  Wm[4],
                                   Wm[5] - 1.f, Wm[6], Wm[11], Wm[8], Wm[9], Wm[10] - 1.f};
            float zncc = 0.5f * (2 - znssd);
            const float fiter = (float)iter;
            if (dp_norm >= P.conv && fiter >= P.stop) zncc = -4.f;
            float o0 = cur[0], o4 = cur[4], o8 = cur[8];
            if (isnan(zncc) || isnan(o0) || isnan(o4) || isnan(o8)) {
                o0 = init[0]; o4 = init[4]; o8 = init[8];
                zncc = -5.f;
            }
#pragma unroll
            for (int i = 0; i < 12; i++) poi[poi3d::P + i] = cur[i];
            poi[poi3d::U] = o0;
            poi[poi3d::V] = o4;
            poi[poi3d::W] = o8;
            poi[poi3d::U0] = init[0];
            poi[poi3d::V0] = init[4];
            poi[poi3d::W0] = init[8];
            poi[poi3d::ZNCC] = zncc;
            poi[poi3d::ITER] = fiter;
            poi[poi3d::CONV] = dp_norm;
            poi[poi3d::SRX] = (float)rx;
            poi[poi3d::SRY] = (float)ry;
            poi[poi3d::SRZ] = (float)rz;
            if constexpr (OC_ABLATE & 16) {
#pragma unroll
                for (int i = 0; i < 6; i++) poi[22 + i] = (float)tl[i] * 1.0e-3f;
            }
        }
    }
}

}  // namespace OC_ARITH

#if !OC_FMA
// persistent workgroups (two per CU), each with one scratch slot for the warped subvolume
size_t icgn3d1_scratch_floats(int rx, int ry, int rz, int* blocks) {
    const size_t n = (size_t)(2 * rx + 1) * (2 * ry + 1) * (2 * rz + 1);
    *blocks = 512;
#if OC_BUILD_AB
    // experiments only, A/B build only (tools/icgn3d_occupancy_probe.py): OC_ICGN3D_BLOCKS=256 leaves ONE persistent workgroup per CU
    static const int env_blocks = std::getenv("OC_ICGN3D_BLOCKS") ? std::atoi(std::getenv("OC_ICGN3D_BLOCKS")) : 0;
    if (env_blocks >= 8 && env_blocks <= 512) *blocks = env_blocks / 8 * 8;
#endif
    return n * (size_t)*blocks;
}

// p.arith_fma selects the build whose per-sample multiply-adds are fused (oc_device.h; icgn3d_fma.o)
hipError_t launch_icgn3d1(const Icgn3dParams& p, float* pois, int stride_f, size_t count, hipStream_t stream) {
    return p.arith_fma ? fma::launch_icgn3d1(p, pois, stride_f, count, stream) : sep::launch_icgn3d1(p, pois, stride_f, count, stream);
}
#endif  // !OC_FMA

namespace OC_ARITH {
hipError_t launch_icgn3d1(const Icgn3dParams& p, float* pois, int stride_f, size_t count, hipStream_t stream) {
    if (count == 0) return hipSuccess;
    if (!p.scratch) return hipErrorInvalidValue;
    int blocks = 0;
    (void)icgn3d1_scratch_floats(p.rx, p.ry, p.rz, &blocks);
    unsigned grid = (unsigned)(count < (size_t)blocks ? count : (size_t)blocks);
    grid = (grid + 7) / 8 * 8;  // whole XCD rounds (idle workgroups exit at once); never more than `blocks` slots
    // samples per thread and pass: as many as keep the nominal coefficient box (small deformation
    // gradients) inside the LDS window; passes that still overflow fall back to global taps
    Icgn3dParams q = p;
    q.samples_per_pass = 1;
    // row pitch of the staged box: the nominal box is (2rx+1) + 5 floats wide; a compile-time pitch makes the tap
    // offsets immediates
    const int want = 2 * p.rx + 1 + 5;
    const int px = want <= 40 ? 40 : want <= 48 ? 48 : want <= 64 ? 64 : 0;
    const int tries[] = {16, 12, 10, 8, 6, 4, 3, 2, 1};
    for (int m : tries) {
        const long long sx = 2 * p.rx + 1, sy = 2 * p.ry + 1, sz = 2 * p.rz + 1, len = (long long)m * kBlock3d;
        const long long planes = (len + sx * sy - 1) / (sx * sy) + 1;
        const long long nz = (planes < sz ? planes : sz) + 3 + 1;
        const long long rows = planes > 1 ? sy : (len + sx - 1) / sx + 1;
        const long long ny = (rows < sy ? rows : sy) + 3 + 2, nx = px ? px : sx + 3 + 2;
        if (nx * ny * nz <= kWinCap) {
            q.samples_per_pass = m;
            break;
        }
    }
    (void)hipGetLastError();  // drop stale errors of earlier, unrelated calls
    switch (px) {
        case 40: hipLaunchKernelGGL(icgn3d1_kernel<40>, dim3(grid), dim3(kBlock3d), 0, stream, q, pois, stride_f, (unsigned long long)count); break;
        case 48: hipLaunchKernelGGL(icgn3d1_kernel<48>, dim3(grid), dim3(kBlock3d), 0, stream, q, pois, stride_f, (unsigned long long)count); break;
        case 64: hipLaunchKernelGGL(icgn3d1_kernel<64>, dim3(grid), dim3(kBlock3d), 0, stream, q, pois, stride_f, (unsigned long long)count); break;
        default: hipLaunchKernelGGL(icgn3d1_kernel<0>, dim3(grid), dim3(kBlock3d), 0, stream, q, pois, stride_f, (unsigned long long)count); break;
    }
    return hipGetLastError();
}
}  // namespace OC_ARITH

}  // namespace ochip
