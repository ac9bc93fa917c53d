// icgn3d_rows.hip -- ICGN3D1 with ONE HALF-WAVE PER SUBVOLUME ROW (round 4).  NOT the default mapping: measured 12 - 25 %
// slower than icgn3d.hip, it is the A/B partner behind oc_hip_set_tuning("icgn3d_mapping", 1) and is compiled
// only into the A/B build of the library (-DOC_BUILD_AB=1, opencorr_amd/build.py --ab).  Oracle order: OC_ORDER_ROWS.
//
// Replaces ICGN3D1::compute(POI3D*) (src/oc_icgn.cpp:1270-1490) for a whole POI queue (:1492-1500), like icgn3d.hip, whose
// mapping (sample s owned by thread s mod 512) stays available behind oc_hip_set_tuning("icgn3d_mapping", 0) as the A/B
// partner.  Same phases, same LDS-staged coefficient boxes, same arithmetic per sample; what changes is WHO owns a sample:
//
//   body   the first BW = 32 * NCH columns of every row (i, j) of the (2rz+1)(2ry+1) rows of the subvolume: row r = i * SY + j
//          belongs to half-wave r mod 16, column k to lane k mod 32 of it, chunk c = k / 32 (NCH = 1 for 33 ... 59 samples
//          per row -- 33^3 subvolumes: columns 0 ... 31 --, 2 from 60 on; a last chunk of 28 ... 31 columns counts as a chunk with
//          idle lanes).  A thread keeps ITS x for the whole kernel: the local coordinate and its share of the warp are
//          per-lane constants, the walk over its samples is a row step (16 rows down: one compare), and the 32 lanes of a
//          tap read sit in ONE row of the staged box -- consecutive LDS addresses instead of two rows 40 floats apart (47 %
//          of the LDS cycles were bank conflicts under the old mapping).
//   tail   the RT = SX - BW remaining columns of every row (one column at 33^3: 1 089 of 35 937 samples), enumerated
//          q = r * RT + (k - BW) and owned by thread q mod 512 -- the old general mapping on a narrow block; it gets its own
//          pass(es) and coefficient box(es) at the end of a sweep.  Subvolumes narrower than 28 samples are all tail.
//
// Reductions: ONE running sum per thread -- its body samples in ascending (row, chunk), then its tail samples in ascending
// q --, then the xor butterfly inside each wave and the balanced tree over the 8 wave sums: the association the oracle
// implements as OC_ORDER_ROWS (oracle/oc_oracle.cpp AccRows).  Subvolumes without a body (fewer than 28 samples per row)
// ARE the old mapping, bit for bit: they run the kernel of icgn3d.hip.
#include <cstdio>
#include <cstdlib>

#include "icgn3d_device.h"

namespace ochip {

namespace {

constexpr int kHalves = kBlock3d / 32;  // 16 half-waves = rows in flight per step

// chunks of 32 columns handled by the row mapping (must match oracle AccRows)
__host__ __device__ constexpr int rows_body_chunks(int SX) {
    const int fc = SX / 32, rc = SX % 32;
    return fc >= 2 ? 2 : fc + (rc >= 28 ? 1 : 0);
}

// the rows owned by one half-wave: row = h, h + 16, ...; row = i * SY + j
struct RowWalk {
    int i, j, row;
    unsigned off;  // (i * DY + j) * DX: voxel offset of the row's first sample from the subvolume's first voxel
    int SY, di, dj;
    unsigned doff, coff;
    __device__ __forceinline__ RowWalk(int h, int first_step, int SY_, int DX, int DY) : SY(SY_) {
        asm volatile("" : "+v"(h));  // set the walk up on the spot (see Walk3)
        row = h + kHalves * first_step;
        i = row / SY_;
        j = row - i * SY_;
        di = kHalves / SY_;
        dj = kHalves - di * SY_;
        off = (unsigned)((i * DY + j) * DX);
        doff = (unsigned)((di * DY + dj) * DX);
        coff = (unsigned)((DY - SY_) * DX);  // j wrapped: one plane further, SY rows back
    }
    __device__ __forceinline__ void next() {
        row += kHalves;
        j += dj;
        off += doff;
        const bool c = j >= SY;
        j = c ? j - SY : j;
        off += c ? coff : 0u;
        i += di + (c ? 1 : 0);
    }
};

// per-thread constants of the row mapping
template <int NCH>
struct Cols {
    float xl[NCH];     // local x coordinate of the thread's column in chunk c
    unsigned kc[NCH];  // the column itself
    bool act[NCH];     // the column exists (a last chunk may be partial)
};

// The body samples of a thread in ascending (row, chunk), B rows per batch: the B * NCH loads of a batch are issued before
// any of them is used (the streaming sweeps are latency bound, see sweep_batched).  load(point, idx) / use(point, v, idx):
// idx = the sample's slot in the workgroup's scratch array: (step * NCH + chunk) * 512 + tid.
template <int B, int NCH, class Load, class Use>
__device__ __forceinline__ void sweep_rows(RowWalk& rw, int rows_mine, const Cols<NCH>& C, int tid, int ry, int rz, Load&& load, Use&& use) {
    {
        int done = 0;
        auto point = [&]() {
            const WalkPoint q = {rw.off, 0.f, (float)(rw.j - ry), (float)(rw.i - rz)};
            rw.next();
            return q;
        };
        auto at = [&](const WalkPoint& q, int c) { return WalkPoint{q.off + C.kc[c], C.xl[c], q.y, q.z}; };
        int slot = (rw.row / kHalves) * NCH * kBlock3d + tid;  // (first step of this walk)
#pragma unroll 1
        for (; done + B <= rows_mine; done += B) {
            WalkPoint p[B];
#pragma unroll
            for (int u = 0; u < B; u++) p[u] = point();
            decltype(load(p[0], 0)) v[B][NCH];
#pragma unroll
            for (int u = 0; u < B; u++)
#pragma unroll
                for (int c = 0; c < NCH; c++)
                    if (C.act[c]) v[u][c] = load(at(p[u], c), slot + (u * NCH + c) * kBlock3d);
#pragma unroll
            for (int u = 0; u < B; u++)
#pragma unroll
                for (int c = 0; c < NCH; c++)
                    if (C.act[c]) use(at(p[u], c), v[u][c], slot + (u * NCH + c) * kBlock3d);
            slot += B * NCH * kBlock3d;
        }
#pragma unroll 1
        for (; done < rows_mine; done++) {
            const WalkPoint q = point();
#pragma unroll
            for (int c = 0; c < NCH; c++)
                if (C.act[c]) {
                    const WalkPoint qc = at(q, c);
                    use(qc, load(qc, slot + c * kBlock3d), slot + c * kBlock3d);
                }
            slot += NCH * kBlock3d;
        }
    }
}

// geometry of a launch (wave-uniform)
struct RowsGeo {
    int SX, SY, SZ, ROWS, N;
    int BW, RT, NT;  // body columns, tail columns, tail samples
    int MB;          // body steps = ceil(ROWS / 16)
    int tail_base;   // first scratch slot of the tail samples (in samples): MB * NCH * 512
};

template <int NCH>
__device__ __forceinline__ RowsGeo make_geo(int rx, int ry, int rz) {
    RowsGeo g;
    g.SX = 2 * rx + 1;
    g.SY = 2 * ry + 1;
    g.SZ = 2 * rz + 1;
    g.ROWS = g.SY * g.SZ;
    g.N = g.SX * g.ROWS;
    g.BW = min(g.SX, 32 * NCH);
    g.RT = g.SX - g.BW;
    g.NT = g.RT * g.ROWS;
    g.MB = (g.ROWS + kHalves - 1) / kHalves;
    g.tail_base = g.MB * NCH * kBlock3d;
    return g;
}

// A complete sweep over the thread's samples: body, then tail (the order of every per-thread running sum).
// load(point, idx) -> value(s) of the sample, use(point, value, idx) consumes it; idx = the sample's scratch slot.
template <int B, int NCH, class Load, class Use>
__device__ __forceinline__ void sweep_all(const RowsGeo& g, const Cols<NCH>& C, int tid, int rx, int ry, int rz, int DX, int DY, Load&& load,
                                          Use&& use) {
    {
        const int h = tid >> 5;
        const int rows_mine = g.ROWS > h ? (g.ROWS - h + kHalves - 1) / kHalves : 0;
        RowWalk rw(h, 0, g.SY, DX, DY);
        sweep_rows<B, NCH>(rw, rows_mine, C, tid, ry, rz, load, use);
    }
    if (g.RT > 0) {  // wave-uniform
        const int cnt = g.NT > tid ? (g.NT - tid + kBlock3d - 1) / kBlock3d : 0;
        Walk3 w(tid, g.RT, g.SY, 0, DX, DY);
        const unsigned bw = (unsigned)g.BW;
        const int base = g.tail_base;
        // the tail block starts at column BW: local x = (k + BW) - rx, voxel offset + BW
        sweep_batched<B>(
            w, rx - g.BW, ry, rz, cnt,
            [&](const WalkPoint& q, int s) {
                WalkPoint t = q;
                t.off += bw;
                return load(t, base + s);
            },
            [&](const WalkPoint& q, const auto& v, int s) {
                WalkPoint t = q;
                t.off += bw;
                use(t, v, base + s);
            });
    }
}

// Hessian rows [R0, R1): sums of sd[r]*sd[c], c <= r, over all samples (src/oc_icgn.cpp:1299-1337), block-reduced and filed
// into the symmetric matrix A (LDS).  Packed pairs over adjacent columns as in icgn3d.hip (same products, same order).
template <int R0, int R1, int NCH>
__device__ __forceinline__ void hessian_rows_r(const Icgn3dParams& P, const RowsGeo& g, const Cols<NCH>& C, int tid, int wave, int lane, int rx, int ry,
                                               int rz, int cx, int cy, int cz, int DX, int DY, float* red, float* __restrict__ A) {
    constexpr int NE = (R1 * (R1 + 1) - R0 * (R0 + 1)) / 2;
    f2 hp[12][6];
    float hd[12];
#pragma unroll
    for (int r = 0; r < 12; r++) {
        hd[r] = 0.f;
#pragma unroll
        for (int q = 0; q < 6; q++) hp[r][q] = mk2(0.f, 0.f);
    }
    const size_t gbase = ((size_t)(cz - rz) * DY + (cy - ry)) * DX + (cx - rx);
    const float* __restrict__ pgx = P.gx + gbase;
    const float* __restrict__ pgy = P.gy + gbase;
    const float* __restrict__ pgz = P.gz + gbase;
    struct G3 {
        float x, y, z;
    };
    sweep_all<4, NCH>(
        g, C, tid, rx, ry, rz, DX, DY, [&](const WalkPoint& q, int) { return G3{pgx[q.off], pgy[q.off], pgz[q.off]}; },
        [&](const WalkPoint& q, const G3& gv, int) {
            const float g_x = gv.x, g_y = gv.y, g_z = gv.z;
            const f2 m01 = mk2(1.f, q.x), m23 = mk2(q.y, q.z);  // g * 1.f is exact
            const f2 sdp[6] = {g_x * m01, g_x * m23, g_y * m01, g_y * m23, g_z * m01, g_z * m23};
#pragma unroll
            for (int r = R0; r < R1; r++) {
                const float sr = (r & 1) ? sdp[r / 2].y : sdp[r / 2].x;
#pragma unroll
                for (int q2 = 0; q2 < (r + 1) / 2; q2++) hp[r][q2] = hp[r][q2] + sr * sdp[q2];
                if ((r & 1) == 0) hd[r] = hd[r] + sr * sr;
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
    constexpr int NCHK = (NE + kRedChunk - 1) / kRedChunk;
#pragma unroll
    for (int ch = 0; ch < NCHK; ch++) {
        float part[kRedChunk];
#pragma unroll
        for (int q = 0; q < kRedChunk; q++) part[q] = (ch * kRedChunk + q < NE) ? h[(ch * kRedChunk + q) % NE] : 0.f;
        block_allreduce<kRedChunk>(part, red, wave, lane);
#pragma unroll
        for (int q = 0; q < kRedChunk; q++)
            if (ch * kRedChunk + q < NE) h[(ch * kRedChunk + q) % NE] = part[q];
    }
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

// PX: row pitch of the staged coefficient box of BODY passes (0 = the box's own width); tail passes always use their
// box's own width (a tail block is a few columns wide)
template <int PX, int NCH>
__global__ __launch_bounds__(kBlock3d, 4) void icgn3d1_rows_kernel(Icgn3dParams P, float* __restrict__ pois, int stride_f,
                                                                  unsigned long long count) {
    __shared__ __attribute__((aligned(16))) float lds[kRedChunk * kWaves3d + 12 * kWave + kWinCap + 6 * kBoxSlots];
    float* red = lds;                               // kRedChunk * 8 floats
    float* lds_hinv = lds + kRedChunk * kWaves3d;   // 12 x 64 floats: column j of H^-1 in lane j (parked between solves)
    float* win = lds_hinv + 12 * kWave;             // staged coefficient box of the current pass
    int* boxes = reinterpret_cast<int*>(win + kWinCap);  // origin + extent of the box of each pass (kBoxSlots x 6)
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1), wave = tid >> 6;
    const int rx = P.rx, ry = P.ry, rz = P.rz, DX = P.dx, DY = P.dy, DZ = P.dz;
    const RowsGeo g = make_geo<NCH>(rx, ry, rz);
    const int SY = g.SY, N = g.N;
    const float fN = (float)N;
    // per-thread columns of the row mapping
    Cols<NCH> C;
    {
        const int kk = tid & 31;
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            const int k = 32 * c + kk;
            C.kc[c] = (unsigned)k;
            C.xl[c] = (float)(k - rx);
            C.act[c] = k < g.BW;
        }
    }
    const int h = tid >> 5;
    // scratch slot of this workgroup: body slots (step * NCH + chunk) * 512 + tid, tail slots tail_base + q
    float* __restrict__ ts = P.scratch + (size_t)blockIdx.x * (size_t)(g.tail_base + (g.NT + kBlock3d - 1) / kBlock3d * kBlock3d);

    const unsigned long long xcd_chunk = (count + 7) / 8, xcd_lo = (blockIdx.x & 7u) * xcd_chunk;
    const unsigned long long xcd_hi = min(count, xcd_lo + xcd_chunk);
    for (unsigned long long idx = xcd_lo + (blockIdx.x >> 3); idx < xcd_hi; idx += gridDim.x >> 3) {
        // the k-th solve of the launch takes POI perm[k] (a locality schedule, poi_order.hip) or simply POI k
        float* poi = pois + (P.perm ? (unsigned long long)P.perm[idx] : idx) * (unsigned long long)stride_f;
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

        // ---- reference subvolume mean + norm (src/oc_subset.cpp:89-135)
        const float sxf = px - rx, syf = py - ry, szf = pz - rz;
        // Subset3D::fill reads voxel (int(start.z + i), int(start.y + j), int(start.x + k)): one box starting at (int)start
        // unless a float addition rounds across an integer -- checked per POI (see icgn3d.hip)
        bool ref_box = true;
        for (int q = tid; q < g.SX + g.SY + g.SZ; q += kBlock3d) {
            const int ax = q < g.SX ? 0 : (q < g.SX + g.SY ? 1 : 2);
            const int e = ax == 0 ? q : (ax == 1 ? q - g.SX : q - g.SX - g.SY);
            const float st = ax == 0 ? sxf : (ax == 1 ? syf : szf);
            ref_box = ref_box && ((int)(st + e) == (int)st + e);
        }
        ref_box = __syncthreads_and(ref_box ? 1 : 0) != 0;
        const float* __restrict__ pref = P.ref + (((size_t)(int)szf * DY + (int)syf) * DX + (int)sxf);
        float ref_mean, ref_norm;
        auto ref_fast = [&](const WalkPoint& q, int) { return pref[q.off]; };
        auto ref_slow = [&](const WalkPoint& q, int) {
            return P.ref[((size_t)(int)(szf + ((int)q.z + rz)) * DY + (int)(syf + ((int)q.y + ry))) * DX + (int)(sxf + ((int)q.x + rx))];
        };
        auto ref_stats = [&](auto&& ref_of) {
            float acc[1] = {0.f};
            sweep_all<8, NCH>(g, C, tid, rx, ry, rz, DX, DY, ref_of, [&](const WalkPoint&, float v, int) { acc[0] += v; });
            block_allreduce<1>(acc, red, wave, lane);
            ref_mean = acc[0] / fN;
            acc[0] = 0.f;
            sweep_all<8, NCH>(g, C, tid, rx, ry, rz, DX, DY, ref_of, [&](const WalkPoint&, float v, int) {
                const float d = v - ref_mean;
                acc[0] += d * d;
            });
            block_allreduce<1>(acc, red, wave, lane);
            ref_norm = sqrtf(acc[0]);
        };
        if (ref_box) ref_stats(ref_fast);
        else ref_stats(ref_slow);

        // ---- SD image + Hessian (src/oc_icgn.cpp:1299-1337) and its inverse (:1339)
        const int cx = (int)px, cy = (int)py, cz = (int)pz;
        {
            float* A = win;  // the coefficient window is idle before the first sweep of the Gauss-Newton loop
            hessian_rows_r<0, 8, NCH>(P, g, C, tid, wave, lane, rx, ry, rz, cx, cy, cz, DX, DY, red, A);
            hessian_rows_r<8, 12, NCH>(P, g, C, tid, wave, lane, rx, ry, rz, cx, cy, cz, DX, DY, red, A);
            if (wave == 0) lu_inverse12_lds(A, reinterpret_cast<int*>(win + 144), lds_hinv, lane);
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
                auto warp_x = [&](float xl, float yl, float zl) { return px + (((Wm[0] * xl + Wm[1] * yl) + Wm[2] * zl) + Wm[3] * 1.f); };
                auto warp_y = [&](float xl, float yl, float zl) { return py + (((Wm[4] * xl + Wm[5] * yl) + Wm[6] * zl) + Wm[7] * 1.f); };
                auto warp_z = [&](float xl, float yl, float zl) { return pz + (((Wm[8] * xl + Wm[9] * yl) + Wm[10] * zl) + Wm[11] * 1.f); };
                // ---- passes of this sweep: body passes of MB_ steps (16 rows each), then tail passes of MT_ steps (512
                // tail samples each); their coefficient boxes are computed kBoxSlots at a time, one pass per thread
                const int MBs = P.samples_per_pass, MTs = P.tail_steps_per_pass;
                const int nbp = (g.MB + MBs - 1) / MBs;
                const int tail_steps = (g.NT + kBlock3d - 1) / kBlock3d;
                const int ntp = g.RT > 0 ? (tail_steps + MTs - 1) / MTs : 0;
                const int npass = nbp + ntp;
                RowWalk rw(h, 0, SY, 0, 0);                    // body rows, continued from pass to pass
                Walk3 tw(tid, g.RT > 0 ? g.RT : 1, SY, 0);     // tail samples, continued from pass to pass
                for (int round0 = 0; round0 < npass; round0 += kBoxSlots) {
                    __syncthreads();  // the previous round's boxes are no longer needed
                    for (int pass = round0 + tid; pass < min(npass, round0 + kBoxSlots); pass += kBlock3d) {
                        // index box of the pass: rows [r0, r1] x columns [k0, k1] -- whole rows and (when several planes
                        // are touched) whole planes: conservative, still a box
                        int r0, r1, k0, k1;
                        if (pass < nbp) {
                            r0 = pass * MBs * kHalves;
                            r1 = min(r0 + MBs * kHalves, g.ROWS) - 1;
                            k0 = 0;
                            k1 = g.BW - 1;
                        } else {
                            const int q0 = (pass - nbp) * MTs * kBlock3d, q1 = min(q0 + MTs * kBlock3d, g.NT) - 1;
                            r0 = q0 / g.RT;
                            r1 = q1 / g.RT;
                            const bool one = r0 == r1;
                            k0 = g.BW + (one ? q0 - r0 * g.RT : 0);
                            k1 = g.BW + (one ? q1 - r1 * g.RT : g.RT - 1);
                        }
                        const int i0 = r0 / SY, i1 = r1 / SY;
                        const int ja = r0 - i0 * SY, jb = r1 - i1 * SY;
                        const int j0 = i0 == i1 ? ja : 0, j1 = i0 == i1 ? jb : SY - 1;
                        // its image under the warp: every coordinate is monotone in each index (also in floating point), so
                        // the 8 corners bound what any sample of the pass computes
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
                        // n[0] = 0 marks "do not stage": nothing of the pass is in range, or the box does not fit -> global taps
                        const int pitch = (pass < nbp && PX) ? PX : n[0];
                        const bool stage = usable && n[0] >= 4 && n[1] >= 4 && n[2] >= 4 && n[0] <= pitch &&
                                           (long long)pitch * n[1] * n[2] <= kWinCap;
                        int* slot = boxes + (pass - round0) * 6;
                        slot[0] = o[0]; slot[1] = o[1]; slot[2] = o[2];
                        slot[3] = stage ? n[0] : 0; slot[4] = n[1]; slot[5] = n[2];
                    }
                    __syncthreads();
                    for (int pass = round0; pass < min(npass, round0 + kBoxSlots); pass++) {
                        const int* slot = boxes + (pass - round0) * 6;
                        int o[3], n[3];
#pragma unroll
                        for (int a = 0; a < 3; a++) {
                            o[a] = __builtin_amdgcn_readfirstlane(slot[a]);
                            n[a] = __builtin_amdgcn_readfirstlane(slot[3 + a]);
                        }
                        const bool body = pass < nbp;
                        const bool staged = n[0] > 0;
                        const int nx = n[0];                           // floats fetched per row
                        const int pitch = (body && PX) ? PX : n[0];    // floats between rows in LDS
                        const int nxy = pitch * n[1];
                        if (staged) {
                            __syncthreads();  // the previous pass has finished reading the box
                            const int rows = n[1] * n[2];
                            int zr = wave / n[1], yr = wave - zr * n[1];
                            const int dzr = kWaves3d / n[1], dyr = kWaves3d - dzr * n[1];
                            if (nx <= kWave) {
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
                        if (body) {
                            {
                                for (int m = 0; m < MBs; m++) {
                                    if (rw.row < g.ROWS) {
                                        const float yl = (float)(rw.j - ry), zl = (float)(rw.i - rz);
                                        const int slot0 = (rw.row / kHalves) * NCH * kBlock3d + tid;
#pragma unroll
                                        for (int c = 0; c < NCH; c++) {
                                            if (C.act[c]) {
                                                const float xl = C.xl[c];
                                                const float x = warp_x(xl, yl, zl), y = warp_y(xl, yl, zl), z = warp_z(xl, yl, zl);
                                                const float v = staged ? bspline3d_eval_lds<PX>(win, o[0], o[1], o[2], pitch, nxy, DZ, DY, DX, x, y, z)
                                                                       : bspline3d_eval(P.coef, DZ, DY, DX, x, y, z);
                                                out_of_range = out_of_range || (v < 0.f);
                                                ts[slot0 + c * kBlock3d] = v;
                                                acc[0] += v;
                                            }
                                        }
                                    }
                                    rw.next();
                                }
                            }
                        } else {
                            for (int m = 0; m < MTs; m++, tw.next()) {
                                if (tw.s < g.NT) {
                                    const float xl = (float)(tw.k + g.BW - rx), yl = (float)(tw.j - ry), zl = (float)(tw.i - rz);
                                    const float x = warp_x(xl, yl, zl), y = warp_y(xl, yl, zl), z = warp_z(xl, yl, zl);
                                    const float v = staged ? bspline3d_eval_lds<0>(win, o[0], o[1], o[2], pitch, nxy, DZ, DY, DX, x, y, z)
                                                           : bspline3d_eval(P.coef, DZ, DY, DX, x, y, z);
                                    out_of_range = out_of_range || (v < 0.f);
                                    ts[g.tail_base + tw.s] = v;
                                    acc[0] += v;
                                }
                            }
                        }
                    }
                }
            }
            // src/oc_icgn.cpp:1396-1400
            if (__syncthreads_or(out_of_range ? 1 : 0)) {
                failed = true;
                break;
            }
            block_allreduce<1>(acc, red, wave, lane);
            const float tmean = acc[0] / fN;
            acc[0] = 0.f;
            sweep_all<8, NCH>(
                g, C, tid, rx, ry, rz, DX, DY, [&](const WalkPoint&, int sidx) { return ts[sidx]; },
                [&](const WalkPoint&, float v, int) {
                    const float d = v - tmean;
                    acc[0] += d * d;
                });
            block_allreduce<1>(acc, red, wave, lane);
            const float tar_norm = sqrtf(acc[0]);
            // error image, ZNSSD, numerator (src/oc_icgn.cpp:1403-1433)
            const float factor = ref_norm / tar_norm;
            float num[13];
#pragma unroll
            for (int i = 0; i < 13; i++) num[i] = 0.f;
            {
                const size_t gbase = ((size_t)(cz - rz) * DY + (cy - ry)) * DX + (cx - rx);
                const float* __restrict__ pgx = P.gx + gbase;
                const float* __restrict__ pgy = P.gy + gbase;
                const float* __restrict__ pgz = P.gz + gbase;
                struct S5 {
                    float r, t, x, y, z;
                };
                auto numerator = [&](auto&& ref_of) {
                    sweep_all<4, NCH>(
                        g, C, tid, rx, ry, rz, DX, DY,
                        [&](const WalkPoint& q, int sidx) { return S5{ref_of(q, sidx), ts[sidx], pgx[q.off], pgy[q.off], pgz[q.off]}; },
                        [&](const WalkPoint& q, const S5& v, int) {
                            const float rsv = v.r - ref_mean;
                            const float tz = v.t - tmean;
                            const float e = factor * tz - rsv;
                            const float g_x = v.x, g_y = v.y, g_z = v.z;
                            const float fx = q.x, fy = q.y, fz = q.z;
                            num[12] += e * e;
                            num[0] += g_x * e; num[1] += (g_x * fx) * e; num[2] += (g_x * fy) * e; num[3] += (g_x * fz) * e;
                            num[4] += g_y * e; num[5] += (g_y * fx) * e; num[6] += (g_y * fy) * e; num[7] += (g_y * fz) * e;
                            num[8] += g_z * e; num[9] += (g_z * fx) * e; num[10] += (g_z * fy) * e; num[11] += (g_z * fz) * e;
                        });
                };
                if (ref_box) numerator(ref_fast);
                else numerator(ref_slow);
            }
            block_allreduce<13>(num, red, wave, lane);
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
        } while (iter < P.stop && dp_norm >= P.conv);

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
        }
    }
}

}  // namespace

// scratch floats per workgroup slot of the row mapping: body slots (MB steps x NCH chunks x 512) + tail slots (rounded up)
size_t icgn3d1_rows_slot_floats(int rx, int ry, int rz) {
    const int SX = 2 * rx + 1, ROWS = (2 * ry + 1) * (2 * rz + 1), nch = rows_body_chunks(SX);
    const int bw = SX < 32 * nch ? SX : 32 * nch, rt = SX - bw;
    const size_t mb = (size_t)(ROWS + kHalves - 1) / kHalves;
    const size_t nt = (size_t)rt * ROWS;
    return mb * nch * kBlock3d + (nt + kBlock3d - 1) / kBlock3d * kBlock3d;
}

// `blocks` = persistent workgroups = scratch slots of icgn3d1_rows_slot_floats() floats each that p.scratch holds (the caller
// sized the allocation with it: icgn3d1_scratch_floats)
hipError_t launch_icgn3d1_rows(const Icgn3dParams& p, float* pois, int stride_f, size_t count, int blocks, hipStream_t stream) {
    if (count == 0) return hipSuccess;
    if (!p.scratch || blocks < 8) return hipErrorInvalidValue;
    // no row has 28 samples: everything is "tail", i.e. the mapping of icgn3d.hip itself (same bits: OC_ORDER_ROWS == OC_ORDER_LANES there)
    if (rows_body_chunks(2 * p.rx + 1) == 0) return launch_icgn3d1(p, pois, stride_f, count, stream);
    unsigned grid = (unsigned)(count < (size_t)blocks ? count : (size_t)blocks);
    grid = (grid + 7) / 8 * 8;  // whole XCD rounds (idle workgroups exit at once); never more than `blocks` slots
    Icgn3dParams q = p;
    const long long sx = 2 * p.rx + 1, sy = 2 * p.ry + 1, sz = 2 * p.rz + 1;
    const int nch = rows_body_chunks((int)sx);
    const long long bw = sx < 32 * nch ? sx : 32 * nch, rt = sx - bw;
    // body: steps (of 16 rows) per pass so that the nominal coefficient box (small deformation gradients) stays inside the
    // LDS window; passes that still overflow fall back to global taps
    const int px = nch == 1 ? 40 : 0;
    const int tries[] = {16, 12, 10, 8, 6, 4, 3, 2, 1};
    q.samples_per_pass = 1;
    if (nch > 0) {
        for (int m : tries) {
            const long long rows = (long long)m * kHalves;             // subvolume rows of a pass
            const long long planes = (rows + sy - 1) / sy + 1;          // subvolume planes they can touch
            const long long nz = (planes < sz ? planes : sz) + 3 + 1;
            const long long ny = (planes > 1 ? sy : (rows < sy ? rows : sy)) + 3 + 2;
            const long long nx = px ? px : bw + 3 + 2;
            if (nx * ny * nz <= kWinCap) {
                q.samples_per_pass = m;
                break;
            }
        }
    }
    // tail: steps (of 512 tail samples) per pass, box as wide as the tail block
    q.tail_steps_per_pass = 1;
    if (rt > 0) {
        for (int m : tries) {
            const long long len = (long long)m * kBlock3d;
            const long long rows = (len + rt - 1) / rt + 1;
            const long long planes = (rows + sy - 1) / sy + 1;
            const long long nz = (planes < sz ? planes : sz) + 3 + 1;
            const long long ny = (planes > 1 ? sy : (rows < sy ? rows : sy)) + 3 + 2;
            const long long nx = rt + 3 + 2;
            if (nx * ny * nz <= kWinCap) {
                q.tail_steps_per_pass = m;
                break;
            }
        }
    }
    (void)hipGetLastError();  // drop stale errors of earlier, unrelated calls
    switch (nch) {
        case 1: hipLaunchKernelGGL((icgn3d1_rows_kernel<40, 1>), dim3(grid), dim3(kBlock3d), 0, stream, q, pois, stride_f, (unsigned long long)count); break;
        default: hipLaunchKernelGGL((icgn3d1_rows_kernel<0, 2>), dim3(grid), dim3(kBlock3d), 0, stream, q, pois, stride_f, (unsigned long long)count); break;
    }
    return hipGetLastError();
}

}  // namespace ochip
