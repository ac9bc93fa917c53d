// fftcc3d_fused.hip -- FFTCC3D for 32 x 32 x 32 windows (subset radius 16) in ONE kernel.
//
// FFTCC3D::compute(POI3D*) (src/oc_fftcc.cpp:327-427) needs, per POI, two 3-D real FFTs, a spectrum product and one
// inverse FFT of a 128 KB window.  The rocFFT pipeline of fftcc3d.hip moves ~1.6 MB per POI through HBM between its
// kernels (windows, two spectra, product, correlation volume); here one 1024-thread workgroup keeps the POI on chip:
//   gather  ->  z = ref + i*tar (zero-mean)  ->  ONE complex 32^3 FFT  ->
//   R(k) = (Z(k) + conj Z(-k))/2, T(k) = (Z(k) - conj Z(-k))/(2i), C = conj(R) T  ->
//   inverse complex FFT (unnormalised, like FFTW's c2r)  ->  arg-max with the first-max rule, wrap, ZNCC.
// The 32^3 complex volume (256 KB) lives in REGISTERS, one 32-point line per thread (64 VGPRs): every axis pass is a
// 32-point FFT in registers (fft_device.h), and between passes the volume is re-distributed through LDS, half of it
// (16 planes x 32 x 33 complex = 132 KB) at a time:
//   LX: thread (z, y) holds the x-line   --[z-halves]-->   LY: thread (z, x) holds the y-line
//   LY                                   --[y-halves]-->   LZ: thread (y, x) holds the z-line
// The spectrum product needs Z(-k): the z-lines are exchanged through LDS in two halves of equal y-parity (k -> -k
// preserves the parity of every index, so each half is closed under the mirror).  Row pitch 33 complex keeps the
// line-wise accesses of both sides of every exchange conflict-free.
// HBM traffic: the two windows once (256 KB, mostly L2 hits between neighbouring POIs) and 28 bytes of results.
// Integer outputs (u, v, w) are what the reference computes; the float ZNCC differs from FFTW's in the last bits
// like any other FFT implementation (tested to 1e-5 against the oracle's double-precision DFT).
#include "oc_device.h"
#include "fft_device.h"
#include "oc_kernels.h"

// OC_FUSED32_WAVE_XY: 1 = the x <-> y exchanges synchronise inside the half-wave that owns a z-plane (a wave-level fence) and
// keep only the workgroup barrier between the two z-halves; 0 = two workgroup barriers per half (rounds 1 - 3).  Config E:
// 9.08 against 9.32 ms, bit-identical (profiles/r4s_fftcc3d_fused32_ab_wave_local_xy.txt).
#ifndef OC_FUSED32_WAVE_XY
#define OC_FUSED32_WAVE_XY 1
#endif

#if defined(OC_F32_ABL) && (OC_F32_ABL & 1)   // timing experiment (results invalid): no workgroup barriers
#define __syncthreads() __builtin_amdgcn_wave_barrier()
#endif
// OC_FUSED32_HERM (round 6): 1 = the decomposition described under "Round 6" below (mirror-closed waves, the spectrum product's
// partner line fetched with ds_bpermute, the inverse on the kx <= 16 half of the Hermitian product); 0 = rounds 1 - 5.
#ifndef OC_FUSED32_HERM
#define OC_FUSED32_HERM 1
#endif

namespace ochip {

namespace {

using namespace fftdev;

constexpr int TN = 32;                    // window side (2 * radius)
constexpr int TP = TN + 1;                // LDS row pitch in complex elements
constexpr int kThreads3 = TN * TN;        // one line per thread
constexpr int kWaves = kThreads3 / kWave;  // 16
constexpr int kHalf = 16 * TN * TP;       // complex elements of half a volume in LDS
constexpr int HP = TN / 2 + 1;            // kx = 0 ... 16: the half of the Hermitian product the inverse works on; also its row pitch
constexpr int kTileH = TN * TN * HP;      // [z][y][kx <= 16] complex: the whole half-spectrum at once (136 KB)
constexpr int kSlotZ = 16 * TN * HP + 16;  // one slot of the exchange to z-lines: [z & 15][ky][kx & 15 (+1)], the second one skewed by 16 elements
constexpr int kLdsC = OC_FUSED32_HERM ? (2 * kSlotZ > kHalf ? 2 * kSlotZ : kHalf) : kHalf;

typedef float float4u __attribute__((ext_vector_type(4), aligned(4)));  // 4-byte aligned 16-byte load

__device__ __forceinline__ int clampi3(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// orders this wave's earlier LDS writes before its later LDS reads (data exchanged between lanes of ONE wave: the LDS executes
// a wave's instructions in issue order, so all that is needed is that the compiler keeps the order and waits for the writes)
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// two block-wide sums at once; every thread returns the same values.  ONE barrier: `red` must not be in use by an earlier
// call (each call site has its own 2 * kWaves floats)
__device__ __forceinline__ void block_sum2(float& x, float& y, float* red, int lane, int wave) {
    x = wave_allreduce_sum(x);
    y = wave_allreduce_sum(y);
    if (lane == 0) {
        red[wave] = x;
        red[kWaves + wave] = y;
    }
    __syncthreads();
    float sx = 0.f, sy = 0.f;
#pragma unroll
    for (int i = 0; i < kWaves; i++) {
        sx += red[i];
        sy += red[kWaves + i];
    }
    x = sx;
    y = sy;
}

// One POI by one workgroup.  CLAMPED = false (the launch that does the work): windows whose x indices are contiguous in both
// volumes -- every window that is not clamped at a volume border -- gathered with 16-byte loads; a workgroup that finds its
// window clamped only raises needs_clamped[idx] and leaves.  CLAMPED = true (a second, small launch whose workgroups scan the
// flags): the flagged POIs, gathered element by element through the index tables.  Two instantiations because the scalar
// gather's 64 addresses raise the register pressure of the WHOLE kernel when both paths live in one: 104 B of scratch per
// thread against 76 B for the contiguous-only instantiation -- and at 1 024 threads x 50 000 POIs every scratch byte is
// 50 MB written to memory and read back (round 5, profiles/r5e_fftcc3d_block_schedule_ab.json: WRITE_SIZE 4.98 GB per launch
// for 28 B of results per POI).
template <bool CLAMPED>
__device__ __forceinline__ void fftcc3d_fused32_poi(const Fftcc3dParams& P, float* __restrict__ pois, int stride_f, unsigned long long idx,
                                                    unsigned char* __restrict__ needs_clamped, unsigned* __restrict__ any_clamped) {
    __shared__ c2 lds[kLdsC];
    __shared__ int tab[6][TN];  // voxel index of window coordinate k: ref x, y, z, tar x, y, z
    __shared__ float red[2 * kWaves], red2[2 * kWaves];
    __shared__ int redi[kWaves];
    __shared__ float norms[2];  // sums of squares of the two windows: formed early, needed by thread 0 at the very end
    const int tid = threadIdx.x;
    const int a = tid >> 5, b = tid & 31;
    const int lane = tid & (kWave - 1), wave = tid >> 6;
    float* poi = pois + idx * (unsigned long long)stride_f;
    constexpr int R = TN / 2;
    constexpr int M = TN * TN * TN;
#if defined(OC_F32_TIMELINE)   // timing experiment (A/B builds): thread 0 leaves the cycles (s_memtime) between its marks in the record's unused fields 19 ... 30
    unsigned long long tl_mark = __builtin_readcyclecounter();
    int tl_slot = 19;
    auto stamp = [&]() {
        const unsigned long long now = __builtin_readcyclecounter();
        if (tid == 0 && tl_slot < 31) poi[tl_slot] = (float)(now - tl_mark);
        tl_slot++;
        tl_mark = now;
    };
#else
    auto stamp = [] {};
#endif

    // ---- window coordinates -> voxel indices (src/oc_fftcc.cpp:349-358: Point3D(poi->x + k - rx, ...) truncated, the
    // target window displaced by the initial guess).  Separable, so one table per axis and window.  The reference has
    // no bounds guard in 3D; indices are clamped like in fftcc3d_gather_kernel.
    constexpr bool kOwnIndices = OC_FUSED32_HERM && !CLAMPED;   // (round 6) no table, no barrier: see below
    auto voxel = [&](int axis, int k) {   // axis 0 ... 2: reference x, y, z; 3 ... 5: target
        const int which = axis % 3;
        const float p = poi[which == 0 ? poi3d::X : which == 1 ? poi3d::Y : poi3d::Z];
        const float g = poi[which == 0 ? poi3d::U : which == 1 ? poi3d::V : poi3d::W];
        const int D = which == 0 ? P.dx : which == 1 ? P.dy : P.dz;
        float c = p + k - R;
        if (axis >= 3) c = c + g;
        return clampi3((int)c, 0, D - 1);
    };
    if constexpr (!kOwnIndices) {
        if (tid < 6 * TN) tab[tid >> 5][tid & 31] = voxel(tid >> 5, tid & 31);
    }
    // both windows' x indices contiguous (true unless a window is clamped at the border): 16-byte loads.  Every thread forms
    // the four table entries the test needs itself, so that the vote's barrier is also the one that publishes `tab`
    const bool mine_contig = voxel(0, b) == voxel(0, 0) + b && voxel(3, b) == voxel(3, 0) + b;
    bool contig;
    if constexpr (kOwnIndices) {
        // the test depends on b alone and every wave holds every b: a vote inside the wave gives the workgroup's answer, and a
        // thread needs six table entries only, which it forms itself -- neither table nor barrier
        contig = __builtin_amdgcn_ballot_w64(mine_contig) == ~0ull;
    } else {
        contig = __syncthreads_and(mine_contig) != 0;
    }
    if constexpr (!CLAMPED) {
        if (tid == 0) {
            needs_clamped[idx] = contig ? 0 : 1;
            if (!contig) atomicOr(any_clamped, 1u);
        }
        if (!contig) return;
    }

#if defined(OC_F32_ABL) && (OC_F32_ABL & 2)   // timing experiment: what launching 50 000 sixteen-wave workgroups with 137 KB of LDS costs by itself
    if (tid == 0) lds[0] = mkc(poi[poi3d::X], 0.f);
    if (P.dx > 0) return;
#endif
#if defined(OC_F32_TIMELINE) && OC_F32_TIMELINE == 2
    stamp();   // fine: record loads, indices, vote
#endif
#if OC_FUSED32_HERM
    // ---- gather (round 6): thread (z = a, x = b) reads its Y-line of both windows; z = ref + i*tar.  A load instruction of a wave
    // then covers two 128-byte rows (the 32 lanes of a half-wave sit side by side in x) instead of 64 rows, one per lane: the
    // texture path handles one cache line per cycle whatever the lanes take from it, and the x-line form cost 27 % of the kernel
    // there (profiles/r6o_*).  The y-pass therefore comes first, the x-pass second.
    c2 v[TN];
    {
        const int my_ry = kOwnIndices ? voxel(1, b) : tab[1][b], my_ty = kOwnIndices ? voxel(4, b) : tab[4][b];   // lane y holds row y's index
        const int rz = kOwnIndices ? voxel(2, a) : tab[2][a], tz = kOwnIndices ? voxel(5, a) : tab[5][a];
        const int rx = CLAMPED ? tab[0][b] : voxel(0, 0) + b, tx = CLAMPED ? tab[3][b] : voxel(3, 0) + b;
        const float* __restrict__ rp = P.ref + (size_t)rz * P.dy * P.dx + rx;
        const float* __restrict__ tp = P.tar + (size_t)tz * P.dy * P.dx + tx;
#pragma unroll
        for (int y = 0; y < TN; y++) {
            const int ry = __builtin_amdgcn_readlane(my_ry, y), ty = __builtin_amdgcn_readlane(my_ty, y);   // wave-uniform row offsets
#if defined(OC_F32_ABL) && (OC_F32_ABL & 4)   // timing experiment (results invalid): every thread gathers the same two rows
            v[y] = mkc(P.ref[y], P.tar[y]);
#else
            v[y] = mkc(rp[(size_t)ry * P.dx], tp[(size_t)ty * P.dx]);
#endif
        }
    }
#else
    // ---- gather: thread (z = a, y = b) reads its x-line of both windows; z = ref + i*tar
    c2 v[TN];
    {
        const int rz = kOwnIndices ? voxel(2, a) : tab[2][a], ry = kOwnIndices ? voxel(1, b) : tab[1][b];
        const int tz = kOwnIndices ? voxel(5, a) : tab[5][a], ty = kOwnIndices ? voxel(4, b) : tab[4][b];
        const float* rrow = P.ref + ((size_t)rz * P.dy + ry) * P.dx;
        const float* trow = P.tar + ((size_t)tz * P.dy + ty) * P.dx;
#if defined(OC_F32_ABL) && (OC_F32_ABL & 4)   // timing experiment (results invalid): every thread gathers the same two rows
        rrow = P.ref; trow = P.tar;
#endif
        if (!CLAMPED) {
            const float* __restrict__ rp = rrow + (kOwnIndices ? voxel(0, 0) : tab[0][0]);
            const float* __restrict__ tp = trow + (kOwnIndices ? voxel(3, 0) : tab[3][0]);
#pragma unroll
            for (int q = 0; q < TN / 4; q++) {
                const float4u r4 = *reinterpret_cast<const float4u*>(rp + 4 * q);
                const float4u t4 = *reinterpret_cast<const float4u*>(tp + 4 * q);
                v[4 * q + 0] = mkc(r4.x, t4.x);
                v[4 * q + 1] = mkc(r4.y, t4.y);
                v[4 * q + 2] = mkc(r4.z, t4.z);
                v[4 * q + 3] = mkc(r4.w, t4.w);
            }
        } else {
#pragma unroll
            for (int k = 0; k < TN; k++) v[k] = mkc(rrow[tab[0][k]], trow[tab[3][k]]);
        }
    }
#endif
    // means, zero-mean, sums of squares (src/oc_fftcc.cpp:360-376)
    {
        float rn, tn;
        float rs = 0.f, ts = 0.f;
#pragma unroll
        for (int k = 0; k < TN; k++) {
            rs += v[k].x;
            ts += v[k].y;
        }
#if defined(OC_F32_TIMELINE) && OC_F32_TIMELINE == 2
        stamp();   // fine: gather arrived, 64 additions
#endif
        block_sum2(rs, ts, red, lane, wave);
#if defined(OC_F32_TIMELINE) && OC_F32_TIMELINE == 2
        stamp();   // fine: wave sums, barrier, 32 LDS reads
#endif
        const c2 mean = mkc(rs / M, ts / M);
        rn = 0.f;
        tn = 0.f;
#pragma unroll
        for (int k = 0; k < TN; k++) {
            v[k] = v[k] - mean;
            rn += v[k].x * v[k].x;
            tn += v[k].y * v[k].y;
        }
#if OC_FUSED32_HERM
        // needed only at the very end, by thread 0: the waves' partial sums are parked in LDS now and added up there, in
        // block_sum2's order, behind the arg-max barrier -- no barrier of their own
        rn = wave_allreduce_sum(rn);
        tn = wave_allreduce_sum(tn);
        if (lane == 0) {
            red2[wave] = rn;
            red2[kWaves + wave] = tn;
        }
    }
#else
        block_sum2(rn, tn, red2, lane, wave);
        // needed only at the very end, by thread 0: parked in LDS instead of two registers of every thread
        if (tid == 0) {
            norms[0] = rn;
            norms[1] = tn;
        }
    }
#endif

    stamp();   // 19: indices, gather, means, zero-mean, sums of squares
    const int zz = a & 15, half = a >> 4;
#if OC_FUSED32_HERM
    // ---- Round 6.  What changed against rounds 1 - 5 (the #else branch), and why: 22 % of the kernel were its 22 workgroup
    // barriers (one 16-wave workgroup per CU: nothing else runs while they drain; profiles/r6o_*), another half LDS traffic.
    //  (1) The thread that transforms the z-line (ky, kx) is chosen so that a WAVE is closed under k -> -k: wave W holds
    //      ky = W and 32 - W (wave 0: ky = 0 and 16), all kx.  Z(-k) for the spectrum product then sits in a lane of the same
    //      wave and comes through ds_bpermute: no LDS memory, no barrier (was: a full-volume exchange, 4 barriers).
    //  (2) The two half-waves of such a wave have DIFFERENT y-halves, which would turn the register index of the 16 x 16 block
    //      scheme into a per-lane select.  Instead a thread of the upper y-half keeps its z-line rotated by 16: round d fills
    //      w[16 d ...] for every lane, the transform of the rotated line is (-1)^kz times the transform of the line -- exactly,
    //      bit for bit: the first butterfly level pairs n with n + 16, a + b commutes, a - b changes sign, and everything
    //      behind it is odd in its input -- and a sign flip of the odd outputs restores it.
    //  (3) conj(R) T is Hermitian, so the inverse needs the lines kx = 0 ... 16 only; after the z and y passes
    //      D(z, y, -kx) = conj D(z, y, kx) completes the x-line inside the thread.  The half-spectrum [z][y][kx <= 16] fits the
    //      LDS at once: one barrier for z -> y, the y-pass writes its column back IN PLACE, and y -> x stays inside the
    //      half-wave that owns the z-plane.  (The correlation values change in their last bits; the arg-max does not.)
    //  (4) The sums of squares wait in LDS for thread 0 (above).
    // Barriers: 8 (means, x -> y done, 4 in y -> z, z -> y, arg-max).  LDS accesses per thread: 128 four-byte ones, 64 + (32 + 32 + 32 on
    // 17 of 32 lanes) + 17 eight-byte ones, and 64 ds_bpermute.
    // ---- forward y (thread (z = a, x = b)), then LY -> LX inside the half-wave that owns plane z = a
    fft32<false>(v);
    stamp();   // 20: forward y
    {
        // (5) real parts first, then imaginary parts: 32 planes x 32 x 33 FLOATS are the 132 KB that held 16 complex planes, so
        // every half-wave has a slot of its own for its plane and the whole exchange needs no workgroup barrier
        float* __restrict__ ldsf = reinterpret_cast<float*>(lds);
#pragma unroll
        for (int k = 0; k < TN; k++) ldsf[(a * TN + b) * TP + k] = v[bitrev5(k)].x;
        wave_lds_fence();
#pragma unroll
        for (int j = 0; j < TN; j++) v[j].x = ldsf[(a * TN + j) * TP + b];
        wave_lds_fence();
#pragma unroll
        for (int k = 0; k < TN; k++) ldsf[(a * TN + b) * TP + k] = v[bitrev5(k)].y;
        wave_lds_fence();
#pragma unroll
        for (int j = 0; j < TN; j++) v[j].y = ldsf[(a * TN + j) * TP + b];
    }
    stamp();   // 21: y -> x exchange (real planes, imaginary planes)
    // ---- forward x (thread (z = a, ky = b)), then -> LZ in 16 x 16 blocks of the (kx, z) plane: slot s is written by the z-half s
    // (round d: its kx-half s ^ d) as [z & 15][ky][kx & 15], row pitch 17; the READER is thread (ky(a), kx = b) of the
    // mirror-closed layout, whose kx-half p = b >> 4 differs from lane to lane: it finds its block in slot p ^ d (the second slot
    // starts 16 elements late, so that the two 16-lane groups of a half-wave meet different banks)
    fft32<false>(v);
    __syncthreads();   // every plane has left its slot
    stamp();   // 22: forward x, barrier
    const int mw = a >> 1, ms = a & 1;
    const int ky = mw == 0 ? (ms << 4) : (ms ? TN - mw : mw);
    const int xh = b >> 4, xl = b & 15;
    c2 w[TN];
#pragma unroll
    for (int d = 0; d < 2; d++) {
        {   // writer (z = a, ky = b): block (kx-half = half ^ d, z-half = half) -> slot `half`
            c2* __restrict__ dst = lds + half * kSlotZ + (zz * TN + b) * HP;
            if ((half ^ d) == 0) {
#pragma unroll
                for (int x = 0; x < 16; x++) dst[x] = v[bitrev5(x)];
            } else {
#pragma unroll
                for (int x = 0; x < 16; x++) dst[x] = v[bitrev5(x + 16)];
            }
        }
        __syncthreads();
        {   // reader (ky, kx = b): block (kx-half = xh, z-half = xh ^ d) sits in slot xh ^ d; it becomes w[16 d ...]: z = i ^ (16 xh)
            const c2* __restrict__ src = lds + (xh ^ d) * kSlotZ + ky * HP + xl;
#pragma unroll
            for (int z = 0; z < 16; z++) w[z + 16 * d] = src[z * TN * HP];
        }
        __syncthreads();
    }
    stamp();   // 23: -> z exchange (two rounds, four barriers)
    // ---- forward z: Z(kz, ky, kx = b) in w[bitrev5(kz)] (the odd kz -- registers 16 ... 31 -- with the sign of (2))
    fft32<false>(w);
    stamp();   // 24: forward z
    {
        const unsigned flip = xh ? 0x80000000u : 0u;
#pragma unroll
        for (int i = 16; i < TN; i++) w[i] = mkc(__uint_as_float(__float_as_uint(w[i].x) ^ flip), __uint_as_float(__float_as_uint(w[i].y) ^ flip));
    }
    // ---- spectra of the two real windows and their product conj(R) * T (src/oc_fftcc.cpp:378-386); Z(-k) is register
    // bitrev5(-kz) of the lane that holds line (-ky, -kx)
    {
        const int plane = ((mw == 0 ? ms : (ms ^ 1)) << 5) | ((TN - b) & (TN - 1));
        const int paddr = plane << 2;
        auto from_partner = [&](c2 x) {
            return mkc(__int_as_float(__builtin_amdgcn_ds_bpermute(paddr, __float_as_int(x.x))),
                       __int_as_float(__builtin_amdgcn_ds_bpermute(paddr, __float_as_int(x.y))));
        };
        auto product = [](c2 zk, c2 zm) {
            const float rr = 0.5f * (zk.x + zm.x), ri = 0.5f * (zk.y - zm.y);
            const float tr = 0.5f * (zk.y + zm.y), ti = -0.5f * (zk.x - zm.x);
            return mkc((rr * tr) + (ri * ti), (rr * ti) - (ri * tr));
        };
#pragma unroll
        for (int z = 0; z <= TN / 2; z++) {
            const int zc = (TN - z) & (TN - 1);            // -kz
            const c2 zk = w[bitrev5(z)], zk2 = w[bitrev5(zc)];
            const c2 zm = from_partner(zk2);               // the partner's Z at -kz: what this lane needs at kz = z
            if (zc != z) {
                const c2 zm2 = from_partner(zk);           // ... and at kz = -z
                w[bitrev5(zc)] = product(zk2, zm2);
            }
            w[bitrev5(z)] = product(zk, zm);
        }
    }
    stamp();   // 25: sign flip, partner values through ds_bpermute, spectrum product
    // ---- inverse z on the lines kx <= 16, LZ -> LY through the half-spectrum tile [z][ky][kx]
    if (b < HP) {
        c2 t[TN];
#pragma unroll
        for (int k = 0; k < TN; k++) t[k] = w[bitrev5(k)];
        fft32<true>(t);
#pragma unroll
        for (int z = 0; z < TN; z++) lds[(z * TN + ky) * HP + b] = t[bitrev5(z)];
    }
    __syncthreads();
    stamp();   // 26: inverse z (kx <= 16), tile write, barrier
    // ---- inverse y by thread (z = a, kx = b <= 16), written back in place (every thread owns its column)
    if (b < HP) {
        c2 u[TN];
#pragma unroll
        for (int j = 0; j < TN; j++) u[j] = lds[(a * TN + j) * HP + b];
        fft32<true>(u);
#pragma unroll
        for (int j = 0; j < TN; j++) lds[(a * TN + j) * HP + b] = u[bitrev5(j)];
    }
    wave_lds_fence();
    // ---- LY -> LX inside the half-wave that owns plane z = a; D(z, y, 32 - kx) = conj D(z, y, kx); inverse x
    c2 q[TN];
    {
        const c2* __restrict__ row = lds + (a * TN + b) * HP;
#pragma unroll
        for (int k = 0; k < HP; k++) q[k] = row[k];
#pragma unroll
        for (int k = 1; k < TN / 2; k++) q[TN - k] = mkc(q[k].x, -q[k].y);
    }
    stamp();   // 27: inverse y on the column, written back, fence, row read + Hermitian completion
    fft32<true>(q);
    stamp();   // 28: inverse x

#else
    // ---- forward x, then LX -> LY through LDS [z & 15][y][x], one z-half at a time
    fft32<false>(v);
    // (a z-plane is written and read by the 32 threads of ONE half-wave: inside the plane a wave-level fence orders its
    // LDS writes before its reads -- the LDS serves a wave's instructions in order -- and the workgroup barrier is only
    // needed where the two z-halves hand the 16 plane slots over)
#pragma unroll
    for (int h = 0; h < 2; h++) {
        if (half == h) {
#pragma unroll
            for (int k = 0; k < TN; k++) lds[(zz * TN + b) * TP + k] = v[bitrev5(k)];
#if OC_FUSED32_WAVE_XY
            wave_lds_fence();
#else
        }
        __syncthreads();
        if (half == h) {
#endif
#pragma unroll
            for (int j = 0; j < TN; j++) v[j] = lds[(zz * TN + j) * TP + b];
        }
        __syncthreads();
    }
    // ---- forward y (thread (z = a, x = b)), then LY -> LZ.  The (y, z) plane of every x is cut into four 16 x 16 blocks
    // (y-half, z-half); the tile holds two of them: [z-half][y & 15][z & 15][x].  Round 0 moves the DIAGONAL blocks, round 1 the
    // off-diagonal ones: a thread (z-half = its own `half` as a writer, y-half = `half` as a reader) hands over 16 values and
    // receives 16 values per round, so it never holds more than one line's worth of data (64 registers).  Round 5: until
    // then round g moved y-half g -- every thread wrote 16 values, half of the threads read 32 -- and a thread that had
    // received its whole z-line in round 0 still held the 16 values it owed round 1: 96 live data registers of the 128 the
    // workgroup size leaves, 16 of them in scratch (64 of the kernel's 76 B per thread).  `half` is uniform over a wave, so
    // the two register-index patterns are two branches, not selects.
    fft32<false>(v);
    c2 w[TN];
#pragma unroll
    for (int d = 0; d < 2; d++) {
        {   // writer (z = a): block (y-half = half ^ d, z-half = half) -> slot `half`
            c2* __restrict__ dst = lds + ((half * 16) * 16 + zz) * TP + b;
            if ((half ^ d) == 0) {
#pragma unroll
                for (int yy = 0; yy < 16; yy++) dst[yy * 16 * TP] = v[bitrev5(yy)];
            } else {
#pragma unroll
                for (int yy = 0; yy < 16; yy++) dst[yy * 16 * TP] = v[bitrev5(yy + 16)];
            }
        }
        __syncthreads();
        {   // reader (y = a): block (y-half = half, z-half = half ^ d) -> slot `half ^ d`
            const c2* __restrict__ src = lds + (((half ^ d) * 16 + zz) * 16) * TP + b;
            if ((half ^ d) == 0) {
#pragma unroll
                for (int z = 0; z < 16; z++) w[z] = src[z * TP];
            } else {
#pragma unroll
                for (int z = 0; z < 16; z++) w[z + 16] = src[z * TP];
            }
        }
        __syncthreads();
    }
    // ---- forward z (thread (y = a, x = b)): Z(kz, ky = a, kx = b) in w[bitrev5(kz)]
    fft32<false>(w);
    // ---- spectra of the two real windows and their product conj(R) * T (src/oc_fftcc.cpp:378-386); Z(-k) comes
    // from the thread that owns line (-ky, -kx), through LDS [ky >> 1][kx][kz], one ky-parity at a time
#pragma unroll
    for (int p = 0; p < 2; p++) {
        if ((a & 1) == p) {
#pragma unroll
            for (int z = 0; z < TN; z++) lds[((a >> 1) * TN + b) * TP + z] = w[bitrev5(z)];
        }
        __syncthreads();
        if ((a & 1) == p) {
            const int my = (TN - a) & (TN - 1), mx = (TN - b) & (TN - 1);
            const c2* __restrict__ mline = lds + ((my >> 1) * TN + mx) * TP;
#pragma unroll
            for (int z = 0; z < TN; z++) {
                const c2 zm = mline[(TN - z) & (TN - 1)];
                const c2 zk = w[bitrev5(z)];
                const float rr = 0.5f * (zk.x + zm.x), ri = 0.5f * (zk.y - zm.y);
                const float tr = 0.5f * (zk.y + zm.y), ti = -0.5f * (zk.x - zm.x);
                w[bitrev5(z)] = mkc((rr * tr) + (ri * ti), (rr * ti) - (ri * tr));
            }
        }
        __syncthreads();
    }
    // ---- inverse z (natural-order input is a renaming of registers), LZ -> LY
    c2 t[TN];
#pragma unroll
    for (int k = 0; k < TN; k++) t[k] = w[bitrev5(k)];
    fft32<true>(t);
    c2 u[TN];
    // (the same two rounds of 16 x 16 blocks the other way round: writer y = a hands over its z-half `half ^ d`, reader z = a
    // receives its y-half `half ^ d`)
#pragma unroll
    for (int d = 0; d < 2; d++) {
        {   // writer (y = a): block (y-half = half, z-half = half ^ d) -> slot `half ^ d`
            c2* __restrict__ dst = lds + (((half ^ d) * 16 + zz) * 16) * TP + b;
            if ((half ^ d) == 0) {
#pragma unroll
                for (int z = 0; z < 16; z++) dst[z * TP] = t[bitrev5(z)];
            } else {
#pragma unroll
                for (int z = 0; z < 16; z++) dst[z * TP] = t[bitrev5(z + 16)];
            }
        }
        __syncthreads();
        {   // reader (z = a): block (y-half = half ^ d, z-half = half) -> slot `half`
            const c2* __restrict__ src = lds + ((half * 16) * 16 + zz) * TP + b;
            if ((half ^ d) == 0) {
#pragma unroll
                for (int yy = 0; yy < 16; yy++) u[yy] = src[yy * 16 * TP];
            } else {
#pragma unroll
                for (int yy = 0; yy < 16; yy++) u[yy + 16] = src[yy * 16 * TP];
            }
        }
        __syncthreads();
    }
    // ---- inverse y (thread (z = a, x = b)), LY -> LX
    fft32<true>(u);
    c2 q[TN];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        if (half == h) {
#pragma unroll
            for (int j = 0; j < TN; j++) lds[(zz * TN + j) * TP + b] = u[bitrev5(j)];
#if OC_FUSED32_WAVE_XY
            wave_lds_fence();
#else
        }
        __syncthreads();
        if (half == h) {
#endif
#pragma unroll
            for (int k = 0; k < TN; k++) q[k] = lds[(zz * TN + b) * TP + k];
        }
        __syncthreads();
    }
    // ---- inverse x (thread (z = a, y = b)): the correlation volume, real part
    fft32<true>(q);

#endif
    // ---- arg-max with "strict >, scanning from index 0" (src/oc_fftcc.cpp:391-400): the thread's 32 values sit at
    // linear indices (a*32 + b)*32 + x, ascending in x
    float best = -2.f;
    int bidx = 0;
#pragma unroll
    for (int x = 0; x < TN; x++) {
        const float val = q[bitrev5(x)].x;
        if (val > best) {
            best = val;
            bidx = (a * TN + b) * TN + x;
        }
    }
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const float ov = __shfl_xor(best, off, kWave);
        const int oi = __shfl_xor(bidx, off, kWave);
        if (ov > best || (ov == best && oi < bidx)) {
            best = ov;
            bidx = oi;
        }
    }
    if (lane == 0) {
        red[wave] = best;
        redi[wave] = bidx;
    }
    __syncthreads();
    stamp();   // 29: arg-max inside the thread and the wave, barrier
    if (tid == 0) {
        for (int i = 1; i < kWaves; i++)
            if (red[i] > best || (red[i] == best && redi[i] < bidx)) {
                best = red[i];
                bidx = redi[i];
            }
        int du = bidx % TN, dv = (bidx / TN) % TN, dw = bidx / (TN * TN);  // src/oc_fftcc.cpp:401-403
        if (du > R) du -= TN;
        if (dv > R) dv -= TN;
        if (dw > R) dw -= TN;
        const float gu = poi[poi3d::U], gv = poi[poi3d::V], gw = poi[poi3d::W];
        poi[poi3d::U] = (float)du + gu;
        poi[poi3d::V] = (float)dv + gv;
        poi[poi3d::W] = (float)dw + gw;
        poi[poi3d::U0] = gu;
        poi[poi3d::V0] = gv;
        poi[poi3d::W0] = gw;
#if OC_FUSED32_HERM
        float rn = 0.f, tn = 0.f;
        for (int i = 0; i < kWaves; i++) {
            rn += red2[i];
            tn += red2[kWaves + i];
        }
        poi[poi3d::ZNCC] = best / (sqrtf(rn * tn) * M);
#else
        poi[poi3d::ZNCC] = best / (sqrtf(norms[0] * norms[1]) * M);
#endif
    }
}

// the launch that does the work: workgroup -> POI (XCD-contiguous ranges of the visiting order)
__global__ __launch_bounds__(kThreads3, 4) void fftcc3d_fused32_kernel(Fftcc3dParams P, float* __restrict__ pois, int stride_f,
                                                                      unsigned long long count, int xcd_chunk,
                                                                      unsigned char* __restrict__ needs_clamped) {
    unsigned long long idx = blockIdx.x;
    if (xcd_chunk > 0) idx = (unsigned long long)(blockIdx.x & 7u) * xcd_chunk + (blockIdx.x >> 3);
    if (idx >= count) return;
    if (P.perm) idx = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)P.perm[idx]);  // wave-uniform: the record's address stays in SGPRs
    fftcc3d_fused32_poi<false>(P, pois, stride_f, idx, needs_clamped, reinterpret_cast<unsigned*>(needs_clamped + ((count + 3) & ~3ull)));
}

// the windows clamped at a volume border: a few persistent workgroups scan the flags the first launch left (normally none is set)
__global__ __launch_bounds__(kThreads3, 4) void fftcc3d_fused32_clamped_kernel(Fftcc3dParams P, float* __restrict__ pois, int stride_f,
                                                                              unsigned long long count,
                                                                              unsigned char* __restrict__ needs_clamped) {
    // (the first launch raises the word behind the flags when ANY window was clamped -- normally none: nothing to scan)
    if (*reinterpret_cast<const unsigned*>(needs_clamped + ((count + 3) & ~3ull)) == 0u) return;
    for (unsigned long long idx = blockIdx.x; idx < count; idx += gridDim.x) {
        if (needs_clamped[idx]) {   // uniform over the workgroup
            fftcc3d_fused32_poi<true>(P, pois, stride_f, idx, needs_clamped, nullptr);
            __syncthreads();        // the next POI reuses the tables and the tile
        }
    }
}

}  // namespace

bool fftcc3d_fused_supported(int rx, int ry, int rz) { return rx == TN / 2 && ry == TN / 2 && rz == TN / 2; }

// needs_clamped: fftcc3d_fused_flag_bytes(count) bytes of device scratch (one flag per POI of the queue + one "any" word)
size_t fftcc3d_fused_flag_bytes(size_t count) { return ((count + 3) & ~(size_t)3) + 4; }

hipError_t launch_fftcc3d_fused(const Fftcc3dParams& p, float* pois, int stride_f, size_t count, bool xcd, unsigned char* needs_clamped,
                                hipStream_t stream) {
    if (count == 0) return hipSuccess;
    if (!fftcc3d_fused_supported(p.rx, p.ry, p.rz) || !needs_clamped) return hipErrorInvalidValue;
    const int chunk = xcd ? (int)((count + 7) / 8) : 0;
    const size_t grid = xcd ? (size_t)chunk * 8 : count;
    (void)hipGetLastError();  // drop stale errors of earlier, unrelated calls
    hipError_t err = hipMemsetAsync(needs_clamped + ((count + 3) & ~(size_t)3), 0, 4, stream);
    if (err != hipSuccess) return err;
    hipLaunchKernelGGL(fftcc3d_fused32_kernel, dim3((unsigned)grid), dim3(kThreads3), 0, stream, p, pois, stride_f,
                       (unsigned long long)count, chunk, needs_clamped);
    err = hipGetLastError();
    if (err != hipSuccess) return err;   // (nothing to scan behind a launch that failed)
    const unsigned scan = (unsigned)(count < 256 ? count : 256);
    hipLaunchKernelGGL(fftcc3d_fused32_clamped_kernel, dim3(scan), dim3(kThreads3), 0, stream, p, pois, stride_f,
                       (unsigned long long)count, needs_clamped);
    return hipGetLastError();
}

}  // namespace ochip
