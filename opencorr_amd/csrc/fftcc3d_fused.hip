// fftcc3d_fused.hip -- FFTCC3D for 32 x 32 x 32 windows (subset radius 16) in ONE kernel.
//
// FFTCC3D::compute(POI3D*) (src/oc_fftcc.cpp:327-427) needs, per POI, two 3-D real FFTs, a spectrum product and one
// inverse FFT of a 128 KB window.  The rocFFT pipeline of fftcc3d.hip moves ~1.6 MB per POI through HBM between its
// kernels (windows, two spectra, product, correlation volume); here one 1024-thread workgroup keeps the POI on chip:
//   gather  ->  z = ref + i*tar (zero-mean)  ->  ONE complex 32^3 FFT  ->
//   R(k) = (Z(k) + conj Z(-k))/2, T(k) = (Z(k) - conj Z(-k))/(2i), C = conj(R) T  ->
//   inverse FFT of the Hermitian C (unnormalised, like FFTW's c2r)  ->  arg-max with the first-max rule, wrap, ZNCC.
// The 32^3 complex volume (256 KB) lives in REGISTERS, one 32-point line per thread (64 VGPRs): every axis pass is a
// 32-point FFT in registers (fft_device.h); between passes the volume is re-distributed through LDS.  One such workgroup fills a
// CU (half of its registers, 137 KB of its LDS), so nothing else runs while it waits: what the kernel costs is decided by
// how often it waits and on what.  Round 6 rebuilt the decomposition around that (rounds 1 - 5: fftcc3d_fused_r5.hip in the
// A/B build; 8.27 -> 5.7 ms on config E's 50 653 POIs, same integers, ZNCC within 4e-7; profiles/r6o_*):
//
//  (1) GATHER.  Thread (z, x) reads its Y-line: the 32 lanes of a half-wave sit side by side in x, so a wave's load instruction
//      covers two 128-byte rows.  (Rounds 1 - 5: thread (z, y) read its x-line with 16-byte loads -- 64 rows per instruction, one
//      per lane; the texture path handles about one cache line per cycle whatever the lanes take from it: 27 % of the kernel.)
//      The y-pass therefore comes first, the x-pass second.
//  (2) y -> x inside a z-plane: the 32 threads of ONE half-wave own the plane, so a wave-level fence orders its LDS writes before
//      its reads.  Real parts first, then imaginary parts: 32 planes x 32 x 33 FLOATS are 132 KB, every half-wave has a slot of
//      its own and the whole exchange needs no workgroup barrier (rounds 1 - 5: 16 complex plane slots, two z-halves, 2 barriers).
//  (3) -> z-lines.  The thread that transforms the z-line (ky, kx) is chosen so that a WAVE is closed under k -> -k: wave W
//      holds ky = W and 32 - W (wave 0: ky = 0 and 16) and all kx.  Z(-k) for the spectrum product then sits in a lane of the
//      same wave and comes through ds_bpermute: no LDS memory, no barrier (rounds 1 - 5: a fifth full-volume exchange, 4 barriers).
//  (4) The volume does not fit the LDS, so the exchange to z-lines moves 16 x 16 blocks of the (kx, z) plane in two rounds (the
//      diagonal blocks, then the others): a thread hands over 16 values and receives 16 per round and never holds more than one
//      line.  A reader's kx-half differs from lane to lane, which would turn its register index into a per-lane select; instead a
//      lane of the upper kx-half keeps its z-line ROTATED by 16 -- round d fills w[16 d ...] in every lane.  The transform of the
//      rotated line is (-1)^kz times the transform of the line, exactly, bit for bit: the first butterfly level pairs n with
//      n + 16, a + b commutes, a - b changes sign, and everything behind it is odd in its input.  A sign flip of the odd outputs
//      restores it.
//  (5) conj(R) T is Hermitian (exactly: swapping k and -k in the formulas conjugates the result bit for bit), so the inverse
//      needs the lines kx = 0 ... 16 only; after its z and y passes D(z, y, -kx) = conj D(z, y, kx) completes the x-line inside
//      the thread.  The half-spectrum [z][ky][kx <= 16] fits the LDS at once: ONE barrier for z -> y, the y-pass writes its
//      column back in place, and y -> x stays inside the half-wave that owns the z-plane (rounds 1 - 5: 4 + 2 barriers).
//  (6) Window indices: a thread needs six, which it forms itself (rounds 1 - 5: a table in LDS, a workgroup vote on "is a window
//      clamped in x" and a second launch for those windows, whose 16-byte loads needed contiguous x; with one lane per x a lane
//      reads the voxel its own clamped index names: one launch, no table, no flags).  The sums of squares wait in LDS for thread
//      0, which adds them up behind the arg-max barrier (rounds 1 - 5: a barrier of their own).  The factors 1/2 of R and T are
//      left out and the maximum is scaled by 1/4 at the end (exact: powers of two).
// Barriers: 8 (means; x-pass done; 4 in the exchange to z-lines; z -> y; arg-max) instead of 22.  LDS accesses per thread: 128
// four-byte ones, 64 + (96 on 17 of 32 lanes) + 17 eight-byte ones and 64 ds_bpermute, instead of 320 eight-byte ones.
//
// HBM traffic: the two windows once (256 KB, mostly L2 hits between neighbouring POIs) and 28 bytes of results.
// Integer outputs (u, v, w) are what the reference computes; the float ZNCC differs from FFTW's in the last bits
// like any other FFT implementation (tested to 1e-5 against the oracle's double-precision DFT).
#include "oc_device.h"
#include "fft_device.h"
#include "oc_kernels.h"

#if defined(OC_F32_TIMELINE) && !OC_BUILD_AB
#error "OC_F32_TIMELINE is a timing experiment of the A/B builds (tools/ab_build.py ... -DOC_BUILD_AB=1)"
#endif

namespace ochip {

namespace {

using namespace fftdev;

constexpr int TN = 32;                     // window side (2 * radius)
constexpr int TP = TN + 1;                 // row pitch (floats) of the plane slots of the y -> x exchange
constexpr int kThreads3 = TN * TN;         // one line per thread
constexpr int kWaves = kThreads3 / kWave;  // 16
constexpr int HP = TN / 2 + 1;             // kx = 0 ... 16: the half of the Hermitian product the inverse works on; also a row pitch
constexpr int kPlanesF = TN * TN * TP;     // FLOATS: 32 plane slots [y][x (+1)] of the y -> x exchange (132 KB)
constexpr int kSlotZ = 16 * TN * HP + 16;  // complex: one slot of the exchange to z-lines, [z & 15][ky][kx & 15 (+1)]; the second one starts 16 elements late
constexpr int kTileH = TN * TN * HP;       // complex: the half-spectrum [z][ky][kx <= 16] of the inverse, all of it at once (136 KB)
constexpr int kLdsC = 2 * kSlotZ;          // complex elements of the LDS tile
static_assert(kLdsC >= kTileH && 2 * kLdsC >= kPlanesF, "the three uses of the tile");

__device__ __forceinline__ int clampi3(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// orders this wave's earlier LDS writes before its later LDS reads (data exchanged between lanes of ONE wave: the LDS executes
// a wave's instructions in issue order, so all that is needed is that the compiler keeps the order and waits for the writes)
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// two block-wide sums at once; every thread returns the same values.  ONE barrier: `red` must not be in use by an earlier call
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

// One POI by one workgroup.  (Rounds 1 - 5 needed a second launch for windows clamped in x at a volume border, because their
// 16-byte loads wanted contiguous x indices; with one lane per x every lane simply reads the voxel its own index names.)
__device__ __forceinline__ void fftcc3d_fused32_poi(const Fftcc3dParams& P, float* __restrict__ pois, int stride_f, unsigned long long idx) {
    __shared__ c2 lds[kLdsC];
    __shared__ float red[2 * kWaves], red2[2 * kWaves];
    __shared__ int redi[kWaves];
    const int tid = threadIdx.x;
    const int a = tid >> 5, b = tid & 31;
    const int lane = tid & (kWave - 1), wave = tid >> 6;
    float* poi = pois + idx * (unsigned long long)stride_f;
    constexpr int R = TN / 2;
    constexpr int M = TN * TN * TN;
#if defined(OC_F32_TIMELINE)   // thread 0 leaves the cycles (s_memtime) between its marks in the record's unused fields 19 ... 30
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
    // target window displaced by the initial guess).  Separable.  The reference has no bounds guard in 3D; indices are
    // clamped like in fftcc3d_gather_kernel.
    auto voxel = [&](int axis, int k) {   // axis 0 ... 2: reference x, y, z; 3 ... 5: target
        const int which = axis % 3;
        const float p = poi[which == 0 ? poi3d::X : which == 1 ? poi3d::Y : poi3d::Z];
        const float g = poi[which == 0 ? poi3d::U : which == 1 ? poi3d::V : poi3d::W];
        const int D = which == 0 ? P.dx : which == 1 ? P.dy : P.dz;
        float c = p + k - R;
        if (axis >= 3) c = c + g;
        return clampi3((int)c, 0, D - 1);
    };
    const int my_rx = voxel(0, b), my_tx = voxel(3, b);
    stamp();   // 19: record, indices

    // ---- gather: thread (z = a, x = b) reads its y-line of both windows; z = ref + i*tar
    c2 v[TN];
    {
        const int my_ry = voxel(1, b), my_ty = voxel(4, b);   // lane y holds row y's index
        const float* __restrict__ rp = P.ref + (size_t)voxel(2, a) * P.dy * P.dx + my_rx;
        const float* __restrict__ tp = P.tar + (size_t)voxel(5, a) * P.dy * P.dx + my_tx;
#pragma unroll
        for (int y = 0; y < TN; y++) {
            const int ry = __builtin_amdgcn_readlane(my_ry, y), ty = __builtin_amdgcn_readlane(my_ty, y);   // wave-uniform row offsets
            v[y] = mkc(rp[(size_t)ry * P.dx], tp[(size_t)ty * P.dx]);
        }
    }
    // means, zero-mean, sums of squares (src/oc_fftcc.cpp:360-376)
    {
        float rs = 0.f, ts = 0.f;
#pragma unroll
        for (int k = 0; k < TN; k++) {
            rs += v[k].x;
            ts += v[k].y;
        }
        stamp();   // 20: the gather has arrived; 64 additions
        block_sum2(rs, ts, red, lane, wave);
        stamp();   // 21: sums inside the wave, barrier, 32 LDS reads
        const c2 mean = mkc(rs / M, ts / M);
        float rn = 0.f, tn = 0.f;
#pragma unroll
        for (int k = 0; k < TN; k++) {
            v[k] = v[k] - mean;
            rn += v[k].x * v[k].x;
            tn += v[k].y * v[k].y;
        }
        // needed only at the very end, by thread 0: the waves' partial sums are parked in LDS now and added up there, in
        // block_sum2's order, behind the arg-max barrier
        rn = wave_allreduce_sum(rn);
        tn = wave_allreduce_sum(tn);
        if (lane == 0) {
            red2[wave] = rn;
            red2[kWaves + wave] = tn;
        }
    }
    stamp();   // 22: zero-mean, sums of squares

    // ---- forward y (thread (z = a, x = b)), then y -> x inside the half-wave that owns plane z = a (2)
    fft32<false>(v);
    {
        float* __restrict__ plane = reinterpret_cast<float*>(lds) + a * TN * TP;
#pragma unroll
        for (int k = 0; k < TN; k++) plane[b * TP + k] = v[bitrev5(k)].x;
        wave_lds_fence();
#pragma unroll
        for (int j = 0; j < TN; j++) v[j].x = plane[j * TP + b];
        wave_lds_fence();
#pragma unroll
        for (int k = 0; k < TN; k++) plane[b * TP + k] = v[bitrev5(k)].y;
        wave_lds_fence();
#pragma unroll
        for (int j = 0; j < TN; j++) v[j].y = plane[j * TP + b];
    }
    stamp();   // 23: forward y, y -> x exchange
    // ---- forward x (thread (z = a, ky = b)), then the exchange to z-lines (3), (4): slot s is written by the z-half s (round d:
    // its kx-half s ^ d) as [z & 15][ky][kx & 15], row pitch 17; the reader (ky(a), kx = b), kx-half xh, finds its block in slot
    // xh ^ d (the second slot starts 16 elements late, so that the two 16-lane groups of a half-wave meet different banks)
    fft32<false>(v);
    __syncthreads();   // every plane has left its slot
    stamp();   // 24: forward x, barrier
    const int zz = a & 15, half = a >> 4;
    const int mw = a >> 1, ms = a & 1;
    const int ky = mw == 0 ? (ms << 4) : (ms ? TN - mw : mw);
    const int xh = b >> 4, xl = b & 15;
    c2 w[TN];
#pragma unroll
    for (int d = 0; d < 2; d++) {
        {   // writer (z = a, ky = b): block (kx-half = half ^ d, z-half = half) -> slot `half` (`half` is uniform over a wave: two branches, no selects)
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
    stamp();   // 25: exchange to z-lines (two rounds, four barriers)
    // ---- forward z: Z(kz, ky, kx = b) in w[bitrev5(kz)]; the odd kz -- registers 16 ... 31 -- of a rotated line change their sign
    fft32<false>(w);
    {
        const unsigned flip = xh ? 0x80000000u : 0u;
#pragma unroll
        for (int i = 16; i < TN; i++) w[i] = mkc(__uint_as_float(__float_as_uint(w[i].x) ^ flip), __uint_as_float(__float_as_uint(w[i].y) ^ flip));
    }
    // ---- spectra of the two real windows and their product conj(R) * T (src/oc_fftcc.cpp:378-386), times 4; Z(-k) is register
    // bitrev5(-kz) of the lane that holds line (-ky, -kx)
    {
        const int paddr = ((((mw == 0 ? ms : (ms ^ 1)) << 5) | ((TN - b) & (TN - 1)))) << 2;
        auto from_partner = [&](c2 x) {
            return mkc(__int_as_float(__builtin_amdgcn_ds_bpermute(paddr, __float_as_int(x.x))),
                       __int_as_float(__builtin_amdgcn_ds_bpermute(paddr, __float_as_int(x.y))));
        };
        auto product = [](c2 zk, c2 zm) {
            const float rr = zk.x + zm.x, ri = zk.y - zm.y;
            const float tr = zk.y + zm.y, ti = -(zk.x - zm.x);
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
    stamp();   // 26: forward z, sign flip, partner values, spectrum product
    // ---- inverse z on the lines kx <= 16 (natural-order input is a renaming of registers), then z -> y through the half-spectrum tile (5)
    if (b < HP) {
        c2 t[TN];
#pragma unroll
        for (int k = 0; k < TN; k++) t[k] = w[bitrev5(k)];
        fft32<true>(t);
#pragma unroll
        for (int z = 0; z < TN; z++) lds[(z * TN + ky) * HP + b] = t[bitrev5(z)];
    }
    __syncthreads();
    stamp();   // 27: inverse z, tile write, barrier
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
    // ---- y -> x inside the half-wave that owns plane z = a; D(z, y, 32 - kx) = conj D(z, y, kx); inverse x: the correlation
    // volume (times 4), real part
    c2 q[TN];
    {
        const c2* __restrict__ row = lds + (a * TN + b) * HP;
#pragma unroll
        for (int k = 0; k < HP; k++) q[k] = row[k];
#pragma unroll
        for (int k = 1; k < TN / 2; k++) q[TN - k] = mkc(q[k].x, -q[k].y);
    }
    stamp();   // 28: inverse y on the column, written back, row read
    fft32<true>(q);

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
    stamp();   // 29: inverse x, arg-max inside the thread and the wave, barrier
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
        float rn = 0.f, tn = 0.f;
        for (int i = 0; i < kWaves; i++) {
            rn += red2[i];
            tn += red2[kWaves + i];
        }
        poi[poi3d::ZNCC] = (0.25f * best) / (sqrtf(rn * tn) * M);
    }
}

// workgroup -> POI (XCD-contiguous ranges of the visiting order)
__global__ __launch_bounds__(kThreads3, 4) void fftcc3d_fused32_kernel(Fftcc3dParams P, float* __restrict__ pois, int stride_f,
                                                                      unsigned long long count, int xcd_chunk) {
    unsigned long long idx = blockIdx.x;
    if (xcd_chunk > 0) idx = (unsigned long long)(blockIdx.x & 7u) * xcd_chunk + (blockIdx.x >> 3);
    if (idx >= count) return;
    if (P.perm) idx = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)P.perm[idx]);  // wave-uniform: the record's address stays in SGPRs
    fftcc3d_fused32_poi(P, pois, stride_f, idx);
}

}  // namespace

bool fftcc3d_fused_supported(int rx, int ry, int rz) { return rx == TN / 2 && ry == TN / 2 && rz == TN / 2; }

hipError_t launch_fftcc3d_fused(const Fftcc3dParams& p, float* pois, int stride_f, size_t count, bool xcd, hipStream_t stream) {
    if (count == 0) return hipSuccess;
    if (!fftcc3d_fused_supported(p.rx, p.ry, p.rz)) return hipErrorInvalidValue;
    const int chunk = xcd ? (int)((count + 7) / 8) : 0;
    const size_t grid = xcd ? (size_t)chunk * 8 : count;
    (void)hipGetLastError();  // drop stale errors of earlier, unrelated calls
    hipLaunchKernelGGL(fftcc3d_fused32_kernel, dim3((unsigned)grid), dim3(kThreads3), 0, stream, p, pois, stride_f,
                       (unsigned long long)count, chunk);
    return hipGetLastError();
}

}  // namespace ochip
