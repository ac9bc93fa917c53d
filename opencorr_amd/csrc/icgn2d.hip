// icgn2d.hip -- ICGN2D1 / ICGN2D2 (inverse-compositional Gauss-Newton) on gfx950.
//
// Replaces ICGN2D1::compute(POI2D*) (src/oc_icgn.cpp:144-341) and
// ICGN2D2::compute(POI2D*) (src/oc_icgn.cpp:685-898) for a whole POI queue
// (:343-351 / :900-908).
//
// Mapping: ONE 64-lane wavefront per POI (one-wave workgroups, so no barriers).
// Sample s = r*W + c of the (2ry+1) x (2rx+1) subset is owned by lane s % 64; each
// lane keeps ITS samples of the zero-mean reference subset, the two reference
// gradients and the current warped-target subset in registers for the whole
// solve (NT = ceil(N/64) values each, compile-time NT), so the only memory
// traffic inside the Gauss-Newton loop is the 64 B/sample gather from the
// bicubic coefficient LUT -- the "interpolation sweep" that bounds the kernel.
// Reductions (mean, norms, Hessian, numerator, ZNSSD) are per-lane partial sums
// in increasing s followed by the xor butterfly of oc_device.h; the CPU oracle
// uses the same association (OC_ORDER_LANES) and the results are bit-identical.
// Wave-uniform state (warp matrix, inverse Hessian, norms) is held in SGPRs.
#include <cstdlib>

#include "oc_device.h"
#include "oc_kernels.h"

namespace ochip {

__device__ __forceinline__ float uni(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

// ---------------------------------------------------------------------------
// small dense algebra on wave-uniform values (every lane computes the same
// thing).  Same operation order as the oracle (oracle/oc_oracle.cpp lu_inverse,
// inverse3, mat_mul), which restates Eigen's PartialPivLU / cofactor inverse /
// lazy product used at src/oc_icgn.cpp:210,290,759,831.
// ---------------------------------------------------------------------------
// Inverse of an n x n matrix by LU with partial pivoting + solve against the identity,
// distributed over the wave: lane j (j < n) holds COLUMN j of the matrix in col[0..n-1]
// and receives column j of the inverse in inv[0..n-1].  Every scalar operation (pivot
// choice, multipliers f = a_rk / a_kk, eliminations a_rc -= f * a_kc, the two
// triangular solves) is the one the sequential algorithm performs on that element, so
// the result is bit-identical to oracle lu_inverse(); only the element -> lane
// placement differs.  Multipliers and pivots are broadcast with v_readlane (SGPRs).
template <int n>
__device__ __forceinline__ void lu_inverse_lanes(float (&col)[n], float (&inv)[n], int lane) {
    int perm[n];  // wave-uniform row permutation
#pragma unroll
    for (int i = 0; i < n; i++) perm[i] = i;
#pragma unroll
    for (int k = 0; k < n; k++) {
        int piv = k;
        float best = fabsf(wave_bcast(col[k], k));
#pragma unroll
        for (int r = k + 1; r < n; r++) {
            const float v = fabsf(wave_bcast(col[r], k));
            if (v > best) { best = v; piv = r; }
        }
#pragma unroll
        for (int r = k + 1; r < n; r++) {  // swap rows k <-> piv (at most one r matches)
            const bool sw = (piv == r);
            const float a = col[k], b = col[r];
            col[k] = sw ? b : a;
            col[r] = sw ? a : b;
            const int pa = perm[k], pb = perm[r];
            perm[k] = sw ? pb : pa;
            perm[r] = sw ? pa : pb;
        }
        const float d = wave_bcast(col[k], k);
#pragma unroll
        for (int r = k + 1; r < n; r++) {
            const float f = wave_bcast(col[r], k) / d;
            const float upd = col[r] - f * col[k];
            col[r] = lane == k ? f : (lane > k ? upd : col[r]);
        }
    }
    // lane c solves L U x = P e_c
    float y[n];
#pragma unroll
    for (int i = 0; i < n; i++) {
        float v = (perm[i] == lane) ? 1.f : 0.f;
#pragma unroll
        for (int j = 0; j < i; j++) v = v - wave_bcast(col[i], j) * y[j];
        y[i] = v;
    }
#pragma unroll
    for (int i = n - 1; i >= 0; i--) {
        float v = y[i];
#pragma unroll
        for (int j = i + 1; j < n; j++) v = v - wave_bcast(col[i], j) * y[j];
        y[i] = v / wave_bcast(col[i], i);
    }
#pragma unroll
    for (int i = 0; i < n; i++) inv[i] = y[i];
}

template <int n>
__device__ __forceinline__ void mat_mul(const float (&a)[n * n], const float (&b)[n * n], float (&c)[n * n]) {
#pragma unroll
    for (int i = 0; i < n; i++)
#pragma unroll
        for (int j = 0; j < n; j++) {
            float v = a[i * n + 0] * b[0 * n + j];
#pragma unroll
            for (int k = 1; k < n; k++) v = v + a[i * n + k] * b[k * n + j];
            c[i * n + j] = v;
        }
}

__device__ __forceinline__ float cof3(const float (&m)[9], int i, int j) {
    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    return m[i1 * 3 + j1] * m[i2 * 3 + j2] - m[i1 * 3 + j2] * m[i2 * 3 + j1];
}
__device__ __forceinline__ void inverse3(const float (&m)[9], float (&r)[9]) {
    const float c0 = cof3(m, 0, 0), c1 = cof3(m, 1, 0), c2 = cof3(m, 2, 0);
    const float det = (c0 * m[0] + c1 * m[3]) + c2 * m[6];
    const float invdet = 1.f / det;
    r[0] = c0 * invdet; r[1] = c1 * invdet; r[2] = c2 * invdet;
    r[3] = cof3(m, 0, 1) * invdet; r[4] = cof3(m, 1, 1) * invdet; r[5] = cof3(m, 2, 1) * invdet;
    r[6] = cof3(m, 0, 2) * invdet; r[7] = cof3(m, 1, 2) * invdet; r[8] = cof3(m, 2, 2) * invdet;
}

// Deformation2D1::setWarp, src/oc_deformation.cpp:117-128
__device__ __forceinline__ void set_warp_2d1(float (&w)[9], float u, float ux, float uy, float v, float vx, float vy) {
    w[0] = 1.f + ux; w[1] = uy; w[2] = u;
    w[3] = vx; w[4] = 1.f + vy; w[5] = v;
    w[6] = 0.f; w[7] = 0.f; w[8] = 1.f;
}

// walks the samples owned by one lane: s = lane, lane+64, ... as (row r, column c)
struct SampleWalk {
    int r, c, s;
    int W, q64, r64;
    __device__ __forceinline__ SampleWalk(int lane, int r0, int c0, int W_, int q64_, int r64_)
        : r(r0), c(c0), s(lane), W(W_), q64(q64_), r64(r64_) {}
    __device__ __forceinline__ void next() {
        s += kWave;
        c += r64;
        r += q64;
        const bool wrap = c >= W;
        c = wrap ? c - W : c;
        r = wrap ? r + 1 : r;
    }
};

// One LUT entry in flight: address generation and the four 16-byte loads are issued for a
// whole group of G samples before any polynomial is evaluated, so each lane keeps G*64 B
// of gathers outstanding (the interpolation sweep is latency/L1-bandwidth bound).
struct LutFetch {
    float4 c0, c1, c2, c3;
    float x, y;
    int xi, yi;
    bool out;
};

// range rule of BicubicBspline::compute (src/oc_cubic_bspline.cpp:137-142); out-of-range
// samples fetch entry (0,0), which is always mapped, and are replaced by -1.f afterwards
__device__ __forceinline__ void lut_fetch(LutFetch& f, const float* __restrict__ lut, int height, int width, float x,
                                          float y) {
    f.out = (x < 1 || y < 1 || x >= width - 2 || y >= height - 2 || isnan(x) || isnan(y));
    f.x = x;
    f.y = y;
    f.xi = f.out ? 0 : (int)floorf(x);
    f.yi = f.out ? 0 : (int)floorf(y);
    const float4* __restrict__ e = reinterpret_cast<const float4*>(lut) + ((size_t)f.yi * width + f.xi) * 4;
    f.c0 = e[0];
    f.c1 = e[1];
    f.c2 = e[2];
    f.c3 = e[3];
}

// explicit 16-term left-to-right polynomial of src/oc_cubic_bspline.cpp:144-177
__device__ __forceinline__ float lut_eval(const LutFetch& f) {
    const float dx = f.x - (float)f.xi, dy = f.y - (float)f.yi;
    const float dx2 = dx * dx, dy2 = dy * dy;
    const float dx3 = dx2 * dx, dy3 = dy2 * dy;
    float v = f.c0.x;
    v = v + f.c0.y * dx;
    v = v + f.c0.z * dx2;
    v = v + f.c0.w * dx3;
    v = v + f.c1.x * dy;
    v = v + f.c1.y * dy * dx;
    v = v + f.c1.z * dy * dx2;
    v = v + f.c1.w * dy * dx3;
    v = v + f.c2.x * dy2;
    v = v + f.c2.y * dy2 * dx;
    v = v + f.c2.z * dy2 * dx2;
    v = v + f.c2.w * dy2 * dx3;
    v = v + f.c3.x * dy3;
    v = v + f.c3.y * dy3 * dx;
    v = v + f.c3.z * dy3 * dx2;
    v = v + f.c3.w * dy3 * dx3;
    return f.out ? -1.f : v;
}

// ---------------------------------------------------------------------------
// ICGN2D1: 6 DoF, 3x3 warp.  One wave per POI; per-sample state lives in LDS as
// [t][lane] arrays (conflict-free ds_read/write_b32), NT = ceil(N/64) at run time.
//   LDS layout (floats): rs[NT*64] | gx[NT*64] | gy[NT*64] | ts[NT*64]
// G = samples whose LUT gathers are issued back to back (template).
// ---------------------------------------------------------------------------
template <int G>
__global__ __launch_bounds__(64) void icgn2d1_kernel(Icgn2dParams P, float* __restrict__ pois, int stride_f,
                                                     unsigned long long count, int NT) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const unsigned long long idx = blockIdx.x;
    if (idx >= count) return;
    const int lane = threadIdx.x;
    float* __restrict__ l_rs = lds + lane;
    float* __restrict__ l_gx = l_rs + NT * kWave;
    float* __restrict__ l_gy = l_gx + NT * kWave;
    float* __restrict__ l_ts = l_gy + NT * kWave;

    float* poi = pois + idx * (unsigned long long)stride_f;
    const float rec = lane < poi2d::FLOATS ? poi[lane] : 0.f;
    const float px = wave_bcast(rec, poi2d::X), py = wave_bcast(rec, poi2d::Y);
    const float u_in = wave_bcast(rec, poi2d::U), ux_in = wave_bcast(rec, poi2d::UX), uy_in = wave_bcast(rec, poi2d::UY);
    const float v_in = wave_bcast(rec, poi2d::V), vx_in = wave_bcast(rec, poi2d::VX), vy_in = wave_bcast(rec, poi2d::VY);
    const float zncc_in = wave_bcast(rec, poi2d::ZNCC);
    const int rx = P.rx, ry = P.ry, height = P.height, width = P.width;

    // guard, src/oc_icgn.cpp:160-167
    if (py - ry < 0 || px - rx < 0 || py + ry > height - 1 || px + rx > width - 1 || fabsf(u_in) >= width ||
        fabsf(v_in) >= height || zncc_in < 0 || isnan(u_in) || isnan(v_in)) {
        if (lane == 0) poi[poi2d::ZNCC] = zncc_in >= 0 ? -3.f : zncc_in;
        return;
    }
    const int W = 2 * rx + 1, N = W * (2 * ry + 1);
    const float fN = (float)N;
    const int q64 = kWave / W, r64 = kWave - q64 * W;
    const int r0 = lane / W;
    const int c0 = lane - r0 * W;

    // ---- reference subset, zero-mean + norm (src/oc_icgn.cpp:174-176, src/oc_subset.cpp:39-53)
    float ref_norm;
    {
        const int x0 = (int)(px - rx), y0 = (int)(py - ry);
        const float* __restrict__ base = P.ref + (size_t)y0 * width + x0;
        float acc = 0.f;
        SampleWalk w(lane, r0, c0, W, q64, r64);
#pragma unroll 2
        for (int t = 0; t < NT; t++, w.next()) {
            const bool valid = w.s < N;
            const float v = valid ? base[w.r * width + w.c] : 0.f;
            acc = valid ? acc + v : acc;
            l_rs[t * kWave] = v;
        }
        const float mean = wave_allreduce_sum(acc) / fN;
        acc = 0.f;
        int s = lane;
#pragma unroll 2
        for (int t = 0; t < NT; t++, s += kWave) {
            const float d = l_rs[t * kWave] - mean;
            l_rs[t * kWave] = d;
            acc = s < N ? acc + d * d : acc;
        }
        ref_norm = uni(sqrtf(wave_allreduce_sum(acc)));
    }

    // ---- steepest-descent image + Hessian (src/oc_icgn.cpp:179-207), inverse (:210)
    float hinv_col[6];  // lane j < 6: column j of H^-1
    {
        float h[21];
#pragma unroll
        for (int i = 0; i < 21; i++) h[i] = 0.f;
        const size_t goff = (size_t)((int)py - ry) * width + ((int)px - rx);
        const float* __restrict__ bgx = P.gx + goff;
        const float* __restrict__ bgy = P.gy + goff;
        SampleWalk w(lane, r0, c0, W, q64, r64);
#pragma unroll 2
        for (int t = 0; t < NT; t++, w.next()) {
            const bool valid = w.s < N;
            const int off = w.r * width + w.c;
            const float g_x = valid ? bgx[off] : 0.f;
            const float g_y = valid ? bgy[off] : 0.f;
            l_gx[t * kWave] = g_x;
            l_gy[t * kWave] = g_y;
            const float fxl = (float)(w.c - rx), fyl = (float)(w.r - ry);
            const float sd[6] = {g_x, g_x * fxl, g_x * fyl, g_y, g_y * fxl, g_y * fyl};
            int k = 0;
#pragma unroll
            for (int i = 0; i < 6; i++)
#pragma unroll
                for (int j = 0; j <= i; j++, k++) h[k] = valid ? h[k] + sd[i] * sd[j] : h[k];
        }
        // lane j < 6 assembles column j of the symmetric Hessian
        float col[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int k = 0;
#pragma unroll
        for (int i = 0; i < 6; i++)
#pragma unroll
            for (int j = 0; j <= i; j++) {
                const float v = wave_allreduce_sum(h[k++]);
                if (lane == j) col[i] = v;  // H(i,j)
                if (lane == i) col[j] = v;  // H(j,i)
            }
        lu_inverse_lanes<6>(col, hinv_col, lane);
    }

    // ---- IC-GN loop (src/oc_icgn.cpp:216-307)
    float Wm[9];
    set_warp_2d1(Wm, u_in, ux_in, uy_in, v_in, vx_in, vy_in);
    int iter = 0;
    float dp_norm = 0.f, znssd = 0.f;
    float cu = 0.f, cux = 0.f, cuy = 0.f, cv = 0.f, cvx = 0.f, cvy = 0.f;
#pragma nounroll
    do {
        iter++;
        // warped target subset (src/oc_icgn.cpp:230-242)
        bool negative = false;
        float acc = 0.f;
        {
            SampleWalk w(lane, r0, c0, W, q64, r64);
#pragma nounroll
            for (int t0 = 0; t0 < NT; t0 += G) {
                LutFetch f[G];
                bool valid[G];
#pragma unroll
                for (int g = 0; g < G; g++, w.next()) {
                    valid[g] = w.s < N;
                    const float xl = (float)(w.c - rx), yl = (float)(w.r - ry);
                    // Deformation2D1::warp, src/oc_deformation.cpp:94-105
                    const float wx = (Wm[0] * xl + Wm[1] * yl) + Wm[2] * 1.f;
                    const float wy = (Wm[3] * xl + Wm[4] * yl) + Wm[5] * 1.f;
                    // a lane past the end of the subset fetches a harmless in-range point
                    lut_fetch(f[g], P.lut, height, width, valid[g] ? px + wx : 1.f, valid[g] ? py + wy : 1.f);
                }
#pragma unroll
                for (int g = 0; g < G; g++) {
                    const float v = lut_eval(f[g]);
                    negative = negative || (valid[g] && v < 0.f);
                    acc = valid[g] ? acc + v : acc;
                    if (t0 + g < NT) l_ts[(t0 + g) * kWave] = v;
                }
            }
        }
        // src/oc_icgn.cpp:251-255
        if (wave_any(negative)) {
            if (lane == 0) poi[poi2d::ZNCC] = -3.f;
            return;
        }
        // zeroMeanNorm of the target subset (src/oc_icgn.cpp:257)
        const float tmean = wave_allreduce_sum(acc) / fN;
        acc = 0.f;
        {
            int s = lane;
#pragma unroll 4
            for (int t = 0; t < NT; t++, s += kWave) {
                const float d = l_ts[t * kWave] - tmean;
                acc = s < N ? acc + d * d : acc;
            }
        }
        const float tar_norm = uni(sqrtf(wave_allreduce_sum(acc)));
        // error image, ZNSSD, numerator (src/oc_icgn.cpp:260-276)
        const float factor = ref_norm / tar_norm;
        float num[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float ssd = 0.f;
        {
            SampleWalk w(lane, r0, c0, W, q64, r64);
#pragma unroll 2
            for (int t = 0; t < NT; t++, w.next()) {
                const bool valid = w.s < N;
                const float tz = l_ts[t * kWave] - tmean;  // same bits as in the norm pass
                const float e = tz * factor - l_rs[t * kWave];
                const float fxl = (float)(w.c - rx), fyl = (float)(w.r - ry);
                const float g_x = l_gx[t * kWave], g_y = l_gy[t * kWave];
                const float e2 = e * e;
                const float n0 = g_x * e, n1 = (g_x * fxl) * e, n2 = (g_x * fyl) * e;
                const float n3 = g_y * e, n4 = (g_y * fxl) * e, n5 = (g_y * fyl) * e;
                ssd = valid ? ssd + e2 : ssd;
                num[0] = valid ? num[0] + n0 : num[0];
                num[1] = valid ? num[1] + n1 : num[1];
                num[2] = valid ? num[2] + n2 : num[2];
                num[3] = valid ? num[3] + n3 : num[3];
                num[4] = valid ? num[4] + n4 : num[4];
                num[5] = valid ? num[5] + n5 : num[5];
            }
        }
        znssd = uni(wave_allreduce_sum(ssd)) / (ref_norm * ref_norm);
        // dp = H^-1 * numerator (src/oc_icgn.cpp:279-286): lane j forms H^-1(i,j) * num[j], the
        // six products of row i are then added in ascending j exactly like the reference loop
        float numj = 0.f;
#pragma unroll
        for (int j = 0; j < 6; j++) {
            const float v = wave_allreduce_sum(num[j]);
            numj = lane == j ? v : numj;
        }
        float dp[6];
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const float prod = hinv_col[i] * numj;
            float v = 0.f;
#pragma unroll
            for (int j = 0; j < 6; j++) v += wave_bcast(prod, j);
            dp[i] = v;
        }
        // W <- W * (dW)^-1 ; p <- W (src/oc_icgn.cpp:287-293, src/oc_deformation.cpp:107-115)
        float dW[9], dWi[9], Wn[9];
        set_warp_2d1(dW, dp[0], dp[1], dp[2], dp[3], dp[4], dp[5]);
        inverse3(dW, dWi);
        mat_mul<3>(Wm, dWi, Wn);
#pragma unroll
        for (int i = 0; i < 9; i++) Wm[i] = uni(Wn[i]);
        cu = Wm[2]; cux = Wm[0] - 1.f; cuy = Wm[1];
        cv = Wm[5]; cvx = Wm[3]; cvy = Wm[4] - 1.f;
        // convergence norm (src/oc_icgn.cpp:296-306)
        const int rx2 = rx * rx, ry2 = ry * ry;
        const float d = dp[0] * dp[0] + dp[1] * dp[1] * rx2 + dp[2] * dp[2] * ry2 + dp[3] * dp[3] +
                        dp[4] * dp[4] * rx2 + dp[5] * dp[5] * ry2;
        dp_norm = uni(sqrtf(d));
    } while (iter < P.stop && dp_norm >= P.conv);

    // ---- outputs (src/oc_icgn.cpp:310-340)
    if (lane == 0) {
        float zncc = 0.5f * (2 - znssd);
        const float fiter = (float)iter;
        if (dp_norm >= P.conv && fiter >= P.stop) zncc = -4.f;
        float out_u = cu, out_v = cv;
        if (isnan(zncc) || isnan(out_u) || isnan(out_v)) {
            out_u = u_in;
            out_v = v_in;
            zncc = -5.f;
        }
        poi[poi2d::U] = out_u;
        poi[poi2d::UX] = cux;
        poi[poi2d::UY] = cuy;
        poi[poi2d::V] = out_v;
        poi[poi2d::VX] = cvx;
        poi[poi2d::VY] = cvy;
        poi[poi2d::U0] = u_in;
        poi[poi2d::V0] = v_in;
        poi[poi2d::ZNCC] = zncc;
        poi[poi2d::ITER] = fiter;
        poi[poi2d::CONV] = dp_norm;
        poi[poi2d::SRX] = (float)rx;
        poi[poi2d::SRY] = (float)ry;
    }
}

// LDS bytes per one-wave workgroup: 4 per-sample arrays
static size_t icgn2d1_lds_bytes(int nt) { return (size_t)4 * nt * kWave * sizeof(float); }
constexpr int kIcgn2dMaxNT = 128;  // 4 * 128 * 256 B = 128 KiB of the 160 KiB LDS

template <int G>
static hipError_t launch1(const Icgn2dParams& p, float* pois, int stride_f, size_t count, int nt, hipStream_t stream) {
    const size_t lds = icgn2d1_lds_bytes(nt);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(icgn2d1_kernel<G>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, kIcgn2dMaxNT * 4 * kWave * 4);
        if (err != hipSuccess) return err;
        attr_set = true;
    }
    hipLaunchKernelGGL(icgn2d1_kernel<G>, dim3((unsigned)count), dim3(64), lds, stream, p, pois, stride_f,
                       (unsigned long long)count, nt);
    return hipGetLastError();
}

int icgn2d_max_samples(int dof) { return dof == 6 ? kIcgn2dMaxNT * kWave : 0; }

hipError_t launch_icgn2d1(const Icgn2dParams& p, float* pois, int stride_f, size_t count, hipStream_t stream) {
    if (count == 0) return hipSuccess;
    const int N = (2 * p.rx + 1) * (2 * p.ry + 1);
    const int nt = (N + 63) / 64;
    if (nt > kIcgn2dMaxNT) return hipErrorInvalidValue;
    // gather depth G (LUT entries in flight per lane).  OC_HIP_ICGN_GATHER overrides the
    // default, which prefers a group size that divides the per-lane sample count.
    static const int forced = [] {
        const char* e = getenv("OC_HIP_ICGN_GATHER");
        return e ? atoi(e) : 0;
    }();
    int g = forced;
    if (g != 2 && g != 3 && g != 4 && g != 6 && g != 8) g = (nt % 4 == 0) ? 4 : ((nt % 3 == 0) ? 3 : 4);
    switch (g) {
        case 2: return launch1<2>(p, pois, stride_f, count, nt, stream);
        case 3: return launch1<3>(p, pois, stride_f, count, nt, stream);
        case 6: return launch1<6>(p, pois, stride_f, count, nt, stream);
        case 8: return launch1<8>(p, pois, stride_f, count, nt, stream);
        default: return launch1<4>(p, pois, stride_f, count, nt, stream);
    }
}

hipError_t launch_icgn2d2(const Icgn2dParams&, float*, int, size_t, hipStream_t) { return hipErrorNotSupported; }

}  // namespace ochip
