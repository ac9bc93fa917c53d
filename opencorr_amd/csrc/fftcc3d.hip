// fftcc3d.hip -- device kernels around the batched 3D rocFFT transforms of FFTCC3D.
//
// FFTCC3D::compute(POI3D*) (src/oc_fftcc.cpp:327-427): two 2rz x 2ry x 2rx windows -> zero-mean +
// sums of squares -> R2C x2 -> conj(R)*T -> C2R -> arg-max (first maximum wins) -> 3-way index
// decode + wrap -> u, v, w, ZNCC.  One 256-thread workgroup per POI for gather and arg-max.
// The reference has NO bounds guard here (compare src/oc_fftcc.cpp:327-353 with the 2D guard at
// :190-196) and would read out of bounds for a POI too close to the border; the kernel clamps
// the voxel indices instead of faulting (the only deviation, and only for inputs on which the
// reference's behaviour is undefined).
#include "oc_device.h"
#include "oc_kernels.h"

namespace ochip {

constexpr int kBlockF3 = 256;

__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_allreduce_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__global__ __launch_bounds__(kBlockF3) void fftcc3d_gather_kernel(Fftcc3dParams P, const float* __restrict__ pois,
                                                                  int stride_f, unsigned long long count,
                                                                  float* __restrict__ ref_win,
                                                                  float* __restrict__ tar_win,
                                                                  float* __restrict__ norms) {
    __shared__ float red[4];
    const unsigned long long idx = blockIdx.x;
    if (idx >= count) return;
    const float* poi = pois + idx * (unsigned long long)stride_f;
    const float px = poi[poi3d::X], py = poi[poi3d::Y], pz = poi[poi3d::Z];
    const float gu = poi[poi3d::U], gv = poi[poi3d::V], gw = poi[poi3d::W];
    const int rx = P.rx, ry = P.ry, rz = P.rz;
    const int sx = 2 * rx, sy = 2 * ry, sz = 2 * rz;
    const int M = sx * sy * sz;
    float* rw = ref_win + idx * (unsigned long long)M;
    float* tw = tar_win + idx * (unsigned long long)M;
    float rsum = 0.f, tsum = 0.f;
    for (int s = threadIdx.x; s < M; s += kBlockF3) {
        const int i = s / (sx * sy);
        const int rem = s - i * (sx * sy);
        const int j = rem / sx, k = rem - j * sx;
        // Point3D ref_point(poi->x + k - rx, poi->y + j - ry, poi->z + i - rz), truncated (src/oc_fftcc.cpp:349-358)
        const float rxp = px + k - rx, ryp = py + j - ry, rzp = pz + i - rz;
        const int ax = clampi((int)rxp, 0, P.dx - 1), ay = clampi((int)ryp, 0, P.dy - 1), az = clampi((int)rzp, 0, P.dz - 1);
        const float a = P.ref[((size_t)az * P.dy + ay) * P.dx + ax];
        const float txp = rxp + gu, typ = ryp + gv, tzp = rzp + gw;
        const int bx = clampi((int)txp, 0, P.dx - 1), by = clampi((int)typ, 0, P.dy - 1), bz = clampi((int)tzp, 0, P.dz - 1);
        const float b = P.tar[((size_t)bz * P.dy + by) * P.dx + bx];
        rw[s] = a;
        tw[s] = b;
        rsum += a;
        tsum += b;
    }
    const float rmean = block_sum_256(rsum, red) / M;
    const float tmean = block_sum_256(tsum, red) / M;
    float rn = 0.f, tn = 0.f;
    for (int s = threadIdx.x; s < M; s += kBlockF3) {
        const float a = rw[s] - rmean, b = tw[s] - tmean;
        rw[s] = a;
        tw[s] = b;
        rn += a * a;
        tn += b * b;
    }
    rn = block_sum_256(rn, red);
    tn = block_sum_256(tn, red);
    if (threadIdx.x == 0) { norms[2 * idx] = rn; norms[2 * idx + 1] = tn; }
}

__global__ __launch_bounds__(kBlockF3) void fftcc3d_argmax_kernel(Fftcc3dParams P, const float* __restrict__ surf,
                                                                  const float* __restrict__ norms,
                                                                  float* __restrict__ pois, int stride_f,
                                                                  unsigned long long count) {
    __shared__ float rv[4];
    __shared__ int ri[4];
    const unsigned long long idx = blockIdx.x;
    if (idx >= count) return;
    const int rx = P.rx, ry = P.ry, rz = P.rz;
    const int sx = 2 * rx, sy = 2 * ry, sz = 2 * rz;
    const int M = sx * sy * sz;
    const float* z = surf + idx * (unsigned long long)M;
    float best = -2.f;  // src/oc_fftcc.cpp:391-400: strict '>' scanning from index 0
    int bidx = 0;
    for (int s = threadIdx.x; s < M; s += kBlockF3) {
        const float v = z[s];
        if (v > best) { best = v; bidx = s; }
    }
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const float ov = __shfl_xor(best, off, kWave);
        const int oi = __shfl_xor(bidx, off, kWave);
        if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { rv[wave] = best; ri[wave] = bidx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++)
            if (rv[w] > best || (rv[w] == best && ri[w] < bidx)) { best = rv[w]; bidx = ri[w]; }
        float* poi = pois + idx * (unsigned long long)stride_f;
        int du = bidx % sx, dv = (bidx / sx) % sy, dw = bidx / (sx * sy);  // src/oc_fftcc.cpp:401-403
        if (du > rx) du -= sx;
        if (dv > ry) dv -= sy;
        if (dw > rz) dw -= sz;
        const float gu = poi[poi3d::U], gv = poi[poi3d::V], gw = poi[poi3d::W];
        poi[poi3d::U] = (float)du + gu;
        poi[poi3d::V] = (float)dv + gv;
        poi[poi3d::W] = (float)dw + gw;
        poi[poi3d::U0] = gu;
        poi[poi3d::V0] = gv;
        poi[poi3d::W0] = gw;
        poi[poi3d::ZNCC] = best / (sqrtf(norms[2 * idx] * norms[2 * idx + 1]) * M);
    }
}

hipError_t launch_fftcc3d_gather(const Fftcc3dParams& p, const float* pois, int stride_f, size_t count, float* ref_win,
                                 float* tar_win, float* norms, hipStream_t stream) {
    if (count == 0) return hipSuccess;
    (void)hipGetLastError();  // drop stale errors of earlier, unrelated calls
    hipLaunchKernelGGL(fftcc3d_gather_kernel, dim3((unsigned)count), dim3(kBlockF3), 0, stream, p, pois, stride_f,
                       (unsigned long long)count, ref_win, tar_win, norms);
    return hipGetLastError();
}

hipError_t launch_fftcc3d_argmax(const Fftcc3dParams& p, const float* surf, const float* norms, float* pois,
                                 int stride_f, size_t count, hipStream_t stream) {
    if (count == 0) return hipSuccess;
    (void)hipGetLastError();  // drop stale errors of earlier, unrelated calls
    hipLaunchKernelGGL(fftcc3d_argmax_kernel, dim3((unsigned)count), dim3(kBlockF3), 0, stream, p, surf, norms, pois,
                       stride_f, (unsigned long long)count);
    return hipGetLastError();
}

}  // namespace ochip
