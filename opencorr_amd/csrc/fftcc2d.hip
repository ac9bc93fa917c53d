// fftcc2d.hip -- device kernels around the batched rocFFT transforms of FFTCC2D.
//
// FFTCC2D::compute(POI2D*) (src/oc_fftcc.cpp:177-275) per POI:
//   guard -> two 2ry x 2rx windows -> zero-mean + sum of squares -> R2C x2 ->
//   conj(R)*T -> C2R (unnormalised) -> arg-max (first maximum wins) -> wrap -> u,v,ZNCC.
// Here: fftcc2d_gather (one wave per POI) -> rocFFT R2C (batch 2*chunk) ->
// fftcc_conjmul -> rocFFT C2R -> fftcc2d_argmax (one wave per POI).
#include "oc_device.h"
#include "oc_kernels.h"

namespace ochip {

// Window fill, means, zero-mean, sums of squares (src/oc_fftcc.cpp:190-231).
// The window is written in the reference's buffer order [r*2rx + c].
__global__ __launch_bounds__(64) void fftcc2d_gather_kernel(Fftcc2dParams P, const float* __restrict__ pois,
                                                            int stride_f, unsigned long long count,
                                                            float* __restrict__ ref_win, float* __restrict__ tar_win,
                                                            float* __restrict__ norms, int* __restrict__ flags) {
    const unsigned long long idx = blockIdx.x;
    if (idx >= count) return;
    const int lane = threadIdx.x;
    const float* poi = pois + idx * (unsigned long long)stride_f;
    const float px = poi[poi2d::X], py = poi[poi2d::Y];
    const float gu = poi[poi2d::U], gv = poi[poi2d::V];
    const int rx = P.rx, ry = P.ry, width = P.width, height = P.height;
    const int sw = 2 * rx, sh = 2 * ry, M = sw * sh;
    float* rw = ref_win + idx * (unsigned long long)M;
    float* tw = tar_win + idx * (unsigned long long)M;

    // bounds guard: the reference returns silently and leaves the POI untouched (src/oc_fftcc.cpp:190-196)
    if ((int)px < rx || (int)px >= width - rx || (int)py < ry || (int)py >= height - ry || (int)(px + gu) < rx ||
        (int)(px + gu) >= width - rx || (int)(py + gv) < ry || (int)(py + gv) >= height - ry) {
        for (int s = lane; s < M; s += kWave) { rw[s] = 0.f; tw[s] = 0.f; }
        if (lane == 0) { flags[idx] = 1; norms[2 * idx] = 0.f; norms[2 * idx + 1] = 0.f; }
        return;
    }
    float rsum = 0.f, tsum = 0.f;
    for (int s = lane; s < M; s += kWave) {
        const int r = s / sw, c = s - r * sw;
        // Point2D ref_point(poi->x + c - rx, poi->y + r - ry), truncated (src/oc_fftcc.cpp:209-216)
        const float rxp = px + c - rx, ryp = py + r - ry;
        const float a = P.ref[(size_t)(int)ryp * width + (int)rxp];
        const float txp = rxp + gu, typ = ryp + gv;
        const float b = P.tar[(size_t)(int)typ * width + (int)txp];
        rw[s] = a;
        tw[s] = b;
        rsum += a;
        tsum += b;
    }
    const float rmean = wave_allreduce_sum(rsum) / M;
    const float tmean = wave_allreduce_sum(tsum) / M;
    float rn = 0.f, tn = 0.f;
    for (int s = lane; s < M; s += kWave) {
        const float a = rw[s] - rmean, b = tw[s] - tmean;  // each lane re-reads only what it wrote
        rw[s] = a;
        tw[s] = b;
        rn += a * a;
        tn += b * b;
    }
    rn = wave_allreduce_sum(rn);
    tn = wave_allreduce_sum(tn);
    if (lane == 0) { flags[idx] = 0; norms[2 * idx] = rn; norms[2 * idx + 1] = tn; }
}

// zncc_freq = conj(ref_freq) * tar_freq  (src/oc_fftcc.cpp:236-241)
__global__ __launch_bounds__(256) void fftcc_conjmul_kernel(const float2* __restrict__ rf, const float2* __restrict__ tf,
                                                            float2* __restrict__ zf, unsigned long long bins) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= bins) return;
    const float2 r = rf[i], t = tf[i];
    float2 z;
    z.x = (r.x * t.x) + (r.y * t.y);
    z.y = (r.x * t.y) - (r.y * t.x);
    zf[i] = z;
}

// arg-max with "strict >, scanning from index 0" semantics (src/oc_fftcc.cpp:246-255), i.e. the
// lowest index among equal maxima wins; then wrap and result write-back (:256-274).
__global__ __launch_bounds__(64) void fftcc2d_argmax_kernel(Fftcc2dParams P, const float* __restrict__ surf,
                                                            const float* __restrict__ norms,
                                                            const int* __restrict__ flags, float* __restrict__ pois,
                                                            int stride_f, unsigned long long count) {
    const unsigned long long idx = blockIdx.x;
    if (idx >= count) return;
    if (flags[idx]) return;  // guard fired: POI untouched
    const int lane = threadIdx.x;
    const int rx = P.rx, ry = P.ry;
    const int sw = 2 * rx, sh = 2 * ry, M = sw * sh;
    const float* z = surf + idx * (unsigned long long)M;
    float best = -2.f;
    int bidx = 0;
    for (int s = lane; s < M; s += kWave) {
        const float v = z[s];
        if (v > best) { best = v; bidx = s; }
    }
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const float ov = __shfl_xor(best, off, kWave);
        const int oi = __shfl_xor(bidx, off, kWave);
        if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
    }
    if (lane == 0) {
        float* poi = pois + idx * (unsigned long long)stride_f;
        int du = bidx % sw, dv = bidx / sw;
        if (du > rx) du -= sw;
        if (dv > ry) dv -= sh;
        const float gu = poi[poi2d::U], gv = poi[poi2d::V];
        const float rn = norms[2 * idx], tn = norms[2 * idx + 1];
        poi[poi2d::U] = (float)du + gu;
        poi[poi2d::V] = (float)dv + gv;
        poi[poi2d::U0] = gu;
        poi[poi2d::V0] = gv;
        poi[poi2d::ZNCC] = best / (sqrtf(rn * tn) * M);
    }
}

hipError_t launch_fftcc2d_gather(const Fftcc2dParams& p, const float* pois, int stride_f, size_t count, float* ref_win,
                                 float* tar_win, float* norms, int* flags, hipStream_t stream) {
    if (count == 0) return hipSuccess;
    (void)hipGetLastError();  // drop stale errors of earlier, unrelated calls
    hipLaunchKernelGGL(fftcc2d_gather_kernel, dim3((unsigned)count), dim3(64), 0, stream, p, pois, stride_f,
                       (unsigned long long)count, ref_win, tar_win, norms, flags);
    return hipGetLastError();
}

hipError_t launch_fftcc_conjmul(const float2* rf, const float2* tf, float2* zf, size_t bins, hipStream_t stream) {
    if (bins == 0) return hipSuccess;
    const unsigned blocks = (unsigned)((bins + 255) / 256);
    (void)hipGetLastError();  // drop stale errors of earlier, unrelated calls
    hipLaunchKernelGGL(fftcc_conjmul_kernel, dim3(blocks), dim3(256), 0, stream, rf, tf, zf, (unsigned long long)bins);
    return hipGetLastError();
}

hipError_t launch_fftcc2d_argmax(const Fftcc2dParams& p, const float* surf, const float* norms, const int* flags,
                                 float* pois, int stride_f, size_t count, hipStream_t stream) {
    if (count == 0) return hipSuccess;
    (void)hipGetLastError();  // drop stale errors of earlier, unrelated calls
    hipLaunchKernelGGL(fftcc2d_argmax_kernel, dim3((unsigned)count), dim3(64), 0, stream, p, surf, norms, flags, pois,
                       stride_f, (unsigned long long)count);
    return hipGetLastError();
}

}  // namespace ochip
