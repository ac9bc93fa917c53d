// poi_split.hip -- order-preserving partitions of a POI queue by result quality, on the device.
//
// The RegionFit -> re-ICGN loop of the reference's examples (examples/test_3d_reconstruction_sift_icgn2_regfit.cpp:214-260)
// is host code around the engines: it sorts the POIs of a finished queue into "reliable" (ZNCC >= high) and "unreliable"
// (ZNCC < low, or convergence > criterion) vectors, re-initialises the unreliable ones from their reliable neighbours
// (RegionFit), refines them again (ICGN) and moves the ones that now pass into the reliable set and back into the main
// queue.  With the queue resident in HBM those two selections would be the only steps that force the records through the
// host (25 MB each way for 250 000 POIs).  Here they are stream compactions: class per POI, per-block counts, an
// exclusive scan over the blocks, a scatter that keeps queue order (so every later step sees the POIs in the order the
// host loop would produce).  Comparisons are the example's own float comparisons: a NaN ZNCC or convergence makes every
// one of them false, i.e. the POI belongs to neither set.
#include "oc_device.h"
#include "oc_kernels.h"

namespace ochip {

namespace {

constexpr int kSplitBlock = 256;

// 0 = reliable / recovered, 1 = unreliable / still unreliable, 2 = neither
__device__ __forceinline__ int poi_class(const float* rec, const PoiSplitParams& P) {
    const float zncc = rec[P.zncc_at], conv = rec[P.conv_at];
    if (P.mode == 0) {
        // examples/test_3d_reconstruction_sift_icgn2_regfit.cpp:219-228
        if (zncc < P.zncc_low || conv > P.conv) return 1;
        return zncc >= P.zncc_high ? 0 : 2;
    }
    // :245: a refined POI is accepted when zncc >= high && convergence <= criterion; everything else stays unreliable
    return (zncc >= P.zncc_high && conv <= P.conv) ? 0 : 1;
}

__global__ __launch_bounds__(kSplitBlock) void poi_split_count_kernel(const float* __restrict__ pois, int stride_f, unsigned count,
                                                                       PoiSplitParams P, unsigned* __restrict__ block_counts) {
    __shared__ unsigned n[2];
    if (threadIdx.x < 2) n[threadIdx.x] = 0;
    __syncthreads();
    const unsigned i = blockIdx.x * kSplitBlock + threadIdx.x;
    const int c = i < count ? poi_class(pois + (size_t)i * stride_f, P) : 2;
    const unsigned long long m0 = __ballot(c == 0), m1 = __ballot(c == 1);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&n[0], (unsigned)__popcll(m0));
        atomicAdd(&n[1], (unsigned)__popcll(m1));
    }
    __syncthreads();
    if (threadIdx.x < 2) block_counts[2 * blockIdx.x + threadIdx.x] = n[threadIdx.x];
}

// block_counts[2 * b + k] -> exclusive prefix over b, in place; totals[k] = the sums.  One 1024-thread workgroup.
__global__ __launch_bounds__(1024) void poi_split_scan_kernel(unsigned* __restrict__ block_counts, unsigned nblocks,
                                                              unsigned* __restrict__ totals) {
    __shared__ unsigned part[2][1024];
    const unsigned per = (nblocks + 1023) / 1024;
    const unsigned lo = threadIdx.x * per, hi = min(lo + per, nblocks);
    unsigned s0 = 0, s1 = 0;
    for (unsigned b = lo; b < hi; b++) {
        s0 += block_counts[2 * b];
        s1 += block_counts[2 * b + 1];
    }
    part[0][threadIdx.x] = s0;
    part[1][threadIdx.x] = s1;
    __syncthreads();
    // Hillis-Steele over the 1024 partial sums
    for (unsigned off = 1; off < 1024; off <<= 1) {
        const unsigned a0 = threadIdx.x >= off ? part[0][threadIdx.x - off] : 0, a1 = threadIdx.x >= off ? part[1][threadIdx.x - off] : 0;
        __syncthreads();
        part[0][threadIdx.x] += a0;
        part[1][threadIdx.x] += a1;
        __syncthreads();
    }
    unsigned e0 = part[0][threadIdx.x] - s0, e1 = part[1][threadIdx.x] - s1;  // exclusive prefix of this thread's range
    for (unsigned b = lo; b < hi; b++) {
        const unsigned c0 = block_counts[2 * b], c1 = block_counts[2 * b + 1];
        block_counts[2 * b] = e0;
        block_counts[2 * b + 1] = e1;
        e0 += c0;
        e1 += c1;
    }
    if (threadIdx.x == 1023) {
        totals[0] = part[0][1023];
        totals[1] = part[1][1023];
    }
}

// Copies record i to its place: class 0 -> out0[off0 + rank] (and, when a main queue is given, back into
// main[index_in[i]]; index_out0[rank], if wanted, receives that main-queue index), class 1 -> out1[rank] with
// index_out[rank] = the POI's index in the MAIN queue.  A main-queue index >= main_count is never written through:
// the record still goes to out0, *bad_index is raised and the host reports OC_HIP_ERR_INVALID.
__global__ __launch_bounds__(kSplitBlock) void poi_split_scatter_kernel(const float* __restrict__ pois, int stride_f, unsigned count,
                                                                         PoiSplitParams P, const unsigned* __restrict__ block_offsets,
                                                                         const unsigned* __restrict__ index_in, float* __restrict__ out0,
                                                                         unsigned off0, unsigned* __restrict__ index_out0,
                                                                         float* __restrict__ out1, unsigned* __restrict__ index_out,
                                                                         float* __restrict__ main_queue, unsigned main_count,
                                                                         unsigned* __restrict__ bad_index) {
    __shared__ unsigned wave_base[2][kSplitBlock / 64];
    const unsigned i = blockIdx.x * kSplitBlock + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* rec = pois + (size_t)i * stride_f;
    const int c = i < count ? poi_class(rec, P) : 2;
    const unsigned long long m0 = __ballot(c == 0), m1 = __ballot(c == 1);
    if (lane == 0) {
        wave_base[0][wave] = (unsigned)__popcll(m0);
        wave_base[1][wave] = (unsigned)__popcll(m1);
    }
    __syncthreads();
    unsigned b0 = block_offsets[2 * blockIdx.x], b1 = block_offsets[2 * blockIdx.x + 1];
    for (int w = 0; w < wave; w++) {
        b0 += wave_base[0][w];
        b1 += wave_base[1][w];
    }
    const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
    if (c == 0) {
        const unsigned r = b0 + (unsigned)__popcll(m0 & below);
        float* dst = out0 + (size_t)(off0 + r) * stride_f;
        float* back = nullptr;
        if (main_queue) {
            const unsigned at = index_in[i];
            if (at < main_count) back = main_queue + (size_t)at * stride_f;
            else *bad_index = 1u;  // (a plain store of the same value from any number of threads)
        }
        for (int k = 0; k < P.rec_floats; k++) {
            const float v = rec[k];
            dst[k] = v;
            if (back) back[k] = v;
        }
        if (index_out0) index_out0[r] = index_in ? index_in[i] : i;
    } else if (c == 1) {
        const unsigned r = b1 + (unsigned)__popcll(m1 & below);
        float* dst = out1 + (size_t)r * stride_f;
        for (int k = 0; k < P.rec_floats; k++) dst[k] = rec[k];
        index_out[r] = index_in ? index_in[i] : i;
    }
}

}  // namespace

// flag[0] = 1 if any of the n indices is >= limit (cheap pre-pass of oc_hip_merge_recovered on DEVICE lists: nothing of the
// caller's is written before the list is known to be clean)
__global__ __launch_bounds__(256) void poi_index_range_kernel(const unsigned* __restrict__ index, unsigned n, unsigned limit,
                                                              unsigned* __restrict__ flag) {
    bool bad = false;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) bad = bad || index[i] >= limit;
    if (__syncthreads_or(bad ? 1 : 0) && threadIdx.x == 0) atomicOr(flag, 1u);
}

hipError_t launch_poi_index_range(const unsigned* index, size_t n, size_t limit, unsigned* flag, hipStream_t stream) {
    hipError_t err = hipMemsetAsync(flag, 0, sizeof(unsigned), stream);
    if (err != hipSuccess || n == 0) return err;
    if (n > 0xffffffffull) return hipErrorInvalidValue;
    if (limit > 0xffffffffull) return hipSuccess;  // 32-bit indices cannot exceed it
    const unsigned blocks = (unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    (void)hipGetLastError();
    hipLaunchKernelGGL(poi_index_range_kernel, dim3(blocks), dim3(256), 0, stream, index, (unsigned)n, (unsigned)limit, flag);
    return hipGetLastError();
}

// per-block counts (2 per block) | totals[2] | bad-index flag
size_t poi_split_scratch_words(size_t count) { return 2 * ((count + kSplitBlock - 1) / kSplitBlock) + 3; }

hipError_t launch_poi_split(const float* pois, int stride_f, size_t count, const PoiSplitParams& P, const unsigned* index_in,
                            float* out0, size_t off0, unsigned* index_out0, float* out1, unsigned* index_out, float* main_queue,
                            size_t main_count, unsigned* scratch, hipStream_t stream) {
    if (count == 0 || count > 0x7fffffffull || off0 > 0x7fffffffull) return count == 0 ? hipSuccess : hipErrorInvalidValue;
    if (main_count > 0xffffffffull) main_count = 0xffffffffull;  // indices are 32-bit
    const unsigned nblocks = (unsigned)((count + kSplitBlock - 1) / kSplitBlock);
    unsigned* totals = scratch + 2 * (size_t)nblocks;
    hipError_t merr = hipMemsetAsync(totals + 2, 0, sizeof(unsigned), stream);
    if (merr != hipSuccess) return merr;
    (void)hipGetLastError();
    hipLaunchKernelGGL(poi_split_count_kernel, dim3(nblocks), dim3(kSplitBlock), 0, stream, pois, stride_f, (unsigned)count, P, scratch);
    hipLaunchKernelGGL(poi_split_scan_kernel, dim3(1), dim3(1024), 0, stream, scratch, nblocks, totals);
    hipLaunchKernelGGL(poi_split_scatter_kernel, dim3(nblocks), dim3(kSplitBlock), 0, stream, pois, stride_f, (unsigned)count, P, scratch,
                       index_in, out0, (unsigned)off0, index_out0, out1, index_out, main_queue, (unsigned)main_count, totals + 2);
    return hipGetLastError();
}

}  // namespace ochip
