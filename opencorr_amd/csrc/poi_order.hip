// poi_order.hip -- queue-level helpers around the per-POI solvers: (1) the locality schedule of a POI queue (a
// permutation that visits the queue tile by tile), (2) at the end of the file, the best-candidate selection over a
// segmented candidate queue (oc_hip_select_best).
//
// The ICGN2D kernels are served by per-XCD L2 caches (4 MB each); POIs that are neighbours in the image
// share most of their coefficient-table lines.  A caller's queue is usually row-major over the whole
// image, so a workgroup batch covers one long thin strip whose table footprint (33 rows x image width
// x 64 B) overflows the L2.  Visiting the POIs tile by tile (square tiles of `tile_px` pixels, tiles in
// row-major order, queue order inside a tile) keeps the footprint of the POIs in flight near 2 MB.
// Every POI is computed exactly as before -- only the order of the independent per-POI solves changes.
//
// Counting sort in small kernels: histogram of tile ids, exclusive scan (one workgroup), scatter, and a rank pass
// that orders the POIs of a tile by queue index (so the schedule does not depend on the atomics' timing, and the
// waves of a workgroup get neighbouring POIs of a row-major queue).
#include "oc_device.h"
#include "oc_kernels.h"

namespace ochip {

namespace {

// 2D: `height` x `width` pixels, depth = 0; 3D (round 4, ICGN3D1): depth x height x width voxels, POI3D records (x, y, z first)
__device__ __forceinline__ unsigned tile_of(const float* poi, int height, int width, int tile_px, int ntx, int depth = 0, int nty = 0) {
    // NaN / out-of-image coordinates land in an edge tile; such POIs are rejected by the guards later anyway
    const float x = poi[poi2d::X], y = poi[poi2d::Y];
    const int xi = x >= 0.f ? (x < (float)width ? (int)x : width - 1) : 0;
    const int yi = y >= 0.f ? (y < (float)height ? (int)y : height - 1) : 0;
    unsigned t = (unsigned)(yi / tile_px) * (unsigned)ntx + (unsigned)(xi / tile_px);
    if (depth > 0) {
        const float z = poi[poi3d::Z];
        const int zi = z >= 0.f ? (z < (float)depth ? (int)z : depth - 1) : 0;
        t += (unsigned)(zi / tile_px) * (unsigned)ntx * (unsigned)nty;
    }
    return t;
}

// A caller's queue is usually row-major, so neighbouring lanes tend to fall into the same tile: lanes that continue their
// left neighbour's tile form a run, and one atomic per run (by its first lane, for the whole run) replaces one per POI --
// 8 times fewer contended atomics on a regular 8 px grid.  Returns the lane's position inside its run; `head` = the
// run's first lane, `len` = its length (valid in every lane of the run).  Lanes with `live` false form runs of their own
// and must not use the result.
__device__ __forceinline__ int run_of_equal_tiles(unsigned t, bool live, int lane, int& head, int& len) {
    const unsigned prev = (unsigned)__shfl_up((int)t, 1, 64);
    const bool prev_live = __shfl_up(live ? 1 : 0, 1, 64) != 0;
    const bool starts = lane == 0 || prev != t || !live || !prev_live;
    const unsigned long long heads = __ballot(starts);
    const unsigned long long upto = heads & (~0ull >> (63 - lane));       // run starts at or below this lane
    head = 63 - __builtin_clzll(upto);
    const unsigned long long above = lane == 63 ? 0ull : heads & (~0ull << (lane + 1));  // the next run's start
    const int end = above ? __builtin_ctzll(above) : 64;
    len = end - head;
    return lane - head;
}

__global__ __launch_bounds__(256) void tile_histogram_kernel(const float* __restrict__ pois, int stride_f, unsigned count,
                                                             int height, int width, int tile_px, int ntx,
                                                             unsigned* __restrict__ counts, int depth = 0, int nty = 0) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < count;
    const unsigned t = live ? tile_of(pois + (size_t)i * stride_f, height, width, tile_px, ntx, depth, nty) : 0u;
    int head, len;
    const int pos = run_of_equal_tiles(t, live, threadIdx.x & 63, head, len);
    if (live && pos == 0) atomicAdd(counts + t, (unsigned)len);
}

// counts[0..n) -> exclusive prefix sums in place, one 1024-thread workgroup
__global__ __launch_bounds__(1024) void tile_scan_kernel(unsigned* __restrict__ counts, int n) {
    __shared__ unsigned part[1024];
    const int tid = threadIdx.x;
    const int per = (n + 1023) / 1024;
    const int lo = tid * per, hi = min(n, lo + per);
    unsigned sum = 0;
    for (int i = lo; i < hi; i++) sum += counts[i];
    part[tid] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan of the per-thread sums
        const unsigned v = tid >= off ? part[tid - off] : 0u;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    unsigned run = part[tid] - sum;
    for (int i = lo; i < hi; i++) {
        const unsigned c = counts[i];
        counts[i] = run;
        run += c;
    }
}

__global__ __launch_bounds__(256) void tile_scatter_kernel(const float* __restrict__ pois, int stride_f, unsigned count,
                                                           int height, int width, int tile_px, int ntx,
                                                           unsigned* __restrict__ cursors, unsigned* __restrict__ perm, int depth = 0, int nty = 0) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < count;
    const unsigned t = live ? tile_of(pois + (size_t)i * stride_f, height, width, tile_px, ntx, depth, nty) : 0u;
    int head, len;
    const int pos = run_of_equal_tiles(t, live, threadIdx.x & 63, head, len);
    unsigned base = 0;
    if (live && pos == 0) base = atomicAdd(cursors + t, (unsigned)len);  // the run takes `len` consecutive slots
    base = (unsigned)__shfl((int)base, head, 64);
    if (live) perm[base + (unsigned)pos] = i;
}

// counts[] holds the END of every tile after the scatter; order the indices inside each tile ascending.  A workgroup takes
// 256 slots of one tile and ranks them against the whole tile held in LDS (every lane reads the same word per step: a
// broadcast): ~2 us for config B's 1024 tiles of 256 POIs where a rank loop over global memory took 27 us.  blockIdx.y
// spreads the chunks of crowded tiles (a 3 px grid puts 2 000 POIs into a 128 px tile) over several workgroups.
constexpr int kRankCap = 4096;  // a more crowded tile keeps the scatter's order

__global__ __launch_bounds__(256) void tile_rank_kernel(int ntiles, const unsigned* __restrict__ ends, const unsigned* __restrict__ slots,
                                                        unsigned* __restrict__ perm) {
    __shared__ unsigned seg[kRankCap];
    const unsigned tid = threadIdx.x;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const unsigned s = t ? ends[t - 1] : 0u, e = ends[t], n = e - s;
        if (blockIdx.y * 256u >= n) continue;  // (wave-uniform: whole workgroups skip)
        if (n > (unsigned)kRankCap) {
            for (unsigned k = blockIdx.y * 256u + tid; k < n; k += gridDim.y * 256u) perm[s + k] = slots[s + k];
            continue;
        }
        __syncthreads();  // the previous tile's ranks have been formed
        for (unsigned k = tid; k < n; k += 256) seg[k] = slots[s + k];
        __syncthreads();
        for (unsigned k = blockIdx.y * 256u + tid; k < n; k += gridDim.y * 256u) {
            const unsigned i = seg[k];
            unsigned r = 0;
            for (unsigned q = 0; q < n; q++) r += seg[q] < i ? 1u : 0u;
            perm[s + r] = i;
        }
    }
}

}  // namespace

size_t poi2d_tile_count(int height, int width, int tile_px) {
    return (size_t)((width + tile_px - 1) / tile_px) * (size_t)((height + tile_px - 1) / tile_px);
}

// perm[k] = index of the k-th POI to visit.  `tiles` is scratch for poi2d_tile_count() unsigned ints, `slots` for
// `count` unsigned ints.
hipError_t launch_poi2d_tile_order(const float* pois, int stride_f, size_t count, int height, int width, int tile_px,
                                   unsigned* tiles, unsigned* slots, unsigned* perm, hipStream_t stream) {
    if (count == 0) return hipSuccess;
    if (count > 0xffffffffull) return hipErrorInvalidValue;
    const int ntx = (width + tile_px - 1) / tile_px;
    const int ntiles = (int)poi2d_tile_count(height, width, tile_px);
    hipError_t err = hipMemsetAsync(tiles, 0, (size_t)ntiles * sizeof(unsigned), stream);
    if (err != hipSuccess) return err;
    const unsigned blocks = (unsigned)((count + 255) / 256);
    (void)hipGetLastError();
    hipLaunchKernelGGL(tile_histogram_kernel, dim3(blocks), dim3(256), 0, stream, pois, stride_f, (unsigned)count, height, width,
                       tile_px, ntx, tiles);
    hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, stream, tiles, ntiles);
    hipLaunchKernelGGL(tile_scatter_kernel, dim3(blocks), dim3(256), 0, stream, pois, stride_f, (unsigned)count, height, width,
                       tile_px, ntx, tiles, slots);
    // chunks of 256 slots per tile: sized for four times the mean tile population, at most the cap's 16
    const size_t mean4 = (4 * count / (size_t)ntiles + 255) / 256;
    const unsigned chunks = (unsigned)(mean4 < 1 ? 1 : (mean4 > (size_t)(kRankCap / 256) ? (size_t)(kRankCap / 256) : mean4));
    hipLaunchKernelGGL(tile_rank_kernel, dim3((unsigned)(ntiles < 65535 ? ntiles : 65535), chunks), dim3(256), 0, stream, ntiles, tiles,
                       slots, perm);
    return hipGetLastError();
}

// The same schedule for a POI3D queue: cubic tiles of `tile_vox` voxels, tiles in z-major order, queue order inside a tile.
// ICGN3D1 (icgn3d.hip) re-reads the 33^3 neighbourhood of a POI in five volumes (0.72 MB at r = 16) every iteration; with the
// 512 POIs in flight spread along queue rows their union (370 MB at config E) overflows the L2s AND the 256 MB Infinity
// Cache, and every byte comes from HBM (303 GB per launch, PMC); visited in compact blocks the POIs in flight share their
// voxels (85 MB).
size_t poi3d_tile_count(int depth, int height, int width, int tile_vox) {
    return (size_t)((width + tile_vox - 1) / tile_vox) * (size_t)((height + tile_vox - 1) / tile_vox) * (size_t)((depth + tile_vox - 1) / tile_vox);
}

hipError_t launch_poi3d_tile_order(const float* pois, int stride_f, size_t count, int depth, int height, int width, int tile_vox,
                                   unsigned* tiles, unsigned* slots, unsigned* perm, hipStream_t stream) {
    if (count == 0) return hipSuccess;
    if (count > 0xffffffffull) return hipErrorInvalidValue;
    const int ntx = (width + tile_vox - 1) / tile_vox, nty = (height + tile_vox - 1) / tile_vox;
    const size_t nt = poi3d_tile_count(depth, height, width, tile_vox);
    if (nt > 0x7fffffffull) return hipErrorInvalidValue;
    const int ntiles = (int)nt;
    hipError_t err = hipMemsetAsync(tiles, 0, (size_t)ntiles * sizeof(unsigned), stream);
    if (err != hipSuccess) return err;
    const unsigned blocks = (unsigned)((count + 255) / 256);
    (void)hipGetLastError();
    hipLaunchKernelGGL(tile_histogram_kernel, dim3(blocks), dim3(256), 0, stream, pois, stride_f, (unsigned)count, height, width,
                       tile_vox, ntx, tiles, depth, nty);
    hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, stream, tiles, ntiles);
    hipLaunchKernelGGL(tile_scatter_kernel, dim3(blocks), dim3(256), 0, stream, pois, stride_f, (unsigned)count, height, width,
                       tile_vox, ntx, tiles, slots, depth, nty);
    const size_t mean4 = (4 * count / (size_t)ntiles + 255) / 256;
    const unsigned chunks = (unsigned)(mean4 < 1 ? 1 : (mean4 > (size_t)(kRankCap / 256) ? (size_t)(kRankCap / 256) : mean4));
    hipLaunchKernelGGL(tile_rank_kernel, dim3((unsigned)(ntiles < 65535 ? ntiles : 65535), chunks), dim3(256), 0, stream, ntiles, tiles,
                       slots, perm);
    return hipGetLastError();
}

}  // namespace ochip

// ---------------------------------------------------------------------------------------------------------------
// Candidate batching for callers like EpipolarSearch::compute(POI2D*) (src/oc_epipolar_search.cpp:133-195): the
// reference refines a handful of trial positions per POI one compute(POI2D*) at a time and keeps the one with the
// highest ZNCC (std::sort by ZNCC, :186-190).  With the engine, the trials of ALL POIs form one queue (one ICGN launch)
// and this kernel does the "keep the best" step: segment s of the candidate queue belongs to POI s.
// One wave per segment; the winner's deformation and result vectors are copied into the POI (`poi->deformation =
// best.deformation; poi->result = best.result`), nothing else is touched.  Highest ZNCC wins, the earliest candidate
// among equals, NaN never; an empty segment leaves its POI untouched.
// ---------------------------------------------------------------------------------------------------------------
namespace ochip {

__global__ __launch_bounds__(256) void poi2d_best_of_segments_kernel(const float* __restrict__ cand, int cand_stride_f,
                                                                      const unsigned* __restrict__ seg_start, unsigned nseg,
                                                                      float* __restrict__ pois, int stride_f) {
    const unsigned seg = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (seg >= nseg) return;
    const int lane = threadIdx.x & 63;
    const unsigned lo = seg_start[seg], hi = seg_start[seg + 1];
    float best = -INFINITY;
    unsigned bidx = 0xffffffffu;
    for (unsigned c = lo + lane; c < hi; c += 64) {
        const float z = cand[(size_t)c * cand_stride_f + poi2d::ZNCC];
        if (z > best || (z == best && c < bidx) || (bidx == 0xffffffffu && z == z)) {
            best = z;
            bidx = c;
        }
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float ov = __shfl_xor(best, off, 64);
        const unsigned oi = (unsigned)__shfl_xor((int)bidx, off, 64);
        const bool take = oi != 0xffffffffu && (bidx == 0xffffffffu || ov > best || (ov == best && oi < bidx));
        best = take ? ov : best;
        bidx = take ? oi : bidx;
    }
    if (bidx == 0xffffffffu) return;  // no candidate (or only NaNs)
    // deformation.p[12] = floats 2..13, result.r[6] = floats 14..19 (src/oc_poi.h:102-136)
    if (lane < 18) pois[(size_t)seg * stride_f + 2 + lane] = cand[(size_t)bidx * cand_stride_f + 2 + lane];
}

hipError_t launch_poi2d_best_of_segments(const float* cand, int cand_stride_f, const unsigned* seg_start, size_t nseg, float* pois,
                                         int stride_f, hipStream_t stream) {
    if (nseg == 0) return hipSuccess;
    if (nseg > 0x7fffffffull) return hipErrorInvalidValue;
    (void)hipGetLastError();
    hipLaunchKernelGGL(poi2d_best_of_segments_kernel, dim3((unsigned)((nseg + 3) / 4)), dim3(256), 0, stream, cand, cand_stride_f,
                       seg_start, (unsigned)nseg, pois, stride_f);
    return hipGetLastError();
}

}  // namespace ochip
