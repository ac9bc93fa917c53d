// strain.hip -- Strain::prepare + Strain::compute(poi_queue) on gfx950.
//
// Replaces Strain::compute(std::vector<POI2D>&) (src/oc_strain.cpp:236-247 -> :149-234) and
// Strain::compute(std::vector<POI3D>&) (:476-488 -> :372-473) with the neighbour search of Strain::prepare
// (:96-147, a nanoflann kd-tree per thread in the reference).
//
// What the reference fixes and what it leaves to third-party code is stated with the oracle's restatement
// (oracle/oc_oracle.cpp, "Strain"); this file performs the same operations in the same order and is bit-identical to
// it: rows of the plane fit are visited cell by cell over a uniform grid (the 3 x 3 (x 3) block around the POI's
// cell, row-major) and by ascending queue index inside a cell; sums and the solve are IEEE double, one rounding to
// float at the end.
//
// Data layout.  prepare() sorts the queue indices by cell (histogram -> scan -> scatter -> rank-by-index inside the
// cell, so the order does not depend on the atomics' timing).  compute() first gathers the fields the fit needs into
// 32-byte records in that order -- {x, y, z, u | v, w, zncc, queue index} -- so that the fit kernel, one thread per
// POI in cell order, streams contiguous records: the threads of a wave sit in the same or adjacent cells and read the
// same cell segments (L1/L2 hits, 2 x 16-byte loads per neighbour).  HBM traffic is the queue once (gather) plus the
// records once; the kernel is bound by the fp64 accumulation, ~20 (2D) / ~40 (3D) double FMAs-worth per neighbour.
//
// POIs with fewer than neighbor_number_min POIs inside the radius take the KNN path of the reference
// (:177-186) in a second kernel over the (normally empty) list of such POIs: rings of cells are searched outwards
// until the K-th best distance is closer than anything an unvisited ring can hold.
#include "oc_device.h"
#include "oc_kernels.h"

namespace ochip {

namespace {

template <int DIM>
struct Lay;
template <>
struct Lay<2> {
    static constexpr int U = poi2d::U, V = poi2d::V, W = poi2d::V, ZNCC = poi2d::ZNCC, E0 = 20;
};
template <>
struct Lay<3> {
    static constexpr int U = poi3d::U, V = poi3d::V, W = poi3d::W, ZNCC = poi3d::ZNCC, E0 = 22;
};

// cell of one coordinate (the oracle's strain_cell_axis)
__device__ __forceinline__ int cell_axis(float c, float c0, float inv_pitch, int nc) {
    const float t = (c - c0) * inv_pitch;
    return t >= 0.f ? (t < (float)nc ? (int)t : nc - 1) : 0;  // NaN -> 0
}

template <int DIM>
__device__ __forceinline__ int cell_of(const float* p, const StrainGrid& g, int& cx, int& cy, int& cz) {
    cx = cell_axis(p[0], g.x0, g.inv_pitch, g.ncx);
    cy = cell_axis(p[1], g.y0, g.inv_pitch, g.ncy);
    cz = DIM == 3 ? cell_axis(p[2], g.z0, g.inv_pitch, g.ncz) : 0;
    return (cz * g.ncy + cy) * g.ncx + cx;
}

// ---- prepare ------------------------------------------------------------------------------------------------
// float <-> unsigned with the same ordering, for atomicMin / atomicMax
__device__ __forceinline__ unsigned ord(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

template <int DIM>
__global__ __launch_bounds__(256) void strain_bbox_kernel(const float* __restrict__ pois, int stride_f, unsigned count,
                                                          unsigned* __restrict__ box) {
    // box[0..2] = min x, y, z; box[3..5] = max; NaN coordinates take part in neither (as in the oracle's compares)
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x)
#pragma unroll
        for (int a = 0; a < DIM; a++) {
            const float c = pois[(size_t)i * stride_f + a];
            if (c < mn[a]) mn[a] = c;
            if (c > mx[a]) mx[a] = c;
        }
#pragma unroll
    for (int a = 0; a < DIM; a++) {
        for (int off = 32; off > 0; off >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], off));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off));
        }
        if ((threadIdx.x & 63) == 0) {
            atomicMin(box + a, ord(mn[a]));
            atomicMax(box + 3 + a, ord(mx[a]));
        }
    }
}

template <int DIM>
__global__ __launch_bounds__(256) void strain_histogram_kernel(const float* __restrict__ pois, int stride_f, unsigned count,
                                                               StrainGrid g, unsigned* __restrict__ counts) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    int cx, cy, cz;
    atomicAdd(counts + cell_of<DIM>(pois + (size_t)i * stride_f, g, cx, cy, cz), 1u);
}

// counts[0..n) -> start[0..n] (exclusive prefix sums, start[n] = total) and cursor[0..n) = start; one workgroup
__global__ __launch_bounds__(1024) void strain_scan_kernel(const unsigned* __restrict__ counts, unsigned* __restrict__ start,
                                                           unsigned* __restrict__ cursor, int n) {
    __shared__ unsigned part[1024];
    const int tid = threadIdx.x;
    const int per = (n + 1023) / 1024;
    const int lo = min(n, tid * per), hi = min(n, lo + per);
    unsigned sum = 0;
    for (int i = lo; i < hi; i++) sum += counts[i];
    part[tid] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const unsigned v = tid >= off ? part[tid - off] : 0u;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    unsigned run = part[tid] - sum;
    for (int i = lo; i < hi; i++) {
        const unsigned c = counts[i];
        start[i] = run;
        cursor[i] = run;
        run += c;
    }
    if (tid == 1023) start[n] = part[1023];
}

template <int DIM>
__global__ __launch_bounds__(256) void strain_scatter_kernel(const float* __restrict__ pois, int stride_f, unsigned count,
                                                             StrainGrid g, unsigned* __restrict__ cursor,
                                                             unsigned* __restrict__ slots) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    int cx, cy, cz;
    slots[atomicAdd(cursor + cell_of<DIM>(pois + (size_t)i * stride_f, g, cx, cy, cz), 1u)] = i;
}

// inside every cell segment, order the queue indices ascending: the element's rank is the number of smaller indices
// in its segment (segments are short: about (pitch / POI spacing)^DIM entries)
template <int DIM>
__global__ __launch_bounds__(256) void strain_rank_kernel(const float* __restrict__ pois, int stride_f, unsigned count,
                                                          StrainGrid g, const unsigned* __restrict__ start,
                                                          const unsigned* __restrict__ slots, unsigned* __restrict__ order) {
    const unsigned k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    const unsigned i = slots[k];
    int cx, cy, cz;
    const int c = cell_of<DIM>(pois + (size_t)i * stride_f, g, cx, cy, cz);
    const unsigned s = start[c], e = start[c + 1];
    unsigned r = 0;
    for (unsigned q = s; q < e; q++) r += slots[q] < i ? 1u : 0u;
    order[s + r] = i;
}

// ---- compute ------------------------------------------------------------------------------------------------
struct alignas(16) StrainRec {
    float x, y, z, u;
    float v, w, zncc;
    unsigned idx;
};

template <int DIM>
__global__ __launch_bounds__(256) void strain_gather_kernel(const float* __restrict__ pois, int stride_f, unsigned count,
                                                            const unsigned* __restrict__ order, StrainRec* __restrict__ recs) {
    const unsigned k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    const unsigned i = order[k];
    const float* p = pois + (size_t)i * stride_f;
    StrainRec r;
    r.x = p[0];
    r.y = p[1];
    r.z = DIM == 3 ? p[2] : 0.f;
    r.u = p[Lay<DIM>::U];
    r.v = p[Lay<DIM>::V];
    r.w = DIM == 3 ? p[Lay<DIM>::W] : 0.f;
    r.zncc = p[Lay<DIM>::ZNCC];
    r.idx = i;
    recs[k] = r;
}

// normal equations of the plane fit in double: S = sum row row^T (upper triangle), B[r] = sum row * rhs_r
template <int DIM>
struct Fit {
    static constexpr int D = DIM + 1;
    double S[D][D];
    double B[DIM][D];
    int n;
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int a = 0; a < D; a++)
#pragma unroll
            for (int b = 0; b < D; b++) S[a][b] = 0.0;
#pragma unroll
        for (int r = 0; r < DIM; r++)
#pragma unroll
            for (int a = 0; a < D; a++) B[r][a] = 0.0;
        n = 0;
    }
    // dx.. = neighbour - POI as float differences (src/oc_strain.cpp:204-205)
    __device__ __forceinline__ void add(float dx, float dy, float dz, float u, float v, float w) {
        double row[D];
        row[0] = 1.0;
        row[1] = (double)dx;
        row[2] = (double)dy;
        if constexpr (DIM == 3) row[3] = (double)dz;
        // the oracle adds row[a]*row[b] for every (a, b); the products commute, so the lower triangle mirrors the upper
#pragma unroll
        for (int a = 0; a < D; a++)
#pragma unroll
            for (int b = a; b < D; b++) S[a][b] = S[a][b] + row[a] * row[b];
        const double rhs[3] = {(double)u, (double)v, (double)w};
#pragma unroll
        for (int r = 0; r < DIM; r++)
#pragma unroll
            for (int a = 0; a < D; a++) B[r][a] = B[r][a] + row[a] * rhs[r];
        n++;
    }
    // Gaussian elimination in the oracle's order (strain_solve); grad[r][k]
    __device__ __forceinline__ void solve(double (&grad)[DIM][D]) {
        double A[D][D + DIM];
#pragma unroll
        for (int i = 0; i < D; i++) {
#pragma unroll
            for (int j = 0; j < D; j++) A[i][j] = i <= j ? S[i][j] : S[j][i];
#pragma unroll
            for (int r = 0; r < DIM; r++) A[i][D + r] = B[r][i];
        }
        bool dead[D];
#pragma unroll
        for (int k = 0; k < D; k++) {
            const double piv = A[k][k];
            dead[k] = !(piv > 1e-12 * S[k][k]);
            if (!dead[k]) {
#pragma unroll
                for (int i = k + 1; i < D; i++) {
                    const double f = A[i][k] / piv;
#pragma unroll
                    for (int j = k + 1; j < D + DIM; j++) A[i][j] = A[i][j] - f * A[k][j];
                }
            }
        }
#pragma unroll
        for (int r = 0; r < DIM; r++)
#pragma unroll
            for (int k = D - 1; k >= 0; k--) {
                double v = 0.0;
                if (!dead[k]) {
                    v = A[k][D + r];
#pragma unroll
                    for (int j = k + 1; j < D; j++) v = v - A[k][j] * grad[r][j];
                    v = v / A[k][k];
                }
                grad[r][k] = v;
            }
    }
};

// strain formulas, src/oc_strain.cpp:220-234 (2D), :446-466 (3D)
template <int DIM>
__device__ __forceinline__ void write_strain(float* __restrict__ poi, Fit<DIM>& fit, int approximation) {
    constexpr int D = DIM + 1;
    double grad[DIM][D];
    fit.solve(grad);
    float* e = poi + Lay<DIM>::E0;
    if constexpr (DIM == 2) {
        const float ux = (float)grad[0][1], uy = (float)grad[0][2], vx = (float)grad[1][1], vy = (float)grad[1][2];
        if (approximation == 1) {
            e[0] = ux;
            e[1] = vy;
            e[2] = 0.5f * (uy + vx);
        }
        if (approximation == 2) {
            e[0] = ux + 0.5f * (ux * ux + vx * vx);
            e[1] = vy + 0.5f * (uy * uy + vy * vy);
            e[2] = 0.5f * (uy + vx + uy * ux + vy * vx);
        }
    } else {
        const float ux = (float)grad[0][1], uy = (float)grad[0][2], uz = (float)grad[0][3 % D];
        const float vx = (float)grad[1][1], vy = (float)grad[1][2], vz = (float)grad[1][3 % D];
        const float wx = (float)grad[2 % DIM][1], wy = (float)grad[2 % DIM][2], wz = (float)grad[2 % DIM][3 % D];
        if (approximation == 1) {
            e[0] = ux;
            e[1] = vy;
            e[2] = wz;
            e[3] = 0.5f * (uy + vx);
            e[4] = 0.5f * (vz + wy);
            e[5] = 0.5f * (wx + uz);
        }
        if (approximation == 2) {
            e[0] = ux + 0.5f * (ux * ux + vx * vx + wx * wx);
            e[1] = vy + 0.5f * (uy * uy + vy * vy + wy * wy);
            e[2] = wz + 0.5f * (uz * uz + vz * vz + wz * wz);
            e[3] = 0.5f * (uy + vx + uy * ux + vy * vx + wy * wx);
            e[4] = 0.5f * (vz + wy + uz * uy + vz * vy + wz * wy);
            e[5] = 0.5f * (wx + uz + ux * uz + vx * vz + wx * wz);
        }
    }
}

// the POI a thread fits a plane for: a record of the cloud itself (Strain) or a POI of a second queue (RegionFit)
struct Query {
    float x, y, z;
    float* poi;  // record the result is written to
};

template <int DIM>
__device__ __forceinline__ float dist2(const Query& me, float x, float y, float z) {
    // nanoflann L2_Simple: sum over the dimensions of (query - point)^2, in order
    const float dx = me.x - x, dy = me.y - y;
    float d = dx * dx;
    d = d + dy * dy;
    if constexpr (DIM == 3) {
        const float dz = me.z - z;
        d = d + dz * dz;
    }
    return d;
}

// MODE 0 (Strain): query k is record k of the cloud, skipped when its ZNCC is below the threshold.
// MODE 1 (RegionFit): query k is POI k of `pois`.
template <int DIM, int MODE>
__device__ __forceinline__ bool load_query(Query& me, unsigned k, float* __restrict__ pois, int stride_f,
                                           const float4* __restrict__ rv, const StrainParams& P) {
    if constexpr (MODE == 0) {
        const float4 m0 = rv[2 * (size_t)k], m1 = rv[2 * (size_t)k + 1];
        me.x = m0.x;
        me.y = m0.y;
        me.z = m0.z;
        me.poi = pois + (size_t)__float_as_uint(m1.w) * stride_f;
        return m1.z >= P.zncc_threshold;  // src/oc_strain.cpp:241 / :481
    } else {
        me.poi = pois + (size_t)k * stride_f;
        me.x = me.poi[0];
        me.y = me.poi[1];
        me.z = DIM == 3 ? me.poi[2] : 0.f;
        return true;
    }
}

// RegionFit2D/3D::compute: the plane itself becomes the deformation, ZNCC is reset
// (src/oc_region_fit.cpp:153-162, 314-330)
template <int DIM>
__device__ __forceinline__ void write_plane(float* __restrict__ poi, Fit<DIM>& fit) {
    constexpr int D = DIM + 1;
    double grad[DIM][D];
    fit.solve(grad);
    if constexpr (DIM == 2) {
        poi[poi2d::U] = (float)grad[0][0];
        poi[poi2d::UX] = (float)grad[0][1];
        poi[poi2d::UY] = (float)grad[0][2];
        poi[poi2d::V] = (float)grad[1][0];
        poi[poi2d::VX] = (float)grad[1][1];
        poi[poi2d::VY] = (float)grad[1][2];
        poi[poi2d::ZNCC] = 0.f;
    } else {
#pragma unroll
        for (int r = 0; r < DIM; r++)
#pragma unroll
            for (int k = 0; k < D; k++) poi[poi3d::P + r * D + k] = (float)grad[r][k];  // u ux uy uz v ... wz
        poi[poi3d::ZNCC] = 0.f;
    }
}

template <int DIM, int MODE>
__global__ __launch_bounds__(256) void strain_fit_kernel(float* __restrict__ pois, int stride_f, unsigned count, StrainGrid g,
                                                         StrainParams P, const unsigned* __restrict__ start,
                                                         const StrainRec* __restrict__ recs, unsigned* __restrict__ fallback) {
    const unsigned k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    const float4* rv = reinterpret_cast<const float4*>(recs);
    Query me;
    if (!load_query<DIM, MODE>(me, k, pois, stride_f, rv, P)) return;
    const float pos[3] = {me.x, me.y, me.z};
    int cx, cy, cz;
    cell_of<DIM>(pos, g, cx, cy, cz);
    Fit<DIM> fit;
    fit.clear();
    int inside = 0;
    for (int dz = (DIM == 3 ? -1 : 0); dz <= (DIM == 3 ? 1 : 0); dz++)
        for (int dy = -1; dy <= 1; dy++) {
            const int y = cy + dy, z = cz + dz;
            if (y < 0 || z < 0 || y >= g.ncy || z >= g.ncz) continue;
            // the three cells of a row are adjacent in memory: one contiguous run of records
            const int xa = max(cx - 1, 0), xb = min(cx + 1, g.ncx - 1);
            const size_t rowc = ((size_t)z * g.ncy + y) * g.ncx;
            const unsigned s = start[rowc + xa], e = start[rowc + xb + 1];
            for (unsigned q = s; q < e; q++) {
                const float4 a = rv[2 * (size_t)q], b = rv[2 * (size_t)q + 1];
                if (dist2<DIM>(me, a.x, a.y, a.z) < P.radius2) {
                    inside++;
                    if (MODE == 1 || b.z >= P.zncc_threshold) fit.add(a.x - me.x, a.y - me.y, a.z - me.z, a.w, b.x, b.y);
                }
            }
        }
    if (inside < P.neighbor_min) {
        fallback[1 + atomicAdd(fallback, 1u)] = k;  // the KNN path, strain_knn_kernel
        return;
    }
    if (fit.n < P.neighbor_min) return;  // src/oc_strain.cpp:190
    if constexpr (MODE == 0)
        write_strain<DIM>(me.poi, fit, P.approximation);
    else
        write_plane<DIM>(me.poi, fit);
}

// KNN path (src/oc_strain.cpp:177-186): the K = neighbor_number_min nearest POIs by ascending (distance^2, queue
// index), found ring by ring; one thread per POI of the fallback list.
constexpr int kKnnMax = 64;

template <int DIM, int MODE>
__global__ __launch_bounds__(64) void strain_knn_kernel(float* __restrict__ pois, int stride_f, StrainGrid g, StrainParams P,
                                                        const unsigned* __restrict__ start, const StrainRec* __restrict__ recs,
                                                        const unsigned* __restrict__ fallback) {
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= fallback[0]) return;
    const unsigned k = fallback[1 + t];
    const float4* rv = reinterpret_cast<const float4*>(recs);
    Query me;
    load_query<DIM, MODE>(me, k, pois, stride_f, rv, P);
    const float pos[3] = {me.x, me.y, me.z};
    int cx, cy, cz;
    cell_of<DIM>(pos, g, cx, cy, cz);
    const int K = P.neighbor_min;
    float bd[kKnnMax];     // ascending (d2, idx)
    unsigned bi[kKnnMax];  // queue index
    unsigned bq[kKnnMax];  // record slot
    int have = 0;
    const float pitch = 1.f / g.inv_pitch;
    const int reach = max(max(g.ncx, g.ncy), g.ncz);
    for (int ring = 0; ring <= reach; ring++) {
        for (int dz = (DIM == 3 ? -ring : 0); dz <= (DIM == 3 ? ring : 0); dz++)
            for (int dy = -ring; dy <= ring; dy++)
                for (int dx = -ring; dx <= ring; dx++) {
                    if (max(max(abs(dx), abs(dy)), abs(dz)) != ring) continue;  // the shell only
                    const int x = cx + dx, y = cy + dy, z = cz + dz;
                    if (x < 0 || y < 0 || z < 0 || x >= g.ncx || y >= g.ncy || z >= g.ncz) continue;
                    const size_t c = ((size_t)z * g.ncy + y) * g.ncx + x;
                    for (unsigned q = start[c]; q < start[c + 1]; q++) {
                        const float4 a = rv[2 * (size_t)q];
                        const unsigned idx = __float_as_uint(rv[2 * (size_t)q + 1].w);
                        const float d = dist2<DIM>(me, a.x, a.y, a.z);
                        if (!(d == d)) continue;  // NaN coordinates never enter
                        // insert into the sorted list if it precedes the current K-th entry
                        if (have == K && !(d < bd[K - 1] || (d == bd[K - 1] && idx < bi[K - 1]))) continue;
                        int p = have < K ? have : K - 1;
                        while (p > 0 && (d < bd[p - 1] || (d == bd[p - 1] && idx < bi[p - 1]))) {
                            bd[p] = bd[p - 1];
                            bi[p] = bi[p - 1];
                            bq[p] = bq[p - 1];
                            p--;
                        }
                        bd[p] = d;
                        bi[p] = idx;
                        bq[p] = q;
                        if (have < K) have++;
                    }
                }
        // everything in rings > `ring` is at least ring * pitch away (0.999: slack for the rounding of the cell
        // assignment); stop once the K-th best is strictly closer than that
        if (have == K) {
            const float lim = (float)ring * pitch * 0.999f;
            if (bd[K - 1] < lim * lim) break;
        }
    }
    Fit<DIM> fit;
    fit.clear();
    for (int j = 0; j < have; j++) {
        const float4 a = rv[2 * (size_t)bq[j]], b = rv[2 * (size_t)bq[j] + 1];
        if (MODE == 1 || b.z >= P.zncc_threshold) fit.add(a.x - me.x, a.y - me.y, a.z - me.z, a.w, b.x, b.y);
    }
    if (fit.n < K) return;
    if constexpr (MODE == 0)
        write_strain<DIM>(me.poi, fit, P.approximation);
    else
        write_plane<DIM>(me.poi, fit);
}

float unord(unsigned u) {
    const unsigned v = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    float f;
    __builtin_memcpy(&f, &v, 4);
    return f;
}

}  // namespace

int strain_knn_max() { return kKnnMax; }

// bounding box of the queue's coordinates -> box[6] on the device (ordered-uint encoding), decoded by strain_make_grid
template <int DIM>
static hipError_t bbox_t(const float* pois, int stride_f, size_t count, unsigned* box, hipStream_t stream) {
    const unsigned init[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
    hipError_t err = hipMemcpyAsync(box, init, sizeof(init), hipMemcpyHostToDevice, stream);
    if (err != hipSuccess) return err;
    // few workgroups: every wave ends in atomics on the same six words
    const unsigned blocks = (unsigned)((count + 255) / 256 < 128 ? (count + 255) / 256 : 128);
    (void)hipGetLastError();
    hipLaunchKernelGGL(strain_bbox_kernel<DIM>, dim3(blocks), dim3(256), 0, stream, pois, stride_f, (unsigned)count, box);
    return hipGetLastError();
}

hipError_t launch_strain_bbox(int ndim, const float* pois, int stride_f, size_t count, unsigned* box, hipStream_t stream) {
    return ndim == 2 ? bbox_t<2>(pois, stride_f, count, box, stream) : bbox_t<3>(pois, stride_f, count, box, stream);
}

// the oracle's grid rule (strain_queue): pitch a little above the radius, capped cell count per axis
StrainGrid strain_make_grid(int ndim, const unsigned* box_host, float radius) {
    float mn[3], mx[3];
    for (int a = 0; a < 3; a++) {
        mn[a] = a < ndim ? unord(box_host[a]) : INFINITY;
        mx[a] = a < ndim ? unord(box_host[3 + a]) : -INFINITY;
        if (!(mn[a] <= mx[a])) mn[a] = mx[a] = 0.f;
    }
    float pitch = radius * 1.001f;
    const float span = fmaxf(fmaxf(mx[0] - mn[0], mx[1] - mn[1]), mx[2] - mn[2]);
    const float cap = ndim == 2 ? 4096.f : 256.f;
    if (!(pitch > span / cap)) pitch = span / cap;
    if (!(pitch > 0.f)) pitch = 1.f;
    StrainGrid g;
    g.x0 = mn[0];
    g.y0 = mn[1];
    g.z0 = mn[2];
    g.inv_pitch = 1.f / pitch;
    g.ncx = (int)((mx[0] - mn[0]) * g.inv_pitch) + 1;
    g.ncy = (int)((mx[1] - mn[1]) * g.inv_pitch) + 1;
    g.ncz = ndim == 3 ? (int)((mx[2] - mn[2]) * g.inv_pitch) + 1 : 1;
    return g;
}

size_t strain_cell_count(const StrainGrid& g) { return (size_t)g.ncx * g.ncy * g.ncz; }

template <int DIM>
static hipError_t sort_t(const float* pois, int stride_f, size_t count, const StrainGrid& g, unsigned* counts, unsigned* start,
                         unsigned* cursor, unsigned* slots, unsigned* order, hipStream_t stream) {
    const size_t ncell = strain_cell_count(g);
    hipError_t err = hipMemsetAsync(counts, 0, ncell * sizeof(unsigned), stream);
    if (err != hipSuccess) return err;
    const unsigned blocks = (unsigned)((count + 255) / 256);
    (void)hipGetLastError();
    hipLaunchKernelGGL(strain_histogram_kernel<DIM>, dim3(blocks), dim3(256), 0, stream, pois, stride_f, (unsigned)count, g, counts);
    hipLaunchKernelGGL(strain_scan_kernel, dim3(1), dim3(1024), 0, stream, counts, start, cursor, (int)ncell);
    hipLaunchKernelGGL(strain_scatter_kernel<DIM>, dim3(blocks), dim3(256), 0, stream, pois, stride_f, (unsigned)count, g, cursor,
                       slots);
    hipLaunchKernelGGL(strain_rank_kernel<DIM>, dim3(blocks), dim3(256), 0, stream, pois, stride_f, (unsigned)count, g, start,
                       slots, order);
    return hipGetLastError();
}

// Strain::prepare: cell-sorted order of the queue.  counts/cursor: ncell, start: ncell + 1, slots/order: count
hipError_t launch_strain_sort(int ndim, const float* pois, int stride_f, size_t count, const StrainGrid& g, unsigned* counts,
                              unsigned* start, unsigned* cursor, unsigned* slots, unsigned* order, hipStream_t stream) {
    if (count == 0) return hipSuccess;
    return ndim == 2 ? sort_t<2>(pois, stride_f, count, g, counts, start, cursor, slots, order, stream)
                     : sort_t<3>(pois, stride_f, count, g, counts, start, cursor, slots, order, stream);
}

template <int DIM, int MODE>
static hipError_t fit_t(float* pois, int stride_f, size_t count, const StrainGrid& g, const StrainParams& P,
                        const unsigned* start, const void* recs, unsigned* fallback, hipStream_t stream) {
    hipError_t err = hipMemsetAsync(fallback, 0, sizeof(unsigned), stream);
    if (err != hipSuccess) return err;
    const unsigned blocks = (unsigned)((count + 255) / 256);
    (void)hipGetLastError();
    hipLaunchKernelGGL((strain_fit_kernel<DIM, MODE>), dim3(blocks), dim3(256), 0, stream, pois, stride_f, (unsigned)count, g, P,
                       start, static_cast<const StrainRec*>(recs), fallback);
    // the KNN list is normally empty; the kernel sizes itself from the device-side counter
    const unsigned kblocks = (unsigned)((count + 63) / 64);
    hipLaunchKernelGGL((strain_knn_kernel<DIM, MODE>), dim3(kblocks), dim3(64), 0, stream, pois, stride_f, g, P, start,
                       static_cast<const StrainRec*>(recs), fallback);
    return hipGetLastError();
}

// the cloud's fit records in cell order (Strain: at every compute; RegionFit: once at prepare).  recs: count * 32 bytes
hipError_t launch_strain_gather(int ndim, const float* pois, int stride_f, size_t count, const unsigned* order, void* recs,
                                hipStream_t stream) {
    if (count == 0) return hipSuccess;
    const unsigned blocks = (unsigned)((count + 255) / 256);
    (void)hipGetLastError();
    if (ndim == 2)
        hipLaunchKernelGGL(strain_gather_kernel<2>, dim3(blocks), dim3(256), 0, stream, pois, stride_f, (unsigned)count, order,
                           static_cast<StrainRec*>(recs));
    else
        hipLaunchKernelGGL(strain_gather_kernel<3>, dim3(blocks), dim3(256), 0, stream, pois, stride_f, (unsigned)count, order,
                           static_cast<StrainRec*>(recs));
    return hipGetLastError();
}

// Strain::compute(poi_queue): `pois` is the cloud itself.  fallback: count + 1 unsigned
hipError_t launch_strain_compute(int ndim, float* pois, int stride_f, size_t count, const StrainGrid& g, const StrainParams& P,
                                 const unsigned* start, const unsigned* order, void* recs, unsigned* fallback,
                                 hipStream_t stream) {
    if (count == 0) return hipSuccess;
    hipError_t err = launch_strain_gather(ndim, pois, stride_f, count, order, recs, stream);
    if (err != hipSuccess) return err;
    return ndim == 2 ? fit_t<2, 0>(pois, stride_f, count, g, P, start, recs, fallback, stream)
                     : fit_t<3, 0>(pois, stride_f, count, g, P, start, recs, fallback, stream);
}

// RegionFit2D/3D::compute(poi_queue): `pois` are the POIs to initialise, `recs` the reliable cloud gathered at prepare
hipError_t launch_region_fit_compute(int ndim, float* pois, int stride_f, size_t count, const StrainGrid& g,
                                     const StrainParams& P, const unsigned* start, const void* recs, unsigned* fallback,
                                     hipStream_t stream) {
    if (count == 0) return hipSuccess;
    return ndim == 2 ? fit_t<2, 1>(pois, stride_f, count, g, P, start, recs, fallback, stream)
                     : fit_t<3, 1>(pois, stride_f, count, g, P, start, recs, fallback, stream);
}

}  // namespace ochip
