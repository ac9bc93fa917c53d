// oc_kernels.h -- host-side launch interface of the gfx950 kernels (internal).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>

// OC_BUILD_AB = 1: the A/B build of the library (opencorr_amd/build.py --ab -> lib/ab/libopencorr_hip_ab.so) also contains the
// measured losers kept as comparison partners -- icgn2d variants 0 and 6, the ICGN3D1 row mapping (icgn3d_rows.hip) -- and
// honours the experiment environment knobs (OC_ICGN3D_BLOCKS, OC_ICGN2D_LDS_PAD).  The library that ships is built without it.
#ifndef OC_BUILD_AB
#define OC_BUILD_AB 0
#endif

namespace ochip {

// ---- prepare2d.hip ---------------------------------------------------------
hipError_t launch_grad2d(const float* img, int height, int width, float* gx, float* gy, hipStream_t stream);
hipError_t launch_bspline2d_lut(const float* img, int height, int width, float* lut, hipStream_t stream);
hipError_t launch_colmajor_to_rowmajor(const float* src, int height, int width, float* dst, hipStream_t stream);

// ---- icgn2d.hip ------------------------------------------------------------
struct Icgn2dParams {
    const float* ref;  // reference image, row-major
    const float* gx;   // reference gradients
    const float* gy;
    const float* lut;  // target bicubic coefficient LUT, 16 floats per pixel
    int height, width;
    int rx, ry;        // subset radius (self_adaptive: the largest radii of the batch)
    float conv, stop;
    const float* offsets;  // per-POI centre offsets (x, y), or nullptr: compute(poi_queue, center_offset_queue)
    const unsigned* perm;  // visiting order of the queue (poi_order.hip), or nullptr: queue order
    int self_adaptive;     // DIC::setSelfAdaptive: every POI carries its own subset radius
    // IC-LM only (launch_iclm2d*): DampingParameter of src/oc_iclm.h:33-38 with ln(lambda) taken on the host
    double lm_log_lambda;
    float lm_alpha, lm_beta;
    int arith_fma;  // 1: the build whose per-sample multiply-adds are fused (oc_device.h OC_FMA; oracle OC_ORDER_LANES_FMA)
    float* setup;   // variant 8 (split launch shape): icgn2d_setup_record_floats(dof) floats per POI -- mean, norm, H^-1 -- written by
                    // the set-up kernel and read by the iteration kernel; nullptr otherwise
};
// writes max over the queue of (int)subset_radius.x / .y to out2[0], out2[1]
hipError_t launch_poi2d_max_radius(const float* pois, int stride_floats, size_t count, int* out2, hipStream_t stream);
// The ICGN2D kernels exist in several bit-identical variants (gather depth, LDS footprint,
// software pipelining, waves per workgroup; icgn2d.hip).  `variant` indexes that table,
// `xcd` turns on the XCD-contiguous mapping of workgroups to the POI queue.
// Returns hipErrorInvalidValue if the subset does not fit the variant's LDS budget.
// phase (variant 8 only): 0 = set-up kernel then iteration kernel on `stream`, 1 = the set-up kernel alone, 2 = the iteration
// kernel alone (the caller orders it behind the set-up of the same POIs)
hipError_t launch_icgn2d1(const Icgn2dParams& p, float* pois, int stride_floats, size_t count, int variant, bool xcd,
                          hipStream_t stream, int phase = 0);
// ICLM2D1 / ICLM2D2 (src/oc_iclm.cpp): the same kernel with the Levenberg-Marquardt step
hipError_t launch_iclm2d1(const Icgn2dParams& p, float* pois, int stride_floats, size_t count, bool xcd, hipStream_t stream);
hipError_t launch_iclm2d2(const Icgn2dParams& p, float* pois, int stride_floats, size_t count, bool xcd, hipStream_t stream);
int iclm2d_max_samples();
hipError_t launch_icgn2d2(const Icgn2dParams& p, float* pois, int stride_floats, size_t count, int variant, bool xcd,
                          hipStream_t stream, int phase = 0);
int icgn2d_setup_record_floats(int dof);
// the two builds of icgn2d.hip (oc_device.h: OC_FMA = 0 / 1) behind the four launchers above
#define OC_DECLARE_ICGN2D_LAUNCHERS                                                                                          \
    hipError_t launch_icgn2d1(const Icgn2dParams& p, float* pois, int stride_floats, size_t count, int variant, bool xcd,   \
                              hipStream_t stream, int phase);                                                               \
    hipError_t launch_icgn2d2(const Icgn2dParams& p, float* pois, int stride_floats, size_t count, int variant, bool xcd,   \
                              hipStream_t stream, int phase);                                                               \
    hipError_t launch_iclm2d1(const Icgn2dParams& p, float* pois, int stride_floats, size_t count, bool xcd, hipStream_t stream); \
    hipError_t launch_iclm2d2(const Icgn2dParams& p, float* pois, int stride_floats, size_t count, bool xcd, hipStream_t stream);
namespace sep {
OC_DECLARE_ICGN2D_LAUNCHERS
}
namespace fma {
OC_DECLARE_ICGN2D_LAUNCHERS
}
#undef OC_DECLARE_ICGN2D_LAUNCHERS
// icgn2d_band.hip (round 6): the same solvers with the workgroup's band of the bicubic table staged in LDS and the warped subset in
// registers; hipErrorInvalidValue when the subset has more passes than the kernel holds in registers or radii are per POI
hipError_t launch_icgn2d1_band(const Icgn2dParams& p, float* pois, int stride_floats, size_t count, bool xcd, hipStream_t stream);
hipError_t launch_icgn2d2_band(const Icgn2dParams& p, float* pois, int stride_floats, size_t count, bool xcd, hipStream_t stream);
bool icgn2d_band_supported(int dof, int rx, int ry);  // an instantiation for this subset's pass count exists
namespace sep {
hipError_t launch_icgn2d1_band(const Icgn2dParams& p, float* pois, int stride_floats, size_t count, bool xcd, hipStream_t stream);
hipError_t launch_icgn2d2_band(const Icgn2dParams& p, float* pois, int stride_floats, size_t count, bool xcd, hipStream_t stream);
}
namespace fma {
hipError_t launch_icgn2d1_band(const Icgn2dParams& p, float* pois, int stride_floats, size_t count, bool xcd, hipStream_t stream);
hipError_t launch_icgn2d2_band(const Icgn2dParams& p, float* pois, int stride_floats, size_t count, bool xcd, hipStream_t stream);
}
int icgn2d_variant_count();
int icgn2d_variant_info(int variant, int* g, int* mode, int* pipe, int* wpb, int* occ);
// variants 0 and 6 are A/B partners that only the A/B build of the library contains (-DOC_BUILD_AB=1)
bool icgn2d_variant_built(int variant);
// variants with a per-workgroup coordinate table (mode >= 3) need one subset radius per launch
inline bool icgn2d_variant_uses_table(int variant) {
    int g, mode, pipe, wpb, occ;
    return icgn2d_variant_info(variant, &g, &mode, &pipe, &wpb, &occ) == 0 && mode >= 3;
}
// largest (2rx+1)*(2ry+1) a variant accepts
int icgn2d_max_samples(int variant);

// ---- poi_order.hip ---------------------------------------------------------
// locality schedule: perm[k] = index of the k-th POI to visit (tile by tile, queue order inside a tile); tiles =
// scratch of poi2d_tile_count(height, width, tile_px) unsigned ints, slots = scratch of `count` unsigned ints
size_t poi2d_tile_count(int height, int width, int tile_px);
hipError_t launch_poi2d_tile_order(const float* pois, int stride_floats, size_t count, int height, int width, int tile_px,
                                   unsigned* tiles, unsigned* slots, unsigned* perm, hipStream_t stream);
// the same for POI3D queues (cubic tiles of tile_vox voxels): the ICGN3D1 kernels visit the queue block by block
size_t poi3d_tile_count(int depth, int height, int width, int tile_vox);
hipError_t launch_poi3d_tile_order(const float* pois, int stride_floats, size_t count, int depth, int height, int width, int tile_vox,
                                   unsigned* tiles, unsigned* slots, unsigned* perm, hipStream_t stream);

// ---- strain.hip -------------------------------------------------------------
// Strain::prepare / Strain::compute (src/oc_strain.cpp): uniform grid over the queue's coordinates
struct StrainGrid {
    float x0, y0, z0, inv_pitch;
    int ncx, ncy, ncz;
};
struct StrainParams {
    float radius2;         // subregion_radius * subregion_radius, in float
    float zncc_threshold;  // Strain::setZnccThreshold, default 0.9
    int neighbor_min;      // neighbor_number_min
    int approximation;     // 1 = Cauchy, 2 = Green
};
int strain_knn_max();  // largest neighbor_number_min the KNN path holds
hipError_t launch_strain_bbox(int ndim, const float* pois, int stride_floats, size_t count, unsigned* box6, hipStream_t stream);
StrainGrid strain_make_grid(int ndim, const unsigned* box6_host, float radius);
size_t strain_cell_count(const StrainGrid& g);
hipError_t launch_strain_sort(int ndim, const float* pois, int stride_floats, size_t count, const StrainGrid& g,
                              unsigned* counts, unsigned* start, unsigned* cursor, unsigned* slots, unsigned* order,
                              hipStream_t stream);
hipError_t launch_strain_gather(int ndim, const float* pois, int stride_floats, size_t count, const unsigned* order, void* recs,
                                hipStream_t stream);
// RegionFit2D/3D::compute (src/oc_region_fit.cpp): plane through the reliable cloud (recs) for every POI of `pois`
hipError_t launch_region_fit_compute(int ndim, float* pois, int stride_floats, size_t count, const StrainGrid& g,
                                     const StrainParams& P, const unsigned* start, const void* recs, unsigned* fallback,
                                     hipStream_t stream);
hipError_t launch_strain_compute(int ndim, float* pois, int stride_floats, size_t count, const StrainGrid& g,
                                 const StrainParams& P, const unsigned* start, const unsigned* order, void* recs,
                                 unsigned* fallback, hipStream_t stream);

// best candidate (highest ZNCC) of every segment of a candidate queue -> deformation + result of the segment's POI
hipError_t launch_poi2d_best_of_segments(const float* cand, int cand_stride_floats, const unsigned* seg_start, size_t nseg,
                                         float* pois, int stride_floats, hipStream_t stream);

// ---- nr2d.hip --------------------------------------------------------------
struct Nr2dParams {
    const float* ref;     // reference image, row-major
    const float* lut;     // bicubic LUT of the target image
    const float* lut_gx;  // bicubic LUT of d/dx target
    const float* lut_gy;  // bicubic LUT of d/dy target
    int height, width;
    int rx, ry;
    float conv, stop;
    const unsigned* perm;  // visiting order of the queue (poi_order.hip), or nullptr: queue order
};
hipError_t launch_nr2d1(const Nr2dParams& p, float* pois, int stride_floats, size_t count, hipStream_t stream);
int nr2d1_max_samples();

// ---- prepare3d.hip ---------------------------------------------------------
hipError_t launch_grad3d(const float* vol, int dz, int dy, int dx, float* gx, float* gy, float* gz, hipStream_t stream);
// coef <- prefilter_z(prefilter_y(prefilter_x(vol))); tmp is a scratch volume of the same size
hipError_t launch_bspline3d_prefilter(const float* vol, int dz, int dy, int dx, float* coef, float* tmp,
                                      hipStream_t stream);

// ---- icgn3d.hip ------------------------------------------------------------
struct Icgn3dParams {
    const float* ref;
    const float* gx;
    const float* gy;
    const float* gz;
    const float* coef;  // tricubic B-spline coefficient volume of the target
    int dz, dy, dx;
    int rx, ry, rz;
    float conv, stop;
    float* scratch;  // per-workgroup slots for the warped subvolume
    int samples_per_pass;  // samples (row mapping: steps of 16 rows) per thread between two coefficient-box stagings (set by the launchers)
    int tail_steps_per_pass;  // row mapping: steps of 512 tail samples per tail pass (set by launch_icgn3d1_rows)
    const unsigned* perm;     // locality schedule (poi_order.hip launch_poi3d_tile_order): the k-th solve takes POI perm[k]; nullptr = queue order
    int arith_fma;            // 1: the build whose per-sample multiply-adds are fused (icgn3d.hip only; the row mapping has no such build)
};
// floats of global scratch the kernel needs for this radius (0 when the subvolume fits LDS);
// *blocks receives the number of persistent workgroups in scratch mode (0 in LDS mode)
size_t icgn3d1_scratch_floats(int rx, int ry, int rz, int* blocks);
hipError_t launch_icgn3d1(const Icgn3dParams& p, float* pois, int stride_floats, size_t count, hipStream_t stream);
namespace sep {
hipError_t launch_icgn3d1(const Icgn3dParams& p, float* pois, int stride_floats, size_t count, hipStream_t stream);
}
namespace fma {
hipError_t launch_icgn3d1(const Icgn3dParams& p, float* pois, int stride_floats, size_t count, hipStream_t stream);
}
// icgn3d_rows.hip (A/B builds only, OC_BUILD_AB; NOT the default -- the default is icgn3d.hip, oracle order OC_ORDER_LANES): the
// same solver with one half-wave per subvolume row (oracle order OC_ORDER_ROWS), measured 12 - 25 % slower; its scratch
// slots are a little larger (whole steps): icgn3d1_rows_slot_floats floats per workgroup, 512 workgroups
size_t icgn3d1_rows_slot_floats(int rx, int ry, int rz);
hipError_t launch_icgn3d1_rows(const Icgn3dParams& p, float* pois, int stride_floats, size_t count, int blocks, hipStream_t stream);

// ---- fftcc2d.hip -----------------------------------------------------------
struct Fftcc2dParams {
    const float* ref;
    const float* tar;
    int height, width;
    int rx, ry;
};
// gathers zero-mean windows of POIs [first, first+count) into ref_win/tar_win (count x (2ry*2rx)),
// writes norms[2*i], norms[2*i+1] = sum of squares, flags[i] = 1 if the bounds guard fired
hipError_t launch_fftcc2d_gather(const Fftcc2dParams& p, const float* pois, int stride_floats, size_t count,
                                 float* ref_win, float* tar_win, float* norms, int* flags, hipStream_t stream);
// zf = conj(rf) * tf over `bins` complex values
hipError_t launch_fftcc_conjmul(const float2* rf, const float2* tf, float2* zf, size_t bins, hipStream_t stream);
// arg-max of each (2ry*2rx) surface with the first-max rule, writes u,v,u0,v0,zncc into the POIs
hipError_t launch_fftcc2d_argmax(const Fftcc2dParams& p, const float* surf, const float* norms, const int* flags,
                                 float* pois, int stride_floats, size_t count, hipStream_t stream);

// ---- fftcc2d_fusedn.hip ----------------------------------------------------
// the same for square windows of side 20, 24, 30, 36, 40, 48 (radii 10, 12, 15, 18, 20, 24): mixed-radix FFT, one line
// per lane
bool fftcc2d_fusedn_supported(int rx, int ry);
hipError_t launch_fftcc2d_fusedn(const Fftcc2dParams& p, float* pois, int stride_floats, size_t count, bool xcd,
                                 hipStream_t stream);

// ---- fftcc2d_rect.hip ------------------------------------------------------
// the same for every OTHER rectangular window with both radii in 4 ... 32: ONE kernel (two instantiations: sides <= 32, sides
// <= 64), the two sides are run-time values (each axis pass switches to the line transform of its length)
bool fftcc2d_rect_supported(int rx, int ry);
hipError_t launch_fftcc2d_rect(const Fftcc2dParams& p, float* pois, int stride_floats, size_t count, bool xcd, hipStream_t stream);

// ---- fftcc2d_fused.hip -----------------------------------------------------
// whole FFTCC2D::compute for 32 x 32 windows (rx == ry == 16) in one kernel, no rocFFT, no scratch
bool fftcc2d_fused_supported(int rx, int ry);
hipError_t launch_fftcc2d_fused(const Fftcc2dParams& p, float* pois, int stride_floats, size_t count, bool xcd,
                                hipStream_t stream);

// ---- fftcc3d.hip -----------------------------------------------------------
struct Fftcc3dParams {
    const float* ref;
    const float* tar;
    int dz, dy, dx;
    int rx, ry, rz;
    // locality schedule of the single-kernel paths (poi_order.hip launch_poi3d_tile_order): the k-th window of the launch
    // belongs to POI perm[k]; nullptr = queue order.  Round 5: the POIs in flight behind one L2 form a compact block of the
    // volume instead of a 1 x 1 x 32 row of the queue, so that their overlapping windows are fetched from HBM once.
    const unsigned* perm = nullptr;
};
hipError_t launch_fftcc3d_gather(const Fftcc3dParams& p, const float* pois, int stride_floats, size_t count,
                                 float* ref_win, float* tar_win, float* norms, hipStream_t stream);
hipError_t launch_fftcc3d_argmax(const Fftcc3dParams& p, const float* surf, const float* norms, float* pois,
                                 int stride_floats, size_t count, hipStream_t stream);

// ---- fftcc3d_fused.hip -----------------------------------------------------
// all of FFTCC3D::compute(POI3D*) in one kernel for 32 x 32 x 32 windows (radius 16)
bool fftcc3d_fused_supported(int rx, int ry, int rz);
hipError_t launch_fftcc3d_fused(const Fftcc3dParams& p, float* pois, int stride_floats, size_t count, bool xcd, hipStream_t stream);
#if OC_BUILD_AB
// fftcc3d_fused_r5.hip (A/B build): the decomposition of rounds 1 - 5.  needs_clamped: fftcc3d_fused_flag_bytes(count) bytes of device
// scratch (its first launch flags the POIs whose windows are clamped at a volume border and raises one "any" word behind the flags,
// its second one computes those -- and returns at once when the word is 0)
size_t fftcc3d_fused_flag_bytes(size_t count);
hipError_t launch_fftcc3d_fused_r5(const Fftcc3dParams& p, float* pois, int stride_floats, size_t count, bool xcd,
                                   unsigned char* needs_clamped, hipStream_t stream);
#endif

// ---- fftcc3d_fusedn.hip ----------------------------------------------------
// the same for cubic windows of side 8 ... 26 (radius 4 ... 13): the complex volume stays in LDS between the axis passes
bool fftcc3d_fusedn_supported(int rx, int ry, int rz);
hipError_t launch_fftcc3d_fusedn(const Fftcc3dParams& p, float* pois, int stride_floats, size_t count, bool xcd,
                                 hipStream_t stream);

// ---- fftcc3d_box.hip ---------------------------------------------------------
// the same for NON-cubic windows with every radius in 4 ... 16 whose complex volume fits the LDS: ONE kernel, the three sides
// are run-time values (each axis pass switches to the line transform of its length)
bool fftcc3d_box_supported(int rx, int ry, int rz);
hipError_t launch_fftcc3d_box(const Fftcc3dParams& p, float* pois, int stride_floats, size_t count, bool xcd, hipStream_t stream);

// ---- fftcc3d_planes.hip / fftcc3d_planesb.hip -------------------------------
// the same for cubic windows of side 28 ... 64 (except 32): one persistent 512-thread workgroup per scratch slot, the complex
// volume passes through a private N^3 scratch volume between the in-LDS plane transforms and the z pass
bool fftcc3d_planes_supported(int rx, int ry, int rz);
size_t fftcc3d_planes_scratch_bytes(int radius, int blocks);
hipError_t launch_fftcc3d_planes(const Fftcc3dParams& p, float* pois, int stride_floats, size_t count, void* scratch, int blocks,
                                 hipStream_t stream);

// ---- poi_split.hip ----------------------------------------------------------
// order-preserving partition of a POI queue by result quality (oc_hip_split_reliable / oc_hip_merge_recovered)
struct PoiSplitParams {
    int mode;        // 0: reliable | unreliable | neither (first selection), 1: recovered | still unreliable (after a round)
    int rec_floats;  // 25 (POI2D) or 31 (POI3D)
    int zncc_at, conv_at;
    float zncc_low, zncc_high, conv;
};
// flag[0] <- 1 when any of the n device-resident indices is >= limit, else 0
hipError_t launch_poi_index_range(const unsigned* index, size_t n, size_t limit, unsigned* flag, hipStream_t stream);
// unsigned words of device scratch: per-block counts, then the two totals and the bad-index flag (the last three words)
size_t poi_split_scratch_words(size_t count);
// main_queue / main_count: class-0 records are also written back to main_queue[index_in[i]] -- never past main_count
// records (the flag word is raised instead)
hipError_t launch_poi_split(const float* pois, int stride_floats, size_t count, const PoiSplitParams& p, const unsigned* index_in,
                            float* out0, size_t out0_offset, unsigned* index_out0, float* out1, unsigned* index_out,
                            float* main_queue, size_t main_count, unsigned* scratch, hipStream_t stream);

}  // namespace ochip
