/*
 * opencorr_hip.h -- C-ABI of the MI355X-native FFTCC -> ICGN engines.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.
 * Every entry point replaces one piece of OpenCorr's host interface for the
 * hot path (citations are file:line in the OpenCorr tree).  The OpenCorr-shaped
 * C++ classes in include/opencorr_compat/ (FFTCC2D, ICGN2D1, ...) are thin
 * shims over these functions; INTEGRATION.md shows the binding a maintainer of
 * the reference would add.
 *
 * Conventions
 *   - All functions return an oc_hip_status (0 = OK).  On failure
 *     oc_hip_last_error() returns a thread-local description; the C++ shim turns
 *     it into `throw std::string`, the reference's only exception type
 *     (src/oc_fftcc.cpp:145, src/oc_icgn.cpp:65, src/oc_image.cpp:43).
 *   - Per-POI failures stay in-band exactly like the reference: result.zncc is
 *     set to -3 / -4 / -5 (src/oc_dic.h:28-34) and nothing else is written.
 *   - POIs are passed as the reference's own AoS records, updated in place:
 *     POI2D = 25 floats / 100 B (src/oc_poi.h:102-136), POI3D = 31 floats / 124 B
 *     (src/oc_poi.h:187-222).  `stride_bytes` lets a caller embed them in a
 *     larger struct (must be a multiple of 4).
 *   - An engine lives on the device named at creation.  Every entry point makes that
 *     device current for its own duration and hands the calling thread's current
 *     device back on return.
 *   - `memory` says where a buffer lives: OC_HIP_HOST buffers are copied by the
 *     engine (pageable or pinned), OC_HIP_DEVICE buffers are used in place on the
 *     engine's device and stream (no copy).  On a stream the caller chose with
 *     oc_hip_set_stream such a call is asynchronous and stream-ordered (chain the
 *     engines of a pipeline on one stream); on the engine's own private stream it
 *     completes before it returns.  Producer ordering of OC_HIP_DEVICE inputs (queues,
 *     offsets, images used in place): on a caller-chosen stream they are read in that
 *     stream's order; on the private stream the engine first waits for everything the
 *     caller has enqueued on HIP's legacy default stream at the time of the call.  Data
 *     still being written on any OTHER stream must be complete (or that stream be named
 *     with oc_hip_set_stream) before the call.
 *   - Images are snapshotted at set_images time (like the CUDA module of the
 *     reference, examples/test_2d_dic_gpu_icgn.cpp:99-136): later edits of the
 *     host image need another set_images + prepare.
 *   - The engines never fall back to a CPU path.  If the HIP runtime, rocFFT or
 *     the device is unavailable the call fails with OC_HIP_ERR_HIP / _ROCFFT.
 */
#ifndef OPENCORR_HIP_H_
#define OPENCORR_HIP_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum oc_hip_status {
    OC_HIP_OK = 0,
    OC_HIP_ERR_INVALID = 1, /* bad argument or call order (e.g. compute before prepare) */
    OC_HIP_ERR_HIP = 2,     /* HIP runtime error (no device, launch failure, ...) */
    OC_HIP_ERR_ROCFFT = 3,  /* rocFFT plan / execute failure */
    OC_HIP_ERR_NOMEM = 4,   /* device allocation failed */
    OC_HIP_ERR_UNSUPPORTED = 5 /* subset too large for the on-chip working set */
} oc_hip_status;

typedef enum oc_hip_memory { OC_HIP_HOST = 0, OC_HIP_DEVICE = 1 } oc_hip_memory;

/* Image2D::eg_mat is an Eigen::MatrixXf, i.e. column-major (src/oc_image.h:37);
 * Img2D of the CUDA module is row-major (gpu_lib/opencorr_gpu.h:31-35). */
typedef enum oc_hip_layout { OC_HIP_ROW_MAJOR = 0, OC_HIP_COL_MAJOR = 1 } oc_hip_layout;

typedef enum oc_hip_kind {
    OC_HIP_FFTCC2D = 1,
    OC_HIP_ICGN2D1 = 2,
    OC_HIP_ICGN2D2 = 3,
    OC_HIP_FFTCC3D = 4,
    OC_HIP_ICGN3D1 = 5,
    OC_HIP_NR2D1 = 6,
    OC_HIP_ICLM2D1 = 7,
    OC_HIP_ICLM2D2 = 8,
    OC_HIP_STRAIN = 9,
    OC_HIP_REGION_FIT = 10
} oc_hip_kind;

#define OC_HIP_POI2D_BYTES 100
#define OC_HIP_POI3D_BYTES 124

typedef struct oc_hip_engine oc_hip_engine; /* opaque */

/* thread-local text of the last failure on the calling thread */
const char* oc_hip_last_error(void);
/* number of visible HIP devices (0 + OC_HIP_ERR_HIP when the runtime is unusable) */
int oc_hip_device_count(int* count);
/* library / ABI version, bumped on any signature change */
int oc_hip_abi_version(void);

/* ---- construction: one per reference class ------------------------------ */
/* FFTCC2D(int subset_radius_x, int subset_radius_y, int thread_number)  src/oc_fftcc.cpp:151-163 */
int oc_hip_fftcc2d_create(int radius_x, int radius_y, int device, oc_hip_engine** out);
/* ICGN2D1(int rx, int ry, float conv_criterion, float stop_condition, int thread_number)  src/oc_icgn.cpp:71-88 */
int oc_hip_icgn2d1_create(int radius_x, int radius_y, float conv_criterion, float stop_condition, int device,
                          oc_hip_engine** out);
/* ICGN2D2(int rx, int ry, float conv, float stop, int thread_number)  src/oc_icgn.cpp:612-629 */
int oc_hip_icgn2d2_create(int radius_x, int radius_y, float conv_criterion, float stop_condition, int device,
                          oc_hip_engine** out);
/* NR2D1(int rx, int ry, float conv_criterion, float stop_condition, int thread_number)  src/oc_nr.cpp:75-91
 * (forward-additive Newton-Raphson).  prepare() = NR2D1::prepare (:119-158): target gradients and
 * three bicubic tables; compute() = NR2D1::compute(poi_queue) (:324-332).  Per-POI codes: -1 (guard), -4, -5. */
int oc_hip_nr2d1_create(int radius_x, int radius_y, float conv_criterion, float stop_condition, int device,
                        oc_hip_engine** out);
/* ICLM2D1 / ICLM2D2(int rx, int ry, float conv_criterion, float stop_condition, int thread_number)
 * src/oc_iclm.cpp:70-87, 422-439 (inverse-compositional Levenberg-Marquardt).  prepare() as for
 * ICGN2D*; compute() = ICLM2D1::compute(poi_queue) (:360-368) / ICLM2D2::compute(poi_queue) (:733-741); both honour
 * oc_hip_set_self_adaptive.  Per-POI codes: -3 (guard), -4, -5; unlike ICGN there is no abort when the warped subset
 * leaves the image (out-of-range samples take the interpolator's -1.f, src/oc_iclm.cpp:230-243). */
int oc_hip_iclm2d1_create(int radius_x, int radius_y, float conv_criterion, float stop_condition, int device,
                          oc_hip_engine** out);
int oc_hip_iclm2d2_create(int radius_x, int radius_y, float conv_criterion, float stop_condition, int device,
                          oc_hip_engine** out);
/* ICLM2D1::setDamping / ICLM2D2::setDamping(float lambda, float alpha, float beta)  src/oc_iclm.cpp:114-119, 466-471;
 * defaults 100, 0.1, 10 (struct DampingParameter, src/oc_iclm.h:33-38).  lambda must be > 0. */
int oc_hip_set_damping(oc_hip_engine* engine, float lambda, float alpha, float beta);
/* Strain(float subregion_radius, int neighbor_number_min, int thread_number)  src/oc_strain.cpp:31-46
 * (first consumer of the device-resident displacement field).  The handle supports set_stream,
 * synchronize, profile_* and destroy like the other engines; it holds no images. */
int oc_hip_strain_create(float subregion_radius, int neighbor_number_min, int device, oc_hip_engine** out);
/* setSubregionRadius / setNeighborMin / setZnccThreshold / setApproximation  src/oc_strain.cpp:72-95
 * (defaults of the constructor: threshold 0.9, approximation 1 = Cauchy; 2 = Green).  A new radius needs a new
 * oc_hip_strain_prepare. */
int oc_hip_strain_set(oc_hip_engine* engine, float subregion_radius, int neighbor_number_min, float zncc_threshold,
                      int approximation);
/* Strain::prepare(std::vector<POI2D>&) / (std::vector<POI3D>&)  src/oc_strain.cpp:96-107, 136-147: builds the
 * neighbour search over the queue's coordinates (the reference: one kd-tree per thread; here a cell-sorted order on
 * the device).  ndim 2 = POI2D records, 3 = POI3D records; layout and `memory` as for oc_hip_compute. */
int oc_hip_strain_prepare(oc_hip_engine* engine, const void* pois, size_t count, size_t stride_bytes, int ndim, int memory);
/* Strain::compute(std::vector<POI2D>&) src/oc_strain.cpp:236-247 / (std::vector<POI3D>&) :476-488.  Writes
 * strain.exx, eyy, exy (POI2D floats 20..22) or exx, eyy, ezz, exy, eyz, ezx (POI3D floats 22..27) of every POI with
 * ZNCC >= threshold that finds at least neighbor_number_min accepted neighbours; everything else is left untouched.
 * The queue must be the one prepare() saw (same length and coordinates). */
int oc_hip_strain_compute(oc_hip_engine* engine, void* pois, size_t count, size_t stride_bytes, int ndim, int memory);
/* RegionFit2D / RegionFit3D(float neighbor_search_radius, int neighbor_number_min, int thread_number)
 * src/oc_region_fit.cpp:32-46, 189-203: re-initialises POIs from the plane fitted through the
 * reliable POIs around them, before they are handed to ICGN again. */
int oc_hip_region_fit_create(float neighbor_search_radius, int neighbor_number_min, int device, oc_hip_engine** out);
/* setSearchRadius / setNeighborMin  src/oc_region_fit.cpp:65-73, 222-230.  A new radius needs a new prepare. */
int oc_hip_region_fit_set(oc_hip_engine* engine, float neighbor_search_radius, int neighbor_number_min);
/* setNeighbor(reliable_pois) + prepare()  src/oc_region_fit.cpp:75-92, 232-249: neighbour search over the reliable
 * POIs' coordinates.  Their displacements are snapshotted here (the reference keeps a pointer and reads them during
 * compute; editing the reliable queue between prepare and compute needs a new prepare). */
int oc_hip_region_fit_prepare(oc_hip_engine* engine, const void* reliable_pois, size_t count, size_t stride_bytes, int ndim,
                              int memory);
/* RegionFit2D::compute(std::vector<POI2D>&) src/oc_region_fit.cpp:166-174 (-> :94-164) / RegionFit3D :334-342
 * (-> :251-332): every POI that finds at least neighbor_number_min reliable POIs (inside the radius, else the K
 * nearest) gets deformation.{u,ux,uy,v,vx,vy} (POI3D: all twelve) from the fitted plane and result.zncc = 0; other
 * POIs and all other fields are left untouched. */
int oc_hip_region_fit_compute(oc_hip_engine* engine, void* pois, size_t count, size_t stride_bytes, int ndim, int memory);
/* Candidate batching for callers like EpipolarSearch::compute(POI2D*) (src/oc_epipolar_search.cpp:133-195): the
 * reference refines a few trial positions per POI with icgn1->compute(&candidate) one at a time and keeps the one
 * with the highest ZNCC (:181-190).  Here the trials of all POIs are ONE queue for oc_hip_compute, and this call
 * does the selection: candidates segment_starts[s] .. segment_starts[s+1]-1 belong to POI s (n_segments + 1 offsets,
 * uint32, same memory space as the queues); the winner's deformation and result vectors are copied into POI s
 * (poi->deformation = best.deformation; poi->result = best.result), nothing else is touched; highest ZNCC wins, the
 * earliest candidate among equals, NaN never; an empty segment leaves its POI as it was.  POI2D records; any 2D
 * engine handle supplies the device and stream. */
int oc_hip_select_best(oc_hip_engine* engine, const void* candidates, size_t n_candidates, size_t candidate_stride_bytes,
                       const unsigned* segment_starts, size_t n_segments, void* pois, size_t stride_bytes, int memory);
/* The two selections of the RegionFit -> re-ICGN loop (examples/test_3d_reconstruction_sift_icgn2_regfit.cpp:214-260), as
 * order-preserving partitions ON THE DEVICE, so that a queue resident in HBM never travels to the host between ICGN,
 * RegionFit and the next ICGN pass.  Any engine handle supplies the device and the stream; ndim = 2 (POI2D) / 3 (POI3D);
 * all queues share `stride_bytes`; the counts come back to the host (the call completes before it returns).
 *
 * oc_hip_split_reliable: for every POI of `pois` in queue order (:216-229)
 *     result.zncc < zncc_threshold_low || result.convergence > conv_criterion  -> appended to `unreliable`, its queue index to
 *                                                                                 `unreliable_index`
 *     else result.zncc >= zncc_threshold_high                                  -> appended to `reliable` (from record
 *                                                                                 `reliable_offset` on)
 *   (the example's own float comparisons: a NaN makes all of them false, the POI goes to neither set).  The caller's
 *   buffers hold up to `count` records (reliable: reliable_offset + count). */
int oc_hip_split_reliable(oc_hip_engine* engine, const void* pois, size_t count, size_t stride_bytes, int ndim,
                          float zncc_threshold_low, float zncc_threshold_high, float conv_criterion, void* reliable,
                          size_t reliable_offset, void* unreliable, unsigned* unreliable_index, size_t* n_reliable,
                          size_t* n_unreliable, int memory);
/* oc_hip_merge_recovered: after RegionFit + ICGN over `unreliable` (:236-256): every POI with result.zncc >=
 * zncc_threshold_high && result.convergence <= conv_criterion is written back to pois[unreliable_index[j]] and appended to
 * `reliable` (from record `reliable_offset` on); the others are moved to the front of `unreliable` / `unreliable_index`
 * in their old order.  (The example erases inside its loop and thereby skips the POI after every success for one round;
 * here every POI is looked at in every round.)  `count` = records in `pois`: an unreliable_index entry >= count is never
 * written through -- the call fails with OC_HIP_ERR_INVALID (HOST: before anything is touched; DEVICE: the offending
 * records are not written back, `unreliable` / `unreliable_index` are left as they were).  On both paths only the
 * record's own bytes are written: with stride_bytes > the record size the caller's bytes between records are kept. */
int oc_hip_merge_recovered(oc_hip_engine* engine, void* pois, size_t count, size_t stride_bytes, int ndim, void* unreliable,
                           unsigned* unreliable_index, size_t n_unreliable, float zncc_threshold_high, float conv_criterion,
                           void* reliable, size_t reliable_offset, size_t* n_recovered, size_t* n_remaining, int memory);
/* FFTCC3D(int rx, int ry, int rz, int thread_number)  src/oc_fftcc.cpp:300-313 */
int oc_hip_fftcc3d_create(int radius_x, int radius_y, int radius_z, int device, oc_hip_engine** out);
/* ICGN3D1(int rx, int ry, int rz, float conv, float stop, int thread_number)  src/oc_icgn.cpp:1197-1213 */
int oc_hip_icgn3d1_create(int radius_x, int radius_y, int radius_z, float conv_criterion, float stop_condition,
                          int device, oc_hip_engine** out);
/* ~FFTCC2D / ~ICGN2D1 ...  src/oc_fftcc.cpp:165-173, src/oc_icgn.cpp:90-101 */
int oc_hip_destroy(oc_hip_engine* engine);

/* ---- configuration ------------------------------------------------------- */
/* DIC::setImages(Image2D&, Image2D&)  src/oc_dic.cpp:22-26 (pointer stored there; snapshotted here) */
int oc_hip_set_images2d(oc_hip_engine* engine, const float* ref, const float* tar, int height, int width,
                        int layout, int memory);
/* DVC::setImages(Image3D&, Image3D&)  src/oc_dic.cpp:45-49; data = &vol_mat[0][0][0] (z,y,x contiguous) */
int oc_hip_set_images3d(oc_hip_engine* engine, const float* ref, const float* tar, int dim_x, int dim_y, int dim_z,
                        int memory);
/* Reuse the device-resident images of another engine on the same device (an
 * FFTCC and an ICGN engine working on one image pair upload it once).  The
 * donor must outlive the borrower or call set_images again. */
int oc_hip_share_images(oc_hip_engine* engine, oc_hip_engine* donor);
/* DIC::setSubset / DVC::setSubset  src/oc_dic.cpp:28-32,51-56 (radius_z ignored by 2D engines) */
int oc_hip_set_subset(oc_hip_engine* engine, int radius_x, int radius_y, int radius_z);
/* ICGN2D1::setIteration(float, float)  src/oc_icgn.cpp:103-107 (also 2D2 :644-648, 3D1 :1228-1232) */
int oc_hip_set_iteration(oc_hip_engine* engine, float conv_criterion, float stop_condition);
/* Run on a caller-owned hipStream_t (e.g. torch's current stream) instead of
 * the engine's own stream.  The handle is used as given: NULL is HIP's default
 * (null) stream, which is what torch.cuda.current_stream().cuda_stream returns
 * for torch's default stream.  The new stream is ordered behind the work already
 * enqueued on the stream in use so far (prepare()'s kernels, say) ON THE DEVICE --
 * an event, no host-side wait -- so switching never blocks and never races.  The
 * outgoing stream may already have been destroyed by its owner (destroy, then
 * set_stream(other) is a valid sequence): a dead stream has nothing left to order
 * behind, the new one is installed regardless.  Fails only if the NEW handle is
 * unusable. */
int oc_hip_set_stream(oc_hip_engine* engine, void* hip_stream);
/* Go back to the engine's own (non-blocking) stream; ordered like set_stream, always succeeds. */
int oc_hip_reset_stream(oc_hip_engine* engine);

/* ---- device groups: one engine, several GPUs of the node -------------------------------
 * The loop being replaced is the reference's per-POI loop, src/oc_icgn.cpp:343-351 (`#pragma omp parallel for` over
 * poi_queue): every POI is independent, so member g of the group takes the contiguous block
 * [g * ceil(n / G), (g + 1) * ceil(n / G)) of the queue.  device_ids[0] is where the engine itself lives (it moves
 * there if necessary: images must then be set again); for each further id a full engine of the same kind and
 * settings is created on that device.  From then on every call on the handle fans out: setters, set_images (host
 * images are uploaded by every member, device images are copied peer to peer), prepare (every member builds its own
 * gradients / tables: cheaper than moving 64 B per pixel over xGMI) and compute:
 *   - OC_HIP_HOST queues: every member moves and solves its own block from its own host thread; results land in the
 *     caller's vector, no exchange is needed.
 *   - OC_HIP_DEVICE queues (resident on device_ids[0]): members pull their block over xGMI (hipMemcpyPeerAsync),
 *     solve it and push the records back; stream-ordered like a single-device call.  With the tuning key
 *     "group_allgather" = 1 every member additionally ends up with the COMPLETE result queue in its own mirror
 *     (oc_hip_group_queue) -- one ncclAllGather (RCCL) of equal, padded blocks when the members sit on distinct
 *     devices -- for consumers that run on every GPU (Strain / RegionFit per device).
 * Results are bit-identical for every group size (a POI's arithmetic does not depend on the block it travels in).
 * A device may be named more than once (two members on one GPU): only useful to exercise the sharding on a one-GPU
 * machine.  n_devices = 1 dissolves a group.  Strain / RegionFit engines stay on one device. */
int oc_hip_set_devices(oc_hip_engine* engine, const int* device_ids, int n_devices);
/* members of the group (1 for a plain engine); device_ids may be NULL */
int oc_hip_get_devices(const oc_hip_engine* engine, int* device_ids, int capacity, int* n_devices);
/* after a DEVICE-queue compute with "group_allgather" = 1: member `member`'s copy of the whole queue on ITS device --
 * G blocks of block_bytes (ceil(n / G) records; the last block padded), valid once the engine's stream has been
 * synchronised */
int oc_hip_group_queue(const oc_hip_engine* engine, int member, const void** device_ptr, size_t* block_bytes);

/* Knobs.  Every key but "arith_fma" selects among kernels that compute BIT-IDENTICAL results; unknown keys and values
 * outside the stated range fail with OC_HIP_ERR_INVALID, values this build does not contain with OC_HIP_ERR_UNSUPPORTED.
 *   "arith_fma"       ICGN2D1 / ICGN2D2 / ICLM2D1 / ICLM2D2 / ICGN3D1 only (other engines refuse 1).  0 (default): every
 *                     multiply and add of the solver rounds on its own -- the reference built for baseline x86-64; GPU ==
 *                     oracle OC_ORDER_LANES bit for bit.  1: every PER-SAMPLE multiply-add is one fused multiply-add -- what a
 *                     compiler with FMA hardware makes of the reference's source expressions (src/oc_cubic_bspline.cpp:159-177,
 *                     390-401; src/oc_icgn.cpp:198-205, 266-276, 1314-1445; the reference's build files fix no contraction
 *                     mode); GPU == oracle OC_ORDER_LANES_FMA bit for bit.  Results of the two modes differ by rounding only:
 *                     against the reference's separately rounded loop order both keep identical failure flags, >= 99.5 %
 *                     identical iteration counts, |d u, v, w| <= 1e-4 and |d ZNCC| <= 1e-5 (all BASELINE configs:
 *                     tests/test_gpu_fullsize.py).  Kernel time on one MI355X: ICGN2D1 -3 ... -9 %, ICGN2D2 -13 ... -17 %,
 *                     ICGN3D1 -5 ... -7 %.  The once-per-POI dense algebra is never fused
 *   "icgn2d_variant"  launch shape of the ICGN2D kernel (gather depth, LDS footprint, per-workgroup coordinate table, waves per
 *                     workgroup); -1 (default) lets the engine choose: 5 / 4 (6 / 12 DoF: coordinate table, lockstep sweeps,
 *                     8-wave workgroups) for queues >= 32768 POIs of subsets up to 35 x 34 / 41 x 41, 2 / 3 (no table) below,
 *                     7 (target array only) for self-adaptive subsets from 27 passes up (2 / 3 below), 1 (single-wave
 *                     workgroups) for what fits nothing else.  0, 6, 8 (the split launch shape) and 9 (the workgroup's
 *                     coefficient band staged in LDS, icgn2d_band.hip) are measured losers that only the A/B build of the
 *                     library contains
 *   "icgn2d_xcd"      1 (default): workgroups of one XCD serve a contiguous range of the POI queue
 *   "icgn2d_tile_px"  side of the square image tiles the ICGN2D / NR2D1 queue is visited by (L1 / L2 locality; default 128;
 *                     0 = queue order; applied to queues >= 16384 POIs)
 *   "icgn2d_split_chunks"  A/B build, variant 8 only: chunks of the two-stream pipeline (0 = the two kernels back to back)
 *   "fftcc2d_fused"   1 (default): single-kernel FFTCC2D (register / LDS FFT) for EVERY window with both radii in 4 ... 32 (even
 *                     sides 8 ... 64): a template instance per square side and for the 42 rectangular pairs (radius_x !=
 *                     radius_y) out of sides 16, 20, 24, 32, 40, 48, 64, one kernel with run-time sides (fftcc2d_rect.hip) for
 *                     every other rectangular pair; larger windows run the rocFFT pipeline.  0: rocFFT pipeline always.  2: as
 *                     1, but 32 x 32 windows run the generic NR x NC kernel instead of their own (an A/B switch)
 *   "fftcc3d_fused"   1 (default): single-kernel FFTCC3D for every cubic window of even side 8 ... 64 (radius 4 ... 32): LDS kernel
 *                     up to 26^3, register kernel at 32^3, plane-wise kernel for 28^3 ... 64^3; and for every NON-cubic window
 *                     with all three radii in 4 ... 16 whose complex volume [2rx][2ry][2rz + 1] fits 160 KB of LDS (one kernel,
 *                     sides as run-time values: fftcc3d_box.hip); other non-cubic windows and larger sides run the rocFFT
 *                     pipeline.  0: rocFFT pipeline always.  2 (A/B build of the library only): as 1, but 32^3 windows run the
 *                     kernel of rounds 1 - 5 (fftcc3d_fused_r5.hip: same integers, ZNCC within 1e-6, 1.45 x the time)
 *   "fftcc3d_planes_blocks"  persistent workgroups (= private scratch volumes) of the plane-wise FFTCC3D kernel; 0 (default) = 256
 *   "fftcc3d_tile_vox"  FFTCC3D single-kernel paths: queues >= 2048 POIs are visited in cubic blocks of this many voxels
 *                     (default 64; 0 = queue order; >= 8)
 *   "icgn3d_tile_vox" the same for ICGN3D1 (default 64; 0 = queue order; >= 8)
 *   "icgn3d_mapping"  0 (default, the only value the shipped library accepts): sample s of a subvolume is owned by thread
 *                     s mod 512 (icgn3d.hip; oracle OC_ORDER_LANES, lanes = 512); 1 (A/B build): one half-wave per subvolume row
 *                     (oracle OC_ORDER_ROWS; 12 - 25 % slower; no fused-arithmetic form)
 *   "single_combine"  1 (default): concurrent compute_one calls on an engine are combined into one launch per batch; 0: every
 *                     call is a launch of its own (the behaviour up to round 5).  Same bits either way.
 *   "host_chunk"      POIs per chunk of the host-queue pipeline (copies of one chunk overlap the kernels of its
 *                     neighbours); 0 = whole queue at once; default 65536
 *   "group_allgather" 1: device groups leave the complete result queue on every member (see oc_hip_set_devices)
 *   "group_force_rccl" 1 (with "group_allgather"): an engine WITHOUT a group sends its DEVICE queues down the group path
 *                     as a group of one, whose all-gather is an ncclAllGather on a one-rank communicator (queue ->
 *                     oc_hip_group_queue(engine, 0)); fails instead of falling back when librccl is unusable.  Exists
 *                     so that the RCCL binding (dlopen, version check, ncclCommInitAll, ncclAllGather) can be executed
 *                     on a one-GPU machine
 * A device group (oc_hip_set_devices) hands every key on to its members. */
int oc_hip_set_tuning(oc_hip_engine* engine, const char* key, int value);

/* ---- precompute ----------------------------------------------------------- */
/* ICGN2D1::prepare()  src/oc_icgn.cpp:138-142 (prepareRef + prepareTar); FFTCC::prepare() is a no-op
 * in the reference (src/oc_fftcc.cpp:175) and here. */
int oc_hip_prepare(oc_hip_engine* engine);
/* ICGN2D1::prepareRef()  src/oc_icgn.cpp:115-125: reference gradients (Gradient2D4 / Gradient3D4) */
int oc_hip_prepare_ref(oc_hip_engine* engine);
/* ICGN2D1::prepareTar()  src/oc_icgn.cpp:127-136: target B-spline coefficients */
int oc_hip_prepare_tar(oc_hip_engine* engine);

/* ---- compute --------------------------------------------------------------- */
/* FFTCC2D::compute(std::vector<POI2D>&)  src/oc_fftcc.cpp:277-285
 * ICGN2D1::compute(std::vector<POI2D>&)  src/oc_icgn.cpp:343-351   (2D2 :900-908, 3D1 :1492-1500,
 * FFTCC3D :429-436).  `pois` = poi_queue.data(), `count` = poi_queue.size().
 * With OC_HIP_HOST the call returns after the results are back in `pois` (the queue travels in chunks: H2D, kernels
 * and D2H of neighbouring chunks overlap); with OC_HIP_DEVICE it only enqueues work on the engine's stream. */
int oc_hip_compute(oc_hip_engine* engine, void* pois, size_t count, size_t stride_bytes, int memory);
/* Several engines over ONE queue, in the given order -- what examples/test_2d_dic_fftcc_icgn1.cpp:80-99 does with two
 * calls (fftcc2d->compute(poi_queue); icgn2d1->compute(poi_queue);) as ONE: an OC_HIP_HOST queue then crosses PCIe once
 * in each direction instead of once per engine (per chunk: one copy in, every engine's kernels, one copy out; config B:
 * two crossings of the 25 MB AoS instead of four).  Results are bit-identical to the separate calls.  All engines live
 * on one device and take the same record type (POI2D or POI3D); for the duration of the call they run on engines[0]'s
 * stream (device-ordered behind whatever their own streams still hold, and handed back afterwards).  Centre offsets,
 * device groups and Strain / RegionFit are not part of a chain. */
int oc_hip_compute_chain(oc_hip_engine* const* engines, int n_engines, void* pois, size_t count, size_t stride_bytes, int memory);
/* FFTCC2D::compute(POI2D*) / ICGN2D1::compute(POI2D*)  src/oc_fftcc.cpp:177, src/oc_icgn.cpp:144: safe to call from the
 * caller's own OpenMP region (src/oc_epipolar_search.cpp:184-188; the reference keeps one scratch instance per thread,
 * src/oc_icgn.cpp:61-69,147).  Calls that arrive while a launch is in flight are COMBINED: they queue up, one of the waiting
 * threads hands the whole batch to the engine as one queue and every caller returns with its own record filled in -- T
 * concurrent threads cost one launch per ~T POIs.  A POI's result does not depend on the batch it travels in.  A strictly
 * sequential caller still pays a launch and two PCIe copies per POI.  Tuning key "single_combine" = 0 restores one launch per
 * call. */
int oc_hip_compute_one(oc_hip_engine* engine, void* poi);
/* batches / POIs served by the combining front end of compute_one on this engine so far (either pointer may be null) */
int oc_hip_single_stats(oc_hip_engine* engine, unsigned long long* batches, unsigned long long* pois);
/* ICGN2D1::compute(std::vector<POI2D>&, std::vector<Point2D>& center_offset_queue)  src/oc_icgn.cpp:549-557
 * (ICGN2D2 :1128-1136): local subset coordinates are shifted by center_offsets[i] = {x, y} (two floats per POI,
 * the reference's Point2D) and the target subset is centred at POI + offset.  The offsets live in the same
 * memory space as the POIs. */
int oc_hip_compute_with_offsets(oc_hip_engine* engine, void* pois, const float* center_offsets, size_t count,
                                size_t stride_bytes, int memory);
/* ICGN2D1::compute(POI2D*, Point2D& center_offset)  src/oc_icgn.cpp:353 (ICGN2D2 :910) */
int oc_hip_compute_one_with_offset(oc_hip_engine* engine, void* poi, const float* center_offset);
/* DIC::setSelfAdaptive(bool)  src/oc_dic.cpp:34-37: when on, ICGN2D1/2D2 take the subset radius of every POI from
 * poi->subset_radius instead of the engine's (src/oc_icgn.cpp:152-158, 697-703) */
int oc_hip_set_self_adaptive(oc_hip_engine* engine, int enable);
/* wait for everything enqueued on the engine's stream */
int oc_hip_synchronize(oc_hip_engine* engine);

/* ---- introspection (tests, bench, profiling) ------------------------------ */
int oc_hip_get_kind(const oc_hip_engine* engine, int* kind);
/* Device pointers of the precomputed fields (row-major float32):
 *   "ref","tar"           height*width            (3D: dz*dy*dx)
 *   "gx","gy"[,"gz"]      same shape              ICGN engines after prepare_ref
 *   "lut"                 4*height*width*4        ICGN2D* / NR2D1 after prepare_tar: the bicubic coefficient table of
 *                                                 src/oc_cubic_bspline.cpp:123-129, stored planar as [k][y][x][l]
 *                                                 (plane k = coef[k][0..3] of every pixel)  (3D: "coef", dz*dy*dx)
 *   "lut_gx","lut_gy"     4*height*width*4        NR2D1 after prepare: tables of the target gradients, same layout
 * Returns OC_HIP_ERR_INVALID for an unknown name or a field not built yet. */
int oc_hip_get_field(const oc_hip_engine* engine, const char* name, const float** device_ptr, size_t* count);
/* Copy a field to host memory (test helper). */
int oc_hip_read_field(oc_hip_engine* engine, const char* name, float* host_dst, size_t count);
/* Per-kernel timing with hipEvents recorded on the engine's stream around every
 * launch of the engine's dominant kernel ("icgn" / "fftcc" pipeline).  Enable,
 * run compute() any number of times, synchronize, then read.  */
int oc_hip_profile_enable(oc_hip_engine* engine, int enable);
int oc_hip_profile_read(oc_hip_engine* engine, double* total_ms, long* launches);
int oc_hip_profile_reset(oc_hip_engine* engine);

#ifdef __cplusplus
}
#endif
#endif /* OPENCORR_HIP_H_ */
