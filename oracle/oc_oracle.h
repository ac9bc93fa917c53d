/*
 * oc_oracle.h -- CPU oracle for the FFTCC -> ICGN hot path of OpenCorr.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is a from-scratch float32 CPU
 * restatement of the reference algorithm (file:line citations are relative to
 * the OpenCorr tree).  It is the checker for the HIP engine and the timed
 * "CPU baseline"; nothing in the product path (opencorr_amd/, include/) may
 * link or call it.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg load it.
 *
 * Parity status of the oracle itself:
 *   - Pinned on the reference ITSELF: oracle/_ref/liboc_ref.so is /root/reference/src/oc_{fftcc,icgn,iclm,nr,
 *     cubic_bspline,gradient,subset,deformation,dic,image}.cpp compiled UNMODIFIED (oracle/Makefile `ref`) against
 *     stand-in Eigen / FFTW / OpenCV headers (oracle/ref_stubs).  tests/test_oracle_vs_ref.py: in OC_ORDER_SEQ this
 *     oracle equals the reference bit for bit -- every float of every POI -- for ICGN2D1, ICGN2D2, both centre-offset
 *     overloads, self-adaptive subsets, ICLM2D1 / ICLM2D2, NR2D1, ICGN3D1, Gradient2D4 / 3D4, BicubicBspline and
 *     TricubicBspline; FFTCC2D / FFTCC3D give identical integer displacements and ZNCC within 1e-5 / 1e-4.
 *     Since round 3 the same library also holds oc_strain.cpp, oc_region_fit.cpp and oc_nearest_neighbor.cpp (against a
 *     stand-in nanoflann and a float colPivHouseholderQr in the stand-in Eigen): tests/test_oracle_vs_ref_strain.py --
 *     Strain (2D and 3D) and RegionFit2D / RegionFit3D compute exactly the same POIs as the reference (ZNCC filters,
 *     radius search, K-nearest fallback, minimum counts; untouched fields bit-identical) with strains / gradients
 *     within 1e-6 and fitted displacements within 1e-5 (this file solves the plane fit in double, the reference by a
 *     float QR).
 *   - Pinned on the reference's golden vectors (its authors' own runs, real Eigen + FFTW): FFTCC2D + ICGN2D1
 *     (examples/2d_dic/oht_cfrp_4_fftcc_icgn1_r16*.csv), NR2D1 (oht_cfrp_4_fftcc_nr1_r16.csv) and Strain (2D) in
 *     tests/test_oracle_golden.py, tests/test_oracle_strain.py; ICGN2D2 soft-anchored on the authors' CUDA results.
 *   - Still "parity unpinned": only what lives INSIDE the third-party libraries -- Eigen's association of mean() /
 *     squaredNorm() and of its LU / cofactor inverses, FFTW's butterflies (the stand-ins restate them the same way
 *     this file does), glibc vs MSVC powf in IC-LM's first damping value, the rounding of Eigen's float Householder QR
 *     in the plane fits, and nanoflann's choice among EQUIDISTANT K-nearest neighbours (tree-traversal order; the
 *     reference has no defined result there).
 *
 * All images are row-major float32 (x fastest): img[y*width + x]; volumes are
 * vol[(z*dim_y + y)*dim_x + x] (same as Image3D::vol_mat, src/oc_array.h:57-74).
 * POIs are passed as the reference's own AoS records:
 *   POI2D = 25 floats {x,y | p[12] | r[6] | e[3] | subset_radius x,y}   (src/oc_poi.h:102-136)
 *   POI3D = 31 floats {x,y,z | p[12] | r[7] | e[6] | subset_radius xyz} (src/oc_poi.h:187-222)
 *
 * Summation order ("order" argument of the ICGN entry points):
 *   OC_ORDER_SEQ    : every reduction is one sequential float32 loop in
 *                     row-major sample order -- the order the reference's hand
 *                     written loops use (src/oc_icgn.cpp:198-205,266-276).
 *                     Reductions the reference delegates to Eigen (mean(),
 *                     squaredNorm()) have no defined order there; SEQ uses the
 *                     same row-major loop for them.
 *   OC_ORDER_LANES  : the association the GPU engine uses: `lanes` strided
 *                     partial sums (sample s belongs to lane s % lanes, each
 *                     lane adds its samples in increasing s) combined by an
 *                     xor-butterfly with ascending offsets 1,2,4,...,lanes/2.
 *                     The 2D HIP kernels (lanes = 64) and the ICGN3D1 kernel of
 *                     icgn3d.hip (lanes = 512) are bit-exact against this mode.
 *   OC_ORDER_ROWS   : ICGN3D1 only -- the association of the row-mapping A/B kernel (NOT the
 *                     default: that is icgn3d.hip / OC_ORDER_LANES with lanes = 512)
 *                     (icgn3d_rows.hip, one half-wave per subvolume row), `lanes`
 *                     = 512 threads: with SX = 2rx+1 samples per row and NCH =
 *                     (SX/32 >= 2 ? 2 : SX/32 + (SX%32 >= 28)) chunks of 32 columns,
 *                     sample (row r = i*SY + j, column k) with k < BW = min(SX, 32 NCH)
 *                     belongs to thread (r % 16)*32 + k % 32 ("body"), the other
 *                     columns to thread q % 512, q = r*(SX-BW) + (k-BW) ("tail");
 *                     a thread adds its body samples in increasing (r, k), then its
 *                     tail samples in increasing q, all into ONE running sum; the
 *                     512 partials are combined by the same xor-butterfly.  With no
 *                     body (SX < 28) this IS OC_ORDER_LANES with lanes = 512.
 * Everything else (interpolation polynomial, warp algebra, LU inverse, guards,
 * flags) is identical in both modes.  No COMPILER contraction anywhere
 * (-ffp-contract=off on both the oracle and the HIP side).
 *
 * Arithmetic contract (round 5): OR the flag OC_ARITH_FMA into `order` (OC_ORDER_LANES_FMA, OC_ORDER_SEQ_FMA) and every
 * PER-SAMPLE multiply-add of ICGN2D1 / ICGN2D2 / ICLM2D1 / ICLM2D2 / ICGN3D1 is ONE explicit fmaf -- the contraction a
 * compiler with FMA hardware enabled makes of the reference's own source expressions (the reference's build files fix no
 * contraction mode: examples/ide_configuration/CMakeLists.txt), association unchanged:
 *   - the 16-term bicubic polynomial (src/oc_cubic_bspline.cpp:159-177): v = fmaf(c * dy^k, dx^l, v), 15 fused of 15 adds;
 *   - the tricubic basis polynomials (:35-53, Horner steps fused) and the 21 four-tap sums (:390-401): b0*r0, then 3 fmaf;
 *   - the warps (src/oc_deformation.cpp:94-105, 268-282, 518-530): the first product stays, the following ones are fused,
 *     the translation column is added last as the reference writes it;
 *   - zero-mean norms (d*d into the running sum), the Hessian (sd[i]*sd[j]), the error e = fmaf(t, factor, -r), ZNSSD (e*e)
 *     and the numerator (sd[i]*e) accumulations (src/oc_icgn.cpp:198-205, 260-276, 1314-1337, 1403-1433).
 * NOT fused: the once-per-POI dense algebra (LU / cofactor inverses, warp products, dp = H^-1 b, convergence norm), the
 * steepest-descent products g * x themselves, prepare() (gradients, tables, prefilter), FFTCC, NR2D1, Strain.
 * IEEE-754 fusedMultiplyAdd is defined bit for bit, so the HIP kernels built with the same fused sites
 * (oc_hip_set_tuning "arith_fma") are bit-identical to OC_ORDER_LANES_FMA; against the reference's separately rounded
 * loop order (OC_ORDER_SEQ) the fused results differ by rounding only (tests/test_order_tolerance.py,
 * tests/test_gpu_fullsize.py assert north_star's bars).  OC_ORDER_ROWS has no fused form.
 */
#ifndef OC_ORACLE_H_
#define OC_ORACLE_H_

#ifdef __cplusplus
extern "C" {
#endif

#define OC_ORDER_SEQ 0
#define OC_ORDER_LANES 1
#define OC_ORDER_ROWS 2
#define OC_ARITH_FMA 0x100
#define OC_ORDER_SEQ_FMA (OC_ORDER_SEQ | OC_ARITH_FMA)
#define OC_ORDER_LANES_FMA (OC_ORDER_LANES | OC_ARITH_FMA)

#define OC_POI2D_FLOATS 25
#define OC_POI3D_FLOATS 31

/* src/oc_gradient.cpp:37-79 */
void oc_oracle_gradient2d(const float* img, int height, int width, float* gx, float* gy, int threads);
/* src/oc_cubic_bspline.cpp:84-132; lut is height*width*16 floats, entry [k][l] at 4*k+l */
void oc_oracle_bspline2d_lut(const float* img, int height, int width, float* lut, int threads);
/* src/oc_cubic_bspline.cpp:134-181 (single sample, for unit tests) */
float oc_oracle_bspline2d_eval(const float* lut, int height, int width, float x, float y);
/* the same under OC_ARITH_FMA */
float oc_oracle_bspline2d_eval_fma(const float* lut, int height, int width, float x, float y);

/* src/oc_fftcc.cpp:177-285.  If surface != NULL it receives, for POI 0 only,
 * the (2rx*2ry) float correlation surface (test hook for tie analysis). */
void oc_oracle_fftcc2d(const float* ref, const float* tar, int height, int width,
                       int rx, int ry, float* pois, long n, int threads, float* surface);

/* src/oc_icgn.cpp:144-351 */
void oc_oracle_icgn2d1(const float* ref, const float* gx, const float* gy, const float* tar_lut,
                       int height, int width, int rx, int ry, float conv, float stop,
                       float* pois, long n, int order, int lanes, int threads);
/* src/oc_icgn.cpp:685-908 */
void oc_oracle_icgn2d2(const float* ref, const float* gx, const float* gy, const float* tar_lut,
                       int height, int width, int rx, int ry, float conv, float stop,
                       float* pois, long n, int order, int lanes, int threads);
/* The same with the per-POI centre offsets of compute(poi_queue, center_offset_queue)
 * (src/oc_icgn.cpp:353-557, 910-1136; center_offsets = n x {x, y} or NULL) and/or
 * DIC::setSelfAdaptive (subset radius from poi->subset_radius, src/oc_icgn.cpp:152-158). */
void oc_oracle_icgn2d1_ex(const float* ref, const float* gx, const float* gy, const float* tar_lut, int height,
                          int width, int rx, int ry, float conv, float stop, float* pois, long n, int order, int lanes,
                          int threads, const float* center_offsets, int self_adaptive);
void oc_oracle_icgn2d2_ex(const float* ref, const float* gx, const float* gy, const float* tar_lut, int height,
                          int width, int rx, int ry, float conv, float stop, float* pois, long n, int order, int lanes,
                          int threads, const float* center_offsets, int self_adaptive);
/* NR2D1::compute(poi_queue), src/oc_nr.cpp:160-332 (forward-additive Newton-Raphson).
 * tar_lut, tar_lut_gx, tar_lut_gy: oc_oracle_bspline2d_lut of the target image and of
 * oc_oracle_gradient2d(target) -- what NR2D1::prepare builds (src/oc_nr.cpp:119-158). */
void oc_oracle_nr2d1(const float* ref, const float* tar_lut, const float* tar_lut_gx, const float* tar_lut_gy, int height,
                     int width, int rx, int ry, float conv, float stop, float* pois, long n, int order, int lanes,
                     int threads);

/* ICLM2D1::compute / ICLM2D2::compute(poi_queue), src/oc_iclm.cpp:150-368, 505-741 (inverse-compositional
 * Levenberg-Marquardt).  dof = 6 or 12; damping = {lambda, alpha, beta} of
 * ICLM2D*::setDamping (defaults 100, 0.1, 10: src/oc_iclm.h:33-38); POI stride in floats. */
void oc_oracle_iclm2d(int dof, const float* ref, const float* gx, const float* gy, const float* tar_lut, int height, int width,
                      int rx, int ry, float conv, float stop, const float* damping, int self_adaptive, float* pois, long n,
                      int stride_floats, int order, int lanes, int threads);
/* the fixed-arithmetic restatement of powf(lambda, q) used for the first damping value (src/oc_iclm.cpp:253) */
float oc_oracle_pow_lambda(float lambda, float q);
/* The small dense algebra the solvers take from Eigen, on its own: inverse of an n x n row-major matrix -- n = 3 and 4 by
 * cofactors (Eigen compute_inverse), n = 1, 2, 5 ... 12 by LU with partial pivoting (PartialPivLU::inverse; the 6 x 6 and
 * 12 x 12 Hessians of src/oc_icgn.cpp:210,759,1339 and the 6 x 6 warp of :831) -- and the coefficient-wise product.
 * Returns -1 for an unsupported n. */
int oc_oracle_inverse(const float* a, float* ainv, int n);
void oc_oracle_mat_mul(const float* a, const float* b, float* c, int n);

/* Strain::compute(poi_queue) for POI2D (src/oc_strain.cpp:149-247) and POI3D (:372-488), after Strain::prepare
 * (:96-147): subregion_radius, neighbor_number_min as in the constructor (:31-46); zncc_threshold default 0.9
 * (:36); approximation 1 = Cauchy, 2 = Green (:220-234).  Writes strain.exx.. in place (2D: floats 20..22,
 * 3D: floats 22..27) for every POI that was fitted; nothing else is touched. */
void oc_oracle_strain2d(float* pois, long n, int stride_floats, float subregion_radius, int neighbor_number_min,
                        float zncc_threshold, int approximation, int threads);
void oc_oracle_strain3d(float* pois, long n, int stride_floats, float subregion_radius, int neighbor_number_min,
                        float zncc_threshold, int approximation, int threads);

/* RegionFit2D::compute(poi_queue) (src/oc_region_fit.cpp:94-174) / RegionFit3D::compute (:251-342) after
 * setNeighbor(reliable) + prepare(): every POI of `pois` gets the plane fitted through the reliable POIs around it as
 * its deformation (2D: u ux uy v vx vy; 3D: all twelve) and result.zncc = 0; POIs without enough neighbours are left
 * untouched.  Same neighbour rule, KNN fallback, row order and double-precision solve as oc_oracle_strain*. */
void oc_oracle_region_fit2d(const float* reliable, long n_reliable, int reliable_stride, float* pois, long n, int stride_floats,
                            float neighbor_search_radius, int neighbor_number_min, int threads);
void oc_oracle_region_fit3d(const float* reliable, long n_reliable, int reliable_stride, float* pois, long n, int stride_floats,
                            float neighbor_search_radius, int neighbor_number_min, int threads);

/* src/oc_gradient.cpp:143-231 */
void oc_oracle_gradient3d(const float* vol, int dz, int dy, int dx, float* gx, float* gy, float* gz, int threads);
/* src/oc_cubic_bspline.cpp:214-351 */
void oc_oracle_bspline3d_prefilter(const float* vol, int dz, int dy, int dx, float* coef, int threads);
/* src/oc_cubic_bspline.cpp:353-405 */
float oc_oracle_bspline3d_eval(const float* coef, int dz, int dy, int dx, float x, float y, float z);
float oc_oracle_bspline3d_eval_fma(const float* coef, int dz, int dy, int dx, float x, float y, float z);
/* 1 when this build executes fmaf as a hardware instruction (built with -mfma), 0 when it goes through libm's fmaf
 * (same bits either way; only the speed differs) */
int oc_oracle_fma_is_hardware(void);
/* src/oc_fftcc.cpp:327-436 */
void oc_oracle_fftcc3d(const float* ref, const float* tar, int dz, int dy, int dx,
                       int rx, int ry, int rz, float* pois, long n, int threads);
/* exact_sums = 1: means, norms and the ZNCC quotient in double (the peak's ZNCC without the rounding noise of the
 * reference's sequential float32 running sums over 10^4 ... 10^5 voxels, src/oc_fftcc.cpp:340-376); 0 = oc_oracle_fftcc3d */
void oc_oracle_fftcc3d_ex(const float* ref, const float* tar, int dz, int dy, int dx,
                          int rx, int ry, int rz, float* pois, long n, int threads, int exact_sums);
/* src/oc_icgn.cpp:1270-1500 */
void oc_oracle_icgn3d1(const float* ref, const float* gx, const float* gy, const float* gz,
                       const float* tar_coef, int dz, int dy, int dx, int rx, int ry, int rz,
                       float conv, float stop, float* pois, long n, int order, int lanes, int threads);

int oc_oracle_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
