// ref_driver.cpp -- C entry points around the REFERENCE's own classes (oracle/_ref/liboc_ref.so).
//
// TEST INFRASTRUCTURE ONLY.  oracle/Makefile (target `ref`) compiles the reference's sources UNMODIFIED, where they lie
// under /root/reference/src -- oc_fftcc.cpp, oc_icgn.cpp, oc_iclm.cpp, oc_nr.cpp, oc_cubic_bspline.cpp,
// oc_gradient.cpp, oc_subset.cpp, oc_deformation.cpp, oc_dic.cpp, oc_image.cpp, and oc_strain.cpp,
// oc_region_fit.cpp, oc_nearest_neighbor.cpp, and (round 5, the EpipolarSearch consumer) oc_calibration.cpp,
// oc_epipolar_search.cpp, oc_stereovision.cpp -- against the stand-in headers of oracle/ref_stubs (mini Eigen, FFTW,
// OpenCV and nanoflann), plus this file.  Nothing of the reference is copied into the
// repository; the library exists only where /root/reference does (this container) and is what pins the oracle's
// reading of the reference's loops: tests/test_oracle_vs_ref.py asserts oracle(OC_ORDER_SEQ) == liboc_ref.
//
// Data conventions are the oracle's (oc_oracle.h): row-major float images, POI2D / POI3D AoS records.
#include <omp.h>

#include <cstring>
#include <string>
#include <vector>

#include "oc_epipolar_search.h"
#include "oc_fftcc.h"
#include "oc_icgn.h"
#include "oc_iclm.h"
#include "oc_nr.h"
#include "oc_region_fit.h"
#include "oc_strain.h"

using namespace opencorr;

static_assert(sizeof(POI2D) == 100, "POI2D is 25 packed floats");
static_assert(sizeof(POI3D) == 124, "POI3D is 31 packed floats");

namespace {

void fill2d(Image2D& img, const float* src) {
    for (int r = 0; r < img.height; r++)
        for (int c = 0; c < img.width; c++) img.eg_mat(r, c) = src[(size_t)r * img.width + c];
}

void fill3d(Image3D& img, const float* src) { std::memcpy(&img.vol_mat[0][0][0], src, sizeof(float) * (size_t)img.dim_x * img.dim_y * img.dim_z); }

std::vector<POI2D> load2d(const float* pois, long n) {
    std::vector<POI2D> q((size_t)n, POI2D(0.f, 0.f));
    if (n) std::memcpy(static_cast<void*>(q.data()), pois, sizeof(POI2D) * (size_t)n);
    return q;
}
std::vector<POI3D> load3d(const float* pois, long n) {
    std::vector<POI3D> q((size_t)n, POI3D(0.f, 0.f, 0.f));
    if (n) std::memcpy(static_cast<void*>(q.data()), pois, sizeof(POI3D) * (size_t)n);
    return q;
}

int threads_or_all(int threads) { return threads > 0 ? threads : omp_get_max_threads(); }

// engine: 0 = ICGN2D1, 1 = ICGN2D2, 2 = ICLM2D1, 3 = ICLM2D2, 4 = NR2D1
template <class Engine>
void run2d(Engine& e, Image2D& ref, Image2D& tar, std::vector<POI2D>& q, const float* offsets, int self_adaptive) {
    e.setImages(ref, tar);
    e.prepare();
    e.setSelfAdaptive(self_adaptive != 0);
    if (offsets) {
        std::vector<Point2D> off(q.size());
        for (size_t i = 0; i < q.size(); i++) off[i] = Point2D(offsets[2 * i], offsets[2 * i + 1]);
        if constexpr (std::is_same<Engine, ICGN2D1>::value || std::is_same<Engine, ICGN2D2>::value) e.compute(q, off);
    } else {
        e.compute(q);
    }
}

}  // namespace

extern "C" {

int oc_ref_available(void) { return 1; }

// FFTCC2D::compute(std::vector<POI2D>&), src/oc_fftcc.cpp:277-285
int oc_ref_fftcc2d(const float* ref, const float* tar, int height, int width, int rx, int ry, float* pois, long n, int threads) {
    try {
        Image2D ref_img(width, height), tar_img(width, height);
        fill2d(ref_img, ref);
        fill2d(tar_img, tar);
        std::vector<POI2D> q = load2d(pois, n);
        FFTCC2D fftcc(rx, ry, threads_or_all(threads));
        fftcc.setImages(ref_img, tar_img);
        fftcc.compute(q);
        if (n) std::memcpy(pois, static_cast<void*>(q.data()), sizeof(POI2D) * (size_t)n);
    } catch (const std::string&) {
        return 1;
    }
    return 0;
}

// ICGN2D1 / ICGN2D2 (src/oc_icgn.cpp), ICLM2D1 / ICLM2D2 (src/oc_iclm.cpp), NR2D1 (src/oc_nr.cpp): prepare() + compute(queue)
// offsets (n x 2, or null): compute(poi_queue, center_offset_queue); self_adaptive: DIC::setSelfAdaptive
// damping (3 floats or null): ICLM*::setDamping(lambda, alpha, beta)
int oc_ref_solve2d(int engine, const float* ref, const float* tar, int height, int width, int rx, int ry, float conv, float stop,
                   float* pois, long n, const float* offsets, int self_adaptive, const float* damping, int threads) {
    try {
        Image2D ref_img(width, height), tar_img(width, height);
        fill2d(ref_img, ref);
        fill2d(tar_img, tar);
        std::vector<POI2D> q = load2d(pois, n);
        const int t = threads_or_all(threads);
        switch (engine) {
            case 0: { ICGN2D1 e(rx, ry, conv, stop, t); run2d(e, ref_img, tar_img, q, offsets, self_adaptive); break; }
            case 1: { ICGN2D2 e(rx, ry, conv, stop, t); run2d(e, ref_img, tar_img, q, offsets, self_adaptive); break; }
            case 2: { ICLM2D1 e(rx, ry, conv, stop, t); if (damping) e.setDamping(damping[0], damping[1], damping[2]); run2d(e, ref_img, tar_img, q, nullptr, self_adaptive); break; }
            case 3: { ICLM2D2 e(rx, ry, conv, stop, t); if (damping) e.setDamping(damping[0], damping[1], damping[2]); run2d(e, ref_img, tar_img, q, nullptr, self_adaptive); break; }
            case 4: { NR2D1 e(rx, ry, conv, stop, t); run2d(e, ref_img, tar_img, q, nullptr, 0); break; }
            default: return 2;
        }
        if (n) std::memcpy(pois, static_cast<void*>(q.data()), sizeof(POI2D) * (size_t)n);
    } catch (const std::string&) {
        return 1;
    }
    return 0;
}

// Timing leg of bench.py's cpu_baseline ("reference_sources"): the reference's OWN ICGN2D1 -- its float**** table, its
// per-thread instance pool, its omp loop (src/oc_icgn.cpp:61-69, 343-351) -- on a queue that already holds initial guesses,
// the way examples/test_2d_dic_fftcc_icgn1.cpp:93-104 runs it.  prepare() and compute(queue) are timed separately, compute
// `reps` times on a fresh copy of the queue (best time returned); the last run's records are written back.
int oc_ref_time_icgn2d1(const float* ref, const float* tar, int height, int width, int rx, int ry, float conv, float stop,
                        float* pois, long n, int threads, int reps, double* prepare_seconds, double* compute_seconds) {
    try {
        Image2D ref_img(width, height), tar_img(width, height);
        fill2d(ref_img, ref);
        fill2d(tar_img, tar);
        ICGN2D1 e(rx, ry, conv, stop, threads_or_all(threads));
        e.setImages(ref_img, tar_img);
        double t0 = omp_get_wtime();
        e.prepare();
        *prepare_seconds = omp_get_wtime() - t0;
        double best = 1e30;
        std::vector<POI2D> q;
        for (int r = 0; r < (reps > 0 ? reps : 1); r++) {
            q = load2d(pois, n);
            t0 = omp_get_wtime();
            e.compute(q);
            const double dt = omp_get_wtime() - t0;
            best = dt < best ? dt : best;
        }
        *compute_seconds = best;
        if (n) std::memcpy(pois, static_cast<void*>(q.data()), sizeof(POI2D) * (size_t)n);
    } catch (const std::string&) {
        return 1;
    }
    return 0;
}

// EpipolarSearch (src/oc_epipolar_search.cpp): two Calibration objects from their 13 intrinsics / 6 extrinsics
// (src/oc_calibration.h:25-49), setSearch, createICGN, setParallax(coefficient_x, coefficient_y), setImages, prepare()
// (-> updateMatrices of both cameras, updateFundementalMatrix, ICGN2D1::prepare) and compute(poi_queue) -- the loop of
// :133-195 per POI: candidates along the epipolar line, icgn1->compute(&candidate) each, std::sort by ZNCC, candidates[0].
// fundamental9_out (row-major) receives the matrix the search used.
int oc_ref_epipolar_search(const float* ref, const float* tar, int height, int width, const float* cam1_intrinsics13,
                           const float* cam1_extrinsics6, const float* cam2_intrinsics13, const float* cam2_extrinsics6,
                           int search_radius, int search_step, const float* parallax_x3, const float* parallax_y3, int rx, int ry,
                           float conv, float stop, float* pois, long n, int threads, float* fundamental9_out) {
    try {
        Image2D ref_img(width, height), tar_img(width, height);
        fill2d(ref_img, ref);
        fill2d(tar_img, tar);
        CameraIntrinsics i1, i2;
        CameraExtrinsics e1, e2;
        std::memcpy(i1.cam_i, cam1_intrinsics13, sizeof(i1.cam_i));
        std::memcpy(i2.cam_i, cam2_intrinsics13, sizeof(i2.cam_i));
        std::memcpy(e1.cam_e, cam1_extrinsics6, sizeof(e1.cam_e));
        std::memcpy(e2.cam_e, cam2_extrinsics6, sizeof(e2.cam_e));
        Calibration cam1(i1, e1), cam2(i2, e2);
        EpipolarSearch search(cam1, cam2, threads_or_all(threads));
        search.setSearch(search_radius, search_step);
        search.createICGN(rx, ry, conv, stop);
        float px[3] = {parallax_x3[0], parallax_x3[1], parallax_x3[2]}, py[3] = {parallax_y3[0], parallax_y3[1], parallax_y3[2]};
        search.setParallax(px, py);
        search.setImages(ref_img, tar_img);
        search.prepare();
        // the matrix prepare() has just built (a protected member): the same four lines on the same two cameras (:99-118)
        {
            Calibration c1 = cam1, c2 = cam2;
            c1.updateMatrices();
            c2.updateMatrices();
            Eigen::Matrix3f right_invK_t = c2.intrinsic_matrix.inverse().transpose();
            Eigen::Matrix3f right_t_antisymmetric;
            right_t_antisymmetric << 0, -c2.translation_vector(2), c2.translation_vector(1), c2.translation_vector(2), 0,
                -c2.translation_vector(0), -c2.translation_vector(1), c2.translation_vector(0), 0;
            Eigen::Matrix3f right_E = right_t_antisymmetric * c2.rotation_matrix;
            Eigen::Matrix3f left_K = c1.intrinsic_matrix.inverse();
            Eigen::Matrix3f F = right_invK_t * right_E * left_K;
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++) fundamental9_out[r * 3 + c] = F(r, c);
        }
        std::vector<POI2D> q = load2d(pois, n);
        search.compute(q);
        if (n) std::memcpy(pois, static_cast<void*>(q.data()), sizeof(POI2D) * (size_t)n);
    } catch (const std::string&) {
        return 1;
    }
    return 0;
}

// what ICGN2D1::prepare() builds, for field-level comparison: Gradient2D4 (src/oc_gradient.cpp:37-79) of `ref` and the
// BicubicBspline table (src/oc_cubic_bspline.cpp:84-132) of `tar`; lut is [y][x][k][l]
int oc_ref_prepare2d(const float* ref, const float* tar, int height, int width, float* gx, float* gy, float* lut) {
    try {
        Image2D ref_img(width, height), tar_img(width, height);
        fill2d(ref_img, ref);
        fill2d(tar_img, tar);
        Gradient2D4 grad(ref_img);
        grad.getGradientX();
        grad.getGradientY();
        BicubicBspline interp(tar_img);
        interp.prepare();
        for (int r = 0; r < height; r++)
            for (int c = 0; c < width; c++) {
                gx[(size_t)r * width + c] = grad.gradient_x(r, c);
                gy[(size_t)r * width + c] = grad.gradient_y(r, c);
            }
        // the table is a private member; its values are observable through compute() only, so the comparison of the
        // table itself is done through interpolated values (oc_ref_bspline2d_eval below) -- lut may be null
        (void)lut;
    } catch (const std::string&) {
        return 1;
    }
    return 0;
}

// BicubicBspline::compute at n points (src/oc_cubic_bspline.cpp:134-181)
int oc_ref_bspline2d_eval(const float* img, int height, int width, const float* xy, long n, float* out) {
    try {
        Image2D im(width, height);
        fill2d(im, img);
        BicubicBspline interp(im);
        interp.prepare();
        for (long i = 0; i < n; i++) {
            Point2D p(xy[2 * i], xy[2 * i + 1]);
            out[i] = interp.compute(p);
        }
    } catch (const std::string&) {
        return 1;
    }
    return 0;
}

// FFTCC3D::compute(std::vector<POI3D>&), src/oc_fftcc.cpp:429-436
int oc_ref_fftcc3d(const float* ref, const float* tar, int dz, int dy, int dx, int rx, int ry, int rz, float* pois, long n, int threads) {
    try {
        Image3D ref_img(dx, dy, dz), tar_img(dx, dy, dz);
        fill3d(ref_img, ref);
        fill3d(tar_img, tar);
        std::vector<POI3D> q = load3d(pois, n);
        {
            FFTCC3D fftcc(rx, ry, rz, threads_or_all(threads));
            fftcc.setImages(ref_img, tar_img);
            fftcc.compute(q);
        }
        ref_img.release();
        tar_img.release();
        if (n) std::memcpy(pois, static_cast<void*>(q.data()), sizeof(POI3D) * (size_t)n);
    } catch (const std::string&) {
        return 1;
    }
    return 0;
}

// ICGN3D1::prepare() + compute(std::vector<POI3D>&), src/oc_icgn.cpp:1240-1500
int oc_ref_icgn3d1(const float* ref, const float* tar, int dz, int dy, int dx, int rx, int ry, int rz, float conv, float stop,
                   float* pois, long n, int threads) {
    try {
        Image3D ref_img(dx, dy, dz), tar_img(dx, dy, dz);
        fill3d(ref_img, ref);
        fill3d(tar_img, tar);
        std::vector<POI3D> q = load3d(pois, n);
        {
            ICGN3D1 e(rx, ry, rz, conv, stop, threads_or_all(threads));
            e.setImages(ref_img, tar_img);
            e.prepare();
            e.compute(q);
        }
        ref_img.release();
        tar_img.release();
        if (n) std::memcpy(pois, static_cast<void*>(q.data()), sizeof(POI3D) * (size_t)n);
    } catch (const std::string&) {
        return 1;
    }
    return 0;
}

// Gradient3D4 (src/oc_gradient.cpp:143-231) and TricubicBspline::prepare + compute (src/oc_cubic_bspline.cpp:214-405)
int oc_ref_prepare3d(const float* ref, const float* tar, int dz, int dy, int dx, float* gx, float* gy, float* gz, const float* xyz,
                     long n, float* interp_out) {
    try {
        Image3D ref_img(dx, dy, dz), tar_img(dx, dy, dz);
        fill3d(ref_img, ref);
        fill3d(tar_img, tar);
        {
            Gradient3D4 grad(ref_img);
            grad.getGradientX();
            grad.getGradientY();
            grad.getGradientZ();
            const size_t vox = (size_t)dx * dy * dz;
            std::memcpy(gx, &grad.gradient_x[0][0][0], sizeof(float) * vox);
            std::memcpy(gy, &grad.gradient_y[0][0][0], sizeof(float) * vox);
            std::memcpy(gz, &grad.gradient_z[0][0][0], sizeof(float) * vox);
            TricubicBspline interp(tar_img);
            interp.prepare();
            for (long i = 0; i < n; i++) {
                Point3D p(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
                interp_out[i] = interp.compute(p);
            }
        }
        ref_img.release();
        tar_img.release();
    } catch (const std::string&) {
        return 1;
    }
    return 0;
}

// Strain(subregion_radius, neighbor_number_min, threads); setZnccThreshold / setApproximation; prepare(poi_queue);
// compute(poi_queue)  -- src/oc_strain.cpp:31-46, 72-107, 136-147, 236-247 (2D), 476-488 (3D)
int oc_ref_strain(int ndim, float* pois, long n, float subregion_radius, int neighbor_number_min, float zncc_threshold,
                  int approximation, int threads) {
    try {
        Strain strain(subregion_radius, neighbor_number_min, threads_or_all(threads));
        strain.setZnccThreshold(zncc_threshold);
        strain.setApproximation(approximation);
        if (ndim == 2) {
            std::vector<POI2D> q = load2d(pois, n);
            strain.prepare(q);
            strain.compute(q);
            if (n) std::memcpy(pois, static_cast<void*>(q.data()), sizeof(POI2D) * (size_t)n);
        } else {
            std::vector<POI3D> q = load3d(pois, n);
            strain.prepare(q);
            strain.compute(q);
            if (n) std::memcpy(pois, static_cast<void*>(q.data()), sizeof(POI3D) * (size_t)n);
        }
    } catch (const std::string&) {
        return 1;
    }
    return 0;
}

// RegionFit2D / RegionFit3D(neighbor_search_radius, neighbor_number_min, threads); setNeighbor(reliable); prepare();
// compute(poi_queue)  -- src/oc_region_fit.cpp:32-46, 75-92, 166-174 (2D), 189-203, 232-249, 334-342 (3D)
int oc_ref_region_fit(int ndim, const float* reliable, long n_reliable, float* pois, long n, float neighbor_search_radius,
                      int neighbor_number_min, int threads) {
    try {
        if (ndim == 2) {
            std::vector<POI2D> rel = load2d(reliable, n_reliable), q = load2d(pois, n);
            RegionFit2D fit(neighbor_search_radius, neighbor_number_min, threads_or_all(threads));
            fit.setNeighbor(rel);
            fit.prepare();
            fit.compute(q);
            if (n) std::memcpy(pois, static_cast<void*>(q.data()), sizeof(POI2D) * (size_t)n);
        } else {
            std::vector<POI3D> rel = load3d(reliable, n_reliable), q = load3d(pois, n);
            RegionFit3D fit(neighbor_search_radius, neighbor_number_min, threads_or_all(threads));
            fit.setNeighbor(rel);
            fit.prepare();
            fit.compute(q);
            if (n) std::memcpy(pois, static_cast<void*>(q.data()), sizeof(POI3D) * (size_t)n);
        }
    } catch (const std::string&) {
        return 1;
    }
    return 0;
}

// The reference's host-side Deformation classes (src/oc_deformation.{h,cpp}) for tests/test_oracle_vs_ref_deformation.py:
// kind 1 = Deformation2D1 (6 parameters, 3 x 3), 2 = Deformation2D2 (12, 6 x 6), 3 = Deformation3D1 (12, 4 x 4).
// mat_out <- warp_matrix after setDeformation(p); warped <- warp(pt); p_back <- the parameters setDeformation() reads back
// from a warp_matrix overwritten with mat_in (row-major).
int oc_ref_deformation(int kind, const float* p, const float* pt, const float* mat_in, float* mat_out, float* warped, float* p_back) {
    try {
        if (kind == 1) {
            float q[6];
            std::memcpy(q, p, sizeof(q));
            Deformation2D1 d(q);
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) mat_out[i * 3 + j] = d.warp_matrix(i, j);
            Point2D a(pt[0], pt[1]);
            Point2D b = d.warp(a);
            warped[0] = b.x; warped[1] = b.y;
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) d.warp_matrix(i, j) = mat_in[i * 3 + j];
            d.setDeformation();
            const float back[6] = {d.u, d.ux, d.uy, d.v, d.vx, d.vy};
            std::memcpy(p_back, back, sizeof(back));
        } else if (kind == 2) {
            float q[12];
            std::memcpy(q, p, sizeof(q));
            Deformation2D2 d(q);
            for (int i = 0; i < 6; i++)
                for (int j = 0; j < 6; j++) mat_out[i * 6 + j] = d.warp_matrix(i, j);
            Point2D b = d.warp(Point2D(pt[0], pt[1]));
            warped[0] = b.x; warped[1] = b.y;
            for (int i = 0; i < 6; i++)
                for (int j = 0; j < 6; j++) d.warp_matrix(i, j) = mat_in[i * 6 + j];
            d.setDeformation();
            const float back[12] = {d.u, d.ux, d.uy, d.uxx, d.uxy, d.uyy, d.v, d.vx, d.vy, d.vxx, d.vxy, d.vyy};
            std::memcpy(p_back, back, sizeof(back));
        } else if (kind == 3) {
            float q[12];
            std::memcpy(q, p, sizeof(q));
            Deformation3D1 d(q);
            for (int i = 0; i < 4; i++)
                for (int j = 0; j < 4; j++) mat_out[i * 4 + j] = d.warp_matrix(i, j);
            Point3D a(pt[0], pt[1], pt[2]);
            Point3D b = d.warp(a);
            warped[0] = b.x; warped[1] = b.y; warped[2] = b.z;
            for (int i = 0; i < 4; i++)
                for (int j = 0; j < 4; j++) d.warp_matrix(i, j) = mat_in[i * 4 + j];
            d.setDeformation();
            const float back[12] = {d.u, d.ux, d.uy, d.uz, d.v, d.vx, d.vy, d.vz, d.w, d.wx, d.wy, d.wz};
            std::memcpy(p_back, back, sizeof(back));
        } else {
            return 2;
        }
    } catch (const std::string&) {
        return 1;
    }
    return 0;
}

}  // extern "C"
