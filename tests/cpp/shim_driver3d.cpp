// shim_driver3d.cpp -- the DVC classes of include/opencorr_compat driven the way the reference's
// examples/test_dvc_fftcc_icgn1.cpp (:45-47, 87-106) and examples/test_dvc_gpu_icgn.cpp (:89-94) drive theirs.
//
//   shim_driver3d <in.bin> <out.bin>
// in.bin : int32 dim_x, dim_y, dim_z, rx, ry, rz, n; float32 conv, stop; ref[dz*dy*dx], tar[...] (z, y, x); x[n], y[n], z[n]
// out.bin: n POI3D records (124 bytes each) after FFTCC3D::compute + ICGN3D1::prepare/compute
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <vector>

#include "opencorr_compat/opencorr.h"

using namespace opencorr;

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 3;
    int hdr[7];
    float it[2];
    if (std::fread(hdr, 4, 7, f) != 7 || std::fread(it, 4, 2, f) != 2) return 4;
    const int dx = hdr[0], dy = hdr[1], dz = hdr[2], rx = hdr[3], ry = hdr[4], rz = hdr[5], n = hdr[6];
    const size_t vox = (size_t)dx * dy * dz;
    std::vector<float> xs(n), ys(n), zs(n);
    try {
        Image3D ref_img(dx, dy, dz), tar_img(dx, dy, dz);
        if (std::fread(&ref_img.vol_mat[0][0][0], 4, vox, f) != vox || std::fread(&tar_img.vol_mat[0][0][0], 4, vox, f) != vox ||
            std::fread(xs.data(), 4, n, f) != (size_t)n || std::fread(ys.data(), 4, n, f) != (size_t)n ||
            std::fread(zs.data(), 4, n, f) != (size_t)n)
            return 5;
        std::fclose(f);
        std::vector<POI3D> poi_queue;
        for (int i = 0; i < n; i++) poi_queue.push_back(POI3D(Point3D(xs[i], ys[i], zs[i])));
        const int cpu_thread_number = 4;

        FFTCC3D* fftcc = new FFTCC3D(rx, ry, rz, cpu_thread_number);
        fftcc->setImages(ref_img, tar_img);
        fftcc->compute(poi_queue);
        const std::vector<POI3D> after_fftcc = poi_queue;

        ICGN3D1* icgn1 = new ICGN3D1(rx, ry, rz, it[0], it[1], cpu_thread_number);
        icgn1->setImages(ref_img, tar_img);
        icgn1->prepare();
        icgn1->compute(poi_queue);

        // single-POI entry points: the first POI recomputed from scratch must agree with the queue
        if (n > 0) {
            POI3D one(Point3D(xs[0], ys[0], zs[0]));
            fftcc->compute(&one);
            if (one.deformation.u != after_fftcc[0].deformation.u || one.result.zncc != after_fftcc[0].result.zncc) {
                std::cerr << "FFTCC3D::compute(POI3D*) disagrees with compute(vector&)" << std::endl;
                return 7;
            }
            icgn1->compute(&one);
            if (std::memcmp(&one, &poi_queue[0], sizeof(POI3D)) != 0) {
                std::cerr << "ICGN3D1::compute(POI3D*) disagrees with compute(vector&)" << std::endl;
                return 7;
            }
        }
        // prepareRef() / prepareTar() separately, then setIteration(POI3D*) with the engine's own values: same bits
        {
            ICGN3D1 again(rx, ry, rz, 0.5f, 1.f, cpu_thread_number);
            again.setImages(ref_img, tar_img);
            again.prepareRef();
            again.prepareTar();
            POI3D cfg(0.f, 0.f, 0.f);
            cfg.result.convergence = it[0];
            cfg.result.iteration = it[1];
            again.setIteration(&cfg);
            std::vector<POI3D> q = after_fftcc;
            again.compute(q);
            if (std::memcmp(q.data(), poi_queue.data(), q.size() * sizeof(POI3D)) != 0) {
                std::cerr << "prepareRef + prepareTar + setIteration(POI3D*) differs from prepare()" << std::endl;
                return 9;
            }
        }
        // the reference's CUDA-module shape (gpu_lib/opencorr_gpu.h:81-101): ICGN3D1GPU fed with Img3D = &vol_mat[0][0][0]
        {
            Img3D ref3{dx, dy, dz, &ref_img.vol_mat[0][0][0]}, tar3{dx, dy, dz, &tar_img.vol_mat[0][0][0]};
            ICGN3D1GPU gpu(rx, ry, rz, it[0], (int)it[1]);
            gpu.setImages(ref3, tar3);
            gpu.prepare();
            std::vector<POI3D> q = after_fftcc;
            gpu.compute(q);
            if (std::memcmp(q.data(), poi_queue.data(), q.size() * sizeof(POI3D)) != 0) {
                std::cerr << "ICGN3D1GPU (Img3D) differs from ICGN3D1 (Image3D)" << std::endl;
                return 13;
            }
        }
        // setSubset on a live FFTCC3D engine re-plans (src/oc_fftcc.cpp:21-139 FFTW::update): a smaller window must still
        // find the same integer displacement on a well-textured POI
        if (n > 0 && rx >= 8) {
            fftcc->setSubset(rx - 2, ry - 2, rz - 2);
            POI3D one(Point3D(xs[n / 2], ys[n / 2], zs[n / 2]));
            fftcc->compute(&one);
            const POI3D& ref_poi = after_fftcc[n / 2];
            if (ref_poi.result.zncc > 0.5f && (one.deformation.u != ref_poi.deformation.u || one.deformation.v != ref_poi.deformation.v ||
                                               one.deformation.w != ref_poi.deformation.w)) {
                std::cerr << "FFTCC3D after setSubset finds another peak" << std::endl;
                return 10;
            }
            fftcc->setSubset(rx, ry, rz);
        }
        // Strain (3D) and RegionFit3D over the DVC result: finite, small strains; the fitted plane reproduces u
        {
            std::vector<POI3D> q = poi_queue;
            Strain strain(40.f, 5, cpu_thread_number);
            strain.setZnccThreshold(0.5f);
            strain.prepare(q);
            strain.compute(q);
            for (size_t i = 0; i < q.size(); i++)
                if (!(std::fabs(q[i].strain.exx) < 0.1f) || !(std::fabs(q[i].strain.ezz) < 0.1f)) {
                    std::cerr << "implausible 3D strain at POI " << i << std::endl;
                    return 15;
                }
            RegionFit3D fit(40.f, 5, cpu_thread_number);
            std::vector<POI3D> reliable = poi_queue;
            fit.setNeighbor(reliable);
            fit.prepare();
            std::vector<POI3D> probe(poi_queue.begin(), poi_queue.begin() + (n < 4 ? n : 4));
            for (POI3D& p : probe) p.deformation.u = p.deformation.v = p.deformation.w = 0.f;
            fit.compute(probe);
            for (size_t i = 0; i < probe.size(); i++)
                if (poi_queue[i].result.zncc > 0.9f && std::fabs(probe[i].deformation.u - poi_queue[i].deformation.u) > 0.1f) {
                    std::cerr << "RegionFit3D does not reproduce the field at POI " << i << std::endl;
                    return 16;
                }
        }
        // error path: compute before prepare must throw std::string
        bool threw = false;
        try {
            ICGN3D1 bad(rx, ry, rz, it[0], it[1], 1);
            bad.setImages(ref_img, tar_img);
            bad.compute(poi_queue);
        } catch (const std::string& msg) {
            threw = true;
        }
        if (!threw) { std::cerr << "missing prepare() was not reported" << std::endl; return 8; }
        delete fftcc;
        delete icgn1;
        FILE* o = std::fopen(argv[2], "wb");
        if (!o) return 6;
        std::fwrite(poi_queue.data(), sizeof(POI3D), poi_queue.size(), o);
        std::fclose(o);
    } catch (const std::string& msg) {
        std::cerr << "OpenCorr shim error: " << msg << std::endl;
        return 1;
    }
    return 0;
}
