// shim_driver.cpp -- exercises the OpenCorr-shaped C++ classes (include/opencorr_compat) the way
// examples/test_2d_dic_fftcc_icgn1.cpp of the reference does, on data handed over by pytest.
//
//   shim_driver <in.bin> <out.bin>
// in.bin : int32 height, width, rx, ry, n; float32 conv, stop; ref[h*w], tar[h*w] (row-major); x[n], y[n]
// out.bin: n POI2D records (100 bytes each) after FFTCC2D::compute + ICGN2D1::prepare/compute
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
    int hdr[5];
    float it[2];
    if (std::fread(hdr, 4, 5, f) != 5 || std::fread(it, 4, 2, f) != 2) return 4;
    const int h = hdr[0], w = hdr[1], rx = hdr[2], ry = hdr[3], n = hdr[4];
    std::vector<float> ref((size_t)h * w), tar((size_t)h * w), xs(n), ys(n);
    if (std::fread(ref.data(), 4, ref.size(), f) != ref.size() || std::fread(tar.data(), 4, tar.size(), f) != tar.size() ||
        std::fread(xs.data(), 4, n, f) != (size_t)n || std::fread(ys.data(), 4, n, f) != (size_t)n)
        return 5;
    std::fclose(f);
    try {
        Image2D ref_img(w, h), tar_img(w, h);
        ref_img.fromRowMajor(ref.data());
        tar_img.fromRowMajor(tar.data());
        std::vector<POI2D> poi_queue;
        for (int i = 0; i < n; i++) poi_queue.push_back(POI2D(Point2D(xs[i], ys[i])));
        const int cpu_thread_number = 4;

        FFTCC2D* fftcc = new FFTCC2D(rx, ry, cpu_thread_number);
        fftcc->setImages(ref_img, tar_img);
        fftcc->compute(poi_queue);

        ICGN2D1* icgn1 = new ICGN2D1(rx, ry, it[0], it[1], cpu_thread_number);
        icgn1->setImages(ref_img, tar_img);
        icgn1->prepare();
        const std::vector<POI2D> after_fftcc = poi_queue;
        icgn1->compute(poi_queue);
        // compute(poi_queue, center_offset_queue) with zero offsets and setSelfAdaptive(true) with every
        // POI carrying the engine's own radius must both reproduce the plain run bit for bit
        {
            std::vector<POI2D> q = after_fftcc;
            std::vector<Point2D> offsets(q.size(), Point2D(0.f, 0.f));
            icgn1->compute(q, offsets);
            if (std::memcmp(q.data(), poi_queue.data(), q.size() * sizeof(POI2D)) != 0) {
                std::cerr << "zero center offsets change the result" << std::endl;
                return 9;
            }
            q = after_fftcc;
            for (POI2D& p : q) p.subset_radius = Point2D((float)rx, (float)ry);
            icgn1->setSelfAdaptive(true);
            icgn1->compute(q);
            icgn1->setSelfAdaptive(false);
            for (size_t i = 0; i < q.size(); i++) {
                // guarded POIs keep the radius they came with; everything else must agree
                if (q[i].result.zncc != poi_queue[i].result.zncc || q[i].deformation.u != poi_queue[i].deformation.u ||
                    q[i].deformation.vy != poi_queue[i].deformation.vy) {
                    std::cerr << "self-adaptive run with the engine's radius differs at POI " << i << std::endl;
                    return 10;
                }
            }
            if (n > 0) {
                POI2D one = after_fftcc[0];
                Point2D off(0.f, 0.f);
                icgn1->compute(&one, off);
                if (one.deformation.u != poi_queue[0].deformation.u) {
                    std::cerr << "compute(POI2D*, Point2D&) disagrees" << std::endl;
                    return 11;
                }
            }
        }
        // FFTCC + ICGN as ONE round trip: computeChain({fftcc, icgn}, queue) and icgn->compute(queue, *fftcc) must give the
        // bits of the two separate calls.  The engines were handed the same Image2D pair: the second one borrowed the
        // first one's device snapshot instead of uploading again (nothing to observe here but the equal results).
        {
            std::vector<POI2D> q;
            for (int i = 0; i < n; i++) q.push_back(POI2D(Point2D(xs[i], ys[i])));
            std::vector<POI2D> q2 = q;
            computeChain({fftcc, icgn1}, q);
            icgn1->compute(q2, *fftcc);
            if (std::memcmp(q.data(), poi_queue.data(), q.size() * sizeof(POI2D)) != 0 ||
                std::memcmp(q2.data(), poi_queue.data(), q2.size() * sizeof(POI2D)) != 0) {
                std::cerr << "computeChain differs from the two separate compute() calls" << std::endl;
                return 17;
            }
            // a fresh target image in the SAME Image2D object (a main that loops over frames reloads tar_img): both
            // engines are told with setImages(), the first to need the pair uploads it, the second borrows THAT snapshot
            // -- never the older one
            Image2D tar_shift(w, h);
            std::vector<float> moved(tar.size());
            for (int r = 0; r < h; r++)
                for (int c = 0; c < w; c++) moved[(size_t)r * w + c] = tar[(size_t)r * w + (c + 1 < w ? c + 1 : c)];
            tar_img.fromRowMajor(moved.data());
            fftcc->setImages(ref_img, tar_img);
            icgn1->setImages(ref_img, tar_img);
            std::vector<POI2D> q3;
            for (int i = 0; i < n; i++) q3.push_back(POI2D(Point2D(xs[i], ys[i])));
            fftcc->compute(q3);
            icgn1->prepare();
            icgn1->compute(q3);
            int moved_by_one = 0;
            for (int i = 0; i < n; i++)
                if (q3[i].result.zncc > 0.9f && std::fabs((q3[i].deformation.u + 1.f) - poi_queue[i].deformation.u) < 0.05f) moved_by_one++;
            if (moved_by_one * 10 < n * 8) {
                std::cerr << "after reloading the target image only " << moved_by_one << " of " << n << " POIs follow it" << std::endl;
                return 18;
            }
            tar_img.fromRowMajor(tar.data());
            fftcc->setImages(ref_img, tar_img);
            icgn1->setImages(ref_img, tar_img);
            icgn1->prepare();
        }
        // single-POI entry point: recompute the first POI from its FFTCC state and check it agrees
        if (n > 0) {
            POI2D one(Point2D(xs[0], ys[0]));
            fftcc->compute(&one);
            icgn1->compute(&one);
            if (one.deformation.u != poi_queue[0].deformation.u || one.result.iteration != poi_queue[0].result.iteration) {
                std::cerr << "compute(POI2D*) disagrees with compute(vector&)" << std::endl;
                return 7;
            }
        }
        // the Newton-Raphson engine through the same interface: must agree with ICGN2D1 to well below the
        // convergence criterion on POIs both solved (they minimise the same ZNSSD)
        {
            NR2D1 nr1(rx, ry, it[0], it[1], cpu_thread_number);
            nr1.setImages(ref_img, tar_img);
            nr1.prepare();
            std::vector<POI2D> q = after_fftcc;
            nr1.compute(q);
            for (size_t i = 0; i < q.size(); i++) {
                if (q[i].result.zncc > 0.9f && poi_queue[i].result.zncc > 0.9f &&
                    (std::fabs(q[i].deformation.u - poi_queue[i].deformation.u) > 5e-3f ||
                     std::fabs(q[i].deformation.v - poi_queue[i].deformation.v) > 5e-3f)) {
                    std::cerr << "NR2D1 and ICGN2D1 disagree at POI " << i << std::endl;
                    return 12;
                }
            }
        }
        // the Levenberg-Marquardt engines (examples/test_2d_dic_fftcc_iclm1.cpp usage): same minimum as ICGN2D1
        {
            ICLM2D1 lm1(rx, ry, it[0], it[1], cpu_thread_number);
            lm1.setImages(ref_img, tar_img);
            lm1.prepare();
            lm1.setDamping(100.f, 0.1f, 10.f);
            std::vector<POI2D> q = after_fftcc;
            lm1.compute(q);
            ICLM2D2 lm2(rx, ry, it[0], it[1], cpu_thread_number);
            lm2.setImages(ref_img, tar_img);
            lm2.prepare();
            std::vector<POI2D> q2 = after_fftcc;
            lm2.compute(q2);
            for (size_t i = 0; i < q.size(); i++) {
                if (q[i].result.zncc > 0.9f && poi_queue[i].result.zncc > 0.9f &&
                    (std::fabs(q[i].deformation.u - poi_queue[i].deformation.u) > 5e-3f ||
                     std::fabs(q[i].deformation.v - poi_queue[i].deformation.v) > 5e-3f ||
                     std::fabs(q2[i].deformation.u - poi_queue[i].deformation.u) > 2e-2f ||
                     std::fabs(q2[i].deformation.v - poi_queue[i].deformation.v) > 2e-2f)) {
                    std::cerr << "ICLM2D1/2D2 and ICGN2D1 disagree at POI " << i << std::endl;
                    return 14;
                }
            }
        }
        // Strain over the ICGN result (examples/test_2d_dic_strain.cpp usage): a smooth field gives small, finite strains
        {
            std::vector<POI2D> q = poi_queue;
            Strain strain(20.f, 5, cpu_thread_number);
            strain.prepare(q);
            strain.compute(q);
            int fitted = 0;
            for (size_t i = 0; i < q.size(); i++) {
                if (q[i].result.zncc < 0.9f) {
                    if (q[i].strain.exx != 0.f) { std::cerr << "Strain touched a POI below the ZNCC threshold" << std::endl; return 15; }
                    continue;
                }
                if (!(std::fabs(q[i].strain.exx) < 0.05f) || !(std::fabs(q[i].strain.eyy) < 0.05f)) {
                    std::cerr << "implausible strain at POI " << i << ": " << q[i].strain.exx << " " << q[i].strain.eyy << std::endl;
                    return 15;
                }
                fitted += q[i].strain.exx != 0.f;
            }
            if (fitted * 2 < (int)q.size()) { std::cerr << "Strain fitted only " << fitted << " POIs" << std::endl; return 15; }
            POI2D probe = q[q.size() / 2];
            std::vector<POI2D> q2 = poi_queue;
            strain.compute(&q2[q2.size() / 2], q2);
            if (q2[q2.size() / 2].strain.exy != probe.strain.exy) { std::cerr << "Strain::compute(POI*, queue) disagrees" << std::endl; return 15; }
        }
        // reliable / unreliable selection and the merge after a RegionFit + ICGN round, against the reference example's host
        // loops (examples/test_3d_reconstruction_sift_icgn2_regfit.cpp:214-260) written out here
        {
            std::vector<POI2D> q = poi_queue;
            for (size_t i = 0; i < q.size(); i += 5) q[i].result.zncc = 0.3f;            // failed
            for (size_t i = 2; i < q.size(); i += 7) q[i].result.convergence = 0.5f;     // did not converge
            for (size_t i = 3; i < q.size(); i += 11) q[i].result.zncc = 0.8f;           // in between: neither set
            const float low = 0.7f, high = 0.9f, crit = 0.001f;
            std::vector<POI2D> rel, unr, rel_want, unr_want;
            std::vector<unsigned> idx;
            std::vector<int> idx_want;
            for (int i = 0; i < (int)q.size(); i++) {
                if (q[i].result.zncc < low || q[i].result.convergence > crit) { unr_want.push_back(q[i]); idx_want.push_back(i); }
                else if (q[i].result.zncc >= high) rel_want.push_back(q[i]);
            }
            splitReliable(*icgn1, q, low, high, crit, rel, unr, idx);
            bool same = rel.size() == rel_want.size() && unr.size() == unr_want.size() && idx.size() == idx_want.size() &&
                        (rel.empty() || std::memcmp(rel.data(), rel_want.data(), rel.size() * sizeof(POI2D)) == 0) &&
                        (unr.empty() || std::memcmp(unr.data(), unr_want.data(), unr.size() * sizeof(POI2D)) == 0);
            for (size_t j = 0; same && j < idx.size(); j++) same = (int)idx[j] == idx_want[j];
            if (!same || unr.empty() || rel.empty()) { std::cerr << "splitReliable differs from the host selection loop" << std::endl; return 19; }
            // pretend a round repaired every second unreliable POI
            for (size_t j = 0; j < unr.size(); j += 2) { unr[j].result.zncc = 0.95f; unr[j].result.convergence = 0.0005f; }
            std::vector<POI2D> q_want = q, rel2_want = rel, unr2_want;
            std::vector<unsigned> idx2_want;
            for (size_t j = 0; j < unr.size(); j++) {
                if (unr[j].result.zncc >= high && unr[j].result.convergence <= crit) { q_want[idx[j]] = unr[j]; rel2_want.push_back(unr[j]); }
                else { unr2_want.push_back(unr[j]); idx2_want.push_back(idx[j]); }
            }
            const size_t rec = mergeRecovered(*icgn1, q, unr, idx, high, crit, rel);
            same = rec == rel2_want.size() - rel_want.size() && rel.size() == rel2_want.size() && unr.size() == unr2_want.size() &&
                   std::memcmp(q.data(), q_want.data(), q.size() * sizeof(POI2D)) == 0 &&
                   std::memcmp(rel.data(), rel2_want.data(), rel.size() * sizeof(POI2D)) == 0 &&
                   (unr.empty() || std::memcmp(unr.data(), unr2_want.data(), unr.size() * sizeof(POI2D)) == 0);
            for (size_t j = 0; same && j < idx.size(); j++) same = idx[j] == idx2_want[j];
            if (!same) { std::cerr << "mergeRecovered differs from the host merge loop" << std::endl; return 20; }
        }
        // candidate batching (the EpipolarSearch pattern): three trial guesses per POI, the middle one is FFTCC's
        {
            std::vector<POI2D> cand;
            std::vector<unsigned> starts;
            const size_t m = after_fftcc.size() < 40 ? after_fftcc.size() : 40;
            for (size_t i = 0; i < m; i++) {
                starts.push_back((unsigned)cand.size());
                for (int t = -1; t <= 1; t++) {
                    POI2D c = after_fftcc[i];
                    c.deformation.u += 4.f * t;
                    cand.push_back(c);
                }
            }
            starts.push_back((unsigned)cand.size());
            std::vector<POI2D> best(after_fftcc.begin(), after_fftcc.begin() + m);
            icgn1->computeBestOf(cand, starts, best);
            for (size_t i = 0; i < m; i++) {
                float top = cand[3 * i].result.zncc;
                for (int t = 1; t < 3; t++) top = cand[3 * i + t].result.zncc > top ? cand[3 * i + t].result.zncc : top;
                if (best[i].result.zncc != top) { std::cerr << "computeBestOf did not keep the highest ZNCC at POI " << i << std::endl; return 16; }
            }
        }
        // the reference's CUDA-module shape (gpu_lib/opencorr_gpu.h:31-101): ICGN2D1GPU fed with row-major Img2D
        {
            Img2D ref2{w, h, ref.data()}, tar2{w, h, tar.data()};
            ICGN2D1GPU gpu(rx, ry, it[0], (int)it[1]);
            gpu.setImages(ref2, tar2);
            gpu.prepare();
            std::vector<POI2D> q = after_fftcc;
            gpu.compute(q);
            if (std::memcmp(q.data(), poi_queue.data(), q.size() * sizeof(POI2D)) != 0) {
                std::cerr << "ICGN2D1GPU (Img2D) differs from ICGN2D1 (Image2D)" << std::endl;
                return 13;
            }
        }
        // error path: compute before prepare must throw std::string
        bool threw = false;
        try {
            ICGN2D1 bad(rx, ry, it[0], it[1], 1);
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
        std::fwrite(poi_queue.data(), sizeof(POI2D), poi_queue.size(), o);
        std::fclose(o);
    } catch (const std::string& msg) {
        std::cerr << "OpenCorr shim error: " << msg << std::endl;
        return 1;
    }
    return 0;
}
