// An UNMODIFIED caller shape of the reference: compute(&poi) from inside the caller's own OpenMP loop
// (src/oc_epipolar_search.cpp:184-188; the reference keeps one scratch instance per thread, src/oc_icgn.cpp:61-69,147).
// The C-ABI combines the concurrent calls into one launch per batch (oc_hip_compute_one); this driver checks that every
// record equals the queue call bit for bit and times both forms.
//   omp_single_poi <in.bin> <out.bin> <threads>      in.bin: the layout of tests/cpp/shim_driver.cpp
// prints one JSON line: seconds of the combined loop, of the one-launch-per-call loop, of the queue call; batches served.
#include <omp.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

#include "opencorr_compat/opencorr.h"

using namespace opencorr;

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 3;
    int hdr[5];
    float it[2];
    if (fread(hdr, sizeof(int), 5, f) != 5 || fread(it, sizeof(float), 2, f) != 2) return 4;
    const int h = hdr[0], w = hdr[1], rx = hdr[2], ry = hdr[3], n = hdr[4];
    const int threads = atoi(argv[3]);
    std::vector<float> ref((size_t)h * w), tar((size_t)h * w), xs(n), ys(n);
    if (fread(ref.data(), 4, ref.size(), f) != ref.size() || fread(tar.data(), 4, tar.size(), f) != tar.size() ||
        fread(xs.data(), 4, n, f) != (size_t)n || fread(ys.data(), 4, n, f) != (size_t)n)
        return 5;
    fclose(f);
    Image2D ref_img(w, h), tar_img(w, h);
    for (int r = 0; r < h; r++)
        for (int c = 0; c < w; c++) {
            ref_img.eg_mat(r, c) = ref[(size_t)r * w + c];
            tar_img.eg_mat(r, c) = tar[(size_t)r * w + c];
        }
    try {
        std::vector<POI2D> start;
        for (int i = 0; i < n; i++) start.push_back(POI2D(xs[i], ys[i]));
        FFTCC2D fftcc(rx, ry, threads);
        fftcc.setImages(ref_img, tar_img);
        fftcc.compute(start);
        ICGN2D1 icgn(rx, ry, it[0], it[1], threads);
        icgn.setImages(ref_img, tar_img);
        icgn.prepare();
        // the queue call: the yardstick for bits and time
        std::vector<POI2D> want = start;
        icgn.compute(want);
        want = start;
        double t0 = now();
        icgn.compute(want);
        const double t_queue = now() - t0;
        // the reference's caller shape, combined
        std::vector<POI2D> got = start;
        unsigned long long b0 = 0, p0 = 0, b1 = 0, p1 = 0;
        oc_hip_single_stats(icgn.handle(), &b0, &p0);
        t0 = now();
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
        for (int i = 0; i < n; i++) icgn.compute(&got[i]);
        const double t_comb = now() - t0;
        oc_hip_single_stats(icgn.handle(), &b1, &p1);
        const bool same = memcmp(got.data(), want.data(), sizeof(POI2D) * (size_t)n) == 0;
        // ... and one launch per call (the round-5 behaviour) on a slice, scaled
        const int m = n < 2000 ? n : 2000;
        std::vector<POI2D> slow(start.begin(), start.begin() + m);
        oc_hip_set_tuning(icgn.handle(), "single_combine", 0);
        t0 = now();
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
        for (int i = 0; i < m; i++) icgn.compute(&slow[i]);
        const double t_serial = (now() - t0) * (double)n / m;
        oc_hip_set_tuning(icgn.handle(), "single_combine", 1);
        const bool same_slow = memcmp(slow.data(), want.data(), sizeof(POI2D) * (size_t)m) == 0;
        printf("{\"pois\": %d, \"threads\": %d, \"same_bits_combined\": %s, \"same_bits_one_launch_per_call\": %s, \"seconds_combined\": %.6f, "
               "\"seconds_one_launch_per_call_scaled\": %.6f, \"seconds_queue_call\": %.6f, \"batches\": %llu, \"pois_batched\": %llu}\n",
               n, threads, same ? "true" : "false", same_slow ? "true" : "false", t_comb, t_serial, t_queue, b1 - b0, p1 - p0);
        FILE* o = fopen(argv[2], "wb");
        if (!o) return 6;
        fwrite(got.data(), sizeof(POI2D), got.size(), o);
        fclose(o);
        return same && same_slow ? 0 : 7;
    } catch (std::string& e) {
        fprintf(stderr, "opencorr: %s\n", e.c_str());
        return 8;
    }
}
