// C wrapper of include/opencorr_compat/oc_epipolar.h for the Python tests (compiled with plain g++: the candidate
// generation is host code and needs neither HIP nor the engine library).
#include <cstring>

#include "opencorr_compat/oc_epipolar.h"

using namespace opencorr;

extern "C" long oc_test_epipolar_candidates(const float* pois, long n, const float* F9, const float* px3, const float* py3, int radius,
                                            int step, int rx, int ry, int width, int height, float* cand_out, long cand_cap,
                                            unsigned* starts_out) {
    static_assert(sizeof(POI2D) == 100, "POI2D is 25 packed floats");
    EpipolarSearchSetting s;
    std::memcpy(s.fundamental_matrix, F9, sizeof(s.fundamental_matrix));
    std::memcpy(s.parallax_x, px3, sizeof(s.parallax_x));
    std::memcpy(s.parallax_y, py3, sizeof(s.parallax_y));
    s.search_radius = radius;
    s.search_step = step;
    s.subset_radius_x = rx;
    s.subset_radius_y = ry;
    s.image_width = width;
    s.image_height = height;
    std::vector<POI2D> q((size_t)n, POI2D(0.f, 0.f));
    if (n) std::memcpy(static_cast<void*>(q.data()), pois, sizeof(POI2D) * (size_t)n);
    std::vector<POI2D> cand;
    std::vector<unsigned> starts;
    epipolarCandidates(q, s, cand, starts);
    if ((long)cand.size() > cand_cap) return -(long)cand.size();
    if (!cand.empty()) std::memcpy(cand_out, static_cast<const void*>(cand.data()), sizeof(POI2D) * cand.size());
    std::memcpy(starts_out, starts.data(), sizeof(unsigned) * starts.size());
    return (long)cand.size();
}
