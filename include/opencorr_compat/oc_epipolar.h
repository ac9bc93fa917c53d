// oc_epipolar.h -- the candidate generation of EpipolarSearch::compute(POI2D*) (src/oc_epipolar_search.cpp:133-180) as a
// host-side helper, so that the reference's per-POI loop
//     for each POI:  build <= 2 * radius / step + 1 trial positions along the epipolar line;
//                    icgn1->compute(&candidate) for each (an omp loop over a handful of candidates);
//                    std::sort by ZNCC, keep candidates[0]                                     (:133-195)
// becomes ONE batch for the GPU engine:
//     epipolarCandidates(...) for every POI -> one candidate queue + segment starts
//     ICGN2D1::computeBestOf(candidates, segment_starts, poi_queue)   (oc_engines.h: one launch + one selection kernel)
// Only plain floats cross this header: the fundamental matrix (row-major, what EpipolarSearch::updateFundementalMatrix
// builds from the two cameras' calibration, :99-118) comes from the caller's calibration code -- Calibration / Stereovision
// are outside this library's scope.  Every expression below is the reference's, operand for operand; its Eigen product
// fundamental_matrix * view1_vector is restated as the coefficient-wise sum with ascending inner index, (a0 + a1) + a2.  The
// candidates equal those of the reference's compiled EpipolarSearch bit for bit where that reference is built against
// oracle/ref_stubs' stand-in Eigen (tests/test_oracle_vs_ref_epipolar.py) -- which sums the three terms in the same order.  Real
// Eigen 3.4 may associate a 3-term row product differently (a0 + (a1 + a2)): a one-ulp change of the line's coefficients, and
// since x_view2 / y_view2 are truncated to int (src/oc_epipolar_search.cpp:166-179) a trial position can then move by a pixel for
// a POI that sits exactly on a truncation boundary.  No real Eigen exists in this image to decide it; ICGN refines every trial to
// the same sub-pixel minimum either way.
#pragma once

#include <vector>

#include "oc_types.h"

namespace opencorr {

struct EpipolarSearchSetting {
    float fundamental_matrix[9];          // row-major 3 x 3
    float parallax_x[3], parallax_y[3];   // EpipolarSearch::setParallax(coefficient_x, coefficient_y) (:84-93); a constant parallax is {0, 0, p}
    int search_radius, search_step;       // EpipolarSearch::setSearch (:43-52)
    int subset_radius_x, subset_radius_y; // of the ICGN2D1 the search runs (createICGN, :54-57)
    int image_width, image_height;
};

// Appends the trial POIs of `poi` to `candidates` (the centre of the search region first, then +step, -step, +2 step, ... as the
// reference pushes them) and returns how many were appended.
inline int epipolarCandidates(const POI2D& poi, const EpipolarSearchSetting& s, std::vector<POI2D>& candidates) {
    const float* F = s.fundamental_matrix;
    // estimate parallax (:136-137)
    const float parallax_x = s.parallax_x[0] * (poi.x - int(s.image_width / 2)) + s.parallax_x[1] * (poi.y - int(s.image_height / 2)) + s.parallax_x[2];
    const float parallax_y = s.parallax_y[0] * (poi.x - int(s.image_width / 2)) + s.parallax_y[1] * (poi.y - int(s.image_height / 2)) + s.parallax_y[2];
    // location of the left POI as a homogeneous vector, its epipolar line in the secondary view (:140-146)
    const float v0 = poi.x + poi.deformation.u, v1 = poi.y + poi.deformation.v, v2 = 1;
    float e[3];
    for (int i = 0; i < 3; i++) {
        float a = F[i * 3 + 0] * v0;
        a = a + F[i * 3 + 1] * v1;
        a = a + F[i * 3 + 2] * v2;
        e[i] = a;
    }
    const float line_slope = -e[0] / e[1];
    const float line_intercept = -e[2] / e[1];
    const int x_view2 = (int)((line_slope * (poi.y + poi.deformation.v + parallax_y - line_intercept) + poi.x + poi.deformation.u + parallax_x) / (line_slope * line_slope + 1));
    const int y_view2 = (int)(line_slope * x_view2 + line_intercept);
    // the centre of the searching region (:150-155: pushed without a bounds test)
    const size_t before = candidates.size();
    POI2D current_poi(poi.x, poi.y);
    current_poi.deformation.u = x_view2 - poi.x;
    current_poi.deformation.v = y_view2 - poi.y;
    candidates.push_back(current_poi);
    // the other trial locations (:158-180)
    for (int i = s.search_step; i < s.search_radius; i += s.search_step) {
        for (int sign = 1; sign >= -1; sign -= 2) {
            const int x_trial = x_view2 + sign * i;
            const int y_trial = (int)(line_slope * x_trial + line_intercept);
            current_poi.deformation.u = x_trial - poi.x;
            current_poi.deformation.v = y_trial - poi.y;
            if (x_trial - s.subset_radius_x > 0 && x_trial + s.subset_radius_x < s.image_width - 1 && y_trial - s.subset_radius_y > 0 &&
                y_trial + s.subset_radius_y < s.image_height - 1)
                candidates.push_back(current_poi);
        }
    }
    return (int)(candidates.size() - before);
}

// The whole queue: candidates of all POIs in one vector, those of poi_queue[k] at [segment_starts[k], segment_starts[k + 1]).
inline void epipolarCandidates(const std::vector<POI2D>& poi_queue, const EpipolarSearchSetting& s, std::vector<POI2D>& candidates,
                               std::vector<unsigned>& segment_starts) {
    candidates.clear();
    segment_starts.assign(1, 0u);
    for (const POI2D& poi : poi_queue) {
        epipolarCandidates(poi, s, candidates);
        segment_starts.push_back((unsigned)candidates.size());
    }
}

}  // namespace opencorr
