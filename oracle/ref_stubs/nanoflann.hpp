// nanoflann.hpp -- a MINIMAL stand-in for nanoflann (>= 1.5: ResultItem / SearchParameters), written from scratch so that
// the reference's own src/oc_nearest_neighbor.cpp, oc_strain.cpp and oc_region_fit.cpp compile UNMODIFIED into
// oracle/_ref/liboc_ref.so.  TEST INFRASTRUCTURE ONLY; nothing in the product includes it.
//
// The surface the reference uses (src/oc_nearest_neighbor.h:76, src/oc_nearest_neighbor.cpp:108-190):
//   KDTreeSingleIndexAdaptor<L2_Simple_Adaptor<float, Cloud>, Cloud, 3>(dim, cloud, {max_leaf})
//   radiusSearch(query, squared_radius, std::vector<ResultItem<uint32_t, float>>&, SearchParameters{sorted = false})
//   knnSearch(query, k, uint32_t* indices, float* squared_distances)
// Semantics restated from nanoflann's documentation and result sets:
//   * L2_Simple distance: sum over the dimensions of (a - b)^2, accumulated in float in dimension order;
//   * radius search keeps points with distance STRICTLY below the squared radius (RadiusResultSet::addPoint:
//     `if (dist < radius)`); with sorted = false the order of the matches is the tree's traversal order -- an
//     implementation detail nobody may rely on; this stand-in returns them in index order;
//   * knn search returns up to k points by ascending distance; among equal distances the one met first stays first
//     (KNNResultSet::addPoint shifts only entries with a strictly larger distance) -- here: lower index first.
// A brute-force scan replaces the tree: same result SETS as nanoflann by definition of the queries.
#pragma once

#include <cstddef>
#include <cstdint>
#include <vector>

namespace nanoflann {

template <typename IndexType = size_t, typename DistanceType = double>
struct ResultItem {
    ResultItem() = default;
    ResultItem(const IndexType index, const DistanceType distance) : first(index), second(distance) {}
    IndexType first;
    DistanceType second;
};

struct SearchParameters {
    SearchParameters(float eps_ = 0, bool sorted_ = true) : eps(eps_), sorted(sorted_) {}
    float eps;
    bool sorted;
};

struct KDTreeSingleIndexAdaptorParams {
    KDTreeSingleIndexAdaptorParams(size_t leaf_max_size_ = 10) : leaf_max_size(leaf_max_size_) {}
    size_t leaf_max_size;
};

template <class T, class DataSource, typename DistT = T, typename IndexT = uint32_t>
struct L2_Simple_Adaptor {
    using ElementType = T;
    using DistanceType = DistT;
    using IndexType = IndexT;
};

template <typename Distance, class DatasetAdaptor, int DIM = -1, typename IndexT = uint32_t>
class KDTreeSingleIndexAdaptor {
    const DatasetAdaptor& data_;
    int dim_;

    float dist(const float* q, size_t i) const {
        float d = 0.f;
        for (int k = 0; k < dim_; k++) {
            const float diff = q[k] - data_.kdtree_get_pt(i, (size_t)k);
            d += diff * diff;
        }
        return d;
    }

public:
    KDTreeSingleIndexAdaptor(int dimensionality, const DatasetAdaptor& data, const KDTreeSingleIndexAdaptorParams& = {})
        : data_(data), dim_(DIM > 0 ? DIM : dimensionality) {}

    size_t radiusSearch(const float* query, float squared_radius, std::vector<ResultItem<IndexT, float>>& matches,
                        const SearchParameters& = {}) const {
        matches.clear();
        const size_t n = data_.kdtree_get_point_count();
        for (size_t i = 0; i < n; i++) {
            const float d = dist(query, i);
            if (d < squared_radius) matches.emplace_back((IndexT)i, d);
        }
        return matches.size();
    }

    size_t knnSearch(const float* query, size_t k, IndexT* indices, float* dists) const {
        const size_t n = data_.kdtree_get_point_count();
        size_t count = 0;
        for (size_t i = 0; i < n; i++) {
            const float d = dist(query, i);
            size_t pos = count;
            while (pos > 0 && dists[pos - 1] > d) pos--;
            if (pos >= k) continue;
            const size_t last = count < k ? count : k - 1;
            for (size_t j = last; j > pos; j--) {
                dists[j] = dists[j - 1];
                indices[j] = indices[j - 1];
            }
            dists[pos] = d;
            indices[pos] = (IndexT)i;
            if (count < k) count++;
        }
        return count;
    }
};

}  // namespace nanoflann
