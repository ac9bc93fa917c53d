// opencv2/core/eigen.hpp -- cv2eigen of the stand-in (see opencv.hpp): 8-bit gray levels -> float matrix.
#pragma once

#include <Eigen/Eigen>

#include "../opencv.hpp"

namespace cv {
inline void cv2eigen(const Mat& src, Eigen::MatrixXf& dst) {
    dst.resize(src.rows, src.cols);
    for (int r = 0; r < src.rows; r++)
        for (int c = 0; c < src.cols; c++) dst(r, c) = src.data ? (float)src.data[(size_t)r * src.cols + c] : 0.f;
}
}  // namespace cv
