// opencv2/opencv.hpp -- stand-in so that src/oc_image.cpp of the reference compiles (oracle/Makefile, target `ref`).
// Image DECODING is off the arithmetic path: imread / imreadmulti always fail here, the oracle/_ref
// driver fills Image2D::eg_mat and Image3D::vol_mat directly.  TEST INFRASTRUCTURE ONLY.
#pragma once

#include <string>
#include <vector>

typedef unsigned char uchar;

#define CV_8UC1 0
#define CV_8UC3 16

namespace cv {

enum { IMREAD_GRAYSCALE = 0, IMREAD_COLOR = 1 };

class Mat {
public:
    int rows = 0, cols = 0, channels_ = 1;
    uchar* data = nullptr;
    std::vector<uchar> store;
    static Mat zeros(int rows, int cols, int type) {
        Mat m;
        m.rows = rows;
        m.cols = cols;
        m.channels_ = type == CV_8UC3 ? 3 : 1;
        m.store.assign((size_t)rows * cols * m.channels_, 0);
        m.data = m.store.data();
        return m;
    }
    Mat() {}
    Mat(const Mat& o) : rows(o.rows), cols(o.cols), channels_(o.channels_), store(o.store) { data = store.empty() ? nullptr : store.data(); }
    Mat& operator=(const Mat& o) {
        rows = o.rows; cols = o.cols; channels_ = o.channels_; store = o.store;
        data = store.empty() ? nullptr : store.data();
        return *this;
    }
    template <class T>
    T& at(int r, int c) { return reinterpret_cast<T*>(data)[(size_t)r * cols + c]; }
};

inline Mat imread(const std::string&, int) { return Mat(); }
inline bool imreadmulti(const std::string&, std::vector<Mat>&, int) { return false; }
inline void split(const Mat& m, std::vector<Mat>& channels) {
    channels.assign(m.channels_, Mat::zeros(m.rows, m.cols, CV_8UC1));
}

}  // namespace cv
