// oc_types.h -- Eigen/OpenCV-free boundary types with OpenCorr's names and memory layout.
//
// These are the PODs that cross the drop-in boundary: Point2D/3D
// (src/oc_point.h), the POI unions and POI2D/POI3D (src/oc_poi.h:25-222; POI2D = 25 floats,
// POI3D = 31 floats, no vptr), Image2D with a column-major eg_mat(r, c) accessor like
// Eigen::MatrixXf (src/oc_image.h:27-45) and Image3D with one contiguous z,y,x block behind
// vol_mat[z][y][x] (src/oc_image.h:47-68, src/oc_array.h:57-74).  Image decoding (OpenCV) is out
// of scope: images are filled from memory by the caller.
#pragma once

#include <cmath>
#include <cstddef>
#include <string>
#include <vector>

namespace opencorr {

class Point2D {
public:
    float x, y;
    Point2D() : x(0.f), y(0.f) {}
    Point2D(float x_, float y_) : x(x_), y(y_) {}
    Point2D(int x_, int y_) : x((float)x_), y((float)y_) {}
    float vectorNorm() const { return std::sqrt(x * x + y * y); }
};
inline Point2D operator+(Point2D a, Point2D b) { return Point2D(a.x + b.x, a.y + b.y); }
inline Point2D operator-(Point2D a, Point2D b) { return Point2D(a.x - b.x, a.y - b.y); }
inline Point2D operator*(float f, Point2D p) { return Point2D(f * p.x, f * p.y); }
inline Point2D operator*(Point2D p, float f) { return f * p; }

class Point3D {
public:
    float x, y, z;
    Point3D() : x(0.f), y(0.f), z(0.f) {}
    Point3D(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    Point3D(int x_, int y_, int z_) : x((float)x_), y((float)y_), z((float)z_) {}
    float vectorNorm() const { return std::sqrt(x * x + y * y + z * z); }
};
inline Point3D operator+(Point3D a, Point3D b) { return Point3D(a.x + b.x, a.y + b.y, a.z + b.z); }
inline Point3D operator-(Point3D a, Point3D b) { return Point3D(a.x - b.x, a.y - b.y, a.z - b.z); }
inline Point3D operator*(float f, Point3D p) { return Point3D(f * p.x, f * p.y, f * p.z); }

union DeformationVector2D {
    struct { float u, ux, uy, uxx, uxy, uyy, v, vx, vy, vxx, vxy, vyy; };
    float p[12];
};
union StrainVector2D {
    struct { float exx, eyy, exy; };
    float e[3];
};
union Result2D {
    struct { float u0, v0, zncc, iteration, convergence, feature; };
    float r[6];
};
union DeformationVector3D {
    struct { float u, ux, uy, uz, v, vx, vy, vz, w, wx, wy, wz; };
    float p[12];
};
union StrainVector3D {
    struct { float exx, eyy, ezz, exy, eyz, ezx; };
    float e[6];
};
union Result3D {
    struct { float u0, v0, w0, zncc, iteration, convergence, feature; };
    float r[7];
};

class POI2D : public Point2D {
public:
    DeformationVector2D deformation;
    Result2D result;
    StrainVector2D strain;
    Point2D subset_radius;
    POI2D(int x_, int y_) : Point2D(x_, y_) { clear(); }
    POI2D(float x_, float y_) : Point2D(x_, y_) { clear(); }
    POI2D(Point2D location) : Point2D(location) { clear(); }
    void clear() {  // everything except the location
        for (float& v : deformation.p) v = 0.f;
        for (float& v : result.r) v = 0.f;
        for (float& v : strain.e) v = 0.f;
        subset_radius = Point2D();
    }
};

class POI3D : public Point3D {
public:
    DeformationVector3D deformation;
    Result3D result;
    StrainVector3D strain;
    Point3D subset_radius;
    POI3D(int x_, int y_, int z_) : Point3D(x_, y_, z_) { clear(); }
    POI3D(float x_, float y_, float z_) : Point3D(x_, y_, z_) { clear(); }
    POI3D(Point3D location) : Point3D(location) { clear(); }
    void clear() {
        for (float& v : deformation.p) v = 0.f;
        for (float& v : result.r) v = 0.f;
        for (float& v : strain.e) v = 0.f;
        subset_radius = Point3D();
    }
};

static_assert(sizeof(POI2D) == 100, "POI2D must be 25 packed floats (src/oc_poi.h:102-136)");
static_assert(sizeof(POI3D) == 124, "POI3D must be 31 packed floats (src/oc_poi.h:187-222)");

// Column-major float matrix with the slice of Eigen::MatrixXf's interface the hot path's
// callers use: (r, c) access, rows(), cols(), data(), setZero().
class ColMajorMatrixXf {
    int rows_ = 0, cols_ = 0;
    std::vector<float> v_;

public:
    ColMajorMatrixXf() {}
    ColMajorMatrixXf(int rows, int cols) { resize(rows, cols); }
    void resize(int rows, int cols) { rows_ = rows; cols_ = cols; v_.assign((size_t)rows * cols, 0.f); }
    void setZero() { v_.assign(v_.size(), 0.f); }
    int rows() const { return rows_; }
    int cols() const { return cols_; }
    float& operator()(int r, int c) { return v_[(size_t)c * rows_ + r]; }
    float operator()(int r, int c) const { return v_[(size_t)c * rows_ + r]; }
    float* data() { return v_.data(); }
    const float* data() const { return v_.data(); }
};

class Image2D {
public:
    int height, width;
    unsigned int size;
    std::string file_path;
    ColMajorMatrixXf eg_mat;  // column-major like Eigen::MatrixXf
    Image2D(int width_, int height_) : height(height_), width(width_), size((unsigned)(width_ * height_)), eg_mat(height_, width_) {}
    // fill from a row-major buffer (e.g. Img2D::data of the reference's CUDA module)
    void fromRowMajor(const float* src) {
        for (int r = 0; r < height; r++)
            for (int c = 0; c < width; c++) eg_mat(r, c) = src[(size_t)r * width + c];
    }
};

class Image3D {
public:
    int dim_x, dim_y, dim_z;
    unsigned long size;
    std::string file_path;
    float*** vol_mat = nullptr;  // vol_mat[z][y][x]; &vol_mat[0][0][0] is one contiguous block
    Image3D(int dim_x_, int dim_y_, int dim_z_) : dim_x(dim_x_), dim_y(dim_y_), dim_z(dim_z_), size((unsigned long)dim_x_ * dim_y_ * dim_z_) {
        data_.assign(size, 0.f);
        rows_.resize((size_t)dim_z * dim_y);
        slabs_.resize(dim_z);
        for (int z = 0; z < dim_z; z++) {
            for (int y = 0; y < dim_y; y++) rows_[(size_t)z * dim_y + y] = data_.data() + ((size_t)z * dim_y + y) * dim_x;
            slabs_[z] = rows_.data() + (size_t)z * dim_y;
        }
        vol_mat = slabs_.data();
    }
    Image3D(const Image3D&) = delete;
    Image3D& operator=(const Image3D&) = delete;
    void release() {}

private:
    std::vector<float> data_;
    std::vector<float*> rows_;
    std::vector<float**> slabs_;
};

}  // namespace opencorr
