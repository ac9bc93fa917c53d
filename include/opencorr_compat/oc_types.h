// oc_types.h -- Eigen/OpenCV-free boundary types with OpenCorr's names and memory layout.
//
// These are the PODs that cross the drop-in boundary: Point2D/3D
// (src/oc_point.h), the POI unions and POI2D/POI3D (src/oc_poi.h:25-222; POI2D = 25 floats,
// POI3D = 31 floats, no vptr), Image2D with a column-major eg_mat(r, c) accessor like
// Eigen::MatrixXf (src/oc_image.h:27-45) and Image3D with one contiguous z,y,x block behind
// vol_mat[z][y][x] (src/oc_image.h:47-68, src/oc_array.h:57-74).  The reference decodes image files with
// OpenCV (cv::imread(path, IMREAD_GRAYSCALE), src/oc_image.cpp:37-58); here Image2D(path) reads the formats
// the reference's own fixtures use without any library -- uncompressed BMP (8-bit palette, 24- and 32-bit,
// converted to 8-bit grey with OpenCV's fixed-point weights) and binary PGM -- and Image3D(path) the `.bin`
// volumes (int[3] header + floats, src/oc_image.cpp:76-110); other formats fail like a failed imread
// (`throw std::string`), or are filled from memory by the caller.
#pragma once

#include <cmath>
#include <cstddef>
#include <cstdint>
#include <fstream>
#include <iostream>
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

namespace detail {

inline uint32_t le32(const unsigned char* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint16_t le16(const unsigned char* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
// 8-bit grey of an (r, g, b) triple the way cv::imread(..., IMREAD_GRAYSCALE) forms it: fixed point, 14 fractional bits
inline unsigned char grey_of(unsigned r, unsigned g, unsigned b) { return (unsigned char)((b * 1868u + g * 9617u + r * 4899u + (1u << 13)) >> 14); }

// Decodes an uncompressed BMP or a binary PGM into row-major 8-bit grey; false = not one of these / damaged.
inline bool decode_grey(const std::vector<unsigned char>& f, int& width, int& height, std::vector<unsigned char>& grey) {
    if (f.size() >= 54 && f[0] == 'B' && f[1] == 'M') {
        const uint32_t data_off = le32(&f[10]), header = le32(&f[14]);
        if (header < 40) return false;
        const int32_t w = (int32_t)le32(&f[18]), hs = (int32_t)le32(&f[22]);
        const int bpp = le16(&f[28]);
        const uint32_t compression = le32(&f[30]);
        if (w <= 0 || hs == 0 || (compression != 0 && !(compression == 3 && bpp == 32)) || (bpp != 8 && bpp != 24 && bpp != 32)) return false;
        const int h = hs < 0 ? -hs : hs;
        const size_t row_bytes = (((size_t)w * bpp + 31) / 32) * 4;
        if ((size_t)data_off + row_bytes * h > f.size()) return false;
        unsigned char pal[256];
        if (bpp == 8) {
            uint32_t colours = le32(&f[46]);
            if (colours == 0 || colours > 256) colours = 256;
            const size_t pal_off = 14 + (size_t)header;
            if (pal_off + 4 * (size_t)colours > f.size()) return false;
            for (uint32_t i = 0; i < 256; i++) pal[i] = i < colours ? grey_of(f[pal_off + 4 * i + 2], f[pal_off + 4 * i + 1], f[pal_off + 4 * i]) : 0;
        }
        width = w;
        height = h;
        grey.resize((size_t)w * h);
        for (int r = 0; r < h; r++) {
            const unsigned char* src = &f[data_off + row_bytes * (size_t)(hs < 0 ? r : h - 1 - r)];  // bottom-up unless the height is negative
            unsigned char* dst = &grey[(size_t)r * w];
            if (bpp == 8)
                for (int c = 0; c < w; c++) dst[c] = pal[src[c]];
            else
                for (int c = 0; c < w; c++) dst[c] = grey_of(src[(bpp / 8) * c + 2], src[(bpp / 8) * c + 1], src[(bpp / 8) * c]);
        }
        return true;
    }
    if (f.size() > 2 && f[0] == 'P' && f[1] == '5') {
        size_t pos = 2;
        int vals[3], got = 0;
        while (got < 3 && pos < f.size()) {
            while (pos < f.size() && (f[pos] == ' ' || f[pos] == '\n' || f[pos] == '\r' || f[pos] == '\t')) pos++;
            if (pos < f.size() && f[pos] == '#') {
                while (pos < f.size() && f[pos] != '\n') pos++;
                continue;
            }
            int v = 0, digits = 0;
            while (pos < f.size() && f[pos] >= '0' && f[pos] <= '9') { v = v * 10 + (f[pos] - '0'); pos++; digits++; }
            if (!digits) return false;
            vals[got++] = v;
        }
        if (got < 3 || vals[0] <= 0 || vals[1] <= 0 || vals[2] <= 0 || vals[2] > 255) return false;
        pos++;  // the single whitespace after maxval
        if (pos + (size_t)vals[0] * vals[1] > f.size()) return false;
        width = vals[0];
        height = vals[1];
        grey.assign(f.begin() + pos, f.begin() + pos + (size_t)width * height);
        return true;
    }
    return false;
}

}  // namespace detail

class Image2D {
public:
    int height, width;
    unsigned int size;
    std::string file_path;
    ColMajorMatrixXf eg_mat;  // column-major like Eigen::MatrixXf
    Image2D(int width_, int height_) : height(height_), width(width_), size((unsigned)(width_ * height_)), eg_mat(height_, width_) {}
    // Image2D(std::string file_path) / load(), src/oc_image.cpp:32-58
    Image2D(std::string path) : height(0), width(0), size(0) { load(path); }
    void load(std::string path) {
        std::ifstream in(path, std::ios::in | std::ios::binary);
        std::vector<unsigned char> bytes;
        if (in.is_open()) bytes.assign(std::istreambuf_iterator<char>(in), std::istreambuf_iterator<char>());
        int w = 0, h = 0;
        std::vector<unsigned char> grey;
        if (!detail::decode_grey(bytes, w, h, grey)) throw std::string("Fail to load file: " + path);
        file_path = path;
        if (width != w || height != h) {
            width = w;
            height = h;
            size = (unsigned)(h * w);
            eg_mat.resize(h, w);
        }
        for (int r = 0; r < h; r++)
            for (int c = 0; c < w; c++) eg_mat(r, c) = (float)grey[(size_t)r * w + c];
    }
    // fill from a row-major buffer (e.g. Img2D::data of the reference's CUDA module)
    void fromRowMajor(const float* src) {
        for (int r = 0; r < height; r++)
            for (int c = 0; c < width; c++) eg_mat(r, c) = src[(size_t)r * width + c];
    }
};

class Image3D {
public:
    int dim_x = 0, dim_y = 0, dim_z = 0;
    unsigned long size = 0;
    std::string file_path;
    float*** vol_mat = nullptr;  // vol_mat[z][y][x]; &vol_mat[0][0][0] is one contiguous block
    Image3D(int dim_x_, int dim_y_, int dim_z_) { allocate(dim_x_, dim_y_, dim_z_); }
    // Image3D(std::string file_path) / load() / loadBin(), src/oc_image.cpp:71-110,147-165
    Image3D(std::string path) { load(path); }
    void load(std::string path) {
        file_path = path;
        const size_t dot = path.find_last_of(".");
        const std::string ext = dot == std::string::npos ? "" : path.substr(dot + 1);
        if (ext == "bin" || ext == "BIN") loadBin(path);
        else throw std::string("Fail to load file (only .bin volumes are decoded without OpenCV): " + path);
    }
    void loadBin(std::string path) {
        std::ifstream in(path, std::ios::in | std::ios::binary);
        if (!in.is_open()) throw std::string("Failed to open bin file: " + path);
        int dims[3] = {0, 0, 0};  // x, y, z
        in.read((char*)dims, sizeof(dims));
        if (!in || dims[0] <= 0 || dims[1] <= 0 || dims[2] <= 0) throw std::string("Not a volume file: " + path);
        allocate(dims[0], dims[1], dims[2]);
        in.read((char*)data_.data(), (std::streamsize)(sizeof(float) * size));
        if ((unsigned long)in.gcount() != sizeof(float) * size) throw std::string("Truncated volume file: " + path);
    }
    Image3D(const Image3D&) = delete;
    Image3D& operator=(const Image3D&) = delete;
    void release() {}

private:
    void allocate(int dx, int dy, int dz) {
        dim_x = dx; dim_y = dy; dim_z = dz;
        size = (unsigned long)dx * dy * dz;
        data_.assign(size, 0.f);
        rows_.resize((size_t)dim_z * dim_y);
        slabs_.resize(dim_z);
        for (int z = 0; z < dim_z; z++) {
            for (int y = 0; y < dim_y; y++) rows_[(size_t)z * dim_y + y] = data_.data() + ((size_t)z * dim_y + y) * dim_x;
            slabs_[z] = rows_.data() + (size_t)z * dim_y;
        }
        vol_mat = slabs_.data();
    }
    std::vector<float> data_;
    std::vector<float*> rows_;
    std::vector<float**> slabs_;
};

}  // namespace opencorr
