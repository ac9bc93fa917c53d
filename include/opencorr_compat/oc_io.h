// oc_io.h -- the reference's result tables, written and read in its own formats.
//
// Host-side only.  IO2D / IO3D keep the reference's names and call shapes (src/oc_io.h:53-142) for the table and
// point functions an FFTCC -> ICGN -> Strain program uses; calibration files and the stereo (POI2DS) tables are not
// offered.  Formats (src/oc_io.cpp:249-392, 1004-1089): one header line, then one row per POI in fixed notation with
// 8 decimals, every field followed by the delimiter (also the last one):
//   saveTable2D             x,y,u,v,u0,v0,ZNCC,iteration,convergence,feature,exx,eyy,exy,subset_rx,subset_ry,
//   saveDeformationTable2D  x,y,u,ux,uy,uxx,uxy,uyy,v,vx,vy,vxx,vxy,vyy,subset_rx,subset_ry,
//   saveTable3D             x,y,z,u,v,w,u0,v0,w0,ZNCC,iteration,convergence,feature,ux,uy,uz,vx,vy,vz,wx,wy,wz,
//                           exx,eyy,ezz,exy,eyz,ezx,subset_rx,subset_ry,subset_rz,
// Loaders accept any prefix of the column list (older files of the reference lack the strain / radius columns) and
// leave the missing fields zero.  The Python twin is opencorr_amd/io.py; tests/test_io_formats.py exchanges files
// between the two.
#pragma once

#include <cstdio>
#include <fstream>
#include <iomanip>
#include <sstream>
#include <string>
#include <vector>

#include "oc_types.h"

namespace opencorr {

enum OutputVariable {  // src/oc_io.h:25-51
    u = 1, v = 2, w = 3, e_xx = 4, e_yy = 5, e_zz = 6, e_xy = 7, e_yz = 8, e_zx = 9, zncc = 10, zncc_r1r2 = 11,
    zncc_r1t2 = 12, deformation_increment = 13, iteration_step = 14, feature_nearby = 15, u_x = 16, u_y = 17, u_z = 18,
    v_x = 19, v_y = 20, v_z = 21, w_x = 22, w_y = 23, w_z = 24
};

namespace iodetail {

inline std::vector<std::vector<float>> read_rows(const std::string& path, const std::string& delimiter) {
    std::ifstream in(path);
    if (!in.is_open()) throw std::string("failed to read file " + path);
    std::vector<std::vector<float>> rows;
    std::string line;
    std::getline(in, line);  // header
    while (std::getline(in, line)) {
        while (!line.empty() && (line.back() == '\r' || line.back() == '\n' || line.back() == ' ')) line.pop_back();
        if (line.empty()) continue;
        std::vector<float> row;
        size_t a = 0;
        while (a <= line.size()) {
            size_t b = line.find(delimiter, a);
            if (b == std::string::npos) b = line.size();
            const std::string tok = line.substr(a, b - a);
            if (!tok.empty()) row.push_back(std::stof(tok));
            a = b + delimiter.size();
        }
        rows.push_back(row);
    }
    return rows;
}

struct Writer {
    std::ofstream out;
    std::string delimiter;
    Writer(const std::string& path, const std::string& delimiter_, int precision) : out(path), delimiter(delimiter_) {
        if (!out.is_open()) throw std::string("failed to write file " + path);
        out.setf(std::ios::fixed);
        out << std::setprecision(precision);
    }
    void header(std::initializer_list<const char*> names) {
        for (const char* n : names) out << n << delimiter;
        out << "\n";
    }
    Writer& operator<<(float v) {
        out << v << delimiter;
        return *this;
    }
    void endrow() { out << "\n"; }
};

}  // namespace iodetail

class IO2D {
    std::string file_path;
    std::string delimiter = ",";
    int width = 0, height = 0;

public:
    OutputVariable out_var = u;

    std::string getPath() const { return file_path; }
    std::string getDelimiter() const { return delimiter; }
    int getWidth() const { return width; }
    int getHeight() const { return height; }
    void setPath(std::string p) { file_path = p; }
    void setDelimiter(std::string d) { delimiter = d; }
    void setWidth(int w_) { width = w_; }
    void setHeight(int h_) { height = h_; }

    // x,y per row (src/oc_io.cpp:65-145; written with 4 decimals)
    std::vector<Point2D> loadPoint2D(std::string path) {
        std::vector<Point2D> q;
        for (const auto& r : iodetail::read_rows(path, delimiter))
            if (r.size() >= 2) q.emplace_back(r[0], r[1]);
        return q;
    }
    void savePoint2D(std::vector<Point2D> q, std::string path) {
        iodetail::Writer wr(path, delimiter, 4);
        wr.header({"x", "y"});
        for (const Point2D& p : q) {
            wr << p.x << p.y;
            wr.endrow();
        }
    }

    std::vector<POI2D> loadTable2D() {
        std::vector<POI2D> q;
        for (const auto& r : iodetail::read_rows(file_path, delimiter)) {
            if (r.size() < 2) continue;
            POI2D poi(r[0], r[1]);
            float* dst[] = {&poi.deformation.u, &poi.deformation.v, &poi.result.u0, &poi.result.v0, &poi.result.zncc,
                            &poi.result.iteration, &poi.result.convergence, &poi.result.feature, &poi.strain.exx,
                            &poi.strain.eyy, &poi.strain.exy, &poi.subset_radius.x, &poi.subset_radius.y};
            for (size_t i = 0; i < sizeof(dst) / sizeof(dst[0]) && i + 2 < r.size(); i++) *dst[i] = r[i + 2];
            q.push_back(poi);
        }
        return q;
    }
    void saveTable2D(std::vector<POI2D>& q) {
        iodetail::Writer wr(file_path, delimiter, 8);
        wr.header({"x", "y", "u", "v", "u0", "v0", "ZNCC", "iteration", "convergence", "feature", "exx", "eyy", "exy",
                   "subset_rx", "subset_ry"});
        for (const POI2D& p : q) {
            wr << p.x << p.y << p.deformation.u << p.deformation.v;
            for (float r : p.result.r) wr << r;
            for (float e : p.strain.e) wr << e;
            wr << p.subset_radius.x << p.subset_radius.y;
            wr.endrow();
        }
    }
    void saveDeformationTable2D(std::vector<POI2D>& q) {
        iodetail::Writer wr(file_path, delimiter, 8);
        wr.header({"x", "y", "u", "ux", "uy", "uxx", "uxy", "uyy", "v", "vx", "vy", "vxx", "vxy", "vyy", "subset_rx",
                   "subset_ry"});
        for (const POI2D& p : q) {
            wr << p.x << p.y;
            for (float d : p.deformation.p) wr << d;
            wr << p.subset_radius.x << p.subset_radius.y;
            wr.endrow();
        }
    }
    // height x width matrix, zero except at ((int)poi.y, (int)poi.x) (src/oc_io.cpp:394-520)
    void saveMap2D(std::vector<POI2D>& q, OutputVariable variable) {
        std::vector<float> map((size_t)height * width, 0.f);
        for (const POI2D& p : q) {
            const int r = (int)p.y, c = (int)p.x;
            if (r < 0 || c < 0 || r >= height || c >= width) continue;
            float val = 0.f;
            switch (variable) {
                case u: val = p.deformation.u; break;
                case v: val = p.deformation.v; break;
                case u_x: val = p.deformation.ux; break;
                case u_y: val = p.deformation.uy; break;
                case v_x: val = p.deformation.vx; break;
                case v_y: val = p.deformation.vy; break;
                case zncc: val = p.result.zncc; break;
                case deformation_increment: val = p.result.convergence; break;
                case iteration_step: val = p.result.iteration; break;
                case feature_nearby: val = p.result.feature; break;
                case e_xx: val = p.strain.exx; break;
                case e_yy: val = p.strain.eyy; break;
                case e_xy: val = p.strain.exy; break;
                default: throw std::string("saveMap2D: variable not available for POI2D");
            }
            map[(size_t)r * width + c] = val;
        }
        iodetail::Writer wr(file_path, delimiter, 8);
        for (int r = 0; r < height; r++) {
            for (int c = 0; c < width; c++) wr << map[(size_t)r * width + c];
            wr.endrow();
        }
    }
};

class IO3D {
    std::string file_path;
    std::string delimiter = ",";
    int dim_x = 0, dim_y = 0, dim_z = 0;

public:
    std::string getPath() const { return file_path; }
    std::string getDelimiter() const { return delimiter; }
    void setPath(std::string p) { file_path = p; }
    void setDelimiter(std::string d) { delimiter = d; }
    int getDimX() { return dim_x; }
    int getDimY() { return dim_y; }
    int getDimZ() { return dim_z; }
    void setDimX(int v_) { dim_x = v_; }
    void setDimY(int v_) { dim_y = v_; }
    void setDimZ(int v_) { dim_z = v_; }

    std::vector<Point3D> loadPoint3D(std::string path) {
        std::vector<Point3D> q;
        for (const auto& r : iodetail::read_rows(path, delimiter))
            if (r.size() >= 3) q.emplace_back(r[0], r[1], r[2]);
        return q;
    }
    void savePoint3D(std::vector<Point3D> q, std::string path) {
        iodetail::Writer wr(path, delimiter, 4);
        wr.header({"x", "y", "z"});
        for (const Point3D& p : q) {
            wr << p.x << p.y << p.z;
            wr.endrow();
        }
    }

    std::vector<POI3D> loadTable3D() {
        std::vector<POI3D> q;
        for (const auto& r : iodetail::read_rows(file_path, delimiter)) {
            if (r.size() < 3) continue;
            POI3D poi(r[0], r[1], r[2]);
            auto& d = poi.deformation;
            float* dst[] = {&d.u, &d.v, &d.w, &poi.result.u0, &poi.result.v0, &poi.result.w0, &poi.result.zncc,
                            &poi.result.iteration, &poi.result.convergence, &poi.result.feature, &d.ux, &d.uy, &d.uz, &d.vx,
                            &d.vy, &d.vz, &d.wx, &d.wy, &d.wz, &poi.strain.exx, &poi.strain.eyy, &poi.strain.ezz,
                            &poi.strain.exy, &poi.strain.eyz, &poi.strain.ezx, &poi.subset_radius.x, &poi.subset_radius.y,
                            &poi.subset_radius.z};
            for (size_t i = 0; i < sizeof(dst) / sizeof(dst[0]) && i + 3 < r.size(); i++) *dst[i] = r[i + 3];
            q.push_back(poi);
        }
        return q;
    }
    void saveTable3D(std::vector<POI3D>& q) {
        iodetail::Writer wr(file_path, delimiter, 8);
        wr.header({"x", "y", "z", "u", "v", "w", "u0", "v0", "w0", "ZNCC", "iteration", "convergence", "feature", "ux", "uy",
                   "uz", "vx", "vy", "vz", "wx", "wy", "wz", "exx", "eyy", "ezz", "exy", "eyz", "ezx", "subset_rx",
                   "subset_ry", "subset_rz"});
        for (const POI3D& p : q) {
            const auto& d = p.deformation;
            wr << p.x << p.y << p.z << d.u << d.v << d.w;
            for (float r : p.result.r) wr << r;
            wr << d.ux << d.uy << d.uz << d.vx << d.vy << d.vz << d.wx << d.wy << d.wz;
            for (float e : p.strain.e) wr << e;
            wr << p.subset_radius.x << p.subset_radius.y << p.subset_radius.z;
            wr.endrow();
        }
    }
};

}  // namespace opencorr
