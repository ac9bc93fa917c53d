// oc_deformation.h -- host-side Deformation2D1 / Deformation2D2 / Deformation3D1 with the reference's public interface
// (src/oc_deformation.h:26-100): the parameter members, `warp_matrix` with (row, column) access, setDeformation() in its
// four forms, setWarp(), warp(point).  The reference stores the warp in Eigen matrices (Matrix3f / Matrix6f / Matrix4f); here
// it is a small row-major fixed matrix with the slice of Eigen's interface a caller of these classes touches: (i, j) access,
// product, inverse.  Same parameter <-> matrix maps (src/oc_deformation.cpp:107-128, 284-350, 416-516) and the same
// matrix-vector association in warp() as the solvers on the device (row i: products added in ascending column).
// Pinned on the reference's own classes, bit for bit: tests/test_oracle_vs_ref_deformation.py.
#pragma once

#include <cmath>

#include "oc_types.h"

namespace opencorr {

namespace hipdetail {

// N x N float matrix, row-major, with Eigen-like element access
template <int N>
struct SmallMatrix {
    float m[N * N];
    SmallMatrix() {
        for (float& v : m) v = 0.f;
    }
    float& operator()(int i, int j) { return m[i * N + j]; }
    float operator()(int i, int j) const { return m[i * N + j]; }
    static SmallMatrix Identity() {
        SmallMatrix r;
        for (int i = 0; i < N; i++) r(i, i) = 1.f;
        return r;
    }
    void setIdentity() { *this = Identity(); }
    SmallMatrix operator*(const SmallMatrix& b) const {  // c(i, j) = a(i, 0) b(0, j) + a(i, 1) b(1, j) + ... in that order
        SmallMatrix c;
        for (int i = 0; i < N; i++)
            for (int j = 0; j < N; j++) {
                float v = (*this)(i, 0) * b(0, j);
                for (int k = 1; k < N; k++) v = v + (*this)(i, k) * b(k, j);
                c(i, j) = v;
            }
        return c;
    }
    // LU with partial pivoting + solve against the identity (what the solvers use for W * (dW)^-1)
    SmallMatrix inverse() const {
        float a[N][N], inv[N][N];
        int perm[N];
        for (int i = 0; i < N; i++) {
            perm[i] = i;
            for (int j = 0; j < N; j++) a[i][j] = (*this)(i, j);
        }
        for (int k = 0; k < N; k++) {
            int piv = k;
            float best = std::fabs(a[k][k]);
            for (int r = k + 1; r < N; r++)
                if (std::fabs(a[r][k]) > best) {
                    best = std::fabs(a[r][k]);
                    piv = r;
                }
            if (piv != k) {
                for (int j = 0; j < N; j++) {
                    const float t = a[k][j];
                    a[k][j] = a[piv][j];
                    a[piv][j] = t;
                }
                const int t = perm[k];
                perm[k] = perm[piv];
                perm[piv] = t;
            }
            for (int r = k + 1; r < N; r++) {
                const float f = a[r][k] / a[k][k];
                a[r][k] = f;
                for (int j = k + 1; j < N; j++) a[r][j] = a[r][j] - f * a[k][j];
            }
        }
        for (int c = 0; c < N; c++) {
            float y[N];
            for (int i = 0; i < N; i++) {
                float v = perm[i] == c ? 1.f : 0.f;
                for (int j = 0; j < i; j++) v = v - a[i][j] * y[j];
                y[i] = v;
            }
            for (int i = N - 1; i >= 0; i--) {
                float v = y[i];
                for (int j = i + 1; j < N; j++) v = v - a[i][j] * y[j];
                y[i] = v / a[i][i];
            }
            for (int i = 0; i < N; i++) inv[i][c] = y[i];
        }
        SmallMatrix r;
        for (int i = 0; i < N; i++)
            for (int j = 0; j < N; j++) r(i, j) = inv[i][j];
        return r;
    }
};

}  // namespace hipdetail

typedef hipdetail::SmallMatrix<3> Matrix3f;
typedef hipdetail::SmallMatrix<4> Matrix4f;
typedef hipdetail::SmallMatrix<6> Matrix6f;

// 2D deformation with the 1st order shape function (src/oc_deformation.h:26-46)
class Deformation2D1 {
public:
    float u, ux, uy;
    float v, vx, vy;
    Matrix3f warp_matrix;

    Deformation2D1() : u(0.f), ux(0.f), uy(0.f), v(0.f), vx(0.f), vy(0.f) { setWarp(); }
    Deformation2D1(float u_, float ux_, float uy_, float v_, float vx_, float vy_) { setDeformation(u_, ux_, uy_, v_, vx_, vy_); }
    Deformation2D1(float p[6]) { setDeformation(p); }

    // p <- warp_matrix (src/oc_deformation.cpp:107-115)
    void setDeformation() {
        u = warp_matrix(0, 2);
        ux = warp_matrix(0, 0) - 1.f;
        uy = warp_matrix(0, 1);
        v = warp_matrix(1, 2);
        vx = warp_matrix(1, 0);
        vy = warp_matrix(1, 1) - 1.f;
    }
    void setDeformation(float u_, float ux_, float uy_, float v_, float vx_, float vy_) {
        u = u_; ux = ux_; uy = uy_;
        v = v_; vx = vx_; vy = vy_;
        setWarp();
    }
    void setDeformation(float p[6]) { setDeformation(p[0], p[1], p[2], p[3], p[4], p[5]); }
    void setDeformation(Deformation2D1& other) { setDeformation(other.u, other.ux, other.uy, other.v, other.vx, other.vy); }

    // warp_matrix <- p (src/oc_deformation.cpp:117-128)
    void setWarp() {
        warp_matrix(0, 0) = 1.f + ux; warp_matrix(0, 1) = uy;       warp_matrix(0, 2) = u;
        warp_matrix(1, 0) = vx;       warp_matrix(1, 1) = 1.f + vy; warp_matrix(1, 2) = v;
        warp_matrix(2, 0) = 0.f;      warp_matrix(2, 1) = 0.f;      warp_matrix(2, 2) = 1.f;
    }
    // W * (x, y, 1)^T (src/oc_deformation.cpp:94-105)
    Point2D warp(Point2D& location) {
        const float pv[3] = {location.x, location.y, 1.f};
        float out[2];
        for (int i = 0; i < 2; i++) {
            float s = warp_matrix(i, 0) * pv[0];
            for (int k = 1; k < 3; k++) s = s + warp_matrix(i, k) * pv[k];
            out[i] = s;
        }
        return Point2D(out[0], out[1]);
    }
};

// 2D deformation with the 2nd order shape function (src/oc_deformation.h:48-71)
class Deformation2D2 {
public:
    float u, ux, uy, uxx, uxy, uyy;
    float v, vx, vy, vxx, vxy, vyy;
    Matrix6f warp_matrix;

    Deformation2D2() : u(0.f), ux(0.f), uy(0.f), uxx(0.f), uxy(0.f), uyy(0.f), v(0.f), vx(0.f), vy(0.f), vxx(0.f), vxy(0.f), vyy(0.f) { setWarp(); }
    Deformation2D2(float u_, float ux_, float uy_, float uxx_, float uxy_, float uyy_, float v_, float vx_, float vy_, float vxx_,
                   float vxy_, float vyy_) {
        setDeformation(u_, ux_, uy_, uxx_, uxy_, uyy_, v_, vx_, vy_, vxx_, vxy_, vyy_);
    }
    Deformation2D2(float p[12]) { setDeformation(p); }

    // p <- rows 3 and 4 of warp_matrix (src/oc_deformation.cpp:284-299)
    void setDeformation() {
        u = warp_matrix(3, 5);
        ux = warp_matrix(3, 3) - 1.f;
        uy = warp_matrix(3, 4);
        uxx = warp_matrix(3, 0) * 2.f;
        uxy = warp_matrix(3, 1);
        uyy = warp_matrix(3, 2) * 2.f;
        v = warp_matrix(4, 5);
        vx = warp_matrix(4, 3);
        vy = warp_matrix(4, 4) - 1.f;
        vxx = warp_matrix(4, 0) * 2.f;
        vxy = warp_matrix(4, 1);
        vyy = warp_matrix(4, 2) * 2.f;
    }
    void setDeformation(float u_, float ux_, float uy_, float uxx_, float uxy_, float uyy_, float v_, float vx_, float vy_, float vxx_,
                        float vxy_, float vyy_) {
        u = u_; ux = ux_; uy = uy_; uxx = uxx_; uxy = uxy_; uyy = uyy_;
        v = v_; vx = vx_; vy = vy_; vxx = vxx_; vxy = vxy_; vyy = vyy_;
        setWarp();
    }
    void setDeformation(float p[12]) { setDeformation(p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8], p[9], p[10], p[11]); }
    void setDeformation(Deformation2D2& o) { setDeformation(o.u, o.ux, o.uy, o.uxx, o.uxy, o.uyy, o.v, o.vx, o.vy, o.vxx, o.vxy, o.vyy); }
    // a first-order deformation promoted to second order (src/oc_deformation.cpp:249-266)
    void setDeformation(Deformation2D1& o) { setDeformation(o.u, o.ux, o.uy, 0.f, 0.f, 0.f, o.v, o.vx, o.vy, 0.f, 0.f, 0.f); }

    // warp_matrix <- p: the 18 polynomial entries of rows 0-2, the two parameter rows, the unit row (src/oc_deformation.cpp:301-350)
    void setWarp() {
        Matrix6f& w = warp_matrix;
        w(0, 0) = 1.f + 2.f * ux + ux * ux + u * uxx;
        w(0, 1) = 2.f * u * uxy + 2.f * (1.f + ux) * uy;
        w(0, 2) = uy * uy + u * uyy;
        w(0, 3) = 2.f * u * (1 + ux);
        w(0, 4) = 2.f * u * uy;
        w(0, 5) = u * u;
        w(1, 0) = 0.5f * (v * uxx + 2.f * (1.f + ux) * vx + u * vxx);
        w(1, 1) = 1.f + uy * vx + ux * vy + v * uxy + u * vxy + vy + ux;
        w(1, 2) = 0.5f * (v * uyy + 2.f * uy * (1.f + vy) + u * vyy);
        w(1, 3) = v + v * ux + u * vx;
        w(1, 4) = u + v * uy + u * vy;
        w(1, 5) = u * v;
        w(2, 0) = vx * vx + v * vxx;
        w(2, 1) = 2.f * v * vxy + 2.f * vx * (1.f + vy);
        w(2, 2) = 1.f + 2.f * vy + vy * vy + v * vyy;
        w(2, 3) = 2.f * v * vx;
        w(2, 4) = 2.f * v * (1.f + vy);
        w(2, 5) = v * v;
        w(3, 0) = 0.5f * uxx; w(3, 1) = uxy; w(3, 2) = 0.5f * uyy; w(3, 3) = 1.f + ux; w(3, 4) = uy;       w(3, 5) = u;
        w(4, 0) = 0.5f * vxx; w(4, 1) = vxy; w(4, 2) = 0.5f * vyy; w(4, 3) = vx;       w(4, 4) = 1.f + vy; w(4, 5) = v;
        w(5, 0) = 0.f; w(5, 1) = 0.f; w(5, 2) = 0.f; w(5, 3) = 0.f; w(5, 4) = 0.f; w(5, 5) = 1.f;
    }
    // rows 3 and 4 of W * (x^2, xy, y^2, x, y, 1)^T (src/oc_deformation.cpp:268-282)
    Point2D warp(Point2D location) {
        const float pv[6] = {location.x * location.x, location.x * location.y, location.y * location.y, location.x, location.y, 1.f};
        float out[2];
        for (int i = 0; i < 2; i++) {
            float s = warp_matrix(3 + i, 0) * pv[0];
            for (int k = 1; k < 6; k++) s = s + warp_matrix(3 + i, k) * pv[k];
            out[i] = s;
        }
        return Point2D(out[0], out[1]);
    }
};

// 3D deformation with the 1st order shape function (src/oc_deformation.h:73-98)
class Deformation3D1 {
public:
    float u, ux, uy, uz;
    float v, vx, vy, vz;
    float w, wx, wy, wz;
    Matrix4f warp_matrix;

    Deformation3D1() : u(0.f), ux(0.f), uy(0.f), uz(0.f), v(0.f), vx(0.f), vy(0.f), vz(0.f), w(0.f), wx(0.f), wy(0.f), wz(0.f) { setWarp(); }
    Deformation3D1(float u_, float ux_, float uy_, float uz_, float v_, float vx_, float vy_, float vz_, float w_, float wx_, float wy_,
                   float wz_) {
        setDeformation(u_, ux_, uy_, uz_, v_, vx_, vy_, vz_, w_, wx_, wy_, wz_);
    }
    Deformation3D1(float p[12]) { setDeformation(p); }

    // p <- warp_matrix (src/oc_deformation.cpp:416-432)
    void setDeformation() {
        u = warp_matrix(0, 3); ux = warp_matrix(0, 0) - 1.f; uy = warp_matrix(0, 1);       uz = warp_matrix(0, 2);
        v = warp_matrix(1, 3); vx = warp_matrix(1, 0);       vy = warp_matrix(1, 1) - 1.f; vz = warp_matrix(1, 2);
        w = warp_matrix(2, 3); wx = warp_matrix(2, 0);       wy = warp_matrix(2, 1);       wz = warp_matrix(2, 2) - 1.f;
    }
    void setDeformation(float u_, float ux_, float uy_, float uz_, float v_, float vx_, float vy_, float vz_, float w_, float wx_,
                        float wy_, float wz_) {
        u = u_; ux = ux_; uy = uy_; uz = uz_;
        v = v_; vx = vx_; vy = vy_; vz = vz_;
        w = w_; wx = wx_; wy = wy_; wz = wz_;
        setWarp();
    }
    void setDeformation(float p[12]) { setDeformation(p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8], p[9], p[10], p[11]); }
    void setDeformation(Deformation3D1& o) { setDeformation(o.u, o.ux, o.uy, o.uz, o.v, o.vx, o.vy, o.vz, o.w, o.wx, o.wy, o.wz); }

    // warp_matrix <- p (src/oc_deformation.cpp:495-516)
    void setWarp() {
        Matrix4f& m = warp_matrix;
        m(0, 0) = 1.f + ux; m(0, 1) = uy;       m(0, 2) = uz;       m(0, 3) = u;
        m(1, 0) = vx;       m(1, 1) = 1.f + vy; m(1, 2) = vz;       m(1, 3) = v;
        m(2, 0) = wx;       m(2, 1) = wy;       m(2, 2) = 1.f + wz; m(2, 3) = w;
        m(3, 0) = 0.f;      m(3, 1) = 0.f;      m(3, 2) = 0.f;      m(3, 3) = 1.f;
    }
    // W * (x, y, z, 1)^T (src/oc_deformation.cpp:518-530)
    Point3D warp(Point3D& location) {
        const float pv[4] = {location.x, location.y, location.z, 1.f};
        float out[3];
        for (int i = 0; i < 3; i++) {
            float s = warp_matrix(i, 0) * pv[0];
            for (int k = 1; k < 4; k++) s = s + warp_matrix(i, k) * pv[k];
            out[i] = s;
        }
        return Point3D(out[0], out[1], out[2]);
    }
};

}  // namespace opencorr
