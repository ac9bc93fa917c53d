// C wrapper of include/opencorr_compat/oc_deformation.h with the signature of oracle/ref_driver.cpp's oc_ref_deformation
// (tests/test_oracle_vs_ref_deformation.py compiles this with plain g++ and compares the two bit for bit).
#include <cstring>

#include "opencorr_compat/oc_deformation.h"

using namespace opencorr;

extern "C" int oc_test_deformation(int kind, const float* p, const float* pt, const float* mat_in, float* mat_out, float* warped,
                                   float* p_back) {
    if (kind == 1) {
        float q[6];
        std::memcpy(q, p, sizeof(q));
        Deformation2D1 d(q);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) mat_out[i * 3 + j] = d.warp_matrix(i, j);
        Point2D a(pt[0], pt[1]);
        Point2D b = d.warp(a);
        warped[0] = b.x; warped[1] = b.y;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) d.warp_matrix(i, j) = mat_in[i * 3 + j];
        d.setDeformation();
        const float back[6] = {d.u, d.ux, d.uy, d.v, d.vx, d.vy};
        std::memcpy(p_back, back, sizeof(back));
        // the other construction paths end in the same state
        Deformation2D1 e(q[0], q[1], q[2], q[3], q[4], q[5]), f;
        f.setDeformation(e);
        if (std::memcmp(f.warp_matrix.m, mat_out, sizeof(float) * 9) != 0) return 3;
    } else if (kind == 2) {
        float q[12];
        std::memcpy(q, p, sizeof(q));
        Deformation2D2 d(q);
        for (int i = 0; i < 6; i++)
            for (int j = 0; j < 6; j++) mat_out[i * 6 + j] = d.warp_matrix(i, j);
        Point2D b = d.warp(Point2D(pt[0], pt[1]));
        warped[0] = b.x; warped[1] = b.y;
        for (int i = 0; i < 6; i++)
            for (int j = 0; j < 6; j++) d.warp_matrix(i, j) = mat_in[i * 6 + j];
        d.setDeformation();
        const float back[12] = {d.u, d.ux, d.uy, d.uxx, d.uxy, d.uyy, d.v, d.vx, d.vy, d.vxx, d.vxy, d.vyy};
        std::memcpy(p_back, back, sizeof(back));
        Deformation2D2 f;
        Deformation2D2 e(q);
        f.setDeformation(e);
        if (std::memcmp(f.warp_matrix.m, mat_out, sizeof(float) * 36) != 0) return 3;
        // a first-order deformation promoted to second order keeps its six parameters and zeroes the others
        Deformation2D1 first(q[0], q[1], q[2], q[6], q[7], q[8]);
        f.setDeformation(first);
        if (f.u != q[0] || f.vy != q[8] || f.uxx != 0.f || f.vyy != 0.f || f.warp_matrix(3, 5) != q[0]) return 4;
    } else if (kind == 3) {
        float q[12];
        std::memcpy(q, p, sizeof(q));
        Deformation3D1 d(q);
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) mat_out[i * 4 + j] = d.warp_matrix(i, j);
        Point3D a(pt[0], pt[1], pt[2]);
        Point3D b = d.warp(a);
        warped[0] = b.x; warped[1] = b.y; warped[2] = b.z;
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) d.warp_matrix(i, j) = mat_in[i * 4 + j];
        d.setDeformation();
        const float back[12] = {d.u, d.ux, d.uy, d.uz, d.v, d.vx, d.vy, d.vz, d.w, d.wx, d.wy, d.wz};
        std::memcpy(p_back, back, sizeof(back));
        Deformation3D1 f, e(q);
        f.setDeformation(e);
        if (std::memcmp(f.warp_matrix.m, mat_out, sizeof(float) * 16) != 0) return 3;
    } else {
        return 2;
    }
    return 0;
}

// W * (dW)^-1 with the header's small matrices (what a caller composing warps by hand computes): returned row-major
extern "C" void oc_test_warp_compose(int n, const float* w, const float* dw, float* out) {
    if (n == 3) {
        Matrix3f a, b;
        std::memcpy(a.m, w, sizeof(a.m));
        std::memcpy(b.m, dw, sizeof(b.m));
        const Matrix3f c = a * b.inverse();
        std::memcpy(out, c.m, sizeof(c.m));
    } else if (n == 4) {
        Matrix4f a, b;
        std::memcpy(a.m, w, sizeof(a.m));
        std::memcpy(b.m, dw, sizeof(b.m));
        const Matrix4f c = a * b.inverse();
        std::memcpy(out, c.m, sizeof(c.m));
    } else {
        Matrix6f a, b;
        std::memcpy(a.m, w, sizeof(a.m));
        std::memcpy(b.m, dw, sizeof(b.m));
        const Matrix6f c = a * b.inverse();
        std::memcpy(out, c.m, sizeof(c.m));
    }
}
