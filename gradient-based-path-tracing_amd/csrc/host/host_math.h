// host_math.h — fp64 vector/matrix helpers for scene ingest (host only).
// Behaviour follows the reference's src/vector.h, src/matrix.h, src/transform.cpp
// (formulas restated; e.g. normalize multiplies by 1/length as src/vector.h:194-197,250-257 does).
#pragma once
#include <cmath>

namespace gdpt {

constexpr double kPi = 3.14159265358979323846; // c_PI, src/lajolla.h:25

struct V2 { double x = 0, y = 0; };
struct V3 {
    double x = 0, y = 0, z = 0;
    double &operator[](int i) { return (&x)[i]; }
    const double &operator[](int i) const { return (&x)[i]; }
};
inline V3 operator+(const V3 &a, const V3 &b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(const V3 &a, const V3 &b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator-(const V3 &a) { return {-a.x, -a.y, -a.z}; }
inline V3 operator*(const V3 &a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(double s, const V3 &a) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator/(const V3 &a, double s) { double inv = 1.0 / s; return {a.x * inv, a.y * inv, a.z * inv}; }
inline double dot(const V3 &a, const V3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(const V3 &a, const V3 &b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline double length(const V3 &a) { return std::sqrt(dot(a, a)); }
inline V3 normalize(const V3 &a) {
    double l = length(a);
    if (l <= 0) return {0, 0, 0};
    return a / l;
}
inline double radians(double deg) { return (kPi / 180.0) * deg; }
inline double degrees(double rad) { return (180.0 / kPi) * rad; }

struct M4 {
    double m[4][4];
    double &operator()(int i, int j) { return m[i][j]; }
    const double &operator()(int i, int j) const { return m[i][j]; }
    static M4 identity() {
        M4 r{};
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) r.m[i][j] = (i == j) ? 1.0 : 0.0;
        return r;
    }
};
inline M4 operator*(const M4 &a, const M4 &b) {
    M4 r{};
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            double s = 0;
            for (int k = 0; k < 4; k++) s += a.m[i][k] * b.m[k][j];
            r.m[i][j] = s;
        }
    return r;
}

// General 4x4 inverse by cofactors (adjugate / determinant), as src/matrix.h:80-200 does.
inline M4 inverse(const M4 &a) {
    auto minor3 = [&](int r0, int r1, int r2, int c0, int c1, int c2) {
        return a.m[r0][c0] * (a.m[r1][c1] * a.m[r2][c2] - a.m[r1][c2] * a.m[r2][c1]) -
               a.m[r0][c1] * (a.m[r1][c0] * a.m[r2][c2] - a.m[r1][c2] * a.m[r2][c0]) +
               a.m[r0][c2] * (a.m[r1][c0] * a.m[r2][c1] - a.m[r1][c1] * a.m[r2][c0]);
    };
    M4 adj{};
    for (int i = 0; i < 4; i++) {
        int r[3], n = 0;
        for (int k = 0; k < 4; k++) if (k != i) r[n++] = k;
        for (int j = 0; j < 4; j++) {
            int c[3]; n = 0;
            for (int k = 0; k < 4; k++) if (k != j) c[n++] = k;
            double cof = minor3(r[0], r[1], r[2], c[0], c[1], c[2]);
            if ((i + j) & 1) cof = -cof;
            adj.m[j][i] = cof; // transpose of the cofactor matrix
        }
    }
    double det = a.m[0][0] * adj.m[0][0] + a.m[0][1] * adj.m[1][0] + a.m[0][2] * adj.m[2][0] + a.m[0][3] * adj.m[3][0];
    M4 inv{};
    if (det == 0) return inv; // src/matrix.h returns a zero matrix for singular input
    double inv_det = 1.0 / det;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) inv.m[i][j] = adj.m[i][j] * inv_det;
    return inv;
}

// src/transform.cpp:5-80
inline M4 translate(const V3 &d) { M4 r = M4::identity(); r(0, 3) = d.x; r(1, 3) = d.y; r(2, 3) = d.z; return r; }
inline M4 scale(const V3 &s) { M4 r = M4::identity(); r(0, 0) = s.x; r(1, 1) = s.y; r(2, 2) = s.z; return r; }
inline M4 rotate(double angle_deg, const V3 &axis) {
    V3 a = normalize(axis);
    double s = std::sin(radians(angle_deg)), c = std::cos(radians(angle_deg));
    M4 m = M4::identity();
    m(0, 0) = a.x * a.x + (1 - a.x * a.x) * c;
    m(0, 1) = a.x * a.y * (1 - c) - a.z * s;
    m(0, 2) = a.x * a.z * (1 - c) + a.y * s;
    m(1, 0) = a.x * a.y * (1 - c) + a.z * s;
    m(1, 1) = a.y * a.y + (1 - a.y * a.y) * c;
    m(1, 2) = a.y * a.z * (1 - c) - a.x * s;
    m(2, 0) = a.x * a.z * (1 - c) - a.y * s;
    m(2, 1) = a.y * a.z * (1 - c) + a.x * s;
    m(2, 2) = a.z * a.z + (1 - a.z * a.z) * c;
    return m;
}
inline M4 look_at(const V3 &pos, const V3 &look, const V3 &up) {
    V3 dir = normalize(look - pos);
    V3 left = normalize(cross(normalize(up), dir));
    V3 new_up = cross(dir, left);
    M4 m = M4::identity();
    m(0, 0) = left.x;   m(1, 0) = left.y;   m(2, 0) = left.z;
    m(0, 1) = new_up.x; m(1, 1) = new_up.y; m(2, 1) = new_up.z;
    m(0, 2) = dir.x;    m(1, 2) = dir.y;    m(2, 2) = dir.z;
    m(0, 3) = pos.x;    m(1, 3) = pos.y;    m(2, 3) = pos.z;
    return m;
}
inline M4 perspective(double fov_deg) {
    double cot = 1.0 / std::tan(radians(fov_deg / 2.0));
    M4 m{};
    m(0, 0) = cot; m(1, 1) = cot; m(2, 2) = 1; m(2, 3) = -1; m(3, 2) = 1;
    return m;
}
inline V3 xform_point(const M4 &x, const V3 &p) {
    double tx = x(0, 0) * p.x + x(0, 1) * p.y + x(0, 2) * p.z + x(0, 3);
    double ty = x(1, 0) * p.x + x(1, 1) * p.y + x(1, 2) * p.z + x(1, 3);
    double tz = x(2, 0) * p.x + x(2, 1) * p.y + x(2, 2) * p.z + x(2, 3);
    double tw = x(3, 0) * p.x + x(3, 1) * p.y + x(3, 2) * p.z + x(3, 3);
    double inv_w = 1.0 / tw;
    return {tx * inv_w, ty * inv_w, tz * inv_w};
}
inline V3 xform_vector(const M4 &x, const V3 &v) {
    return {x(0, 0) * v.x + x(0, 1) * v.y + x(0, 2) * v.z,
            x(1, 0) * v.x + x(1, 1) * v.y + x(1, 2) * v.z,
            x(2, 0) * v.x + x(2, 1) * v.y + x(2, 2) * v.z};
}
inline V3 xform_normal(const M4 &inv_x, const V3 &n) {
    return normalize(V3{inv_x(0, 0) * n.x + inv_x(1, 0) * n.y + inv_x(2, 0) * n.z,
                        inv_x(0, 1) * n.x + inv_x(1, 1) * n.y + inv_x(2, 1) * n.z,
                        inv_x(0, 2) * n.x + inv_x(1, 2) * n.y + inv_x(2, 2) * n.z});
}

} // namespace gdpt
