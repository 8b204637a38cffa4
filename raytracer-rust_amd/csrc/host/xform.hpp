// xform.hpp -- f32 math the loader needs: the reference's own Vec3 (src/vec3.rs) and the glam 0.30.3
// pieces it calls (Cargo.lock pin; crate source is not under /root/reference, so these restate the
// crate's published algorithms):
//   Quat::from_euler(EulerRot::YXZ, a, b, c) = Ry(a) * Rx(b) * Rz(c)  (parser.rs:663-668), in f32 like glam (an f64 evaluation
//       rounded once differs from it by an ulp here and there; against the reference's committed render the f32 form matches
//       1.7 % more pixels exactly -- tests/test_oracle_golden.py).
//   Mat4::from_scale_rotation_translation, Mat4::inverse (cofactor form), Mat4 * Vec4 (column-major).
#pragma once
#include <cmath>

namespace mi355rt_host {

constexpr float EPSILON = 1e-4f;                         // renderer.rs:17
constexpr float PI_F = 3.14159265358979323846f;          // std::f32::consts::PI

struct V3 { float x, y, z; };
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float length(V3 a) { return std::sqrt(dot(a, a)); }
inline V3 normalized(V3 a) { float l = length(a); if (l < EPSILON) return a; return a * (1.0f / l); }   // vec3.rs:37-44

struct Quat { float x, y, z, w; };
struct Mat4 { float m[16]; };                            // m[4*col + row]

inline float to_radians(float deg) { return deg * (PI_F / 180.0f); }   // f32::to_radians

inline Quat quat_from_euler_yxz(float a, float b, float c) {
    // all in f32, as glam computes it: sin_cos of the f32 half angles (correctly rounded here: via double, so the result does not
    // depend on the platform's sinf), then the scalar Quat * Quat with every product and sum rounded to f32, left to right
    auto sc = [](float angle, float& s_out, float& c_out) { const float h = angle * 0.5f; s_out = (float)std::sin((double)h); c_out = (float)std::cos((double)h); };
    auto mul = [](Quat p, Quat q) {
        return Quat{((p.w * q.x + p.x * q.w) + p.y * q.z) - p.z * q.y, ((p.w * q.y - p.x * q.z) + p.y * q.w) + p.z * q.x,
                    ((p.w * q.z + p.x * q.y) - p.y * q.x) + p.z * q.w, ((p.w * q.w - p.x * q.x) - p.y * q.y) - p.z * q.z};
    };
    float sa, ca, sb, cb, sc_, cc; sc(a, sa, ca); sc(b, sb, cb); sc(c, sc_, cc);
    const Quat qy{0.0f, sa, 0.0f, ca}, qx{sb, 0.0f, 0.0f, cb}, qz{0.0f, 0.0f, sc_, cc};
    return mul(mul(qy, qx), qz);
}

inline Mat4 mat4_from_scale_rotation_translation(V3 s, Quat q, V3 t) {
    const float x2 = q.x + q.x, y2 = q.y + q.y, z2 = q.z + q.z;
    const float xx = q.x * x2, xy = q.x * y2, xz = q.x * z2;
    const float yy = q.y * y2, yz = q.y * z2, zz = q.z * z2;
    const float wx = q.w * x2, wy = q.w * y2, wz = q.w * z2;
    const float xa[4] = {1.0f - (yy + zz), xy + wz, xz - wy, 0.0f};
    const float ya[4] = {xy - wz, 1.0f - (xx + zz), yz + wx, 0.0f};
    const float za[4] = {xz + wy, yz - wx, 1.0f - (xx + yy), 0.0f};
    Mat4 r;
    for (int i = 0; i < 4; ++i) { r.m[i] = xa[i] * s.x; r.m[4 + i] = ya[i] * s.y; r.m[8 + i] = za[i] * s.z; }
    r.m[12] = t.x; r.m[13] = t.y; r.m[14] = t.z; r.m[15] = 1.0f;
    return r;
}

inline Mat4 mat4_inverse(const Mat4& a) {
    const float* m = a.m;
    const float m00 = m[0], m01 = m[1], m02 = m[2], m03 = m[3], m10 = m[4], m11 = m[5], m12 = m[6], m13 = m[7];
    const float m20 = m[8], m21 = m[9], m22 = m[10], m23 = m[11], m30 = m[12], m31 = m[13], m32 = m[14], m33 = m[15];
    const float coef00 = m22 * m33 - m32 * m23, coef02 = m12 * m33 - m32 * m13, coef03 = m12 * m23 - m22 * m13;
    const float coef04 = m21 * m33 - m31 * m23, coef06 = m11 * m33 - m31 * m13, coef07 = m11 * m23 - m21 * m13;
    const float coef08 = m21 * m32 - m31 * m22, coef10 = m11 * m32 - m31 * m12, coef11 = m11 * m22 - m21 * m12;
    const float coef12 = m20 * m33 - m30 * m23, coef14 = m10 * m33 - m30 * m13, coef15 = m10 * m23 - m20 * m13;
    const float coef16 = m20 * m32 - m30 * m22, coef18 = m10 * m32 - m30 * m12, coef19 = m10 * m22 - m20 * m12;
    const float coef20 = m20 * m31 - m30 * m21, coef22 = m10 * m31 - m30 * m11, coef23 = m10 * m21 - m20 * m11;
    const float fac0[4] = {coef00, coef00, coef02, coef03}, fac1[4] = {coef04, coef04, coef06, coef07};
    const float fac2[4] = {coef08, coef08, coef10, coef11}, fac3[4] = {coef12, coef12, coef14, coef15};
    const float fac4[4] = {coef16, coef16, coef18, coef19}, fac5[4] = {coef20, coef20, coef22, coef23};
    const float vec0[4] = {m10, m00, m00, m00}, vec1[4] = {m11, m01, m01, m01};
    const float vec2[4] = {m12, m02, m02, m02}, vec3[4] = {m13, m03, m03, m03};
    const float sign_a[4] = {1.0f, -1.0f, 1.0f, -1.0f}, sign_b[4] = {-1.0f, 1.0f, -1.0f, 1.0f};
    float c0[4], c1[4], c2[4], c3[4];
    for (int i = 0; i < 4; ++i) {
        c0[i] = ((vec1[i] * fac0[i] - vec2[i] * fac1[i]) + vec3[i] * fac2[i]) * sign_a[i];
        c1[i] = ((vec0[i] * fac0[i] - vec2[i] * fac3[i]) + vec3[i] * fac4[i]) * sign_b[i];
        c2[i] = ((vec0[i] * fac1[i] - vec1[i] * fac3[i]) + vec3[i] * fac5[i]) * sign_a[i];
        c3[i] = ((vec0[i] * fac2[i] - vec1[i] * fac4[i]) + vec2[i] * fac5[i]) * sign_b[i];
    }
    const float d0 = m[0] * c0[0], d1 = m[1] * c1[0], d2 = m[2] * c2[0], d3 = m[3] * c3[0];
    const float det = ((d0 + d1) + d2) + d3;
    const float rcp = 1.0f / det;
    Mat4 r;
    for (int i = 0; i < 4; ++i) { r.m[i] = c0[i] * rcp; r.m[4 + i] = c1[i] * rcp; r.m[8 + i] = c2[i] * rcp; r.m[12 + i] = c3[i] * rcp; }
    return r;
}

inline V3 mat4_mul_point(const Mat4& a, V3 p) {          // (a * Vec4(p, 1)).truncate()
    float out[3];
    for (int r = 0; r < 3; ++r) {
        float acc = a.m[r] * p.x;
        acc = acc + a.m[4 + r] * p.y;
        acc = acc + a.m[8 + r] * p.z;
        acc = acc + a.m[12 + r] * 1.0f;
        out[r] = acc;
    }
    return {out[0], out[1], out[2]};
}

}  // namespace mi355rt_host
