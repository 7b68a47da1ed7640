// glam.hpp — the subset of glam 0.31.0 (Cargo.lock:552-555; scalar f32 code path, as built for wasm32 without
// simd128) that the reference's host-side hot path uses, restated for the C++ host layer:
//   Mat4::from_scale_rotation_translation   /root/reference/crates/renderer/src/transforms.rs:495-497
//   Mat4::mul_mat4                          transforms.rs:396-404, meshes/skins.rs:170
//   Mat4::inverse / transpose, Mat3::from_mat4     transforms.rs:412-413, camera.rs:160-162
//   Mat4::transform_point3 (no perspective divide)  renderable.rs:127-131, bounds.rs:52-59
// glam is not vendored under /root/reference, so this is its published algorithm restated ("parity unpinned").
// Compiled with -ffp-contract=off: every operation rounds to f32 exactly once, in the written order.
#pragma once
#include <cmath>
#include <cstring>

namespace awsm_host {

struct Vec3 { float x, y, z; };
struct Vec4 { float x, y, z, w; };
struct Quat { float x, y, z, w; };
struct Mat4 { Vec4 c[4]; };   // column-major: c[col]

inline Vec4 v4_scale(Vec4 a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
inline Vec4 v4_add(Vec4 a, Vec4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline Vec4 v4_sub(Vec4 a, Vec4 b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
inline Vec4 v4_mul(Vec4 a, Vec4 b) { return {a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
inline Vec3 v3_min(Vec3 a, Vec3 b) { return {std::fmin(a.x, b.x), std::fmin(a.y, b.y), std::fmin(a.z, b.z)}; }
inline Vec3 v3_max(Vec3 a, Vec3 b) { return {std::fmax(a.x, b.x), std::fmax(a.y, b.y), std::fmax(a.z, b.z)}; }

inline Mat4 mat4_identity() { return {{{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}}}; }

inline Mat4 mat4_from_srt(Vec3 scale, Quat r, Vec3 t) {
    const float x = r.x, y = r.y, z = r.z, w = r.w;
    const float x2 = x + x, y2 = y + y, z2 = z + z;
    const float xx = x * x2, xy = x * y2, xz = x * z2;
    const float yy = y * y2, yz = y * z2, zz = z * z2;
    const float wx = w * x2, wy = w * y2, wz = w * z2;
    const Vec4 xa = {1.0f - (yy + zz), xy + wz, xz - wy, 0.0f};
    const Vec4 ya = {xy - wz, 1.0f - (xx + zz), yz + wx, 0.0f};
    const Vec4 za = {xz + wy, yz - wx, 1.0f - (xx + yy), 0.0f};
    Mat4 m;
    m.c[0] = v4_scale(xa, scale.x);
    m.c[1] = v4_scale(ya, scale.y);
    m.c[2] = v4_scale(za, scale.z);
    m.c[3] = {t.x, t.y, t.z, 1.0f};
    return m;
}

inline Vec4 mat4_mul_vec4(const Mat4& m, Vec4 v) {
    Vec4 res = v4_scale(m.c[0], v.x);
    res = v4_add(res, v4_scale(m.c[1], v.y));
    res = v4_add(res, v4_scale(m.c[2], v.z));
    res = v4_add(res, v4_scale(m.c[3], v.w));
    return res;
}
inline Mat4 mat4_mul(const Mat4& a, const Mat4& b) {
    Mat4 r;
    for (int i = 0; i < 4; i++) r.c[i] = mat4_mul_vec4(a, b.c[i]);
    return r;
}
inline Mat4 mat4_transpose(const Mat4& m) {
    return {{{m.c[0].x, m.c[1].x, m.c[2].x, m.c[3].x}, {m.c[0].y, m.c[1].y, m.c[2].y, m.c[3].y},
             {m.c[0].z, m.c[1].z, m.c[2].z, m.c[3].z}, {m.c[0].w, m.c[1].w, m.c[2].w, m.c[3].w}}};
}
inline Mat4 mat4_inverse(const Mat4& m) {
    const float m00 = m.c[0].x, m01 = m.c[0].y, m02 = m.c[0].z, m03 = m.c[0].w;
    const float m10 = m.c[1].x, m11 = m.c[1].y, m12 = m.c[1].z, m13 = m.c[1].w;
    const float m20 = m.c[2].x, m21 = m.c[2].y, m22 = m.c[2].z, m23 = m.c[2].w;
    const float m30 = m.c[3].x, m31 = m.c[3].y, m32 = m.c[3].z, m33 = m.c[3].w;
    const float coef00 = m22 * m33 - m32 * m23, coef02 = m12 * m33 - m32 * m13, coef03 = m12 * m23 - m22 * m13;
    const float coef04 = m21 * m33 - m31 * m23, coef06 = m11 * m33 - m31 * m13, coef07 = m11 * m23 - m21 * m13;
    const float coef08 = m21 * m32 - m31 * m22, coef10 = m11 * m32 - m31 * m12, coef11 = m11 * m22 - m21 * m12;
    const float coef12 = m20 * m33 - m30 * m23, coef14 = m10 * m33 - m30 * m13, coef15 = m10 * m23 - m20 * m13;
    const float coef16 = m20 * m32 - m30 * m22, coef18 = m10 * m32 - m30 * m12, coef19 = m10 * m22 - m20 * m12;
    const float coef20 = m20 * m31 - m30 * m21, coef22 = m10 * m31 - m30 * m11, coef23 = m10 * m21 - m20 * m11;
    const Vec4 fac0 = {coef00, coef00, coef02, coef03}, fac1 = {coef04, coef04, coef06, coef07}, fac2 = {coef08, coef08, coef10, coef11};
    const Vec4 fac3 = {coef12, coef12, coef14, coef15}, fac4 = {coef16, coef16, coef18, coef19}, fac5 = {coef20, coef20, coef22, coef23};
    const Vec4 vec0 = {m10, m00, m00, m00}, vec1 = {m11, m01, m01, m01}, vec2 = {m12, m02, m02, m02}, vec3 = {m13, m03, m03, m03};
    const Vec4 inv0 = v4_add(v4_sub(v4_mul(vec1, fac0), v4_mul(vec2, fac1)), v4_mul(vec3, fac2));
    const Vec4 inv1 = v4_add(v4_sub(v4_mul(vec0, fac0), v4_mul(vec2, fac3)), v4_mul(vec3, fac4));
    const Vec4 inv2 = v4_add(v4_sub(v4_mul(vec0, fac1), v4_mul(vec1, fac3)), v4_mul(vec3, fac5));
    const Vec4 inv3 = v4_add(v4_sub(v4_mul(vec0, fac2), v4_mul(vec1, fac4)), v4_mul(vec2, fac5));
    const Vec4 sign_a = {1.0f, -1.0f, 1.0f, -1.0f}, sign_b = {-1.0f, 1.0f, -1.0f, 1.0f};
    Mat4 inverse = {{v4_mul(inv0, sign_a), v4_mul(inv1, sign_b), v4_mul(inv2, sign_a), v4_mul(inv3, sign_b)}};
    const Vec4 col0 = {inverse.c[0].x, inverse.c[1].x, inverse.c[2].x, inverse.c[3].x};
    const Vec4 dot0 = v4_mul(m.c[0], col0);
    const float dot1 = ((dot0.x + dot0.y) + dot0.z) + dot0.w;
    const float rcp_det = 1.0f / dot1;
    for (int i = 0; i < 4; i++) inverse.c[i] = v4_scale(inverse.c[i], rcp_det);
    return inverse;
}
inline Vec3 mat4_transform_point3(const Mat4& m, Vec3 p) {
    Vec4 res = v4_scale(m.c[0], p.x);
    res = v4_add(v4_scale(m.c[1], p.y), res);
    res = v4_add(v4_scale(m.c[2], p.z), res);
    res = v4_add(m.c[3], res);
    return {res.x, res.y, res.z};
}
inline Vec3 v3_normalize(Vec3 v) {
    const float len = std::sqrt((v.x * v.x + v.y * v.y) + v.z * v.z);
    return {v.x / len, v.y / len, v.z / len};
}
inline float v3_dot(Vec3 a, Vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }

}  // namespace awsm_host
