// device_math.hpp — gfx950 device helpers for the awsm-renderer hot path.
//
// Hand-written for the HIP kernels (not shared with oracle/).  The op ORDER of every function that
// feeds the visibility key or the f16-quantised G-buffer values is part of the arithmetic contract
// (DESIGN.md §"Arithmetic contract") and is kept identical to what the reference's WGSL expresses:
//   crates/renderer/src/render_passes/shared/shared_wgsl/math.wgsl:1-121
// The whole library is compiled with -ffp-contract=off (no FMA contraction), IEEE div/sqrt.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

namespace awsm {

struct f2 { float x, y; };
struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };
struct m4 { f4 c[4]; };   // column-major
struct m3 { f3 c[3]; };

#define AWSM_DI __device__ __forceinline__

constexpr float kPi = 3.1415926535897932384626433832795f;
constexpr float kTau = 6.283185307179586476925286766559f;
constexpr float kEps = 1e-4f;

AWSM_DI f2 mk2(float x, float y) { return {x, y}; }
AWSM_DI f3 mk3(float x, float y, float z) { return {x, y, z}; }
AWSM_DI f4 mk4(float x, float y, float z, float w) { return {x, y, z, w}; }
AWSM_DI f3 splat3(float s) { return {s, s, s}; }
AWSM_DI f3 operator+(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
AWSM_DI f3 operator-(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
AWSM_DI f3 operator*(f3 a, f3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
AWSM_DI f3 operator*(f3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
AWSM_DI f3 operator/(f3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
AWSM_DI f3 operator-(f3 a) { return {-a.x, -a.y, -a.z}; }
AWSM_DI float dot(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
AWSM_DI f3 cross(f3 a, f3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
AWSM_DI float length(f3 a) { return sqrtf(dot(a, a)); }
AWSM_DI f3 normalize(f3 a) { const float inv = 1.0f / length(a); return {a.x * inv, a.y * inv, a.z * inv}; }   // contract: one IEEE reciprocal, three products
AWSM_DI float mixf(float a, float b, float t) { return a * (1.0f - t) + b * t; }
AWSM_DI f3 mix3(f3 a, f3 b, float t) { float s = 1.0f - t; return {a.x * s + b.x * t, a.y * s + b.y * t, a.z * s + b.z * t}; }
AWSM_DI f3 min3(f3 a, f3 b) { return {fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)}; }
AWSM_DI float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
AWSM_DI float saturate(float x) { return clampf(x, 0.0f, 1.0f); }
AWSM_DI float inverse_sqrt(float x) { return 1.0f / sqrtf(x); }
AWSM_DI float signf(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }

AWSM_DI f4 mul(const m4& m, f4 v) {
    f4 r;
    r.x = ((m.c[0].x * v.x + m.c[1].x * v.y) + m.c[2].x * v.z) + m.c[3].x * v.w;
    r.y = ((m.c[0].y * v.x + m.c[1].y * v.y) + m.c[2].y * v.z) + m.c[3].y * v.w;
    r.z = ((m.c[0].z * v.x + m.c[1].z * v.y) + m.c[2].z * v.z) + m.c[3].z * v.w;
    r.w = ((m.c[0].w * v.x + m.c[1].w * v.y) + m.c[2].w * v.z) + m.c[3].w * v.w;
    return r;
}
AWSM_DI f3 mul(const m3& m, f3 v) {
    f3 r;
    r.x = (m.c[0].x * v.x + m.c[1].x * v.y) + m.c[2].x * v.z;
    r.y = (m.c[0].y * v.x + m.c[1].y * v.y) + m.c[2].y * v.z;
    r.z = (m.c[0].z * v.x + m.c[1].z * v.y) + m.c[2].z * v.z;
    return r;
}
AWSM_DI m3 upper3(const m4& m) {
    m3 r;
    r.c[0] = {m.c[0].x, m.c[0].y, m.c[0].z};
    r.c[1] = {m.c[1].x, m.c[1].y, m.c[1].z};
    r.c[2] = {m.c[2].x, m.c[2].y, m.c[2].z};
    return r;
}
AWSM_DI m4 load_m4(const float* p) {
    const float4* q = reinterpret_cast<const float4*>(p);   // every mat4 in the reference's buffers is 16-B aligned
    m4 m;
#pragma unroll
    for (int i = 0; i < 4; i++) { float4 v = q[i]; m.c[i] = {v.x, v.y, v.z, v.w}; }
    return m;
}

// textureStore to rgba16float / rg16float rounds to nearest even (v_cvt_f16_f32, default rounding mode)
AWSM_DI float round_f16(float x) { return __half2float(__float2half_rn(x)); }
AWSM_DI unsigned short f16_bits(float x) { return __half_as_ushort(__float2half_rn(x)); }
AWSM_DI float f16_bits_to_f32(unsigned short h) { return __half2float(__ushort_as_half(h)); }

// Fixed-algorithm atan2 (G-buffer tangent angle).  Same polynomial and unfolding as the contract states;
// must stay free of fma contraction and fast-math.
AWSM_DI float det_atan2f(float y, float x) {
    float ax = fabsf(x), ay = fabsf(y);
    float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    if (mx == 0.0f) return 0.0f;
    float a = mn / mx;
    float s = a * a;
    float r = 0.0027856871f;
    r = r * s - 0.0158660002f;
    r = r * s + 0.0424557589f;
    r = r * s - 0.0749753043f;
    r = r * s + 0.106448799f;
    r = r * s - 0.142070308f;
    r = r * s + 0.199934542f;
    r = r * s - 0.333331466f;
    r = r * s;
    r = r * a + a;
    if (ay > ax) r = 1.57079637f - r;
    if (x < 0.0f) r = 3.14159274f - r;
    if (y < 0.0f) r = -r;
    return r;
}

AWSM_DI f3 safe_normalize(f3 n) {            // math.wgsl:21-28
    float len_sq = dot(n, n);
    if (len_sq > 0.0f) return n * inverse_sqrt(len_sq);
    return {0.0f, 0.0f, 1.0f};
}
AWSM_DI float inverse_square(float range, float dist) {   // math.wgsl:12-19
    if (range == 0.0f) return 1.0f / fmaxf(dist * dist, 0.01f);
    float denom = dist * dist + 1.0f;
    float falloff = 1.0f - (dist * dist) / (range * range);
    return saturate(falloff * falloff) / denom;
}
AWSM_DI f2 encode_octahedral(f3 n_in) {      // math.wgsl:44-53
    const float inv = 1.0f / ((fabsf(n_in.x) + fabsf(n_in.y)) + fabsf(n_in.z));
    f3 n = {n_in.x * inv, n_in.y * inv, n_in.z * inv};
    if (n.z < 0.0f) {
        float wx = (1.0f - fabsf(n.y)) * signf(n.x);
        float wy = (1.0f - fabsf(n.x)) * signf(n.y);
        n.x = wx; n.y = wy;
    }
    return {n.x * 0.5f + 0.5f, n.y * 0.5f + 0.5f};
}
AWSM_DI f3 decode_octahedral(f2 e) {         // math.wgsl:55-67
    float fx = e.x * 2.0f - 1.0f, fy = e.y * 2.0f - 1.0f;
    f3 n = {fx, fy, (1.0f - fabsf(fx)) - fabsf(fy)};
    float t = clampf(-n.z, 0.0f, 1.0f);
    float vx = (n.x >= 0.0f) ? -t : t;
    float vy = (n.y >= 0.0f) ? -t : t;
    n = {n.x + vx, n.y + vy, n.z};
    return normalize(n);
}
struct TB { f3 t, b; };
AWSM_DI TB canonical_tb(f3 n) {              // math.wgsl:73-84
    TB r;
    if (n.z < -0.9999999f) {
        r.t = {0.0f, -1.0f, 0.0f};
        r.b = {-1.0f, 0.0f, 0.0f};
    } else {
        float a = 1.0f / (1.0f + n.z);
        float bb = (-n.x * n.y) * a;
        r.t = {1.0f - (n.x * n.x) * a, bb, -n.x};
        r.b = {bb, 1.0f - (n.y * n.y) * a, -n.y};
    }
    return r;
}
AWSM_DI f4 pack_normal_tangent(f3 N, f3 T, float s) {   // math.wgsl:93-102
    f2 oct = encode_octahedral(N);
    TB tb = canonical_tb(N);
    float x = dot(T, tb.t);
    float y = dot(T, tb.b);
    float theta = det_atan2f(y, x);
    float angle_u = (theta + kPi) / kTau;
    float sign_u = (s > 0.0f) ? 1.0f : 0.0f;
    return {oct.x, oct.y, angle_u, sign_u};
}
struct TBN { f3 N, T, B; };
AWSM_DI TBN unpack_normal_tangent(f4 rgba) {  // math.wgsl:104-116
    TBN r;
    r.N = decode_octahedral({rgba.x, rgba.y});
    float theta = rgba.z * kTau - kPi;
    float s = (rgba.w >= 0.5f) ? 1.0f : -1.0f;
    TB tb0 = canonical_tb(r.N);
    float c = cosf(theta), sn = sinf(theta);
    r.T = normalize(tb0.t * c + tb0.b * sn);
    r.B = normalize(cross(r.N, r.T)) * s;
    return r;
}

// Texture coordinates follow the arithmetic contract up to the texel address even though the rest of the shading is
// relaxed: with a nearest filter a coordinate that lands exactly on a texel boundary (common once vertices are snapped to
// the sub-pixel grid and UVs are simple fractions) must pick the same texel as the oracle.
AWSM_DI float interp3_strict(float b0, float b1, float b2, float x0, float x1, float x2) { return (b0 * x0 + b1 * x1) + b2 * x2; }
AWSM_DI float affine2_strict(float a, float b, float c, float x, float y) { return (a * x + b * y) + c; }

// Blend of the transparent pass into its RGBA16F target (material_transparent/pipeline.rs:96-110: One / OneMinusSrcAlpha,
// colour and alpha): the stored f16 value is read back, the blend is f32 without contraction, the store rounds to f16.
AWSM_DI float blend_over_f16(float src, float dst, float one_minus_a) { return round_f16(src + dst * one_minus_a); }
AWSM_DI float resolve4_f16(float s0, float s1, float s2, float s3) { return round_f16((((s0 + s1) + s2) + s3) * 0.25f); }

}  // namespace awsm
