/*
 * oracle_math.h — TEST INFRASTRUCTURE ONLY (parity oracle).  Scalar C restatement of the small math
 * helpers of the reference's WGSL.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may use anything under oracle/.
 *
 * Follows (paths relative to /root/reference/crates/renderer/src/render_passes/):
 *   shared/shared_wgsl/math.wgsl:1-121        constants, saturate, inverse_square, safe_normalize,
 *                                             join32/split16, octahedral, canonical_tb, pack/unpack TBN
 *   shared/shared_wgsl/color_space.wgsl:1-13  (unused by the opaque pass; kept for texture prep)
 *
 * Arithmetic contract shared with the HIP kernels (DESIGN.md §"Arithmetic contract"):
 *   - IEEE-754 binary32, round-to-nearest-even, NO fused multiply-add contraction
 *     (compile with -ffp-contract=off), division and sqrt correctly rounded.
 *   - dot(a,b)   = (a.x*b.x + a.y*b.y) + a.z*b.z            (left to right)
 *   - M * v      = ((c0*v.x + c1*v.y) + c2*v.z) + c3*v.w    (WGSL: sum of scaled columns)
 *   - normalize(v) = v / sqrt(dot(v,v));  inverseSqrt(x) = 1 / sqrt(x)
 *   - mix(a,b,t) = a*(1-t) + b*t                            (WGSL definition)
 *   - atan2 on the G-buffer path is the fixed polynomial det_atan2f below (WGSL leaves atan2's
 *     precision to the implementation; a fixed algorithm makes the f16-quantised tangent angle
 *     bit-identical on CPU and GPU).
 *   - f32 -> f16 is round-to-nearest-even (what textureStore to rgba16float does).
 */
#ifndef ORACLE_MATH_H
#define ORACLE_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

typedef struct { float x, y; } ovec2;
typedef struct { float x, y, z; } ovec3;
typedef struct { float x, y, z, w; } ovec4;
typedef struct { ovec4 c[4]; } omat4;   /* column-major, like WGSL mat4x4<f32> */
typedef struct { ovec3 c[3]; } omat3;

#define O_PI 3.1415926535897932384626433832795f
#define O_TAU 6.283185307179586476925286766559f
#define O_EPSILON 1e-4f
#define O_U32_MAX 4294967295u

static inline ovec2 ov2(float x, float y) { ovec2 r = {x, y}; return r; }
static inline ovec3 ov3(float x, float y, float z) { ovec3 r = {x, y, z}; return r; }
static inline ovec4 ov4(float x, float y, float z, float w) { ovec4 r = {x, y, z, w}; return r; }
static inline ovec3 ov3s(float s) { return ov3(s, s, s); }

static inline ovec3 ov3_add(ovec3 a, ovec3 b) { return ov3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline ovec3 ov3_sub(ovec3 a, ovec3 b) { return ov3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline ovec3 ov3_mul(ovec3 a, ovec3 b) { return ov3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline ovec3 ov3_scale(ovec3 a, float s) { return ov3(a.x * s, a.y * s, a.z * s); }
static inline ovec3 ov3_div(ovec3 a, float s) { return ov3(a.x / s, a.y / s, a.z / s); }
static inline ovec3 ov3_neg(ovec3 a) { return ov3(-a.x, -a.y, -a.z); }
static inline float ov3_dot(ovec3 a, ovec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline ovec3 ov3_cross(ovec3 a, ovec3 b) {
    return ov3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float ov3_length(ovec3 a) { return sqrtf(ov3_dot(a, a)); }
/* contract: normalize(v) = v * (1 / sqrt(dot(v,v))) — one IEEE reciprocal, three products (WGSL leaves normalize's precision open) */
static inline ovec3 ov3_normalize(ovec3 a) { float inv = 1.0f / ov3_length(a); return ov3(a.x * inv, a.y * inv, a.z * inv); }
static inline ovec3 ov3_mix(ovec3 a, ovec3 b, float t) {
    float s = 1.0f - t;
    return ov3(a.x * s + b.x * t, a.y * s + b.y * t, a.z * s + b.z * t);
}
static inline ovec3 ov3_min(ovec3 a, ovec3 b) { return ov3(fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)); }
static inline float o_mix(float a, float b, float t) { return a * (1.0f - t) + b * t; }
static inline float o_clamp(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
static inline float o_saturate(float x) { return o_clamp(x, 0.0f, 1.0f); }
static inline float o_inverse_sqrt(float x) { return 1.0f / sqrtf(x); }
static inline float o_sign(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }

static inline ovec4 omat4_mul_v4(const omat4* m, ovec4 v) {
    ovec4 r;
    r.x = ((m->c[0].x * v.x + m->c[1].x * v.y) + m->c[2].x * v.z) + m->c[3].x * v.w;
    r.y = ((m->c[0].y * v.x + m->c[1].y * v.y) + m->c[2].y * v.z) + m->c[3].y * v.w;
    r.z = ((m->c[0].z * v.x + m->c[1].z * v.y) + m->c[2].z * v.z) + m->c[3].z * v.w;
    r.w = ((m->c[0].w * v.x + m->c[1].w * v.y) + m->c[2].w * v.z) + m->c[3].w * v.w;
    return r;
}
static inline ovec3 omat3_mul_v3(const omat3* m, ovec3 v) {
    ovec3 r;
    r.x = (m->c[0].x * v.x + m->c[1].x * v.y) + m->c[2].x * v.z;
    r.y = (m->c[0].y * v.x + m->c[1].y * v.y) + m->c[2].y * v.z;
    r.z = (m->c[0].z * v.x + m->c[1].z * v.y) + m->c[2].z * v.z;
    return r;
}
static inline omat3 omat3_from_mat4(const omat4* m) {
    omat3 r;
    r.c[0] = ov3(m->c[0].x, m->c[0].y, m->c[0].z);
    r.c[1] = ov3(m->c[1].x, m->c[1].y, m->c[1].z);
    r.c[2] = ov3(m->c[2].x, m->c[2].y, m->c[2].z);
    return r;
}
static inline omat4 omat4_load(const float* p) {
    omat4 m;
    for (int i = 0; i < 4; i++) m.c[i] = ov4(p[4 * i], p[4 * i + 1], p[4 * i + 2], p[4 * i + 3]);
    return m;
}

static inline uint32_t o_f32_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float o_bits_f32(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* f32 -> f16, round-to-nearest-even, IEEE (denormals kept, overflow -> inf). */
static inline uint16_t o_f32_to_f16(float f) {
    uint32_t x = o_f32_bits(f);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t ax = x & 0x7FFFFFFFu;
    if (ax >= 0x7F800000u) {                       /* inf / nan */
        return (uint16_t)(sign | 0x7C00u | ((ax > 0x7F800000u) ? 0x0200u : 0u));
    }
    if (ax >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u); /* >= 65520 rounds to inf */
    if (ax < 0x33000001u) return (uint16_t)sign;  /* <= 2^-25 rounds to zero (ties-to-even at exactly 2^-25) */
    int32_t e = (int32_t)(ax >> 23) - 127;
    uint32_t m = (ax & 0x007FFFFFu) | 0x00800000u;
    uint32_t shift, half_bits;
    if (e < -14) {                                  /* f16 subnormal */
        shift = (uint32_t)(13 + (-14 - e));
        half_bits = 0;
    } else {
        shift = 13;
        half_bits = (uint32_t)(e + 15) << 10;
        m &= 0x007FFFFFu;
    }
    uint32_t q = m >> shift;
    uint32_t rem = m & ((1u << shift) - 1u);
    uint32_t halfway = 1u << (shift - 1);
    uint32_t r = half_bits + q;                     /* mantissa carry propagates into the exponent */
    if (rem > halfway || (rem == halfway && (r & 1u))) r += 1u;
    return (uint16_t)(sign | r);
}
static inline float o_f16_to_f32(uint16_t h) {
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1Fu;
    uint32_t m = h & 0x3FFu;
    if (e == 0) {
        if (m == 0) return o_bits_f32(sign);
        float f = (float)m * (1.0f / 16777216.0f);  /* m * 2^-24, exact */
        return (sign ? -f : f);
    }
    if (e == 31) return o_bits_f32(sign | 0x7F800000u | (m << 13));
    return o_bits_f32(sign | ((e + 112u) << 23) | (m << 13));
}
static inline float o_round_f16(float f) { return o_f16_to_f32(o_f32_to_f16(f)); }

/* Fixed-algorithm atan2 (see header comment).  atan on [0,1] by an odd minimax polynomial, then
 * octant unfolding.  Max error vs libm ~2 ulp; what matters is that CPU and GPU agree bit for bit. */
static inline float det_atan2f(float y, float x) {
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

/* math.wgsl:12-19 */
static inline float o_inverse_square(float range, float dist) {
    if (range == 0.0f) return 1.0f / fmaxf(dist * dist, 0.01f);
    float denom = dist * dist + 1.0f;
    float falloff = 1.0f - (dist * dist) / (range * range);
    return o_saturate(falloff * falloff) / denom;
}
/* math.wgsl:21-28 */
static inline ovec3 o_safe_normalize(ovec3 n) {
    float len_sq = ov3_dot(n, n);
    if (len_sq > 0.0f) return ov3_scale(n, o_inverse_sqrt(len_sq));
    return ov3(0.0f, 0.0f, 1.0f);
}
/* math.wgsl:30-38 */
static inline uint32_t o_join32(uint32_t lo, uint32_t hi) { return (hi << 16) | (lo & 0xFFFFu); }

/* math.wgsl:44-53 */
static inline ovec2 o_encode_octahedral(ovec3 n_in) {
    float inv = 1.0f / ((fabsf(n_in.x) + fabsf(n_in.y)) + fabsf(n_in.z));
    ovec3 n = ov3(n_in.x * inv, n_in.y * inv, n_in.z * inv);
    if (n.z < 0.0f) {
        float wx = (1.0f - fabsf(n.y)) * o_sign(n.x);
        float wy = (1.0f - fabsf(n.x)) * o_sign(n.y);
        n.x = wx; n.y = wy;
    }
    return ov2(n.x * 0.5f + 0.5f, n.y * 0.5f + 0.5f);
}
/* math.wgsl:55-67 */
static inline ovec3 o_decode_octahedral(ovec2 e) {
    float fx = e.x * 2.0f - 1.0f, fy = e.y * 2.0f - 1.0f;
    ovec3 n = ov3(fx, fy, (1.0f - fabsf(fx)) - fabsf(fy));
    float t = o_clamp(-n.z, 0.0f, 1.0f);
    float vx = (n.x >= 0.0f) ? -t : t;
    float vy = (n.y >= 0.0f) ? -t : t;
    n = ov3(n.x + vx, n.y + vy, n.z);
    return ov3_normalize(n);
}
/* math.wgsl:73-84 */
typedef struct { ovec3 t, b; } o_tb;
static inline o_tb o_canonical_tb(ovec3 n) {
    o_tb r;
    if (n.z < -0.9999999f) {
        r.t = ov3(0.0f, -1.0f, 0.0f);
        r.b = ov3(-1.0f, 0.0f, 0.0f);
    } else {
        float a = 1.0f / (1.0f + n.z);
        float bb = (-n.x * n.y) * a;
        r.t = ov3(1.0f - (n.x * n.x) * a, bb, -n.x);
        r.b = ov3(bb, 1.0f - (n.y * n.y) * a, -n.y);
    }
    return r;
}
/* math.wgsl:93-102 */
static inline ovec4 o_pack_normal_tangent(ovec3 N, ovec3 T, float s) {
    ovec2 oct = o_encode_octahedral(N);
    o_tb tb = o_canonical_tb(N);
    float x = ov3_dot(T, tb.t);
    float y = ov3_dot(T, tb.b);
    float theta = det_atan2f(y, x);
    float angle_u = (theta + O_PI) / O_TAU;
    float sign_u = (s > 0.0f) ? 1.0f : 0.0f;
    return ov4(oct.x, oct.y, angle_u, sign_u);
}
/* math.wgsl:104-116 */
typedef struct { ovec3 N, T, B; } o_tbn;
static inline o_tbn o_unpack_normal_tangent(ovec4 rgba) {
    o_tbn r;
    r.N = o_decode_octahedral(ov2(rgba.x, rgba.y));
    float theta = rgba.z * O_TAU - O_PI;
    float s = (rgba.w >= 0.5f) ? 1.0f : -1.0f;
    o_tb tb0 = o_canonical_tb(r.N);
    float c = cosf(theta), sn = sinf(theta);
    r.T = ov3_normalize(ov3_add(ov3_scale(tb0.t, c), ov3_scale(tb0.b, sn)));
    r.B = ov3_scale(ov3_normalize(ov3_cross(r.N, r.T)), s);
    return r;
}
/* math.wgsl:118-121 */
static inline uint32_t o_abs_index(uint32_t base_index, uint32_t relative_index) {
    return relative_index != 0u ? base_index + relative_index : 0u;
}

#endif /* ORACLE_MATH_H */
