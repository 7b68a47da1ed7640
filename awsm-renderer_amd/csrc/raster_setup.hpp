// raster_setup.hpp — triangle setup + per-pixel coverage for the software rasteriser (device side).
//
// Replaces WebGPU's fixed-function rasteriser as configured by
//   crates/renderer/src/render_passes/geometry/pipeline.rs:337-344
//   (TriangleList, FrontFace::Ccw, CullMode::{None,Back}, depth write, CompareFunction::LessEqual).
// The raster contract (DESIGN.md §"Raster contract"): homogeneous edge functions in f32, evaluated at
// pixel centres, top-left rule, per-pixel 0 <= z_ndc <= 1 clip, z_ndc = (e0*z0 + e1*z1 + e2*z2) * (1/det) with the
// reciprocal taken once per triangle (IEEE division) — one multiply per sample instead of an 11-instruction division.
// Used by the binning, raster and shade kernels so that all three see bit-identical edge values.
#pragma once
#include "device_math.hpp"

namespace awsm {

struct TriSetup {
    float a[3], b[3], c[3];   // e_i(X,Y) = (a*X + b*Y) + c ; inside >= 0 ; e_i is the weight of vertex i
    float z[3];
    float det;                // > 0 after orientation normalisation
    float inv_det;            // 1 / det (IEEE), the factor of the per-sample depth
    int minx, maxx, miny, maxy;   // inclusive, conservative, clamped to the target rect
};

AWSM_DI bool finite4(float4 v) { return isfinite(v.x) && isfinite(v.y) && isfinite(v.z) && isfinite(v.w); }

// Edge-function coefficients, orientation normalisation and det (the part of the setup that per-pixel sampling needs).
// Returns false for a degenerate or culled triangle.  X*/Y* are the homogeneous screen coordinates, for the bbox.
AWSM_DI bool tri_coefficients(float4 v0, float4 v1, float4 v2, bool cull_back, uint32_t width, uint32_t height, TriSetup& t,
                              float& X0, float& Y0, float& X1, float& Y1, float& X2, float& Y2) {
    float hw = 0.5f * (float)width, hh = 0.5f * (float)height;
    const float w0 = v0.w, w1 = v1.w, w2 = v2.w;
    X0 = (v0.x + v0.w) * hw; Y0 = (v0.w - v0.y) * hh;
    X1 = (v1.x + v1.w) * hw; Y1 = (v1.w - v1.y) * hh;
    X2 = (v2.x + v2.w) * hw; Y2 = (v2.w - v2.y) * hh;

    float a0 = Y1 * w2 - Y2 * w1, b0 = X2 * w1 - X1 * w2, c0 = X1 * Y2 - X2 * Y1;
    float a1 = Y2 * w0 - Y0 * w2, b1 = X0 * w2 - X2 * w0, c1 = X2 * Y0 - X0 * Y2;
    float a2 = Y0 * w1 - Y1 * w0, b2 = X1 * w0 - X0 * w1, c2 = X0 * Y1 - X1 * Y0;
    float det = (X0 * a0 + Y0 * b0) + w0 * c0;
    if (!(det != 0.0f) || !isfinite(det)) return false;
    if (cull_back && det > 0.0f) return false;      // y-down framebuffer: det < 0 <=> CCW on screen <=> front
    if (det < 0.0f) {
        a0 = -a0; b0 = -b0; c0 = -c0; a1 = -a1; b1 = -b1; c1 = -c1; a2 = -a2; b2 = -b2; c2 = -c2;
        det = -det;
    }
    t.a[0] = a0; t.b[0] = b0; t.c[0] = c0;
    t.a[1] = a1; t.b[1] = b1; t.c[1] = c1;
    t.a[2] = a2; t.b[2] = b2; t.c[2] = c2;
    t.z[0] = v0.z; t.z[1] = v1.z; t.z[2] = v2.z;
    t.det = det;
    t.inv_det = 1.0f / det;
    return true;
}

// Returns false if the triangle cannot produce a fragment (culled, degenerate, outside, empty bbox).
AWSM_DI bool tri_setup(float4 v0, float4 v1, float4 v2, bool cull_back, uint32_t width, uint32_t height,
                       uint32_t ry0, uint32_t ry1, TriSetup& t) {
    if (!finite4(v0) || !finite4(v1) || !finite4(v2)) return false;
    if (v0.x < -v0.w && v1.x < -v1.w && v2.x < -v2.w) return false;
    if (v0.x > v0.w && v1.x > v1.w && v2.x > v2.w) return false;
    if (v0.y < -v0.w && v1.y < -v1.w && v2.y < -v2.w) return false;
    if (v0.y > v0.w && v1.y > v1.w && v2.y > v2.w) return false;
    if (v0.z < 0.0f && v1.z < 0.0f && v2.z < 0.0f) return false;
    if (v0.z > v0.w && v1.z > v1.w && v2.z > v2.w) return false;

    float X0, Y0, X1, Y1, X2, Y2;
    if (!tri_coefficients(v0, v1, v2, cull_back, width, height, t, X0, Y0, X1, Y1, X2, Y2)) return false;
    const float w0 = v0.w, w1 = v1.w, w2 = v2.w;

    int minx = 0, maxx = (int)width - 1, miny = (int)ry0, maxy = (int)ry1 - 1;
    if (w0 > 0.0f && w1 > 0.0f && w2 > 0.0f) {
        float sx0 = X0 / w0, sx1 = X1 / w1, sx2 = X2 / w2;
        float sy0 = Y0 / w0, sy1 = Y1 / w1, sy2 = Y2 / w2;
        float fminx = fminf(fminf(sx0, sx1), sx2), fmaxx = fmaxf(fmaxf(sx0, sx1), sx2);
        float fminy = fminf(fminf(sy0, sy1), sy2), fmaxy = fmaxf(fmaxf(sy0, sy1), sy2);
        fminx = fminf(fmaxf(fminx, -16777216.0f), 16777216.0f);
        fmaxx = fminf(fmaxf(fmaxx, -16777216.0f), 16777216.0f);
        fminy = fminf(fmaxf(fminy, -16777216.0f), 16777216.0f);
        fmaxy = fminf(fmaxf(fmaxy, -16777216.0f), 16777216.0f);
        int bx0 = (int)floorf(fminx) - 1, bx1 = (int)floorf(fmaxx) + 1;
        int by0 = (int)floorf(fminy) - 1, by1 = (int)floorf(fmaxy) + 1;
        minx = max(minx, bx0); maxx = min(maxx, bx1);
        miny = max(miny, by0); maxy = min(maxy, by1);
    }
    if (minx > maxx || miny > maxy) return false;
    t.minx = minx; t.maxx = maxx; t.miny = miny; t.maxy = maxy;
    return true;
}

// Per-triangle setup record, written once per frame by the counting pass of the binner and read by the fill pass, the
// raster kernel and the shade kernel (which would otherwise each redo the setup: ~250 VALU incl. six IEEE divisions).
// Same bits everywhere by construction.  64 bytes, 16-byte aligned.
struct alignas(16) TriRec {
    float a[3], b[3], c[3];
    float z[3];
    float inv_det;
    uint32_t bbox_x;   // minx | maxx << 16   (inclusive, clamped to the target rect)
    uint32_t bbox_y;   // miny | maxy << 16
    uint32_t valid;    // 0: the triangle cannot produce a fragment in this shard
};
static_assert(sizeof(TriRec) == 64, "TriRec must be 64 bytes");

AWSM_DI void tri_rec_store(TriRec* __restrict__ dst, const TriSetup& t, bool ok) {
    float4* q = reinterpret_cast<float4*>(dst);
    q[0] = make_float4(t.a[0], t.a[1], t.a[2], t.b[0]);
    q[1] = make_float4(t.b[1], t.b[2], t.c[0], t.c[1]);
    q[2] = make_float4(t.c[2], t.z[0], t.z[1], t.z[2]);
    q[3] = make_float4(t.inv_det, __uint_as_float((uint32_t)t.minx | ((uint32_t)t.maxx << 16)), __uint_as_float((uint32_t)t.miny | ((uint32_t)t.maxy << 16)),
                       __uint_as_float(ok ? 1u : 0u));
}
AWSM_DI bool tri_rec_load(const TriRec* __restrict__ src, TriSetup& t) {
    const float4* q = reinterpret_cast<const float4*>(src);
    const float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
    t.a[0] = q0.x; t.a[1] = q0.y; t.a[2] = q0.z; t.b[0] = q0.w;
    t.b[1] = q1.x; t.b[2] = q1.y; t.c[0] = q1.z; t.c[1] = q1.w;
    t.c[2] = q2.x; t.z[0] = q2.y; t.z[1] = q2.z; t.z[2] = q2.w;
    t.inv_det = q3.x;
    const uint32_t bx = __float_as_uint(q3.y), by = __float_as_uint(q3.z);
    t.minx = (int)(bx & 0xFFFFu); t.maxx = (int)(bx >> 16); t.miny = (int)(by & 0xFFFFu); t.maxy = (int)(by >> 16);
    return __float_as_uint(q3.w) != 0u;
}
// the edge coefficients only (shade kernel)
AWSM_DI void tri_rec_load_edges(const TriRec* __restrict__ src, TriSetup& t) {
    const float4* q = reinterpret_cast<const float4*>(src);
    const float4 q0 = q[0], q1 = q[1];
    const float c2 = src->c[2];
    t.a[0] = q0.x; t.a[1] = q0.y; t.a[2] = q0.z; t.b[0] = q0.w;
    t.b[1] = q1.x; t.b[2] = q1.y; t.c[0] = q1.z; t.c[1] = q1.w;
    t.c[2] = c2;
}

AWSM_DI bool edge_inside(float e, float a, float b) {
    return e > 0.0f || (e == 0.0f && (a > 0.0f || (a == 0.0f && b > 0.0f)));   // top-left rule
}

AWSM_DI void tri_edges_at(const TriSetup& t, float X, float Y, float& e0, float& e1, float& e2) {
    e0 = (t.a[0] * X + t.b[0] * Y) + t.c[0];
    e1 = (t.a[1] * X + t.b[1] * Y) + t.c[1];
    e2 = (t.a[2] * X + t.b[2] * Y) + t.c[2];
}
AWSM_DI void tri_edges(const TriSetup& t, int px, int py, float& e0, float& e1, float& e2) {
    tri_edges_at(t, (float)px + 0.5f, (float)py + 0.5f, e0, e1, e2);
}

// WebGPU's standard 4x sample pattern (GPUMultisampleState count = 4; the D3D standard pattern), pixel-relative.
// Exactly representable, so px + offset is exact in f32 for any frame size.
AWSM_DI float msaa4_x(int k) { return k == 0 ? 0.375f : (k == 1 ? 0.875f : (k == 2 ? 0.125f : 0.625f)); }
AWSM_DI float msaa4_y(int k) { return k == 0 ? 0.125f : (k == 1 ? 0.375f : (k == 2 ? 0.625f : 0.875f)); }

// Coverage + depth at sample position (X, Y) in pixel units.  Returns the packed 64-bit key or ~0 if not covered.
AWSM_DI unsigned long long tri_sample_key_at(const TriSetup& t, float X, float Y, uint32_t rank) {
    float e0, e1, e2;
    tri_edges_at(t, X, Y, e0, e1, e2);
    if (!edge_inside(e0, t.a[0], t.b[0]) || !edge_inside(e1, t.a[1], t.b[1]) || !edge_inside(e2, t.a[2], t.b[2]))
        return ~0ull;
    float zn = ((e0 * t.z[0] + e1 * t.z[1]) + e2 * t.z[2]) * t.inv_det;
    if (!(zn >= 0.0f && zn <= 1.0f)) return ~0ull;
    if (zn == 0.0f) zn = 0.0f;   // -0 -> +0 so the bits order as an unsigned integer
    // depth LessEqual + submission order: smaller depth wins, equal depth -> LATER primitive wins
    return ((unsigned long long)__float_as_uint(zn) << 32) | (unsigned long long)(0xFFFFFFFFu - rank);
}
// Pixel centre (single-sampled targets).
AWSM_DI unsigned long long tri_sample_key(const TriSetup& t, int px, int py, uint32_t rank) {
    return tri_sample_key_at(t, (float)px + 0.5f, (float)py + 0.5f, rank);
}

// Conservative "tile can contain a covered sample" test: evaluates every edge at the tile corner that maximises it.
// The corners are at least half a pixel (pixel centres) or an eighth of a pixel (MSAA sample positions) away from the
// outermost samples, which dwarfs the rounding error of the evaluation, so no covered sample is ever rejected.
AWSM_DI bool tile_may_overlap(const TriSetup& t, int tx0, int ty0, int tx1, int ty1 /* pixel bounds, exclusive max */) {
#pragma unroll
    for (int i = 0; i < 3; i++) {
        float X = t.a[i] > 0.0f ? (float)tx1 : (float)tx0;
        float Y = t.b[i] > 0.0f ? (float)ty1 : (float)ty0;
        float ax = t.a[i] * X, by = t.b[i] * Y;
        float e = (ax + by) + t.c[i];
        float slack = 4e-7f * ((fabsf(ax) + fabsf(by)) + fabsf(t.c[i]));   // > 3 ulp of the largest term
        if (e < -slack) return false;
    }
    return true;
}

}  // namespace awsm
