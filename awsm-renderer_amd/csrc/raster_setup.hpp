// raster_setup.hpp — triangle setup + per-pixel coverage for the software rasteriser (device side).
//
// Replaces WebGPU's fixed-function rasteriser as configured by
//   crates/renderer/src/render_passes/geometry/pipeline.rs:337-344
//   (TriangleList, FrontFace::Ccw, CullMode::{None,Back}, depth write, CompareFunction::LessEqual).
// The raster contract (DESIGN.md §"Raster contract"), two kinds of triangle setup:
//   kind 0 — w > 0 at all three vertices and inside a +-32768-pixel guard band: vertices projected and SNAPPED to a 1/256-pixel
//     grid; facing, edge functions and the top-left rule are exact (integers < 2^49, carried in f64 where every operation on
//     them is exact), as in a hardware rasteriser: shared edges are watertight and small distant triangles keep their depth;
//   kind 1 — triangles touching w <= 0 (crossing the near plane): homogeneous clip-less edge functions, f32 coefficients.
// Both kinds evaluate a sample with the same instructions: E_i = fma(a_i, X, fma(b_i, Y, c_i)) in f64 with X, Y in pixels
// (exact for kind 0), e_i = (float)E_i, depth = (e0*zq0 + e1*zq1) + e2*zq2 in f32, 0 <= depth <= 1 clip per sample (a sample of a
// multisampled target: that form at the pixel's corner plus the sample's step along the plane's gradient, msaa_depth_steps below);
// perspective-correct barycentrics for the opaque pass: u_i = e_i * iw_i, b_i = u_i * (1 / ((u0 + u1) + u2)).
// Used by the binning, raster and shade kernels so that all three see bit-identical edge values.
#pragma once
#include "device_math.hpp"

namespace awsm {

struct TriSetup {
    float a[3], b[3];         // E_i(X,Y) = a*X + b*Y + c, X/Y in pixels ; inside >= 0 ; E_i is the (screen-space) weight of vertex i
    double c[3];
    float zq[3];              // kind 0: (z_i / w_i) / |2 area| ; kind 1: z_i / det
    float iw[3];              // kind 0: 1 / w_i ; kind 1: 1
    int minx, maxx, miny, maxy;   // inclusive, conservative, clamped to the target rect
    bool front;                   // @builtin(front_facing): counter-clockwise in NDC (FrontFace::Ccw)
    bool exact;                   // kind 0: a, b, c and every E at a pixel centre are integers below 2^49 (E can be stepped by adding: still exact)
    bool small;                   // exact, and every |E| within 16 pixels of the triangle's bounding box is below 2^30: the walk may step in 32-bit integers
};

AWSM_DI bool finite4(float4 v) { return isfinite(v.x) && isfinite(v.y) && isfinite(v.z) && isfinite(v.w); }

constexpr float kSubPix = 256.0f;            // 8 fractional bits
constexpr float kGuardBand = 8388608.0f;     // |coordinate| * 256 <= 2^23

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

    const float hw = 0.5f * (float)width, hh = 0.5f * (float)height;
    int minx = 0, maxx = (int)width - 1, miny = (int)ry0, maxy = (int)ry1 - 1;

    bool snapped = v0.w > 0.0f && v1.w > 0.0f && v2.w > 0.0f;
    int x[3] = {0, 0, 0}, y[3] = {0, 0, 0};
    float iw[3] = {1.0f, 1.0f, 1.0f};
    const float4 v[3] = {v0, v1, v2};
    if (snapped) {
#pragma unroll
        for (int i = 0; i < 3; i++) {
            iw[i] = 1.0f / v[i].w;
            const float fx = ((v[i].x * iw[i] + 1.0f) * hw) * kSubPix;      // screen x, y-down screen y, in 1/256 pixel
            const float fy = ((1.0f - v[i].y * iw[i]) * hh) * kSubPix;
            if (!(fabsf(fx) <= kGuardBand && fabsf(fy) <= kGuardBand)) snapped = false;
            x[i] = (int)rintf(fx); y[i] = (int)rintf(fy);                     // round to nearest even; garbage if !snapped, unused
        }
    }
    if (snapped) {
        // 2*area; y-down screen: negative <=> counter-clockwise in NDC <=> front facing (FrontFace::Ccw)
        const long long A2 = (long long)(x[1] - x[0]) * (long long)(y[2] - y[0]) - (long long)(x[2] - x[0]) * (long long)(y[1] - y[0]);
        if (A2 == 0) return false;
        if (cull_back && A2 > 0) return false;
        const bool flip = A2 < 0;
        t.front = flip;
        t.exact = true;
        {   // E_i(P) = a_i (Px - x_j) + b_i (Py - y_j) in sub-pixel units, |a_i|, |b_i| <= S (the extent), |P - v_j| <= S + 4096 within 16 pixels of
            // the box (the walk's 8x8 blocks reach 7 pixels past it, and the stepped value one block further before the loop ends):
            // |E| <= 2 S (S + 4096) = 1.054e9 < 2^30 for S <= 21000 (82 pixels)
            const int ext = max(max(max(x[0], x[1]), x[2]) - min(min(x[0], x[1]), x[2]), max(max(y[0], y[1]), y[2]) - min(min(y[0], y[1]), y[2]));
            t.small = ext <= 21000;
        }
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const int j = (i + 1) % 3, k = (i + 2) % 3;                       // weight of vertex i = edge j -> k
            const int ai = y[j] - y[k], bi = x[k] - x[j];                     // |.| <= 2^24: exact in f32, and so is * 256
            const long long ci = (long long)x[j] * (long long)y[k] - (long long)x[k] * (long long)y[j];   // |.| < 2^48: exact in f64
            t.a[i] = (float)(flip ? -ai : ai) * kSubPix;                      // pixel units: a*256 * (Px/256)
            t.b[i] = (float)(flip ? -bi : bi) * kSubPix;
            t.c[i] = (double)(flip ? -ci : ci);
        }
        const float inv_area = 1.0f / (float)(double)(flip ? -A2 : A2);      // i64 -> f64 exact, one rounding to f32
#pragma unroll
        for (int i = 0; i < 3; i++) { t.zq[i] = (v[i].z * iw[i]) * inv_area; t.iw[i] = iw[i]; }
        // pixels that can hold a sample inside [min, max] of the snapped vertices
        const int mnx = min(min(x[0], x[1]), x[2]), mxx = max(max(x[0], x[1]), x[2]);
        const int mny = min(min(y[0], y[1]), y[2]), mxy = max(max(y[0], y[1]), y[2]);
        minx = max(minx, mnx >> 8); maxx = min(maxx, (mxx - 1) >> 8);        // arithmetic shifts: floor
        miny = max(miny, mny >> 8); maxy = min(maxy, (mxy - 1) >> 8);
    } else {
        const float w0 = v0.w, w1 = v1.w, w2 = v2.w;
        const float X0 = (v0.x + v0.w) * hw, Y0 = (v0.w - v0.y) * hh;
        const float X1 = (v1.x + v1.w) * hw, Y1 = (v1.w - v1.y) * hh;
        const float X2 = (v2.x + v2.w) * hw, Y2 = (v2.w - v2.y) * hh;
        float a0 = Y1 * w2 - Y2 * w1, b0 = X2 * w1 - X1 * w2, c0 = X1 * Y2 - X2 * Y1;
        float a1 = Y2 * w0 - Y0 * w2, b1 = X0 * w2 - X2 * w0, c1 = X2 * Y0 - X0 * Y2;
        float a2 = Y0 * w1 - Y1 * w0, b2 = X1 * w0 - X0 * w1, c2 = X0 * Y1 - X1 * Y0;
        float det = (X0 * a0 + Y0 * b0) + w0 * c0;
        if (!(det != 0.0f) || !isfinite(det)) return false;
        if (cull_back && det > 0.0f) return false;      // y-down framebuffer: det < 0 <=> CCW on screen <=> front
        t.front = det < 0.0f;
        t.exact = false; t.small = false;
        if (det < 0.0f) {
            a0 = -a0; b0 = -b0; c0 = -c0; a1 = -a1; b1 = -b1; c1 = -c1; a2 = -a2; b2 = -b2; c2 = -c2;
            det = -det;
        }
        t.a[0] = a0; t.b[0] = b0; t.c[0] = (double)c0;
        t.a[1] = a1; t.b[1] = b1; t.c[1] = (double)c1;
        t.a[2] = a2; t.b[2] = b2; t.c[2] = (double)c2;
        const float inv_det = 1.0f / det;
        t.zq[0] = v0.z * inv_det; t.zq[1] = v1.z * inv_det; t.zq[2] = v2.z * inv_det;
        t.iw[0] = 1.0f; t.iw[1] = 1.0f; t.iw[2] = 1.0f;
    }
    if (minx > maxx || miny > maxy) return false;
    t.minx = minx; t.maxx = maxx; t.miny = miny; t.maxy = maxy;
    return true;
}

// Per-triangle setup record, written once per frame by the counting pass of the binner and read by the fill pass, the
// raster kernel and the shade kernel (which would otherwise each redo the setup: ~250 VALU incl. IEEE divisions).
// Same bits everywhere by construction.  80 bytes, 16-byte aligned.  An empty bbox (minx > maxx) marks a triangle that
// cannot produce a fragment in this shard.
struct alignas(16) TriRec {
    float a[3], b[3];
    float zq[3];
    float iw[3];
    double c[3];
    uint32_t bbox_x;   // minx | small << 15 | maxx << 16 | exact << 31   (inclusive, clamped to the target rect; frame width <= 16384)
    uint32_t bbox_y;   // miny | maxy << 16 | front_facing << 31   (frame height <= 32768)
};
static_assert(sizeof(TriRec) == 80, "TriRec must be 80 bytes");

AWSM_DI void tri_rec_store(TriRec* __restrict__ dst, const TriSetup& t, bool ok) {
    float4* q = reinterpret_cast<float4*>(dst);
    const uint32_t bx = ok ? ((uint32_t)t.minx | (t.small ? 0x8000u : 0u) | ((uint32_t)t.maxx << 16) | (t.exact ? 0x80000000u : 0u)) : 1u;   // minx 1 > maxx 0
    const uint32_t by = ok ? ((uint32_t)t.miny | ((uint32_t)t.maxy << 16) | (t.front ? 0x80000000u : 0u)) : 1u;
    q[0] = make_float4(t.a[0], t.a[1], t.a[2], t.b[0]);
    q[1] = make_float4(t.b[1], t.b[2], t.zq[0], t.zq[1]);
    q[2] = make_float4(t.zq[2], t.iw[0], t.iw[1], t.iw[2]);
    double2* d = reinterpret_cast<double2*>(dst);
    d[3] = make_double2(t.c[0], t.c[1]);
    d[4] = make_double2(t.c[2], __hiloint2double((int)by, (int)bx));
}
AWSM_DI void tri_rec_store_invalid(TriRec* __restrict__ dst) {     // only the words tri_rec_unpack's validity test reads
    reinterpret_cast<double2*>(dst)[4] = make_double2(0.0, __hiloint2double(1, 1));
}
struct TriRecRaw { float4 q0, q1, q2; double2 d3, d4; };      // the record as loaded; lets a serial walk fetch the next one early
AWSM_DI TriRecRaw tri_rec_fetch(const TriRec* __restrict__ src) {
    const float4* q = reinterpret_cast<const float4*>(src);
    const double2* d = reinterpret_cast<const double2*>(src);
    TriRecRaw r;
    r.q0 = q[0]; r.q1 = q[1]; r.q2 = q[2]; r.d3 = d[3]; r.d4 = d[4];
    return r;
}
AWSM_DI bool tri_rec_unpack(const TriRecRaw& r, TriSetup& t) {
    t.a[0] = r.q0.x; t.a[1] = r.q0.y; t.a[2] = r.q0.z; t.b[0] = r.q0.w;
    t.b[1] = r.q1.x; t.b[2] = r.q1.y; t.zq[0] = r.q1.z; t.zq[1] = r.q1.w;
    t.zq[2] = r.q2.x; t.iw[0] = r.q2.y; t.iw[1] = r.q2.z; t.iw[2] = r.q2.w;
    t.c[0] = r.d3.x; t.c[1] = r.d3.y; t.c[2] = r.d4.x;
    const uint32_t bx = (uint32_t)__double2loint(r.d4.y), by = (uint32_t)__double2hiint(r.d4.y);
    t.minx = (int)(bx & 0x7FFFu); t.small = (bx & 0x8000u) != 0u; t.maxx = (int)((bx >> 16) & 0x7FFFu); t.exact = (bx >> 31) != 0u; t.miny = (int)(by & 0xFFFFu); t.maxy = (int)((by >> 16) & 0x7FFFu);
    t.front = (by >> 31) != 0u;
    return t.minx <= t.maxx;
}
AWSM_DI bool tri_rec_load(const TriRec* __restrict__ src, TriSetup& t) { return tri_rec_unpack(tri_rec_fetch(src), t); }

// Edge values at a sample given in 1/256-pixel units (pixel centre = px*256 + 128).
struct EdgeVals { double E[3]; };
AWSM_DI double sample_coord(int p_sub) { return (double)p_sub * 0.00390625; }    // exact
AWSM_DI EdgeVals tri_edges_d(const TriSetup& t, double X, double Y) {
    EdgeVals r;
#pragma unroll
    for (int i = 0; i < 3; i++) r.E[i] = fma((double)t.a[i], X, fma((double)t.b[i], Y, t.c[i]));
    return r;
}
AWSM_DI bool edge_inside(double e, float a, float b) {
    return e > 0.0 || (e == 0.0 && (a > 0.0f || (a == 0.0f && b > 0.0f)));   // top-left rule
}

// WebGPU's standard 4x sample pattern (GPUMultisampleState count = 4; the D3D standard pattern) in 1/256 pixel:
// (0.375, 0.125) (0.875, 0.375) (0.125, 0.625) (0.625, 0.875).
AWSM_DI int msaa4_x(int k) { return k == 0 ? 96 : (k == 1 ? 224 : (k == 2 ? 32 : 160)); }
AWSM_DI int msaa4_y(int k) { return k == 0 ? 32 : (k == 1 ? 96 : (k == 2 ? 160 : 224)); }

// The top-left rule as one comparison per edge: an edge that owns its zero line (a > 0, or a == 0 and b > 0) accepts E >= 0, the others
// E > 0.  E >= 0 is E > -(smallest f64 subnormal): nothing lies between that and zero (f64 subnormals are never flushed on gfx950), so
// edge_inside(E, a, b) == (E > edge_threshold(a, b)) for every E, exact integers (kind 0) and rounded values (kind 1) alike.
AWSM_DI double edge_threshold(float a, float b) {
    return (a > 0.0f || (a == 0.0f && b > 0.0f)) ? __longlong_as_double((long long)0x8000000000000001ull) : 0.0;
}

// "zn >= 0 && zn <= 1, and -0 is stored as +0" (the depth test of the key, as the oracle spells it) in two instructions instead of three comparisons and a
// select: -0 + 0 = +0 and x + 0 = x for every other x (round to nearest; no fast-math, so the addition stays), and the bit patterns of the floats in
// [+0, 1] are exactly the unsigned integers up to 0x3F800000 — negative numbers, numbers above 1, infinities and NaNs are all larger as unsigned integers.
AWSM_DI bool depth_key_bits(float zn, uint32_t& bits) { bits = __float_as_uint(zn + 0.0f); return bits <= 0x3F800000u; }
// The same for a sum zc + dz whose first term has been through "+ 0.0f" already: a sum is -0 only when both terms are, so it needs no second one
// (and if zc was -0, +0 + dz gives what the canonicalised -0 + dz would: dz, or +0 for dz = -0).
AWSM_DI bool depth_key_bits_sum(float zc_canonical, float dz, uint32_t& bits) { bits = __float_as_uint(zc_canonical + dz); return bits <= 0x3F800000u; }

// Coverage + depth at the sample (X, Y) (pixels, f64).  Returns the packed 64-bit key or ~0 if not covered.
AWSM_DI unsigned long long tri_key_from_edges(const TriSetup& t, const EdgeVals& ev, uint32_t rank);
AWSM_DI unsigned long long tri_sample_key_at(const TriSetup& t, double X, double Y, uint32_t rank) {
    return tri_key_from_edges(t, tri_edges_d(t, X, Y), rank);
}
// all three edges, then one decision (no short-circuit: a nest of exec-masked branches costs more than the two FMAs it may skip)
AWSM_DI bool tri_covers(const TriSetup& t, const EdgeVals& ev) {
    return ((int)(ev.E[0] > edge_threshold(t.a[0], t.b[0])) & (int)(ev.E[1] > edge_threshold(t.a[1], t.b[1])) & (int)(ev.E[2] > edge_threshold(t.a[2], t.b[2]))) != 0;
}
AWSM_DI float tri_plane_depth(const TriSetup& t, const EdgeVals& ev) {
    const float e0 = (float)ev.E[0], e1 = (float)ev.E[1], e2 = (float)ev.E[2];
    return (e0 * t.zq[0] + e1 * t.zq[1]) + e2 * t.zq[2];
}
// Multisampled targets (the contract of oracle_geometry.c, tri_sample_msaa): a sample's depth is the plane's value at the pixel's corner —
// tri_plane_depth of the edge values there — plus the sample's increment along the plane's gradient, dz_k = gx fx_k + gy fy_k with
// gx = (a0 zq0 + a1 zq1) + a2 zq2, gy likewise from b: one f32 rounding per operation, nothing contracted (the library is built with -ffp-contract=off).
// One add per sample where the three-term form costs three conversions, three products and two sums.
AWSM_DI void msaa_depth_steps(const float a[3], const float b[3], const float zq[3], float dz[4]) {
    const float gx = (a[0] * zq[0] + a[1] * zq[1]) + a[2] * zq[2];
    const float gy = (b[0] * zq[0] + b[1] * zq[1]) + b[2] * zq[2];
    dz[0] = gx * 0.375f + gy * 0.125f; dz[1] = gx * 0.875f + gy * 0.375f; dz[2] = gx * 0.125f + gy * 0.625f; dz[3] = gx * 0.625f + gy * 0.875f;      // msaa4_x / 256, msaa4_y / 256
}
// The key of sample k of pixel (px, py), or ~0: zc = tri_plane_depth at the pixel's corner + 0.0f (depth_key_bits_sum), dz from msaa_depth_steps.
AWSM_DI unsigned long long tri_msaa_sample_key(const TriSetup& t, int px, int py, int k, float zc, const float dz[4], uint32_t rank) {
    if (!tri_covers(t, tri_edges_d(t, sample_coord((px << 8) + msaa4_x(k)), sample_coord((py << 8) + msaa4_y(k))))) return ~0ull;
    uint32_t zbits;
    if (!depth_key_bits_sum(zc, dz[k], zbits)) return ~0ull;
    return ((unsigned long long)zbits << 32) | (unsigned long long)(0xFFFFFFFFu - rank);
}
AWSM_DI unsigned long long tri_key_from_edges(const TriSetup& t, const EdgeVals& ev, uint32_t rank) {
    if (!tri_covers(t, ev)) return ~0ull;
    const float zn = tri_plane_depth(t, ev);
    uint32_t zbits;
    if (!depth_key_bits(zn, zbits)) return ~0ull;      // (-0 -> +0 so the bits order as an unsigned integer)
    // depth LessEqual + submission order: smaller depth wins, equal depth -> LATER primitive wins
    return ((unsigned long long)zbits << 32) | (unsigned long long)(0xFFFFFFFFu - rank);
}

// Conservative "tile can contain a covered sample" test: evaluates every edge at the tile corner that maximises it.
// The corners are at least half a pixel (pixel centres) or an eighth of a pixel (MSAA sample positions) away from the
// outermost samples, which dwarfs the rounding error of the evaluation, so no covered sample is ever rejected.
AWSM_DI bool tile_may_overlap(const TriSetup& t, int tx0, int ty0, int tx1, int ty1 /* pixel bounds, exclusive max */) {
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const double X = t.a[i] > 0.0f ? (double)tx1 : (double)tx0;
        const double Y = t.b[i] > 0.0f ? (double)ty1 : (double)ty0;
        const double ax = (double)t.a[i] * X, by = (double)t.b[i] * Y;
        const double e = (ax + by) + t.c[i];
        const double slack = 1e-15 * ((fabs(ax) + fabs(by)) + fabs(t.c[i]));   // > 3 ulp of the largest term (kind 1; kind 0 is exact)
        if (e < -slack) return false;
    }
    return true;
}

}  // namespace awsm
