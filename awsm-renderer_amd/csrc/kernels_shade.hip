// kernels_shade.hip — Opaque Pass on gfx950: one thread per pixel, single dispatch over the screen.
//
// Replaces (paths relative to /root/reference/crates/renderer/src/render_passes/):
//   material_opaque/shader/material_opaque_wgsl/compute.wgsl:100-322        main (single-sample, MipmapMode::None)
//   material_opaque/shader/material_opaque_wgsl/empty.wgsl:38-59            main when there are no opaque renderables
//   material_opaque/shader/material_opaque_wgsl/helpers/{standard,skybox,texture_uvs,vertex_color_attrib,
//       material_color_calc}.wgsl
//   shared/shared_wgsl/{material,material_mesh_meta,textures}.wgsl, pbr/*.wgsl, unlit/*.wgsl, lighting/*.wgsl
//   geometry/shader/geometry_wgsl/fragment.wgsl:23-54  (fs_main: folded in — the G-buffer targets are never
//       materialised; the interpolants of the visible triangle are recomputed here and rounded to the
//       reference's RG16F / RGBA16F storage formats before use, crates/renderer/src/render_textures.rs:49-54)
//
// In this file, in order: the STRICT G-buffer reconstruction and MSAA edge predicates; the samplers (2D arrays: level 0 / gradient mips / anisotropic
// probes; cubes; the BRDF LUT); the material and lighting code of the general route (shade_material, shade_surface) with the transparent pass's three
// kernels (k_forward_cover / shade / blend); k_resolve_draws; the general opaque kernels (k_shade, k_shade_todo); the lean route (k_shade_lean: level-1
// records staged through LDS, scalar draw records, lean::fetch*, the aproned cube sampler); the MSAA kernels (k_shade_msaa, k_msaa_detect,
// k_shade_msaa_resolve); k_brdf_lut, k_cube_border and the small service kernels; the launch wrappers.  Kernels that sample are instantiated per mip mode
// (0 MipmapMode::None, 1 Gradient, 2 Gradient with anisotropic probes), the lean kernel also per MSAA.
//
// Pure gather + ALU: no MFMA.  Compulsory HBM traffic per pixel is the 8-byte key read and the 8-byte
// RGBA16F write; everything else (vertices, metas, materials, texels) is reused across neighbouring pixels
// and served by L1 / the XCD's L2 — workgroups are dealt to XCDs in contiguous screen runs for that.
#include <type_traits>
#include "frame_params.hpp"
#include "raster_setup.hpp"

#ifndef AWSM_LEAN_WAVES
#define AWSM_LEAN_WAVES 6
#endif

namespace awsm {

// ================================================================================================
// STRICT section (arithmetic contract, -ffp-contract=off, IEEE div/sqrt): what fs_main wrote for this pixel
// (fragment.wgsl:23-54), rounded to the G-buffer storage formats.  Bit-identical to the CPU oracle (tests only).
// ================================================================================================
struct GBufferTexel {
    f4 packed_nt;    // RGBA16F normal_tangent, already rounded to f16
    float bx, by;    // RG16F barycentric, already rounded to f16
    f4 bary_derivs;  // RGBA16F barycentric_derivatives (db0/dx, db0/dy, db1/dx, db1/dy), rounded to f16; MipmapMode::Gradient only
};
// The interpolants are evaluated at the PIXEL CENTRE (@interpolate(perspective, center)); with MSAA the centre may lie
// outside the triangle and the values extrapolate — every sample the triangle covers in the pixel gets the same texel.
// A key in the visibility buffer means the triangle's setup record is valid; its edge coefficients are the bits the
// raster kernel used.
template <bool DERIVS>
AWSM_DI GBufferTexel reconstruct_core(const TriSetup& t, float4 n0, float4 n1, float4 n2, float4 t0, float4 t1, float4 t2, int cx, int cy) {
    GBufferTexel g;
    g.bary_derivs = {0.0f, 0.0f, 0.0f, 0.0f};
    const double Xc = sample_coord((cx << 8) + 128), Yc = sample_coord((cy << 8) + 128);
    const EdgeVals ev = tri_edges_d(t, Xc, Yc);
    // screen-space edge weights -> perspective-correct barycentrics: one IEEE reciprocal, six products
    const float e0 = (float)ev.E[0] * t.iw[0], e1 = (float)ev.E[1] * t.iw[1], e2 = (float)ev.E[2] * t.iw[2];
    const float inv_esum = 1.0f / ((e0 + e1) + e2);
    const float b0 = e0 * inv_esum, b1 = e1 * inv_esum, b2 = e2 * inv_esum;
    const f3 Ni = {(b0 * n0.x + b1 * n1.x) + b2 * n2.x, (b0 * n0.y + b1 * n1.y) + b2 * n2.y, (b0 * n0.z + b1 * n1.z) + b2 * n2.z};
    const f4 Ti = {(b0 * t0.x + b1 * t1.x) + b2 * t2.x, (b0 * t0.y + b1 * t1.y) + b2 * t2.y,
                   (b0 * t0.z + b1 * t1.z) + b2 * t2.z, (b0 * t0.w + b1 * t1.w) + b2 * t2.w};
    const f4 p = pack_normal_tangent(normalize(Ni), normalize(mk3(Ti.x, Ti.y, Ti.z)), Ti.w);
    g.packed_nt = {round_f16(p.x), round_f16(p.y), round_f16(p.z), round_f16(p.w)};
    g.bx = round_f16(b0);
    g.by = round_f16(b1);
    if (DERIVS) {
        // fragment.wgsl:46-51 dpdx/dpdy of the barycentrics.  Contract ("fine" derivatives of a 2x2 quad): the difference
        // between the two pixels of the quad row / column this pixel sits in, both evaluated for THIS triangle (helper
        // invocations extrapolate), right minus left and bottom minus top; then RGBA16F.
        const EdgeVals eh = tri_edges_d(t, sample_coord(((cx ^ 1) << 8) + 128), Yc), evv = tri_edges_d(t, Xc, sample_coord(((cy ^ 1) << 8) + 128));
        const float h0 = (float)eh.E[0] * t.iw[0], h1 = (float)eh.E[1] * t.iw[1], h2 = (float)eh.E[2] * t.iw[2];
        const float w0 = (float)evv.E[0] * t.iw[0], w1 = (float)evv.E[1] * t.iw[1], w2 = (float)evv.E[2] * t.iw[2];
        const float ish = 1.0f / ((h0 + h1) + h2), isv = 1.0f / ((w0 + w1) + w2);
        const float hb0 = h0 * ish, hb1 = h1 * ish, vb0 = w0 * isv, vb1 = w1 * isv;
        const float ddx0 = (cx & 1) ? b0 - hb0 : hb0 - b0, ddx1 = (cx & 1) ? b1 - hb1 : hb1 - b1;
        const float ddy0 = (cy & 1) ? b0 - vb0 : vb0 - b0, ddy1 = (cy & 1) ? b1 - vb1 : vb1 - b1;
        g.bary_derivs = {round_f16(ddx0), round_f16(ddy0), round_f16(ddx1), round_f16(ddy1)};
    }
    return g;
}
template <bool DERIVS>
AWSM_DI GBufferTexel reconstruct_gbuffer(const FrameDev& f, uint32_t rank, int cx, int cy) {
    TriSetup t;
    tri_rec_load(f.tri_rec + rank, t);
    const float4 n0 = f.nrm[(size_t)rank * 3], n1 = f.nrm[(size_t)rank * 3 + 1], n2 = f.nrm[(size_t)rank * 3 + 2];
    const float4 t0 = f.tan[(size_t)rank * 3], t1 = f.tan[(size_t)rank * 3 + 1], t2 = f.tan[(size_t)rank * 3 + 2];
    return reconstruct_core<DERIVS>(t, n0, n1, n2, t0, t1, t2, cx, cy);
}

// The decoded normal of the pixel's G-buffer texel alone — decode_octahedral(packed_nt.xy) — for the MSAA edge detector: the same operations on the
// same values as reconstruct_core + pack_normal_tangent's octahedral half, without the tangent (its interpolation, normalisation, basis and atan2: a
// third of the reconstruction) and without the tangents' 48 bytes per lane.  Bit-identical to unpack_normal_tangent(g.packed_nt).N by construction.
AWSM_DI f2 strict_oct_of(const FrameDev& f, uint32_t rank, int cx, int cy) {
    TriSetup t;
    tri_rec_load(f.tri_rec + rank, t);
    const float4 n0 = f.nrm[(size_t)rank * 3], n1 = f.nrm[(size_t)rank * 3 + 1], n2 = f.nrm[(size_t)rank * 3 + 2];
    const double Xc = sample_coord((cx << 8) + 128), Yc = sample_coord((cy << 8) + 128);
    const EdgeVals ev = tri_edges_d(t, Xc, Yc);
    const float e0 = (float)ev.E[0] * t.iw[0], e1 = (float)ev.E[1] * t.iw[1], e2 = (float)ev.E[2] * t.iw[2];
    const float inv_esum = 1.0f / ((e0 + e1) + e2);
    const float b0 = e0 * inv_esum, b1 = e1 * inv_esum, b2 = e2 * inv_esum;
    const f3 Ni = {(b0 * n0.x + b1 * n1.x) + b2 * n2.x, (b0 * n0.y + b1 * n1.y) + b2 * n2.y, (b0 * n0.z + b1 * n1.z) + b2 * n2.z};
    const f2 oct = encode_octahedral(normalize(Ni));
    return mk2(round_f16(oct.x), round_f16(oct.y));
}
AWSM_DI f3 strict_normal_of(const FrameDev& f, uint32_t rank, int cx, int cy) { return decode_octahedral(strict_oct_of(f, rank, cx, cy)); }
// An octahedral pair as two f16 in a word (its values ARE f16 values: exact both ways)
AWSM_DI uint32_t oct_word(f2 oct) { return (uint32_t)f16_bits(oct.x) | ((uint32_t)f16_bits(oct.y) << 16); }
AWSM_DI f2 oct_of_word(uint32_t w) { return mk2(__half2float(__ushort_as_half((unsigned short)(w & 0xFFFFu))), __half2float(__ushort_as_half((unsigned short)(w >> 16)))); }

// Is pixel row `py` one this shard shades (row strip: [sy0, sy1); bands: the 32-row tile rows r, r + n, ...)?
AWSM_DI bool row_owned(const FrameDev& f, int py) {
    if (py < (int)f.sy0 || py >= (int)f.sy1) return false;
    return f.band_n <= 1u || (((uint32_t)py >> kTileShift) % f.band_n) == f.band_r;
}

// ---- MSAA edge predicates (helpers/msaa.wgsl), STRICT: a decision that flips between implementations would swap a
// pixel between one-sample and four-sample shading, so every value feeding a threshold follows the arithmetic contract ----
constexpr float kEdgeNormalThreshold = 0.95f, kEdgeDepthThreshold = 0.02f, kEdgeMsaaDepthThreshold = 0.02f;
AWSM_DI float view_space_depth(const m4& inv_proj, float depth, float px, float py, float W, float H) {   // msaa.wgsl:185-199
    // A projection whose view-space z and w depend on the depth alone (every perspective_rh / orthographic_rh: glam's matrices have exact zeros there) makes
    // the x and y terms of those two rows exact zeros, and ((0 x + 0 y) + c2 d) + c3 IS c2 d + c3 bit for bit: the NDC divisions and two thirds of the
    // product drop out (the detector calls this up to nine times per pixel).  Wave-uniform test; anything else takes the full product.
    if (inv_proj.c[0].z == 0.0f && inv_proj.c[1].z == 0.0f && inv_proj.c[0].w == 0.0f && inv_proj.c[1].w == 0.0f)
        return (inv_proj.c[2].z * depth + inv_proj.c[3].z) / (inv_proj.c[2].w * depth + inv_proj.c[3].w);
    const f4 view_pos = mul(inv_proj, mk4((px / W) * 2.0f - 1.0f, 1.0f - (py / H) * 2.0f, depth, 1.0f));
    return view_pos.z / view_pos.w;
}
AWSM_DI float key_depth(unsigned long long k) { return k == ~0ull ? 1.0f : __uint_as_float((uint32_t)(k >> 32)); }   // depth clear = 1.0
AWSM_DI uint32_t key_rank(unsigned long long k) { return 0xFFFFFFFFu - (uint32_t)(k & 0xFFFFFFFFull); }
AWSM_DI bool edge_mask_depth_msaa(const m4& inv_proj, const unsigned long long k4[4], float pcx, float pcy, float W, float H) {   // msaa.wgsl:116-146
    uint32_t count = 0; float dmin = 1e9f, dmax = -1e9f;
#pragma unroll
    for (int s = 0; s < 4; s++) {
        if (k4[s] == ~0ull) continue;
        count++;
        const float vd = view_space_depth(inv_proj, key_depth(k4[s]), pcx, pcy, W, H);
        dmin = fminf(dmin, vd); dmax = fmaxf(dmax, vd);
    }
    if (count < 2u) return false;
    return fabsf(dmax - dmin) > (kEdgeMsaaDepthThreshold * fabsf((dmax + dmin) * 0.5f));
}

// The two depth predicates with the division behind a filter: a projection whose view depth is (a d + b) / (c d + e) (view_space_depth's first form) is
// evaluated with the hardware reciprocal — numerator and denominator as the strict form computes them, so the quotient is within 2 ulps of the IEEE one —
// and the comparison is accepted when it clears the threshold by more than 4e-6 of the larger depth (ten times that error); anything closer, and any
// other projection, takes the strict form.  Same decisions, a fifth of the instructions (an IEEE division is ten, and the detector makes up to nine).
AWSM_DI bool depth_only_projection(const m4& inv_proj) { return inv_proj.c[0].z == 0.0f && inv_proj.c[1].z == 0.0f && inv_proj.c[0].w == 0.0f && inv_proj.c[1].w == 0.0f; }
AWSM_DI float view_depth_approx(const m4& inv_proj, float depth) { return (inv_proj.c[2].z * depth + inv_proj.c[3].z) * __builtin_amdgcn_rcpf(inv_proj.c[2].w * depth + inv_proj.c[3].w); }
AWSM_DI bool edge_mask_depth_msaa_filtered(const m4& inv_proj, const unsigned long long k4[4], float pcx, float pcy, float W, float H) {
    if (depth_only_projection(inv_proj)) {      // wave-uniform
        uint32_t count = 0; float dmin = 1e9f, dmax = -1e9f;
#pragma unroll
        for (int s = 0; s < 4; s++) {
            if (k4[s] == ~0ull) continue;
            count++;
            const float vd = view_depth_approx(inv_proj, key_depth(k4[s]));
            dmin = fminf(dmin, vd); dmax = fmaxf(dmax, vd);
        }
        if (count < 2u) return false;
        const float lhs = fabsf(dmax - dmin), rhs = kEdgeMsaaDepthThreshold * fabsf((dmax + dmin) * 0.5f), margin = 4e-6f * fmaxf(fabsf(dmax), fabsf(dmin));
        if (lhs > rhs + margin) return true;
        if (lhs < rhs - margin) return false;     // (a NaN falls through to the strict form)
    }
    return edge_mask_depth_msaa(inv_proj, k4, pcx, pcy, W, H);
}

// standard.wgsl:17-33 operation by operation (IEEE divisions, no contraction: this function sits in the STRICT part of the file): the world position exactly
// as the oracle forms it.  Used by the experiment AWSM_STRICT_POSITION only (tests/diagnostics/abs_bar_survey.py: which pixels over the absolute colour bar
// come from the position's last bits) — the shipped kernels compose pixel -> view on the host (FrameDev.pix2view).
AWSM_DI f3 strict_world_position(const m4& inv_proj, const m4& inv_view, int cx, int cy, float W, float H, float depth) {
    const float uvx = ((float)cx + 0.5f) / W, uvy = ((float)cy + 0.5f) / H;
    const f4 view_h = mul(inv_proj, mk4(uvx * 2.0f - 1.0f, 1.0f - uvy * 2.0f, depth, 1.0f));
    const float vw = fmaxf(view_h.w, 1e-8f);
    const f4 wp = mul(inv_view, mk4(view_h.x / vw, view_h.y / vw, view_h.z / vw, 1.0f));
    return {wp.x, wp.y, wp.z};
}

// ================================================================================================
// RELAXED section: everything downstream of the quantised G-buffer values only has to stay within 1e-4 of the
// oracle (BASELINE.json north_star), so it may contract to FMA and use the hardware reciprocal / rsqrt / exp2 /
// log2 / sin / cos units (each ~1 ulp).  Helpers are re-defined here under contract(fast); the strict ones in
// device_math.hpp keep their own flags even when inlined.
// ================================================================================================
#pragma clang fp contract(fast)
namespace fm {
AWSM_DI float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
AWSM_DI float rsq(float x) { return __builtin_amdgcn_rsqf(x); }
AWSM_DI float fdiv(float a, float b) { return a * rcp(b); }
AWSM_DI float fdot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
AWSM_DI f3 fnormalize(f3 a) { return a * rsq(fdot(a, a)); }
AWSM_DI f3 fsafe_normalize(f3 n) { const float l = fdot(n, n); return l > 0.0f ? n * rsq(l) : mk3(0.0f, 0.0f, 1.0f); }
AWSM_DI float pow5(float x) { const float x2 = x * x; return x2 * x2 * x; }
AWSM_DI float powp(float x, float y) { return x > 0.0f ? __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x)) : (x == 0.0f ? (y == 0.0f ? 1.0f : 0.0f) : __builtin_nanf("")); }
AWSM_DI f4 fmul(const m4& m, f4 v) {
    return {m.c[0].x * v.x + m.c[1].x * v.y + m.c[2].x * v.z + m.c[3].x * v.w, m.c[0].y * v.x + m.c[1].y * v.y + m.c[2].y * v.z + m.c[3].y * v.w,
            m.c[0].z * v.x + m.c[1].z * v.y + m.c[2].z * v.z + m.c[3].z * v.w, m.c[0].w * v.x + m.c[1].w * v.y + m.c[2].w * v.z + m.c[3].w * v.w};
}
AWSM_DI f3 fdecode_octahedral(f2 e) {         // math.wgsl:55-67
    const float fx = e.x * 2.0f - 1.0f, fy = e.y * 2.0f - 1.0f;
    f3 n = {fx, fy, (1.0f - fabsf(fx)) - fabsf(fy)};
    const float t = clampf(-n.z, 0.0f, 1.0f);
    n.x += (n.x >= 0.0f) ? -t : t;
    n.y += (n.y >= 0.0f) ? -t : t;
    return fnormalize(n);
}
AWSM_DI TBN funpack_normal_tangent(f4 rgba) {  // math.wgsl:104-116
    TBN r;
    r.N = fdecode_octahedral({rgba.x, rgba.y});
    const float theta = rgba.z * kTau - kPi;
    const float s = (rgba.w >= 0.5f) ? 1.0f : -1.0f;
    f3 tt, tb;
    if (r.N.z < -0.98f) {
        // canonical_tb (math.wgsl:73-84) divides by 1 + N.z: towards N = (0, 0, -1) a one-ulp difference in the decoded normal moves the basis
        // by 6e-8 / (1 + N.z) — percent of a radian in the last degrees — so there the normal and the basis are computed with the oracle's
        // operations (IEEE division and square root, no contraction; decode_octahedral / canonical_tb of the STRICT section).  Found by
        // rendering from random viewpoints (tests/diagnostics/viewpoint_survey.py): surfaces facing -z were off by up to 6e-2 in single pixels.
        r.N = decode_octahedral({rgba.x, rgba.y});
        const TB cb = canonical_tb(r.N);
        tt = cb.t; tb = cb.b;
    } else {
        const float a = rcp(1.0f + r.N.z), bb = (-r.N.x * r.N.y) * a;
        tt = {1.0f - (r.N.x * r.N.x) * a, bb, -r.N.x};
        tb = {bb, 1.0f - (r.N.y * r.N.y) * a, -r.N.y};
    }
    const float c = __ocml_native_cos_f32(theta), sn = __ocml_native_sin_f32(theta);
    r.T = fnormalize(tt * c + tb * sn);
    r.B = fnormalize(cross(r.N, r.T)) * s;
    return r;
}

}  // namespace fm

// ---------------- textures.wgsl ----------------
struct TexInfo {
    bool exists;
    uint32_t array_index, layer_index, uv_set_index, sampler_index, uv_transform_index;
};
AWSM_DI TexInfo tex_load(const uint32_t* __restrict__ m, uint32_t i) {      // textures.wgsl:75-114
    TexInfo t;
    const uint32_t array_and_layer = m[i + 1], uv_and_sampler = m[i + 2], extra = m[i + 3], transform_offset = m[i + 4];
    t.array_index = array_and_layer & 0xFFFu; t.layer_index = array_and_layer >> 12;
    t.uv_set_index = uv_and_sampler & 0xFFu; t.sampler_index = uv_and_sampler >> 8;
    t.exists = (extra & 1u) != 0u;
    t.uv_transform_index = transform_offset / 32u;
    return t;
}

// exact integer wrap without an integer divide: power-of-two sizes use masks, other sizes a float quotient + fix-up
AWSM_DI int mod_floor(int i, int n) {
    if ((n & (n - 1)) == 0) return i & (n - 1);
    int r = i - (int)floorf((float)i * fm::rcp((float)n)) * n;
    if (r < 0) r += n;
    if (r >= n) r -= n;
    return r;
}
AWSM_DI int wrap_index(int i, int n, uint32_t mode) {
    if (mode == 1u) return mod_floor(i, n);
    if (mode == 2u) { const int m = mod_floor(i, 2 * n); return m < n ? m : 2 * n - 1 - m; }
    return i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
}
AWSM_DI f4 texel_rgba8(const uint8_t* __restrict__ p) {
    const uint32_t u = *reinterpret_cast<const uint32_t*>(p);
    const float k = 1.0f / 255.0f;
    return {(float)(u & 255u) * k, (float)((u >> 8) & 255u) * k, (float)((u >> 16) & 255u) * k, (float)(u >> 24) * k};
}
AWSM_DI float safe_floor(float x, float& frac) {
    const float fl = floorf(x);
    if (!(fl >= -1073741824.0f && fl <= 1073741824.0f)) { frac = 0.0f; return 0.0f; }
    frac = x - fl;
    return fl;
}
AWSM_DI f4 lerp4(f4 a, f4 b, float t) { const float s = 1.0f - t; return {a.x * s + b.x * t, a.y * s + b.y * t, a.z * s + b.z * t, a.w * s + b.w * t}; }
// textureSampleLevel(tex, sampler, uv, layer, level) on one level of one layer: DESIGN.md §"Texture sampling".  General
// form: any size, any address mode, nearest or linear.  Out of line (one copy for all call sites); the hot path below
// handles the common sampler inline and only falls back here when some lane of the wavefront needs it.
__device__ __attribute__((noinline)) f4 sample_level_generic(const uint8_t* base, uint32_t width, uint32_t height, uint32_t mode_u, uint32_t mode_v,
                                                             uint32_t linear, float u, float v) {
    const int W = (int)width, H = (int)height;
    float fx, fy;
    if (linear == 0u) {
        const int i = wrap_index((int)safe_floor(u * (float)W, fx), W, mode_u);
        const int j = wrap_index((int)safe_floor(v * (float)H, fy), H, mode_v);
        return texel_rgba8(base + ((size_t)j * W + i) * 4u);
    }
    const float x0f = safe_floor(u * (float)W - 0.5f, fx);
    const float y0f = safe_floor(v * (float)H - 0.5f, fy);
    const int i0 = wrap_index((int)x0f, W, mode_u), i1 = wrap_index((int)x0f + 1, W, mode_u);
    const int j0 = wrap_index((int)y0f, H, mode_v), j1 = wrap_index((int)y0f + 1, H, mode_v);
    const uint8_t* r0 = base + (size_t)j0 * W * 4u;
    const uint8_t* r1 = base + (size_t)j1 * W * 4u;
    const f4 c00 = texel_rgba8(r0 + i0 * 4), c10 = texel_rgba8(r0 + i1 * 4), c01 = texel_rgba8(r1 + i0 * 4), c11 = texel_rgba8(r1 + i1 * 4);
    return lerp4(lerp4(c00, c10, fx), lerp4(c01, c11, fx), fy);
}
// The common sampler (linear, repeat/repeat, power-of-two extent) inline: wrap is a mask, no mode selects, no quotients,
// the two taps of a row in one 8-byte load.
AWSM_DI f4 sample_level_fast(const uint32_t* base, uint32_t W, uint32_t H, float u, float v) {
    float fx, fy;
    const float x0f = safe_floor(u * (float)W - 0.5f, fx);
    const float y0f = safe_floor(v * (float)H - 0.5f, fy);
    const uint32_t xi = (uint32_t)(int)x0f, yi = (uint32_t)(int)y0f;
    const uint32_t i0 = xi & (W - 1u), i1 = (xi + 1u) & (W - 1u);
    const uint32_t r0 = (yi & (H - 1u)) * W, r1 = ((yi + 1u) & (H - 1u)) * W;
    uint32_t t00, t10, t01, t11;
    if (i1 == i0 + 1u) {   // neighbours in memory unless the footprint wraps
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2), aligned(4)));
        const u32x2 p0 = *reinterpret_cast<const u32x2*>(base + r0 + i0), p1 = *reinterpret_cast<const u32x2*>(base + r1 + i0);
        t00 = p0.x; t10 = p0.y; t01 = p1.x; t11 = p1.y;
    } else {
        t00 = base[r0 + i0]; t10 = base[r0 + i1]; t01 = base[r1 + i0]; t11 = base[r1 + i1];
    }
    // bilinear on the raw 0..255 values, one scale by 1/255 at the end
    const float gx = 1.0f - fx, gy = 1.0f - fy;
    const float w00 = gx * gy, w10 = fx * gy, w01 = gx * fy, w11 = fx * fy;
    const float k = 1.0f / 255.0f;
    f4 r;
    r.x = ((float)(t00 & 255u) * w00 + (float)(t10 & 255u) * w10 + (float)(t01 & 255u) * w01 + (float)(t11 & 255u) * w11) * k;
    r.y = ((float)((t00 >> 8) & 255u) * w00 + (float)((t10 >> 8) & 255u) * w10 + (float)((t01 >> 8) & 255u) * w01 + (float)((t11 >> 8) & 255u) * w11) * k;
    r.z = ((float)((t00 >> 16) & 255u) * w00 + (float)((t10 >> 16) & 255u) * w10 + (float)((t01 >> 16) & 255u) * w01 + (float)((t11 >> 16) & 255u) * w11) * k;
    r.w = ((float)(t00 >> 24) * w00 + (float)(t10 >> 24) * w10 + (float)(t01 >> 24) * w01 + (float)(t11 >> 24) * w11) * k;
    return r;
}

// textureSampleGrad's footprint.  WebGPU leaves level selection and anisotropy to the implementation; the contract here:
//   * max_anisotropy 1 (or a context without AWSM_CFG_ANISOTROPIC — the default, the rule the reference itself documents as "mimics the hardware mip
//     selection", helpers/mipmap.wgsl:419-439): rho = max(|ddx * size|, |ddy * size|), lod = log2(max(rho, 1e-6));
//   * max_anisotropy A > 1 (gltf samplers ask for 16, gltf/populate/material.rs:892-902): N = clamp(rho_max / rho_min, 1, A) — a real number — the
//     level is chosen for rho_max / N, and the footprint is covered by probes along the major axis at t_j = j / N, j = -m..m, m = ceil((N - 1) / 2),
//     each weighted by the part of [-1/2, 1/2] its cell [t_j - 1/2N, t_j + 1/2N] covers (a box filter of the footprint's length sampled at the chosen
//     level's spacing), normalised.  Continuous in N — a probe enters with weight zero — so two implementations that disagree in the last bit of a
//     gradient agree in the colour; N = 1 is the isotropic rule bit for bit.
struct GradFootprint { float lod, n, major_u, major_v; int m; };
AWSM_DI GradFootprint grad_footprint(float dxu, float dxv, float dyu, float dyv, float W, float H, uint32_t max_aniso) {
    const float ax = dxu * W, ay = dxv * H, bx = dyu * W, by = dyv * H;
    const float rx2 = ax * ax + ay * ay, ry2 = bx * bx + by * by;
    const float r2max = fmaxf(rx2, ry2);
    GradFootprint fp;
    fp.lod = 0.5f * __builtin_amdgcn_logf(fmaxf(r2max, 1e-12f));      // log2(max(rho, 1e-6))
    fp.n = 1.0f; fp.major_u = 0.0f; fp.major_v = 0.0f; fp.m = 0;
    if (max_aniso > 1u && r2max > 0.0f) {
        const float r2min = fminf(rx2, ry2), A = (float)min(max_aniso, 16u);
        float nf = r2min * (A * A) <= r2max ? A : __builtin_sqrtf(r2max / r2min);
        nf = fminf(fmaxf(nf, 1.0f), A);
        if (nf > 1.0f) {
            fp.n = nf;
            fp.lod = fp.lod - __builtin_amdgcn_logf(nf);
            fp.m = (int)ceilf((nf - 1.0f) * 0.5f);
            const bool xmajor = rx2 >= ry2;
            fp.major_u = xmajor ? dxu : dyu; fp.major_v = xmajor ? dxv : dyv;
        }
    }
    return fp;
}

// grad_footprint's probes: 2 m + 1 trilinear samples along the major axis, weighted and normalised.  levels_modes: lo | hi << 8 | address mode u << 16 |
// v << 18 | linear << 20.  Out of line, per lane: only pixels with an anisotropic footprint on an AWSM_CFG_ANISOTROPIC context come here.
__device__ __attribute__((noinline)) f4 sample_probes(const uint32_t* texels, const uint32_t* level_off, uint32_t W, uint32_t H, uint32_t layer, uint32_t levels_modes, float f,
                                                      float u, float v, float major_u, float major_v, float nf, int m) {
    const uint32_t lo = levels_modes & 255u, hi = (levels_modes >> 8) & 255u, mode_u = (levels_modes >> 16) & 3u, mode_v = (levels_modes >> 18) & 3u, linear = (levels_modes >> 20) & 1u;
    const float inv_n = 1.0f / nf;
    f4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    float wsum = 0.0f;
    const int n = f > 0.0f ? 2 : 1;
    for (int j = -m; j <= m; j++) {
        const float t = (float)j * inv_n, wp = saturate((0.5f - fabsf(t)) * nf + 0.5f);
        const float pu = u + major_u * t, pv = v + major_v * t;
        for (int k = 0; k < n; k++) {
            const uint32_t level = k ? hi : lo;
            const float w = (k ? f : 1.0f - f) * wp;
            const uint32_t Wl = max(W >> level, 1u), Hl = max(H >> level, 1u);
            const uint32_t* base = texels + level_off[level] + (size_t)layer * Wl * Hl;
            const f4 c = sample_level_generic(reinterpret_cast<const uint8_t*>(base), Wl, Hl, mode_u, mode_v, linear, pu, pv);
            acc = {acc.x + c.x * w, acc.y + c.y * w, acc.z + c.z * w, acc.w + c.w * w};
        }
        wsum += wp;
    }
    const float iw = 1.0f / wsum;
    return {acc.x * iw, acc.y * iw, acc.z * iw, acc.w * iw};
}

// ---------------- per-pixel attribute context ----------------
struct Attr {
    const DevScene* sc;
    const float* ad;          // attribute_data (f32 view)
    uint32_t v0, v1, v2;      // vertex_start of the three corners (floats)
    uint32_t uv_sets_index;
    f3 bary;
    f2 uv0;                   // interpolated TEXCOORD_0, computed once per pixel when a core texture uses it
    bool has_uv0;
    f4 bary_derivs;           // MipmapMode::Gradient: the RGBA16F barycentric_derivatives texel (db0/dx, db0/dy, db1/dx, db1/dy)
    f2 duv0_dx, duv0_dy;      // ... and d(TEXCOORD_0)/d(screen), alongside uv0
};
// texture_uvs.wgsl:64-84 (+ helpers/mipmap.wgsl:113-205 get_uv_derivatives when GRAD: chain rule over the vertex UVs)
template <int GRAD>
AWSM_DI f2 attr_uv(const Attr& a, uint32_t set, f2& ddx, f2& ddy) {
    const uint32_t o = a.uv_sets_index + set * 2u;
    const float x0 = a.ad[a.v0 + o], y0 = a.ad[a.v0 + o + 1], x1 = a.ad[a.v1 + o], y1 = a.ad[a.v1 + o + 1];
    const float x2 = a.ad[a.v2 + o], y2 = a.ad[a.v2 + o + 1];
    if (GRAD) {
        const float dAlphaDx = a.bary_derivs.x, dAlphaDy = a.bary_derivs.y, dBetaDx = a.bary_derivs.z, dBetaDy = a.bary_derivs.w;
        const float dGammaDx = -dAlphaDx - dBetaDx, dGammaDy = -dAlphaDy - dBetaDy;
        ddx = {x0 * dAlphaDx + x1 * dBetaDx + x2 * dGammaDx, y0 * dAlphaDx + y1 * dBetaDx + y2 * dGammaDx};
        ddy = {x0 * dAlphaDy + x1 * dBetaDy + x2 * dGammaDy, y0 * dAlphaDy + y1 * dBetaDy + y2 * dGammaDy};
        const bool tiny = (fabsf(dAlphaDx) + fabsf(dAlphaDy) + fabsf(dBetaDx) + fabsf(dBetaDy)) < 1e-20f;
        const bool ok = (ddx.x == ddx.x) && (ddx.y == ddx.y) && (ddy.x == ddy.x) && (ddy.y == ddy.y);   // NaN guard
        if (tiny || !ok) { ddx = {0.0f, 0.0f}; ddy = {0.0f, 0.0f}; }
    }
    return {interp3_strict(a.bary.x, a.bary.y, a.bary.z, x0, x1, x2), interp3_strict(a.bary.x, a.bary.y, a.bary.z, y0, y1, y2)};
}
// texture_uvs.wgsl:64-187 + textures.wgsl:131-150.  GRAD = MipmapMode::Gradient: textureSampleGrad by the contract the
// reference documents as "mimics the hardware mip selection" (helpers/mipmap.wgsl:419-439): rho = max(|ddx*size|, |ddy*size|),
// lod = log2(max(rho, 1e-6)) clamped to the chain; magnification -> mag filter on level 0; otherwise min filter on
// floor(lod) and floor(lod)+1 blended by the fraction (mipmap filter linear) or round(lod) (nearest).  Isotropic.
template <int GRAD>
AWSM_DI f4 sample_tex(const Attr& a, const TexInfo& t) {
    f2 uv = a.uv0, ddx = a.duv0_dx, ddy = a.duv0_dy;
    if (!(a.has_uv0 && t.uv_set_index == 0u)) uv = attr_uv<GRAD>(a, t.uv_set_index, ddx, ddy);
    const float* tt = reinterpret_cast<const float*>(a.sc->buf[AWSM_BUF_TEXTURE_TRANSFORMS] + (size_t)t.uv_transform_index * 32u);
    const float u = affine2_strict(tt[0], tt[1], tt[4], uv.x, uv.y), v = affine2_strict(tt[2], tt[3], tt[5], uv.x, uv.y);
    if (t.array_index >= a.sc->n_tex || t.sampler_index >= a.sc->n_samplers) return {0.0f, 0.0f, 0.0f, 0.0f};
    const TexArrayDev& arr = a.sc->tex[t.array_index];
    const AwsmSampler& smp = a.sc->samplers[t.sampler_index];
    const uint32_t W = arr.width, H = arr.height, layers = arr.layers;
    const uint8_t* texels = arr.texels;
    if (texels == nullptr || W == 0u || H == 0u || layers == 0u) return {0.0f, 0.0f, 0.0f, 0.0f};
    const uint32_t layer = min(t.layer_index, layers - 1u);
    const bool common = smp.address_mode_u == 1u && smp.address_mode_v == 1u && (W & (W - 1u)) == 0u && (H & (H - 1u)) == 0u;
    if (!GRAD) {
        // Hot path taken when ALL lanes of the wavefront qualify (one scalar branch)
        const bool fast = common && smp.mag_filter != 0u;
        if (__builtin_amdgcn_ballot_w64(!fast) != 0ull)
            return sample_level_generic(texels + (size_t)layer * W * H * 4u, W, H, smp.address_mode_u, smp.address_mode_v, smp.mag_filter, u, v);
        return sample_level_fast(reinterpret_cast<const uint32_t*>(texels) + (size_t)layer * W * H, W, H, u, v);
    }
    // ---- level selection ----
    const float dxu = tt[0] * ddx.x + tt[1] * ddx.y, dxv = tt[2] * ddx.x + tt[3] * ddx.y;     // texture_uvs.wgsl:27-35
    const float dyu = tt[0] * ddy.x + tt[1] * ddy.y, dyv = tt[2] * ddy.x + tt[3] * ddy.y;
    GradFootprint fp = grad_footprint(dxu, dxv, dyu, dyv, (float)W, (float)H, (GRAD == 2 && smp.mag_filter != 0u && smp.min_filter != 0u && smp.mipmap_filter != 0u) ? smp.max_anisotropy : 1u);
    const uint32_t levels = max(arr.mips, 1u);
    uint32_t lo = 0u, hi = 0u, linear = smp.mag_filter;
    float f = 0.0f;
    if (fp.lod > 0.0f && levels > 1u) {
        const float lod = fminf(fp.lod, (float)(levels - 1u));
        linear = smp.min_filter;
        if (smp.mipmap_filter == 0u) { lo = hi = (uint32_t)floorf(lod + 0.5f); }
        else { const float fl = floorf(lod); lo = (uint32_t)fl; hi = min(lo + 1u, levels - 1u); f = (hi != lo) ? lod - fl : 0.0f; }
    }
    const bool fast = common && linear != 0u;
    const bool all_fast = __builtin_amdgcn_ballot_w64(!fast) == 0ull;
    if (GRAD == 2 && fp.m > 0)      // anisotropic footprint on a context that honours max_anisotropy (the kernels' <2> instantiations): the probes, out of line
        return sample_probes(reinterpret_cast<const uint32_t*>(texels), arr.level_off, W, H, layer, lo | (hi << 8) | (smp.address_mode_u << 16) | (smp.address_mode_v << 18) | (linear << 20), f, u, v, fp.major_u, fp.major_v, fp.n, fp.m);
    f4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    const int n = f > 0.0f ? 2 : 1;
    for (int k = 0; k < n; k++) {            // not unrolled: one copy of the samplers per call site
        const uint32_t level = k ? hi : lo;
        const float w = k ? f : 1.0f - f;
        const uint32_t Wl = max(W >> level, 1u), Hl = max(H >> level, 1u);
        const uint32_t* base = reinterpret_cast<const uint32_t*>(texels) + arr.level_off[level] + (size_t)layer * Wl * Hl;
        const f4 c = all_fast ? sample_level_fast(base, Wl, Hl, u, v)
                              : sample_level_generic(reinterpret_cast<const uint8_t*>(base), Wl, Hl, smp.address_mode_u, smp.address_mode_v, linear, u, v);
        acc = {acc.x + c.x * w, acc.y + c.y * w, acc.z + c.z * w, acc.w + c.w * w};
    }
    return acc;
}
// A core texture through its per-draw slot.  MipmapMode::None: the fast path needs nothing but the slot; any other sampler / size falls
// back to the general route through the material words.  MipmapMode::Gradient: level selection as sample_tex<true>, with the array's
// layout and the sampler's modes taken from the slot.
template <int GRAD>
AWSM_DI f4 sample_slot(const Attr& a, const TexSlotDev* __restrict__ slot, const uint32_t* __restrict__ M, uint32_t word) {
    const uint4* q = reinterpret_cast<const uint4*>(slot);
    const uint4 q0 = q[0], q1 = q[1], q2 = q[2];        // base lo/hi, width, height | flags, tt0, tt1, tt2 | tt3, tt4, tt5, layer_levels
    const uint32_t flags = q1.x;
    const uint32_t uv_set = flags >> 24;
    const float t0 = __uint_as_float(q1.y), t1 = __uint_as_float(q1.z), t2 = __uint_as_float(q1.w), t3 = __uint_as_float(q2.x), t4 = __uint_as_float(q2.y), t5 = __uint_as_float(q2.z);
    if (!GRAD) {
        if (__builtin_amdgcn_ballot_w64((flags & 6u) != 2u) != 0ull) {       // some lane is not on the fast path
            if (flags & 4u) return {0.0f, 0.0f, 0.0f, 0.0f};
            return sample_tex<false>(a, tex_load(M, word));
        }
        f2 uv = a.uv0, ddx, ddy;
        if (!(a.has_uv0 && uv_set == 0u)) uv = attr_uv<false>(a, uv_set, ddx, ddy);
        const float u = affine2_strict(t0, t1, t4, uv.x, uv.y), v = affine2_strict(t2, t3, t5, uv.x, uv.y);
        const uint32_t* base = reinterpret_cast<const uint32_t*>(((unsigned long long)q0.y << 32) | q0.x);
        return sample_level_fast(base, q0.z, q0.w, u, v);
    }
    if (flags & 4u) return {0.0f, 0.0f, 0.0f, 0.0f};
    f2 uv = a.uv0, ddx = a.duv0_dx, ddy = a.duv0_dy;
    if (!(a.has_uv0 && uv_set == 0u)) uv = attr_uv<GRAD>(a, uv_set, ddx, ddy);
    const float u = affine2_strict(t0, t1, t4, uv.x, uv.y), v = affine2_strict(t2, t3, t5, uv.x, uv.y);
    const uint4 q3 = q[3];                              // level_off pointer, array base
    const uint32_t* level_off = reinterpret_cast<const uint32_t*>(((unsigned long long)q3.y << 32) | q3.x);
    const uint32_t* texels = reinterpret_cast<const uint32_t*>(((unsigned long long)q3.w << 32) | q3.z);
    const uint32_t W = q0.z, H = q0.w, layer = q2.w & 0xFFFFu, levels = q2.w >> 24;
    const uint32_t mode_u = (flags >> 13) & 3u, mode_v = (flags >> 21) & 3u;
    // ---- level selection (texture_uvs.wgsl:27-35 + the LOD contract, grad_footprint) ----
    const float dxu = t0 * ddx.x + t1 * ddx.y, dxv = t2 * ddx.x + t3 * ddx.y;
    const float dyu = t0 * ddy.x + t1 * ddy.y, dyv = t2 * ddy.x + t3 * ddy.y;
    GradFootprint fp = grad_footprint(dxu, dxv, dyu, dyv, (float)W, (float)H, GRAD == 2 ? max((q2.w >> 16) & 31u, 1u) : 1u);
    uint32_t lo = 0u, hi = 0u, linear = (flags >> 4) & 1u;
    float f = 0.0f;
    if (fp.lod > 0.0f && levels > 1u) {
        const float lod = fminf(fp.lod, (float)(levels - 1u));
        linear = (flags >> 5) & 1u;
        if (!(flags & 64u)) { lo = hi = (uint32_t)floorf(lod + 0.5f); }
        else { const float fl = floorf(lod); lo = (uint32_t)fl; hi = min(lo + 1u, levels - 1u); f = (hi != lo) ? lod - fl : 0.0f; }
    }
    const bool fast = (flags & 8u) != 0u && linear != 0u;
    const bool all_fast = __builtin_amdgcn_ballot_w64(!fast) == 0ull;
    if (GRAD == 2 && fp.m > 0) return sample_probes(texels, level_off, W, H, layer, lo | (hi << 8) | (mode_u << 16) | (mode_v << 18) | (linear << 20), f, u, v, fp.major_u, fp.major_v, fp.n, fp.m);
    f4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    const int n = f > 0.0f ? 2 : 1;
    for (int k = 0; k < n; k++) {            // not unrolled: one copy of the samplers per call site
        const uint32_t level = k ? hi : lo;
        const float w = k ? f : 1.0f - f;
        const uint32_t Wl = max(W >> level, 1u), Hl = max(H >> level, 1u);
        const uint32_t* base = texels + level_off[level] + (size_t)layer * Wl * Hl;
        const f4 c = all_fast ? sample_level_fast(base, Wl, Hl, u, v)
                              : sample_level_generic(reinterpret_cast<const uint8_t*>(base), Wl, Hl, mode_u, mode_v, linear, u, v);
        acc = {acc.x + c.x * w, acc.y + c.y * w, acc.z + c.z * w, acc.w + c.w * w};
    }
    return acc;
}

AWSM_DI f4 vertex_color(const Attr& a, uint32_t set_index) {               // vertex_color_attrib.wgsl:1-21
    const uint32_t o = set_index * 4u;
    float r[4];
#pragma unroll
    for (int j = 0; j < 4; j++) r[j] = a.bary.x * a.ad[a.v0 + o + j] + a.bary.y * a.ad[a.v1 + o + j] + a.bary.z * a.ad[a.v2 + o + j];
    return {r[0], r[1], r[2], r[3]};
}

AWSM_DI float mf(const uint32_t* __restrict__ m, uint32_t i) { return __uint_as_float(m[i]); }
AWSM_DI uint32_t abs_index(uint32_t base, uint32_t rel) { return rel != 0u ? base + rel : 0u; }

// pbr_material_color.wgsl:4-32
struct PbrColor {
    f3 base; f2 mr; f3 normal; float occlusion; f3 emissive;
    float specular; f3 specular_color; float ior; float transmission;
    float volume_thickness, volume_attenuation_distance; f3 volume_attenuation_color;
    float clearcoat, clearcoat_roughness; f3 clearcoat_normal;
    f3 sheen_color; float sheen_roughness;
};

template <int GRAD>
AWSM_DI f3 normal_map(const Attr& a, const TexInfo& t, float scale, const TBN& tbn) {   // material_color_calc.wgsl:301-322
    if (!t.exists) return tbn.N;
    const f4 s = sample_tex<GRAD>(a, t);
    const float tx = (s.x * 2.0f - 1.0f) * scale, ty = (s.y * 2.0f - 1.0f) * scale, tz = s.z * 2.0f - 1.0f;
    return fm::fnormalize(tbn.T * tx + tbn.B * ty + tbn.N * tz);
}

// ---------------- brdf.wgsl ----------------
AWSM_DI float ior_to_f0(float ior) { const float v = ior < 1.0f ? 1.5f : ior; const float r = fm::fdiv(v - 1.0f, v + 1.0f); return r * r; }
AWSM_DI f3 volume_attenuation(float distance, f3 color, float att_distance) {           // brdf.wgsl:55-74
    if (distance <= 0.0f) return splat3(1.0f);
    if (att_distance <= 0.0f || att_distance > 1e10f) return splat3(1.0f);
    if (color.x >= 0.999f && color.y >= 0.999f && color.z >= 0.999f) return splat3(1.0f);
    const float e = fm::fdiv(distance, att_distance);
    return {fm::powp(color.x, e), fm::powp(color.y, e), fm::powp(color.z, e)};
}
AWSM_DI bool should_apply_volume_attenuation(float thickness, float att_distance, f3 c) {
    return thickness > 0.0f && att_distance < 1e10f && (c.x < 1.0f || c.y < 1.0f || c.z < 1.0f);
}
AWSM_DI f3 fresnel_schlick_f90(float cos_theta, f3 F0, float f90) {                      // brdf.wgsl:111-115
    const float p = fm::pow5(1.0f - saturate(cos_theta));
    return {F0.x + (f90 - F0.x) * p, F0.y + (f90 - F0.y) * p, F0.z + (f90 - F0.z) * p};
}
AWSM_DI float fresnel_schlick_scalar(float cos_theta, float F0) { return F0 + (1.0f - F0) * fm::pow5(1.0f - saturate(cos_theta)); }
AWSM_DI float distribution_ggx(float n_dot_h, float alpha) {                             // brdf.wgsl:118-124
    const float a = fmaxf(alpha, 0.001f);
    const float a2 = a * a;
    const float ndh = saturate(n_dot_h);
    const float d = (ndh * ndh) * (a2 - 1.0f) + 1.0f;
    return fm::fdiv(a2, (kPi * d) * d + kEps);
}
AWSM_DI float geometry_schlick_ggx(float n_dot_x, float alpha) {                         // brdf.wgsl:127-132
    const float a = fmaxf(alpha, 0.001f);
    const float k = ((a + 1.0f) * (a + 1.0f)) * 0.125f;
    const float ndx = saturate(n_dot_x);
    return fm::fdiv(ndx, ndx * (1.0f - k) + k);
}
constexpr float kClearcoatF0 = 0.04f;
AWSM_DI float clearcoat_fresnel(float clearcoat, float v_dot_h) { return clearcoat <= 0.0f ? 0.0f : clearcoat * fresnel_schlick_scalar(v_dot_h, kClearcoatF0); }
AWSM_DI float sheen_albedo_scaling(f3 sheen_color, float sheen_roughness, float n_dot_v) {   // brdf.wgsl:245-262
    const float sheen_max = fmaxf(fmaxf(sheen_color.x, sheen_color.y), sheen_color.z);
    if (sheen_max <= 0.0f) return 1.0f;
    const float alpha = sheen_roughness * sheen_roughness;
    return 1.0f - sheen_max * (alpha * (0.18f + 0.06f * (1.0f - n_dot_v)));
}
// brdf.wgsl:293-302 — linear, clamp-to-edge, RG of the RGBA16F LUT
AWSM_DI f2 sample_brdf_lut(const DevScene* sc, float n_dot_v, float roughness) {
    const float u = saturate(n_dot_v), v = saturate(roughness);
    const int W = (int)sc->lut_w, H = (int)sc->lut_h;
    float fx, fy;
    const float x0f = safe_floor(u * (float)W - 0.5f, fx);
    const float y0f = safe_floor(v * (float)H - 0.5f, fy);
    const int i0 = wrap_index((int)x0f, W, 0u), i1 = wrap_index((int)x0f + 1, W, 0u);
    const int j0 = wrap_index((int)y0f, H, 0u), j1 = wrap_index((int)y0f + 1, H, 0u);
    const uint32_t* L = reinterpret_cast<const uint32_t*>(sc->lut_rg16f);   // one u32 = (r16, g16)
    const uint32_t t00 = L[(size_t)j0 * W + i0], t10 = L[(size_t)j0 * W + i1], t01 = L[(size_t)j1 * W + i0], t11 = L[(size_t)j1 * W + i1];
    const float gx = 1.0f - fx, gy = 1.0f - fy;
    const float r_top = f16_bits_to_f32((unsigned short)(t00 & 0xFFFFu)) * gx + f16_bits_to_f32((unsigned short)(t10 & 0xFFFFu)) * fx;
    const float r_bot = f16_bits_to_f32((unsigned short)(t01 & 0xFFFFu)) * gx + f16_bits_to_f32((unsigned short)(t11 & 0xFFFFu)) * fx;
    const float g_top = f16_bits_to_f32((unsigned short)(t00 >> 16)) * gx + f16_bits_to_f32((unsigned short)(t10 >> 16)) * fx;
    const float g_bot = f16_bits_to_f32((unsigned short)(t01 >> 16)) * gx + f16_bits_to_f32((unsigned short)(t11 >> 16)) * fx;
    return {r_top * gy + r_bot * fy, g_top * gy + g_bot * fy};
}


// ---------------- cubemaps: textureSampleLevel(texture_cube<f32>, linear / linear / linear sampler, direction, level) ----------------
// (skybox.wgsl:37, brdf.wgsl:268-290).  Contract where WebGPU defers to the hardware: the major axis picks the face (z if |z| >= |x|,|y|,
// else y if |y| >= |x|, else x — the Vulkan / D3D table, as are sc, tc below); bilinear on the level's N x N faces with texel
// centres at (i + 0.5) / N; a tap that falls off the face comes from the face across that edge (seamless, kCubeEdge; a corner tap,
// off in both directions, keeps its row); the level is clamped to the chain and the two nearest levels are blended by its fraction.
// Continuous in the direction everywhere but at the eight corners, which is what lets a relaxed-arithmetic direction stay in tolerance.
// kCubeEdge[face][edge: 0 left (i = -1), 1 right (i = N), 2 up (j = -1), 3 down (j = N)] = face' | swap << 3 | flip << 4 | far << 5:
// the running coordinate k (j for left / right, i for up / down), reversed if flip, becomes j' (swap) or i'; the other one is N - 1 (far) or 0.
__device__ const uint8_t kCubeEdge[6][4] = {{44, 13, 58, 43}, {45, 12, 10, 27}, {1, 16, 21, 4}, {49, 32, 36, 53}, {41, 8, 34, 3}, {40, 9, 18, 51}};
AWSM_DI uint2 cube_texel_raw(const CubeDev& c, uint32_t level_base, int N, uint32_t face, int i, int j) {
    if (i < 0 || i >= N) j = min(max(j, 0), N - 1);     // corner taps keep their row
    if (i < 0 || i >= N || j < 0 || j >= N) {
        const uint32_t e = i < 0 ? 0u : (i >= N ? 1u : (j < 0 ? 2u : 3u));
        const uint32_t t = kCubeEdge[face][e];
        int k = e < 2u ? j : i;
        if (t & 16u) k = N - 1 - k;
        const int far = (t & 32u) ? N - 1 : 0;
        face = t & 7u;
        if (t & 8u) { i = far; j = k; } else { i = k; j = far; }
    }
    return c.texels[level_base + ((size_t)face * (size_t)N + (size_t)j) * (size_t)N + (size_t)i];
}
AWSM_DI f4 half4(uint2 h) { return {f16_bits_to_f32((unsigned short)(h.x & 0xFFFFu)), f16_bits_to_f32((unsigned short)(h.x >> 16)), f16_bits_to_f32((unsigned short)(h.y & 0xFFFFu)), f16_bits_to_f32((unsigned short)(h.y >> 16))}; }
AWSM_DI f4 cube_texel(const CubeDev& c, uint32_t level_base, int N, uint32_t face, int i, int j) { return half4(cube_texel_raw(c, level_base, N, face, i, j)); }
// CubeDev.bordered: one thread per texel of the aproned chain
__global__ __launch_bounds__(256) void k_cube_border(CubeDev c, uint2* __restrict__ out, uint32_t total) {
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= total) return;
    uint32_t level = 0u;
    while (level + 1u < c.mips && idx >= c.b_level_off[level + 1u]) level++;
    const int N = (int)max(c.size >> level, 1u), P = N + 2;
    const uint32_t r = idx - c.b_level_off[level], face = r / (uint32_t)(P * P), q = r % (uint32_t)(P * P);
    out[idx] = cube_texel_raw(c, c.level_off[level], N, face, (int)(q % (uint32_t)P) - 1, (int)(q / (uint32_t)P) - 1);
}
AWSM_DI f4 cube_level(const CubeDev& c, uint32_t level, f3 d) {
    const int N = (int)max(c.size >> level, 1u);
    const float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    uint32_t face; float sc, tc, ma;
    if (az >= ax && az >= ay) { face = d.z < 0.0f ? 5u : 4u; sc = d.z < 0.0f ? -d.x : d.x; tc = -d.y; ma = az; }
    else if (ay >= ax) { face = d.y < 0.0f ? 3u : 2u; sc = d.x; tc = d.y < 0.0f ? -d.z : d.z; ma = ay; }
    else { face = d.x < 0.0f ? 1u : 0u; sc = d.x < 0.0f ? d.z : -d.z; tc = -d.y; ma = ax; }
    const float inv = fm::rcp(ma);
    float x = (0.5f * (sc * inv) + 0.5f) * (float)N - 0.5f, y = (0.5f * (tc * inv) + 0.5f) * (float)N - 0.5f;
    if (!(x >= -0.5f)) x = -0.5f;                 // also NaN (zero / non-finite direction): the face's first texel
    if (!(y >= -0.5f)) y = -0.5f;
    x = fminf(x, (float)N - 0.5f); y = fminf(y, (float)N - 0.5f);
    const float flx = floorf(x), fly = floorf(y), fx = x - flx, fy = y - fly;
    const int i0 = (int)flx, j0 = (int)fly;
    const uint32_t base = c.level_off[level];
    const f4 c00 = cube_texel(c, base, N, face, i0, j0), c10 = cube_texel(c, base, N, face, i0 + 1, j0);
    const f4 c01 = cube_texel(c, base, N, face, i0, j0 + 1), c11 = cube_texel(c, base, N, face, i0 + 1, j0 + 1);
    return lerp4(lerp4(c00, c10, fx), lerp4(c01, c11, fx), fy);
}
__device__ __attribute__((noinline)) f4 sample_cube(const CubeDev* cp, f3 d, float level) {
    const CubeDev& c = *cp;
    const float top = (float)(c.mips - 1u);
    float lod = level > 0.0f ? level : 0.0f;          // also NaN
    lod = fminf(lod, top);
    const float fl = floorf(lod), fr = lod - fl;
    const uint32_t l0 = (uint32_t)fl, l1 = min(l0 + 1u, c.mips - 1u);
    f4 r = cube_level(c, l0, d);
    if (fr > 0.0f && l1 != l0) r = lerp4(r, cube_level(c, l1, d), fr);
    return r;
}
// skybox.wgsl:1-41
AWSM_DI f4 skybox_color(const DevScene* sc, const FrameDev& f, int cx, int cy) {
    if (!sc->cube[kCubeSkybox].texels) return {sc->skybox_rgba[0], sc->skybox_rgba[1], sc->skybox_rgba[2], sc->skybox_rgba[3]};
    const uint8_t* cam = f.camera;
    const m4 proj = load_m4(reinterpret_cast<const float*>(cam + 64)), inv_proj = load_m4(reinterpret_cast<const float*>(cam + 256)), inv_view = load_m4(reinterpret_cast<const float*>(cam + 320));
    const float ux = ((float)cx + 0.5f) * fm::rcp((float)f.width), uy = ((float)cy + 0.5f) * fm::rcp((float)f.height);
    const float nx = ux * 2.0f - 1.0f, ny = 1.0f - uy * 2.0f;
    f3 ray;
    if (proj.c[2].w != 0.0f) {
        const f4 vp = fm::fmul(inv_proj, {nx, ny, 0.0f, 1.0f});
        const float iw = fm::rcp(vp.w);
        ray = {vp.x * iw, vp.y * iw, vp.z * iw};
    } else ray = {nx, ny, -1.0f};
    const f3 w = {inv_view.c[0].x * ray.x + inv_view.c[1].x * ray.y + inv_view.c[2].x * ray.z, inv_view.c[0].y * ray.x + inv_view.c[1].y * ray.y + inv_view.c[2].y * ray.z,
                  inv_view.c[0].z * ray.x + inv_view.c[1].z * ray.y + inv_view.c[2].z * ray.z};
    return sample_cube(&sc->cube[kCubeSkybox], fm::fnormalize(w), 0.0f);
}
// brdf.wgsl:268-290
AWSM_DI f3 sample_irradiance(const DevScene* sc, f3 n) {
    if (!sc->cube[kCubeIrradiance].texels) return {sc->irradiance_rgb[0], sc->irradiance_rgb[1], sc->irradiance_rgb[2]};
    const f4 c = sample_cube(&sc->cube[kCubeIrradiance], n, 0.0f);
    return {c.x, c.y, c.z};
}
AWSM_DI f3 sample_prefiltered(const DevScene* sc, f3 dir, float roughness) {
    if (!sc->cube[kCubePrefiltered].texels) return {sc->prefiltered_rgb[0], sc->prefiltered_rgb[1], sc->prefiltered_rgb[2]};
    const uint32_t mip_count = reinterpret_cast<const uint32_t*>(sc->buf[AWSM_BUF_LIGHTS_INFO])[1];      // IblInfo.prefiltered_env_mip_count (lights.rs:300-305)
    const f4 c = sample_cube(&sc->cube[kCubePrefiltered], dir, roughness * (float)(mip_count - 1u));
    return {c.x, c.y, c.z};
}
AWSM_DI f3 reflect3(f3 i, f3 n) { return i - n * (2.0f * fm::fdot(n, i)); }

// Per-pixel terms shared by the IBL lobe and every punctual light (brdf_direct recomputes them per light in the WGSL;
// hoisting them is value-preserving up to rounding).
struct Surface {
    f3 n, v, F0;
    float metallic, roughness, alpha, f90, n_dot_v_ibl /* saturate */, n_dot_v_dir /* max(.,1e-4) */, sheen_scaling_dir, g1_v;
    f3 cc_n;
    // direct-light invariants
    f3 df90;            // f90 - F0
    f3 base_diffuse;    // base * (1 - metallic) / pi
    float a2, a2m1;     // GGX alpha^2 (alpha clamped at 0.001) and alpha^2 - 1
    float gk, one_m_gk; // Schlick-GGX k = (alpha + 1)^2 / 8
    bool has_sheen, has_clearcoat;
};

// brdf.wgsl:308-381 for one light.  `l` is the unit vector towards the light (the WGSL normalises it again on entry; callers here pass unit vectors).
AWSM_DI f3 brdf_direct(const PbrColor& c, const Surface& sf, f3 l, f3 radiance) {
    const f3 n = sf.n, v = sf.v;
    const f3 sum = v + l;
    const float len_sq = fm::fdot(sum, sum);
    const bool has_half = len_sq > 1e-8f;
    const float inv_len = has_half ? fm::rsq(len_sq) : 0.0f;          // h = sum * inv_len (never materialised for the base lobe)
    const float ndl_raw = fm::fdot(n, l);
    const float n_dot_l = fmaxf(ndl_raw, 0.0f);
    const float n_dot_v = sf.n_dot_v_dir;
    const float n_dot_h = fmaxf(fm::fdot(n, sum) * inv_len, 0.0f);
    const float v_dot_h = fmaxf(fm::fdot(v, sum) * inv_len, 0.0f);
    const float p5 = fm::pow5(1.0f - saturate(has_half ? v_dot_h : n_dot_v));   // brdf.wgsl:111-115
    const f3 F = {sf.F0.x + sf.df90.x * p5, sf.F0.y + sf.df90.y * p5, sf.F0.z + sf.df90.z * p5};
    const float ndh = saturate(n_dot_h);
    const float dd = (ndh * ndh) * sf.a2m1 + 1.0f;                      // brdf.wgsl:118-124
    const float D = sf.a2 * fm::rcp((kPi * dd) * dd + kEps);
    const float ndl = saturate(ndl_raw);
    const float g1_l = ndl * fm::rcp(ndl * sf.one_m_gk + sf.gk);        // brdf.wgsl:127-139
    const float spec = has_half ? (D * (sf.g1_v * g1_l)) * fm::rcp(fmaxf((4.0f * n_dot_l) * n_dot_v, kEps)) : 0.0f;
    const float k_d = 1.0f - fmaxf(fmaxf(F.x, F.y), F.z);
    const float w = n_dot_l * c.occlusion;
    f3 result = {((sf.base_diffuse.x * k_d + F.x * spec) * radiance.x) * w, ((sf.base_diffuse.y * k_d + F.y * spec) * radiance.y) * w,
                 ((sf.base_diffuse.z * k_d + F.z * spec) * radiance.z) * w};
    if (sf.has_sheen) {   // brdf.wgsl:198-240
        f3 sheen = {0.0f, 0.0f, 0.0f};
        if (has_half) {
            const float rough = fmaxf(c.sheen_roughness, 0.07f);
            const float inv_alpha = fm::rcp(rough * rough);
            const float sin2h = 1.0f - n_dot_h * n_dot_h;
            const float Ds = ((2.0f + inv_alpha) * fm::powp(sin2h, inv_alpha * 0.5f)) * (1.0f / (2.0f * kPi));
            const float V = fm::rcp(4.0f * ((n_dot_l + n_dot_v) - n_dot_l * n_dot_v));
            sheen = (c.sheen_color * Ds) * V;
        }
        result = result * sf.sheen_scaling_dir + ((sheen * radiance) * n_dot_l) * c.occlusion;
    }
    if (sf.has_clearcoat) {   // brdf.wgsl:149-190
        float clearcoat_spec = 0.0f;
        if (has_half) {
            const f3 h = sum * inv_len;
            const float cc_n_dot_l = fmaxf(fm::fdot(sf.cc_n, l), 0.0f);
            const float cc_n_dot_v = fmaxf(fm::fdot(sf.cc_n, v), 1e-4f);
            const float cc_n_dot_h = fmaxf(fm::fdot(sf.cc_n, h), 0.0f);
            const float cc_alpha = fmaxf(c.clearcoat_roughness * c.clearcoat_roughness, 0.001f);
            const float Fc = fresnel_schlick_scalar(v_dot_h, kClearcoatF0);
            const float Dc = distribution_ggx(cc_n_dot_h, cc_alpha);
            const float Gc = geometry_schlick_ggx(saturate(fm::fdot(sf.cc_n, v)), cc_alpha) * geometry_schlick_ggx(saturate(fm::fdot(sf.cc_n, l)), cc_alpha);
            clearcoat_spec = fm::fdiv(((c.clearcoat * Fc) * Dc) * Gc, fmaxf((4.0f * cc_n_dot_l) * cc_n_dot_v, kEps));
        }
        const float cc_fresnel = clearcoat_fresnel(c.clearcoat, v_dot_h);
        result = result * (1.0f - cc_fresnel) + (radiance * clearcoat_spec) * n_dot_l;
    }
    return result;
}

// brdf.wgsl:389-576 (brdf_ibl -> brdf_ibl_with_transmission)
AWSM_DI f3 brdf_ibl(const DevScene* sc, const PbrColor& c, const Surface& sf, f3 transmission_background) {
    const f3 prefiltered = sample_prefiltered(sc, reflect3(-sf.v, sf.n), sf.roughness);
    const f3 irradiance = sample_irradiance(sc, sf.n);
    const float n_dot_v = sf.n_dot_v_ibl;
    const f3 F_view = fresnel_schlick_f90(n_dot_v, sf.F0, sf.f90);
    const float F_view_max = fmaxf(fmaxf(F_view.x, F_view.y), F_view.z);
    const float effective_transmission = c.transmission * (1.0f - sf.metallic);
    f3 base_layer = (c.base * (1.0f / kPi)) * irradiance;
    if (effective_transmission > 0.0f) {
        f3 attenuation = splat3(1.0f);
        if (should_apply_volume_attenuation(c.volume_thickness, c.volume_attenuation_distance, c.volume_attenuation_color))
            attenuation = volume_attenuation(c.volume_thickness, c.volume_attenuation_color, c.volume_attenuation_distance);
        const f3 transmission_btdf = (transmission_background * c.base) * attenuation;   // brdf.wgsl:531-561 (opaque pass: the uniform cube)
        base_layer = mix3(base_layer, transmission_btdf, effective_transmission);
    }
    const float k_d = (1.0f - F_view_max) * (1.0f - sf.metallic);
    const f3 base_contribution = (base_layer * k_d) * c.occlusion;
    const f2 lut = sample_brdf_lut(sc, n_dot_v, sf.roughness);
    const f3 spec_term = sf.F0 * lut.x + splat3(sf.f90 * lut.y);
    const f3 specular = (prefiltered * spec_term) * mixf(1.0f, c.occlusion, 0.5f);
    f3 base_with_sheen = base_contribution * sheen_albedo_scaling(c.sheen_color, c.sheen_roughness, n_dot_v);
    if (c.sheen_color.x > 0.0f || c.sheen_color.y > 0.0f || c.sheen_color.z > 0.0f) {
        const float alpha = c.sheen_roughness * c.sheen_roughness;
        const float om = 1.0f - n_dot_v;
        base_with_sheen = base_with_sheen + (((c.sheen_color * irradiance) * alpha) * (om * om * om)) * c.occlusion;
    }
    f3 result = (base_with_sheen + specular) + c.emissive;
    if (c.clearcoat > 0.0f) {
        const float cc_n_dot_v = saturate(fm::fdot(sf.cc_n, sf.v));
        const float cc_roughness = fmaxf(c.clearcoat_roughness, 0.04f);
        const f2 cc_lut = sample_brdf_lut(sc, cc_n_dot_v, cc_roughness);
        const f3 cc_prefiltered = sc->cube[kCubePrefiltered].texels ? sample_prefiltered(sc, reflect3(-sf.v, sf.cc_n), cc_roughness) : prefiltered;
        const f3 cc_specular = cc_prefiltered * (kClearcoatF0 * cc_lut.x + cc_lut.y);
        const float cc_fresnel = clearcoat_fresnel(c.clearcoat, n_dot_v);
        result = result * (1.0f - cc_fresnel) + cc_specular * c.clearcoat;
    }
    return result;
}

// lights.wgsl:70-152
AWSM_DI f3 apply_lighting(const DevScene* sc, const float4* __restrict__ lights_pre, const PbrColor& mc, f3 surface_to_camera, f3 world_position, uint32_t n_lights, f3 transmission_background) {
    Surface sf;
    sf.n = fm::fsafe_normalize(mc.normal);
    sf.v = fm::fsafe_normalize(surface_to_camera);
    sf.metallic = clampf(mc.mr.x, 0.0f, 1.0f);
    sf.roughness = fmaxf(clampf(mc.mr.y, 0.0f, 1.0f), 0.04f);
    sf.alpha = sf.roughness * sf.roughness;
    const float ndv = fm::fdot(sf.n, sf.v);
    sf.n_dot_v_ibl = saturate(ndv);
    sf.n_dot_v_dir = fmaxf(ndv, 1e-4f);
    const float f0b = ior_to_f0(mc.ior);
    const f3 dielectric_f0 = min3(splat3(f0b) * mc.specular_color, splat3(1.0f)) * mc.specular;
    sf.F0 = mix3(dielectric_f0, mc.base, sf.metallic);
    sf.f90 = mixf(mc.specular, 1.0f, sf.metallic);
    sf.sheen_scaling_dir = sheen_albedo_scaling(mc.sheen_color, mc.sheen_roughness, sf.n_dot_v_dir);
    sf.g1_v = geometry_schlick_ggx(saturate(ndv), sf.alpha);
    sf.cc_n = fm::fsafe_normalize(mc.clearcoat_normal);
    sf.df90 = splat3(sf.f90) - sf.F0;
    sf.base_diffuse = mc.base * ((1.0f - sf.metallic) * (1.0f / kPi));
    const float ac = fmaxf(sf.alpha, 0.001f);
    sf.a2 = ac * ac; sf.a2m1 = sf.a2 - 1.0f;
    sf.gk = ((ac + 1.0f) * (ac + 1.0f)) * 0.125f; sf.one_m_gk = 1.0f - sf.gk;
    sf.has_sheen = mc.sheen_color.x > 0.0f || mc.sheen_color.y > 0.0f || mc.sheen_color.z > 0.0f;
    sf.has_clearcoat = mc.clearcoat > 0.0f;

    f3 color = brdf_ibl(sc, mc, sf, transmission_background);
    const float4* lights = reinterpret_cast<const float4*>(sc->buf[AWSM_BUF_LIGHTS]);
    for (uint32_t i = 0; i < n_lights; i++) {
        const float4 pre0 = lights_pre[i * 2], pre1 = lights_pre[i * 2 + 1];      // k_resolve_draws: unit direction / spot axis + kind, colour * intensity
        const uint32_t kind = (uint32_t)pre0.w;
        f3 light_dir = {pre0.x, pre0.y, pre0.z}, radiance = {pre1.x, pre1.y, pre1.z};
        if (kind == 2u || kind == 3u) {
            const float4 pos_range = lights[i * 4 + 0];
            const f3 stl = mk3(pos_range.x, pos_range.y, pos_range.z) - world_position;
            const float d2 = fm::fdot(stl, stl);
            const float inv_d = d2 > 0.0f ? fm::rsq(d2) : 0.0f;
            const float dist = d2 * inv_d;
            float att;   // math.wgsl:12-19 inverse_square
            if (pos_range.w == 0.0f) att = fm::rcp(fmaxf(dist * dist, 0.01f));
            else { const float fo = 1.0f - fm::fdiv(dist * dist, pos_range.w * pos_range.w); att = fm::fdiv(saturate(fo * fo), dist * dist + 1.0f); }
            const f3 to_light = stl * inv_d;
            if (kind == 3u) {
                const float4 dir_inner = lights[i * 4 + 1], kind_outer = lights[i * 4 + 3];
                const float cos_l = fm::fdot(to_light, -light_dir);                 // light_dir holds the unit spot axis here
                const float sm = saturate(fm::fdiv(cos_l - kind_outer.y, dir_inner.w - kind_outer.y));
                att = att * (sm * sm);
            }
            light_dir = to_light;
            radiance = radiance * att;
        } else if (kind != 1u) { light_dir = {0.0f, 0.0f, 0.0f}; radiance = {0.0f, 0.0f, 0.0f}; }
        color = color + brdf_direct(mc, sf, light_dir, radiance);
    }
    return color;
}

AWSM_DI void store_pixel(const FrameDev& f, size_t p, f4 c) {
    ushort4 h = make_ushort4(f16_bits(c.x), f16_bits(c.y), f16_bits(c.z), f16_bits(c.w));
    __builtin_nontemporal_store(*reinterpret_cast<unsigned long long*>(&h), reinterpret_cast<unsigned long long*>(f.out_rgba16f) + p);      // the image is not read again by this pass
    if (f.out_rgba32f) reinterpret_cast<float4*>(f.out_rgba32f)[p] = make_float4(c.x, c.y, c.z, c.w);
}

// One of the five core textures of a draw's material, ready to sample (TexSlotDev).  words = the TextureInfo word index of each.
AWSM_DI TexSlotDev resolve_tex_slot(const DevScene* __restrict__ sc, const uint32_t* __restrict__ M, const uint32_t (&words)[kCoreTextures], int k, bool unlit) {
    TexSlotDev s;
    s.base = nullptr; s.width = 0u; s.height = 0u; s.flags = 0u; s.layer_levels = 0u; s.level_off = nullptr; s.array_base = nullptr;
    for (int j = 0; j < 6; j++) s.tt[j] = 0.0f;
    if (!(unlit && k >= 2)) {
        const TexInfo t = tex_load(M, words[k]);
        if (t.exists) {
            s.flags = 1u | (t.uv_set_index << 24);
            const float* tt = reinterpret_cast<const float*>(sc->buf[AWSM_BUF_TEXTURE_TRANSFORMS] + (size_t)t.uv_transform_index * 32u);
            for (int j = 0; j < 6; j++) s.tt[j] = tt[j];
            bool ok = t.array_index < sc->n_tex && t.sampler_index < sc->n_samplers;
            if (ok) {
                const TexArrayDev& arr = sc->tex[t.array_index];
                const AwsmSampler& smp = sc->samplers[t.sampler_index];
                ok = arr.texels != nullptr && arr.width != 0u && arr.height != 0u && arr.layers != 0u;
                if (ok) {
                    const uint32_t layer = min(t.layer_index, arr.layers - 1u);
                    s.base = reinterpret_cast<const uint32_t*>(arr.texels) + (size_t)layer * arr.width * arr.height;
                    s.width = arr.width; s.height = arr.height;
                    const bool common = smp.address_mode_u == 1u && smp.address_mode_v == 1u && (arr.width & (arr.width - 1u)) == 0u && (arr.height & (arr.height - 1u)) == 0u;
                    if (common && smp.mag_filter != 0u) s.flags |= 2u;
                    if (common) s.flags |= 8u;
                    s.flags |= (smp.mag_filter != 0u ? 16u : 0u) | (smp.min_filter != 0u ? 32u : 0u) | (smp.mipmap_filter != 0u ? 64u : 0u);
                    s.flags |= ((smp.address_mode_u & 3u) << 13) | ((smp.address_mode_v & 3u) << 21);
                    // layer (< 65536) | max_anisotropy 1..16, counted only with three linear filters (SamplerCacheKey::allowed_ansiotropy) << 16 | levels << 24
                    const uint32_t an = (smp.mag_filter != 0u && smp.min_filter != 0u && smp.mipmap_filter != 0u) ? min(max(smp.max_anisotropy, 1u), 16u) : 1u;
                    s.layer_levels = (layer & 0xFFFFu) | (an << 16) | (max(arr.mips, 1u) << 24);
                    s.level_off = &sc->tex[t.array_index].level_off[0];
                    s.array_base = reinterpret_cast<const uint32_t*>(arr.texels);
                }
            }
            if (!ok) s.flags |= 4u;
        }
    }
    return s;
}

// ------------------------------------------------------------------------------------------------
// k_resolve_draws: one thread per draw.  Follows geometry meta -> material mesh meta once per frame
// (compute.wgsl:171-181 does it per pixel) and leaves the per-draw constants of the opaque pass in one 32-byte record.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_resolve_draws(const DevScene* __restrict__ sc, FrameDev f) {
    const uint32_t d = blockIdx.x * 256u + threadIdx.x;
    // Per-light constants, once per frame instead of once per pixel and light (lights.wgsl:70-118 recomputes them in every invocation):
    // the unit vector towards a directional light / the unit axis of a spot, and colour * intensity.
    if (f.lights_pre && sc->buf[AWSM_BUF_LIGHTS_INFO] && sc->buf[AWSM_BUF_LIGHTS]) {
        const uint32_t n_lights = min(*reinterpret_cast<const uint32_t*>(sc->buf[AWSM_BUF_LIGHTS_INFO]), f.lights_cap);
        const float4* lights = reinterpret_cast<const float4*>(sc->buf[AWSM_BUF_LIGHTS]);
        for (uint32_t i = d; i < n_lights; i += gridDim.x * 256u) {
            const float4 dir_inner = lights[i * 4 + 1], color_intensity = lights[i * 4 + 2], kind_outer = lights[i * 4 + 3];
            const uint32_t kind = (uint32_t)kind_outer.x;
            f3 v = {dir_inner.x, dir_inner.y, dir_inner.z};
            if (kind == 1u) v = fm::fsafe_normalize(-v);          // towards the light
            else v = fm::fnormalize(v);                           // spot axis (unused for point lights)
            f.lights_pre[i * 2] = make_float4(v.x, v.y, v.z, kind_outer.x);
            f.lights_pre[i * 2 + 1] = make_float4(color_intensity.x * color_intensity.w, color_intensity.y * color_intensity.w, color_intensity.z * color_intensity.w, 0.0f);
        }
    }
    // eight threads per draw: roles 0..4 resolve one core texture each, role 5 the per-draw records
    if (d == 0u && f.shade_todo) f.shade_todo[0] = 0u;      // the list k_shade_lean leaves for k_shade_todo (same stream, this frame)
    if (d < 64u && f.lean_next) f.lean_next[d * 16u] = 0u;  // the persistent lean grid's strip counters
    const uint32_t draw = d >> 3, role = d & 7u;
    if (draw >= f.n_draws || role > 6u) return;
    const DrawDev dr = f.draws[draw];
    const uint32_t material_meta_offset = *reinterpret_cast<const uint32_t*>(sc->buf[AWSM_BUF_GEOM_META] + dr.geom_meta_off + 36);
    const uint32_t* mm = reinterpret_cast<const uint32_t*>(sc->buf[AWSM_BUF_MATERIAL_META] + (size_t)(material_meta_offset / 256u) * 256u);
    const uint32_t* M = reinterpret_cast<const uint32_t*>(sc->buf[AWSM_BUF_MATERIALS]);
    const uint32_t material_word = mm[6] / 4u, b = material_word + 1u;
    const uint32_t shader_id = M[material_word];
    const bool unlit = shader_id == 2u;
    const uint32_t words[kCoreTextures] = {b + 2u, b + 11u, b + 18u, b + 24u, b + 30u};
    if (role == 6u) {   // the lean route's record (LeanDrawDev)
        if (!f.draw_lean) return;
        LeanDrawDev L;
        L.flags = 0u; L.normal_bias = 1.0f; L.occlusion_bias = 1.0f; L.pad1 = 0u; L.pad2[0] = 0u; L.pad2[1] = 0u;
        L.tt[0] = 1.0f; L.tt[1] = 0.0f; L.tt[2] = 0.0f; L.tt[3] = 1.0f; L.tt[4] = 0.0f; L.tt[5] = 0.0f;
        bool have_tt = false;
        L.metallic = 0.0f; L.roughness = 0.0f; L.normal_scale = 1.0f; L.occlusion_strength = 1.0f;
        for (int j = 0; j < 3; j++) { L.base_color[j] = 0.0f; L.emissive[j] = 0.0f; }
        for (int k = 0; k < kCoreTextures; k++) { L.tex[k][0] = 0u; L.tex[k][1] = 0u; for (int j = 0; j < 4; j++) L.gtex[k][j] = 0u; }
        bool lean_grad = true, lean_aniso = true;      // ... under MipmapMode::Gradient; ... with anisotropic probes too: one max_anisotropy for all its textures
        uint32_t aniso = 0u;
        bool lean = shader_id == 1u && (mm[16] & 1u) == 0u && M[b + 38] == 0u;     // PBR, not a hud mesh, no debug view
        if (lean) {
            const uint32_t fi = b + 39u;
            for (int j = 0; j < 12; j++) if (M[fi + j] != 0u) lean = false;      // no optional block
        }
        if (lean) {
            uint32_t exists = 0u;
            for (int k = 0; k < kCoreTextures && lean; k++) {
                const TexSlotDev s = resolve_tex_slot(sc, M, words, k, false);
                if (!(s.flags & 1u)) continue;
                const unsigned long long addr = (unsigned long long)s.base;
                bool same_tt = true;      // one transform for all of the draw's textures (bit for bit; the first one's)
                for (int j = 0; j < 6; j++) { if (!have_tt) L.tt[j] = s.tt[j]; else if (__float_as_uint(L.tt[j]) != __float_as_uint(s.tt[j])) same_tt = false; }
                have_tt = true;
                if ((s.flags & 6u) != 2u || (s.flags >> 24) != 0u || !same_tt || s.width > 32768u || s.height > 32768u || (addr >> 48) != 0ull || (addr & 3ull) != 0ull) { lean = false; break; }
                exists |= 1u << k;
                L.tex[k][0] = (uint32_t)addr;
                L.tex[k][1] = (uint32_t)(addr >> 32) | ((uint32_t)(31 - __clz((int)s.width)) << 16) | ((uint32_t)(31 - __clz((int)s.height)) << 20);
                {   // MipmapMode::Gradient: square layers, linear min + mipmap filters, the whole chain addressable with 32-bit byte offsets
                    const TexInfo ti = tex_load(M, words[k]);
                    const TexArrayDev& arr = sc->tex[ti.array_index];
                    const unsigned long long abase = (unsigned long long)s.array_base;
                    const uint32_t levels = s.layer_levels >> 24, layer = s.layer_levels & 0xFFFFu, lw = (uint32_t)(31 - __clz((int)s.width));
                    unsigned long long chain_texels = 0ull;
                    for (uint32_t l = 0; l < levels; l++) chain_texels += (unsigned long long)arr.layers * max(arr.width >> l, 1u) * max(arr.height >> l, 1u);
                    const bool ok = s.width == s.height && (s.flags & (32u | 64u)) == (32u | 64u) && levels >= 1u && levels <= 15u && levels <= lw + 1u && arr.layers < 4096u &&
                                    chain_texels * 4ull < 0xFFFFFFF0ull && (abase >> 48) == 0ull && (abase & 3ull) == 0ull;
                    if (!ok) lean_grad = false;
                    const uint32_t an = max((s.layer_levels >> 16) & 31u, 1u);
                    if (aniso != 0u && an != aniso) lean_aniso = false;
                    aniso = an;
                    L.gtex[k][0] = (uint32_t)abase; L.gtex[k][1] = (uint32_t)(abase >> 32) | (levels << 16) | (lw << 24);
                    L.gtex[k][2] = layer; L.gtex[k][3] = arr.layers;
                }
            }
            if (lean) {
                const bool identity = L.tt[0] == 1.0f && L.tt[1] == 0.0f && L.tt[2] == 0.0f && L.tt[3] == 1.0f && L.tt[4] == 0.0f && L.tt[5] == 0.0f;
                L.flags = 1u | (lean_grad ? 2u : 0u) | (lean_grad && lean_aniso ? 4u : 0u) | (identity ? 0u : 8u) | (exists << 8);
                L.pad1 = f.aniso ? max(aniso, 1u) : 1u;      // max_anisotropy of the draw's textures (k_shade_lean<.., 2, ..>)
                // factors ready for raw 0..255 bilinear sums wherever the texture exists (LeanDrawDev)
                const float k255 = 1.0f / 255.0f;
                const float s_base = (exists & 1u) ? k255 : 1.0f, s_mr = (exists & 2u) ? k255 : 1.0f, s_em = (exists & 16u) ? k255 : 1.0f;
                L.metallic = mf(M, b + 16) * s_mr; L.roughness = mf(M, b + 17) * s_mr;
                const float nscale = mf(M, b + 23), ostrength = mf(M, b + 29);
                L.normal_scale = (nscale * 2.0f) * k255; L.normal_bias = nscale;
                L.occlusion_strength = (exists & 8u) ? ostrength * k255 : 0.0f; L.occlusion_bias = (exists & 8u) ? 1.0f - ostrength : 1.0f;
                const float strength = M[b + 39u + 1u] != 0u ? mf(M, b + M[b + 39u + 1u]) : 1.0f;    // (no optional block: 1)
                for (int j = 0; j < 3; j++) { L.base_color[j] = mf(M, b + 7 + j) * s_base; L.emissive[j] = (mf(M, b + 35 + j) * strength) * s_em; }
            }
        }
        if (!(L.flags & 1u)) for (int k = 0; k < kCoreTextures; k++) { L.tex[k][0] = 0u; L.tex[k][1] = 0u; for (int j = 0; j < 4; j++) L.gtex[k][j] = 0u; }
        f.draw_lean[draw] = L;
        return;
    }
    if (role == 5u) {
        DrawShadeDev o;
        o.first_tri = dr.first_tri;
        o.material_word = material_word;
        o.attr_indices_word = mm[9] / 4u; o.attr_data_word = mm[10] / 4u; o.stride_words = mm[11] / 4u;
        o.uv_sets_index = mm[12];
        // bit 0: hud mesh; bit 1: ALPHA_MODE_MASK material (the transparent pass may discard its fragments, so their depth write waits for the shading)
        o.flags = (mm[16] & 1u) | (M[b] == 1u ? 2u : 0u);
        o.color_sets = mm[14];
        f.draw_shade[draw] = o;
        DrawMatDev m;
        for (int j = 0; j < 4; j++) m.base_color[j] = mf(M, b + 7 + j);
        m.shader_alpha = shader_id | (M[b] << 8);
        m.alpha_cutoff = mf(M, b + 1);
        m.ext_mask = 0u; m.debug_bitmask = 0u; m.ior = 1.5f;
        m.metallic = 0.0f; m.roughness = 0.0f; m.normal_scale = 1.0f; m.occlusion_strength = 1.0f;
        if (unlit) {
            for (int j = 0; j < 3; j++) m.emissive[j] = mf(M, b + 16 + j);
        } else {
            m.metallic = mf(M, b + 16); m.roughness = mf(M, b + 17); m.normal_scale = mf(M, b + 23); m.occlusion_strength = mf(M, b + 29);
            m.debug_bitmask = M[b + 38];
            const uint32_t fi = b + 39u;
            for (int j = 0; j < 12; j++) if (M[fi + j] != 0u) m.ext_mask |= 1u << j;
            const float strength = M[fi + 1] != 0u ? mf(M, b + M[fi + 1]) : 1.0f;
            for (int j = 0; j < 3; j++) m.emissive[j] = mf(M, b + 35 + j) * strength;
            if (M[fi + 2] != 0u) m.ior = mf(M, b + M[fi + 2]);
        }
        f.draw_mat[draw] = m;
        return;
    }
    // one of the five core textures of the draw's material, ready to sample (TexSlotDev)
    const int k = (int)role;
    TexSlotDev s = resolve_tex_slot(sc, M, words, k, unlit);
    if (k == 0) {   // slot 0 also carries which of the five exist and which use TEXCOORD_0
        uint32_t exists_mask = 0u, uv0_mask = 0u;
        for (int j = 0; j < kCoreTextures; j++) {
            if (unlit && j >= 2) break;
            const uint32_t uv_and_sampler = M[words[j] + 2], extra = M[words[j] + 3];
            if (extra & 1u) { exists_mask |= 1u << j; if ((uv_and_sampler & 0xFFu) == 0u) uv0_mask |= 1u << j; }
        }
        s.flags |= (exists_mask << 8) | (uv0_mask << 16);
    }
    f.tex_slots[(size_t)draw * kCoreTextures + k] = s;
}

// fragment.wgsl:27-186 (transparent pass): the opaque image behind a transmissive surface, refracted through the volume
// (KHR_materials_volume) and blurred by roughness with up to three rings of eight taps (transmission_blur_rings = 3,
// material_transparent/shader/template.rs:170).  Outside the screen: the (uniform) prefiltered environment.
struct OpaqueImage { const uint2* texels; const uint8_t* camera; int width, height; };      // by value: a FrameDev& into a real call would spill all of it
AWSM_DI f3 opaque_texel(const OpaqueImage& f, int x, int y) {
    x = min(max(x, 0), f.width - 1); y = min(max(y, 0), f.height - 1);
    const uint2 h = f.texels[(size_t)y * (size_t)f.width + (size_t)x];
    return {f16_bits_to_f32((unsigned short)(h.x & 0xFFFFu)), f16_bits_to_f32((unsigned short)(h.x >> 16)), f16_bits_to_f32((unsigned short)(h.y & 0xFFFFu))};
}
__device__ __attribute__((noinline)) f3 sample_transmission_background(const DevScene* sc, const OpaqueImage f, float frag_x, float frag_y, f3 world_position, f3 normal,
                                                                        f3 view_dir, float ior, float roughness, float thickness) {
    const float Wf = (float)f.width, Hf = (float)f.height;
    f2 screen_uv = {frag_x / Wf, frag_y / Hf};
    const float ior_val = ior < 1.0f ? 1.5f : ior;
    f3 sample_dir = view_dir;            // fragment.wgsl:39,50: direction of the IBL fallback
    if (thickness > 0.0f && ior_val != 1.0f) {
        // brdf.wgsl:30-47 refract_direction
        const float eta = 1.0f / ior_val;
        const float cos_i = -fm::fdot(normal, view_dir);
        const float k = 1.0f - eta * eta * (1.0f - cos_i * cos_i);
        f3 refracted = {0.0f, 0.0f, 0.0f};
        if (!(k < 0.0f)) refracted = view_dir * eta + normal * (eta * cos_i - sqrtf(k));
        if (fm::fdot(refracted, refracted) > 1e-6f) {
            sample_dir = refracted;
            const f3 exit = world_position + normalize(refracted) * thickness;
            const m4 view_proj = load_m4(reinterpret_cast<const float*>(f.camera + 128));
            const f4 clip_pos = mul(view_proj, {exit.x, exit.y, exit.z, 1.0f});
            screen_uv = {(clip_pos.x / clip_pos.w + 1.0f) * 0.5f, (1.0f - clip_pos.y / clip_pos.w) * 0.5f};
        }
    }
    if (!(screen_uv.x >= 0.0f && screen_uv.x <= 1.0f && screen_uv.y >= 0.0f && screen_uv.y <= 1.0f))
        return sample_prefiltered(sc, sample_dir, roughness);      // fragment.wgsl:68-81
    const float sx = screen_uv.x * Wf, sy = screen_uv.y * Hf;
    const int tx = (int)sx, ty = (int)sy;
    const float blur_roughness = roughness * clampf(ior * 2.0f - 2.0f, 0.0f, 1.0f);
    if (blur_roughness > 0.05f) {
        const float target_mip = __builtin_amdgcn_logf(Wf) * blur_roughness;
        const float blur_radius = __builtin_amdgcn_exp2f(clampf(target_mip, 0.0f, 8.0f));
        const float sigma = blur_radius * 0.5f, sigma_sq_2 = 2.0f * sigma * sigma;
        f3 sum = opaque_texel(f, tx, ty);
        float wsum = 1.0f;
        for (int k = 0; k < 3; k++) {
            const float r = blur_radius * (k == 0 ? 0.33f : (k == 1 ? 0.67f : 1.0f));
            const float w = __expf(-(r * r) / sigma_sq_2);
            for (int i = 0; i < 8; i++) {
                const float ox = (i == 0 ? 1.0f : i == 1 ? 0.707f : i == 2 ? 0.0f : i == 3 ? -0.707f : i == 4 ? -1.0f : i == 5 ? -0.707f : i == 6 ? 0.0f : 0.707f);
                const float oy = (i == 0 ? 0.0f : i == 1 ? 0.707f : i == 2 ? 1.0f : i == 3 ? 0.707f : i == 4 ? 0.0f : i == 5 ? -0.707f : i == 6 ? -1.0f : -0.707f);
                const float fx = sx + ox * r, fy = sy + oy * r;
                const int cx = (int)fx, cy = (int)fy;           // vec2<i32>(): truncation toward zero
                if (cx >= 0 && cx <= f.width - 1 && cy >= 0 && cy <= f.height - 1) { sum = sum + opaque_texel(f, cx, cy) * w; wsum += w; }
            }
        }
        return sum * fm::rcp(wsum);
    }
    return opaque_texel(f, tx, ty);
}

// The material half of the shading, shared by the opaque pass (FWD = false: compute.wgsl:212-299, material_color_calc.wgsl of
// material_opaque) and the transparent pass (FWD = true: fragment.wgsl:217-281, material_color_calc.wgsl of
// material_transparent): same textures, factors and lighting; they differ in the alpha rules, the vertex-colour rule and
// where the transmission background comes from.  out.color.w = alpha.
struct SurfaceOut { f4 color; uint32_t kind; bool discard; };
template <int GRAD, bool FWD>
AWSM_DI SurfaceOut shade_material(const DevScene* __restrict__ sc, const FrameDev& f, Attr& a, uint32_t material_word, const TexSlotDev* __restrict__ slots,
                                  const DrawMatDev* __restrict__ draw_mat, const TBN& tbn,
                                  f3 world_position, f3 surface_to_camera, uint32_t color_sets, float frag_x, float frag_y, bool mask_resolved = false) {
    SurfaceOut out;
    out.color = {0.0f, 0.0f, 0.0f, 0.0f};
    out.kind = 0u; out.discard = false;
    const uint32_t n_lights = min(*reinterpret_cast<const uint32_t*>(sc->buf[AWSM_BUF_LIGHTS_INFO]), f.lights_cap);
    const uint32_t* M = reinterpret_cast<const uint32_t*>(sc->buf[AWSM_BUF_MATERIALS]);
    const uint32_t b = material_word + 1u;
    // the factor half of the material: four aligned loads of the per-draw record instead of ~25 scattered words of the stream
    const float4* dmq = reinterpret_cast<const float4*>(draw_mat);
    const float4 dm0 = dmq[0], dm1 = dmq[1], dm2 = dmq[2], dm3 = dmq[3];   // base colour | metallic roughness normal_scale occlusion_strength | emissive ior | shader_alpha debug ext cutoff
    const uint32_t shader_alpha = __float_as_uint(dm3.x), shader_id = shader_alpha & 0xFFu, alpha_mode = shader_alpha >> 8;
    const float alpha_cutoff = dm3.w;
    if (shader_id == 2u) {   // unlit_material.wgsl:28-73 + material_color_calc.wgsl:517-580
        f4 base = {dm0.x, dm0.y, dm0.z, dm0.w};
        f3 em = {dm2.x, dm2.y, dm2.z};
        {
            const uint32_t em_mask = slots[0].flags >> 8;
            if (em_mask & 1u) { const f4 s = sample_slot<GRAD>(a, slots, M, b + 2); base = {base.x * s.x, base.y * s.y, base.z * s.z, base.w * s.w}; }
            if (em_mask & 2u) { const f4 s = sample_slot<GRAD>(a, slots + 1, M, b + 11); em = {em.x * s.x, em.y * s.y, em.z * s.z}; }
        }
        float alpha = 1.0f;
        if (FWD) {            // transparent material_color_calc.wgsl:344-372: alpha kept; ALPHA_MODE_MASK discards or forces 1
            if (alpha_mode == 1u) { if (!mask_resolved && base.w < alpha_cutoff) { out.discard = true; return out; } base.w = 1.0f; }
            alpha = base.w;
        }
        out.color = {base.x + em.x, base.y + em.y, base.z + em.z, alpha};
        return out;
    }

    // ---- pbr_material.wgsl:110-216 + material_color_calc.wgsl:25-265 ----
    PbrColor c;
    float base_alpha = 1.0f;
    const uint32_t debug_bitmask = __float_as_uint(dm3.y), ext_mask = __float_as_uint(dm3.z);
    uint32_t idx_vertex_color = 0u, idx_specular = 0u, idx_transmission = 0u, idx_volume = 0u, idx_clearcoat = 0u, idx_sheen = 0u;
    if (ext_mask != 0u) {     // optional blocks: their word indices come from the stream (pbr.rs:358-362); most materials have none
        const uint32_t fi = b + 39u;
        idx_vertex_color = abs_index(b, M[fi + 0]); idx_specular = abs_index(b, M[fi + 3]); idx_transmission = abs_index(b, M[fi + 4]);
        idx_volume = abs_index(b, M[fi + 6]); idx_clearcoat = abs_index(b, M[fi + 7]); idx_sheen = abs_index(b, M[fi + 8]);
    }
    // the five core textures, through the draw's resolved slots
    const uint32_t slot_flags = slots[0].flags;
    const uint32_t exists_mask = (slot_flags >> 8) & 31u, uv0_mask = (slot_flags >> 16) & 31u;
    auto core = [&](int k, uint32_t word) -> f4 { return sample_slot<GRAD>(a, slots + k, M, word); };
    // TEXCOORD_0 is interpolated once for all the textures that use it (the WGSL re-derives it per texture)
    if (uv0_mask != 0u) {
        a.uv0 = attr_uv<GRAD>(a, 0u, a.duv0_dx, a.duv0_dy);
        a.has_uv0 = true;
    }
    {
        f4 base = {dm0.x, dm0.y, dm0.z, dm0.w};
        if (exists_mask & 1u) { const f4 s = core(0, b + 2); base = {base.x * s.x, base.y * s.y, base.z * s.z, base.w * s.w}; }
        if (!FWD) {
            base.w = 1.0f;
            if (idx_vertex_color != 0u) { const f4 vc = vertex_color(a, M[idx_vertex_color]); base = {base.x * vc.x, base.y * vc.y, base.z * vc.z, base.w * vc.w}; }
        } else {
            // transparent material_color_calc.wgsl:38-52: a mesh with colour sets always multiplies (set 0 unless the material names
            // one; a set the mesh lacks reads as 1); alpha is kept; ALPHA_MODE_MASK discards below the cutoff, else alpha = 1
            if (color_sets != 0u) {
                const uint32_t set_index = idx_vertex_color != 0u ? M[idx_vertex_color] : 0u;
                if (set_index < color_sets) { const f4 vc = vertex_color(a, set_index); base = {base.x * vc.x, base.y * vc.y, base.z * vc.z, base.w * vc.w}; }
            }
            if (alpha_mode == 1u) { if (!mask_resolved && base.w < alpha_cutoff) { out.discard = true; return out; } base.w = 1.0f; }
            base_alpha = base.w;
        }
        c.base = {base.x, base.y, base.z};
    }
    c.mr = {dm1.x, dm1.y};
    if (exists_mask & 2u) { const f4 s = core(1, b + 11); c.mr = {c.mr.x * s.z, c.mr.y * s.y}; }
    c.normal = tbn.N;
    if (exists_mask & 4u) {   // material_color_calc.wgsl:301-322
        const f4 s = core(2, b + 18);
        const float scale = dm1.z;
        const float ntx = (s.x * 2.0f - 1.0f) * scale, nty = (s.y * 2.0f - 1.0f) * scale, ntz = s.z * 2.0f - 1.0f;
        c.normal = fm::fnormalize(tbn.T * ntx + tbn.B * nty + tbn.N * ntz);
    }
    c.occlusion = 1.0f;
    if (exists_mask & 8u) { const f4 s = core(3, b + 24); c.occlusion = mixf(1.0f, s.x, dm1.w); }
    {
        f3 em = {dm2.x, dm2.y, dm2.z};                 // factor * emissive_strength (k_resolve_draws)
        if (exists_mask & 16u) { const f4 s = core(4, b + 30); em = {em.x * s.x, em.y * s.y, em.z * s.z}; }
        c.emissive = em;
    }
    c.ior = dm2.w;
    c.specular = 1.0f; c.specular_color = {1.0f, 1.0f, 1.0f};
    if (idx_specular != 0u) {
        const uint32_t i = idx_specular;
        const TexInfo tx = tex_load(M, i), ctx = tex_load(M, i + 6);
        c.specular = mf(M, i + 5);
        if (tx.exists) c.specular = c.specular * sample_tex<GRAD>(a, tx).w;
        c.specular_color = {mf(M, i + 11), mf(M, i + 12), mf(M, i + 13)};
        if (ctx.exists) { const f4 s = sample_tex<GRAD>(a, ctx); c.specular_color = {c.specular_color.x * s.x, c.specular_color.y * s.y, c.specular_color.z * s.z}; }
    }
    c.transmission = 0.0f;
    if (idx_transmission != 0u) {
        const uint32_t i = idx_transmission;
        const TexInfo tx = tex_load(M, i);
        const float factor = mf(M, i + 5);
        if (!(!tx.exists && factor == 0.0f)) { c.transmission = factor; if (tx.exists) c.transmission = c.transmission * sample_tex<GRAD>(a, tx).x; }
    }
    c.volume_thickness = 0.0f; c.volume_attenuation_distance = 0.0f; c.volume_attenuation_color = {1.0f, 1.0f, 1.0f};
    if (idx_volume != 0u) {
        const uint32_t i = idx_volume;
        const TexInfo tx = tex_load(M, i);
        const float factor = mf(M, i + 5);
        if (!(!tx.exists && factor == 0.0f)) { c.volume_thickness = factor; if (tx.exists) c.volume_thickness = c.volume_thickness * sample_tex<GRAD>(a, tx).y; }
        c.volume_attenuation_distance = mf(M, i + 6);
        c.volume_attenuation_color = {mf(M, i + 7), mf(M, i + 8), mf(M, i + 9)};
    }
    c.clearcoat = 0.0f; c.clearcoat_roughness = 0.0f; c.clearcoat_normal = tbn.N;
    if (idx_clearcoat != 0u) {
        const uint32_t i = idx_clearcoat;
        const TexInfo tx = tex_load(M, i), rtx = tex_load(M, i + 6);
        const float factor = mf(M, i + 5);
        if (!(!tx.exists && factor == 0.0f)) { c.clearcoat = factor; if (tx.exists) c.clearcoat = c.clearcoat * sample_tex<GRAD>(a, tx).x; }
        c.clearcoat_roughness = mf(M, i + 11);
        if (rtx.exists) c.clearcoat_roughness = c.clearcoat_roughness * sample_tex<GRAD>(a, rtx).y;
        c.clearcoat_normal = normal_map<GRAD>(a, tex_load(M, i + 12), mf(M, i + 17), tbn);
    }
    c.sheen_color = {0.0f, 0.0f, 0.0f}; c.sheen_roughness = 0.0f;
    if (idx_sheen != 0u) {
        const uint32_t i = idx_sheen;
        const TexInfo rtx = tex_load(M, i), ctx = tex_load(M, i + 6);
        c.sheen_roughness = mf(M, i + 5);
        if (rtx.exists) c.sheen_roughness = c.sheen_roughness * sample_tex<GRAD>(a, rtx).w;
        c.sheen_color = {mf(M, i + 11), mf(M, i + 12), mf(M, i + 13)};
        if (ctx.exists) { const f4 s = sample_tex<GRAD>(a, ctx); c.sheen_color = {c.sheen_color.x * s.x, c.sheen_color.y * s.y, c.sheen_color.z * s.z}; }
    }

    if (debug_bitmask != 0u) {   // pbr_material_color.wgsl:34-60
        f3 dc = {1.0f, 0.0f, 1.0f};
        if (debug_bitmask & 1u) dc = c.base;
        else if (debug_bitmask & 2u) dc = {c.mr.x, c.mr.y, 0.0f};
        else if (debug_bitmask & 4u) dc = {c.normal.x * 0.5f + 0.5f, c.normal.y * 0.5f + 0.5f, c.normal.z * 0.5f + 0.5f};
        else if (debug_bitmask & 8u) dc = splat3(c.occlusion);
        else if (debug_bitmask & 16u) dc = c.emissive;
        else if (debug_bitmask & 32u) dc = c.specular_color * c.specular;
        out.color = {dc.x, dc.y, dc.z, base_alpha};
        out.kind = 1u;
        return out;
    }
    f3 background = {sc->prefiltered_rgb[0], sc->prefiltered_rgb[1], sc->prefiltered_rgb[2]};   // opaque pass: IBL only (brdf.wgsl:531-561)
    if (!FWD && sc->cube[kCubePrefiltered].texels && c.transmission * (1.0f - clampf(c.mr.x, 0.0f, 1.0f)) > 0.0f) {
        // brdf.wgsl:531-561 with a texel cube: straight through, or the refracted direction of a volume
        const f3 n = fm::fsafe_normalize(c.normal), v = fm::fsafe_normalize(surface_to_camera);
        f3 dir = -v;
        const float ior_val = c.ior < 1.0f ? 1.5f : c.ior;
        if (c.volume_thickness > 0.0f && ior_val != 1.0f) {
            const float eta = fm::rcp(ior_val);
            f3 refracted = v;                                            // refract_direction(v, n, eta), brdf.wgsl:30-47
            if (!(fabsf(eta - 1.0f) < 0.001f)) {
                const float cos_i = -fm::fdot(v, n), sin_t2 = eta * eta * (1.0f - cos_i * cos_i);
                refracted = sin_t2 > 1.0f ? mk3(0.0f, 0.0f, 0.0f) : v * eta + n * (eta * cos_i - sqrtf(1.0f - sin_t2));
            }
            if (fm::fdot(refracted, refracted) > 1e-6f) dir = refracted;
        }
        background = sample_prefiltered(sc, dir, fmaxf(clampf(c.mr.y, 0.0f, 1.0f), 0.04f));
    }
    if (FWD) {     // fragment.wgsl:245-270: screen-space transmission from the opaque image
        const float metallic = clampf(c.mr.x, 0.0f, 1.0f);
        if (c.transmission * (1.0f - metallic) > 0.0f)
            background = sample_transmission_background(sc, OpaqueImage{reinterpret_cast<const uint2*>(f.opaque_rgba16f), f.camera, (int)f.width, (int)f.height}, frag_x, frag_y, world_position, c.normal, -surface_to_camera, c.ior,
                                                        fmaxf(clampf(c.mr.y, 0.0f, 1.0f), 0.04f), c.volume_thickness);
    }
    const f3 color = apply_lighting(sc, f.lights_pre, c, surface_to_camera, world_position, n_lights, background);
    out.color = {color.x, color.y, color.z, base_alpha};
    return out;
}

// ------------------------------------------------------------------------------------------------
// shade_surface: the shading of one visibility sample — compute.wgsl:171-299 for the main sample,
// material_shading.wgsl:69-168 (msaa_process_sample) for the samples of an edge pixel.  `g` is the G-buffer texel of
// (triangle `rank`, pixel), `depth_sample` the depth the standard coordinates are built from (always sample 0's,
// standard.wgsl:17).  kind: 0 lit/unlit colour, 1 PBR debug view, 2 hud mesh (only reported when check_hud).
// ------------------------------------------------------------------------------------------------
template <int GRAD>
AWSM_DI SurfaceOut shade_surface(const DevScene* __restrict__ sc, const FrameDev& f, uint32_t rank, int cx, int cy, float depth_sample,
                                 const GBufferTexel& g, bool check_hud) {
    SurfaceOut out;
    out.color = {0.0f, 0.0f, 0.0f, 0.0f};
    out.kind = 0u; out.discard = false;
    const uint4* dsp = reinterpret_cast<const uint4*>(f.draw_shade + (f.tri_info[rank] & 0x00FFFFFFu));
    const uint4 ds0 = dsp[0], ds1 = dsp[1];   // first_tri, material_word, attr_indices_word, attr_data_word | stride_words, uv_sets_index, flags
    const uint32_t triangle_index = rank - ds0.x;
    if (check_hud && (ds1.z & 1u)) { out.kind = 2u; return out; }   // is_hud (compute.wgsl:176-179); msaa_process_sample has no such test
    const uint32_t material_word = ds0.y;
    const uint32_t attr_indices_off = ds0.z, attr_data_off = ds0.w, stride = ds1.x, uv_sets_index = ds1.y;

    // ---- compute.wgsl:182-211 ----
    Attr a;
    a.sc = sc;
    a.ad = reinterpret_cast<const float*>(sc->buf[AWSM_BUF_ATTR_DATA]);
    a.bary = {g.bx, g.by, (1.0f - g.bx) - g.by};
    a.uv_sets_index = uv_sets_index;
    const uint32_t* attr_idx = reinterpret_cast<const uint32_t*>(sc->buf[AWSM_BUF_ATTR_INDEX]) + attr_indices_off + triangle_index * 3u;
    a.v0 = attr_data_off + attr_idx[0] * stride;
    a.v1 = attr_data_off + attr_idx[1] * stride;
    a.v2 = attr_data_off + attr_idx[2] * stride;
    a.has_uv0 = false; a.uv0 = {0.0f, 0.0f};
    a.bary_derivs = g.bary_derivs; a.duv0_dx = {0.0f, 0.0f}; a.duv0_dy = {0.0f, 0.0f};

    // ---- standard.wgsl:11-62 ----
    const uint8_t* cam = f.camera;
    const m4 inv_proj = load_m4(reinterpret_cast<const float*>(cam + 256));
    const m4 inv_view = load_m4(reinterpret_cast<const float*>(cam + 320));
    const float proj33 = *reinterpret_cast<const float*>(cam + 64 + 60);
    const float* cam_pos = reinterpret_cast<const float*>(cam + 384);
    const f2 uv = {((float)cx + 0.5f) * fm::rcp((float)f.width), ((float)cy + 0.5f) * fm::rcp((float)f.height)};
    const f4 view_h = fm::fmul(inv_proj, {uv.x * 2.0f - 1.0f, 1.0f - uv.y * 2.0f, depth_sample, 1.0f});
    const float ivw = fm::rcp(fmaxf(view_h.w, 1e-8f));
    const f4 wp = fm::fmul(inv_view, {view_h.x * ivw, view_h.y * ivw, view_h.z * ivw, 1.0f});
    const f3 world_position = {wp.x, wp.y, wp.z};
    f3 surface_to_camera;
    if (proj33 > 0.9f) {
        surface_to_camera = fm::fnormalize(mk3(inv_view.c[2].x, inv_view.c[2].y, inv_view.c[2].z));
    } else {
        const f3 to_camera = mk3(cam_pos[0], cam_pos[1], cam_pos[2]) - world_position;
        surface_to_camera = fm::fdot(to_camera, to_camera) > 0.0f ? fm::fsafe_normalize(to_camera) : mk3(0.0f, 0.0f, 1.0f);
    }
    const TBN tbn = fm::funpack_normal_tangent(g.packed_nt);

    return shade_material<GRAD, false>(sc, f, a, material_word, f.tex_slots + (size_t)(f.tri_info[rank] & 0x00FFFFFFu) * kCoreTextures, f.draw_mat + (f.tri_info[rank] & 0x00FFFFFFu), tbn, world_position, surface_to_camera, 0u, 0.0f, 0.0f);
}


// ------------------------------------------------------------------------------------------------
// The world transparent pass (render.rs:224-297; material_transparent/{pipeline,render_pass}.rs;
// material_transparent_wgsl/fragment.wgsl) as three kernels over per-pixel fragment lists:
//
//   k_forward_cover   coverage + depth, in submission order.  Blending is order dependent (the host sorts the meshes back to front,
//                     renderable.rs:90,131-135), and so is the depth test (depth write is on), so one wavefront owns an 8x8 block of a
//                     tile, one lane a pixel: per-sample depth lives in that lane's registers and the wavefront walks the triangles
//                     that touch its block in rank order.  One workgroup (16 wavefronts) serves a 32x32 binning tile: it reads the
//                     tile's list once, sets bit (rank - base) in an LDS bitmap per block (ranks are unique, so scanning a bitmap IS
//                     the sorted list; windows of kFwdWindow ranks) and stages the setup records in LDS, so a step of the serial walk
//                     has no HBM/L2 round trip on its path.  A fragment that passes is appended to the pixel's list — triangle, sample
//                     mask, link to the next — and the depth is written.  ALPHA_MODE_MASK materials decide here, with the base-colour
//                     alpha alone (their depth write depends on it); nothing else of the material is evaluated in this kernel.
//   k_forward_shade   one thread per fragment, in no particular order: the material code of the opaque pass on interpolated varyings
//                     (shade_material<GRAD, true>), the screen-space transmission taps, premultiplied colour out.  This is where the
//                     time goes, and it runs like k_shade: full wavefronts, high occupancy, no ordering constraints.
//   k_forward_blend   one thread per pixel: the opaque colour (that IS the opaque -> transparent blit), the pixel's fragments in list
//                     order through the RGBA16F "over" blend per covered sample, the MSAA resolve, one store.  The multisampled colour
//                     target of the reference never exists in HBM.
//
// Contract (DESIGN.md "Transparent pass"): coverage/facing/depth as the geometry pass; varyings and implicit derivatives from the
// pixel centre's barycentrics; RGBA16F target rounding at every blend.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kFwdWindow = 16384;       // ranks per bitmap window (2 KB of LDS)
constexpr int kFwdBlock = 8;                 // one wavefront owns an 8x8 pixel block of the binning tile, one pixel per lane
constexpr uint32_t kFragNone = 0xFFFFFFFFu;
constexpr uint32_t kFragChunk = 64;          // fragment slots a wavefront reserves at a time

struct FwdVary { float b0, b1, b2; f4 derivs; uint4 ds0, ds1; uint32_t draw; };

// STRICT: barycentrics (and their quad differences) exactly as the oracle derives them from the setup record; then the attribute context
template <int GRAD>
AWSM_DI FwdVary forward_attr(const DevScene* __restrict__ sc, const FrameDev& f, const TriSetup& t, uint32_t rank, int px, int py, Attr& a) {
    FwdVary o;
    const double Xc = sample_coord((px << 8) + 128), Yc = sample_coord((py << 8) + 128);
    const EdgeVals ev = tri_edges_d(t, Xc, Yc);
    const float e0 = (float)ev.E[0] * t.iw[0], e1 = (float)ev.E[1] * t.iw[1], e2 = (float)ev.E[2] * t.iw[2];
    const float inv_esum = 1.0f / ((e0 + e1) + e2);
    const float b0 = e0 * inv_esum, b1 = e1 * inv_esum, b2 = e2 * inv_esum;
    f4 derivs = {0.0f, 0.0f, 0.0f, 0.0f};
    if (GRAD) {
        const EdgeVals eh = tri_edges_d(t, sample_coord(((px ^ 1) << 8) + 128), Yc), evv = tri_edges_d(t, Xc, sample_coord(((py ^ 1) << 8) + 128));
        const float h0 = (float)eh.E[0] * t.iw[0], h1 = (float)eh.E[1] * t.iw[1], h2 = (float)eh.E[2] * t.iw[2];
        const float w0 = (float)evv.E[0] * t.iw[0], w1 = (float)evv.E[1] * t.iw[1], w2 = (float)evv.E[2] * t.iw[2];
        const float ish = 1.0f / ((h0 + h1) + h2), isv = 1.0f / ((w0 + w1) + w2);
        const float hb0 = h0 * ish, hb1 = h1 * ish, vb0 = w0 * isv, vb1 = w1 * isv;
        derivs = {(px & 1) ? b0 - hb0 : hb0 - b0, (py & 1) ? b0 - vb0 : vb0 - b0, (px & 1) ? b1 - hb1 : hb1 - b1, (py & 1) ? b1 - vb1 : vb1 - b1};
    }
    o.b0 = b0; o.b1 = b1; o.b2 = b2; o.derivs = derivs;
    o.draw = f.tri_info[rank] & 0x00FFFFFFu;
    // Keep the masked index opaque to the optimiser.  hipcc 7.2 (clang 22) on gfx950 turned `(x & 0xFFFFFF) * 320` of the slot address
    // below into a 24-bit multiply, dropped the mask as redundant, and then emitted v_mad_u64_u32 on the UNMASKED word: a triangle
    // whose info word carries flag bits addressed memory 2^31 draws away (seen in k_forward_cover's ISA; memory access fault).
    asm volatile("" : "+v"(o.draw));
    const uint4* dsp = reinterpret_cast<const uint4*>(f.draw_shade + o.draw);
    o.ds0 = dsp[0]; o.ds1 = dsp[1];
    const uint32_t triangle_index = rank - o.ds0.x;
    a.sc = sc;
    a.ad = reinterpret_cast<const float*>(sc->buf[AWSM_BUF_ATTR_DATA]);
    a.bary = {b0, b1, b2};
    a.uv_sets_index = o.ds1.y;
    const uint32_t* attr_idx = reinterpret_cast<const uint32_t*>(sc->buf[AWSM_BUF_ATTR_INDEX]) + o.ds0.z + triangle_index * 3u;
    a.v0 = o.ds0.w + attr_idx[0] * o.ds1.x;
    a.v1 = o.ds0.w + attr_idx[1] * o.ds1.x;
    a.v2 = o.ds0.w + attr_idx[2] * o.ds1.x;
    a.has_uv0 = false; a.uv0 = {0.0f, 0.0f};
    a.bary_derivs = derivs; a.duv0_dx = {0.0f, 0.0f}; a.duv0_dy = {0.0f, 0.0f};
    return o;
}

// ALPHA_MODE_MASK (transparent material_color_calc.wgsl:38-52,344-362): the base colour's alpha against the cutoff, nothing else of the
// material.  The coverage kernel decides with this; the shading kernel then treats the fragment's alpha as 1 without testing again.
template <int GRAD>
AWSM_DI bool forward_alpha_test(const DevScene* __restrict__ sc, const FrameDev& f, const TriSetup& t, uint32_t rank, int px, int py) {
    Attr a;
    const FwdVary vy = forward_attr<GRAD>(sc, f, t, rank, px, py, a);
    const TexSlotDev* slots = f.tex_slots + (size_t)vy.draw * kCoreTextures;
    const DrawMatDev* dm = f.draw_mat + vy.draw;
    const uint32_t* M = reinterpret_cast<const uint32_t*>(sc->buf[AWSM_BUF_MATERIALS]);
    const uint32_t b = vy.ds0.y + 1u;
    float alpha = dm->base_color[3];
    if ((slots[0].flags >> 8) & 1u) alpha = alpha * sample_slot<GRAD>(a, slots, M, b + 2).w;
    if ((dm->shader_alpha & 0xFFu) != 2u && vy.ds1.w != 0u) {        // PBR meshes with colour sets multiply by the vertex colour
        const uint32_t set_index = (dm->ext_mask & 1u) ? M[b + M[b + 39u]] : 0u;
        if (set_index < vy.ds1.w) alpha = alpha * vertex_color(a, set_index).w;
    }
    return !(alpha < dm->alpha_cutoff);
}

template <int GRAD>
AWSM_DI SurfaceOut forward_fragment(const DevScene* __restrict__ sc, const FrameDev& f, const TriSetup& t, uint32_t rank, int px, int py, bool mask_resolved) {
    Attr a;
    const FwdVary vy = forward_attr<GRAD>(sc, f, t, rank, px, py, a);
    const float b0 = vy.b0, b1 = vy.b1, b2 = vy.b2;
    // ---- RELAXED from here ----
    const size_t v = (size_t)rank * 3;
    const float4 n0 = f.nrm[v], n1 = f.nrm[v + 1], n2 = f.nrm[v + 2];
    const float4 t0 = f.tan[v], t1 = f.tan[v + 1], t2 = f.tan[v + 2];
    const float4 w0 = f.wpos[v], w1 = f.wpos[v + 1], w2 = f.wpos[v + 2];
    const f3 world_position = {b0 * w0.x + b1 * w1.x + b2 * w2.x, b0 * w0.y + b1 * w1.y + b2 * w2.y, b0 * w0.z + b1 * w1.z + b2 * w2.z};
    f3 world_normal = {b0 * n0.x + b1 * n1.x + b2 * n2.x, b0 * n0.y + b1 * n1.y + b2 * n2.y, b0 * n0.z + b1 * n1.z + b2 * n2.z};
    const f3 tangent_xyz = {b0 * t0.x + b1 * t1.x + b2 * t2.x, b0 * t0.y + b1 * t1.y + b2 * t2.y, b0 * t0.z + b1 * t1.z + b2 * t2.z};
    float handedness = b0 * t0.w + b1 * t1.w + b2 * t2.w;
    if (!t.front) { world_normal = -world_normal; handedness = -handedness; }     // fragment.wgsl:195-203

    const uint8_t* cam = f.camera;
    const m4 inv_view = load_m4(reinterpret_cast<const float*>(cam + 320));
    const float proj33 = *reinterpret_cast<const float*>(cam + 64 + 60);
    const float* cam_pos = reinterpret_cast<const float*>(cam + 384);
    const f3 surface_to_camera = fabsf(proj33 - 1.0f) < 0.001f ? fm::fnormalize(mk3(inv_view.c[2].x, inv_view.c[2].y, inv_view.c[2].z))     // fragment.wgsl:205-215
                                                               : fm::fnormalize(mk3(cam_pos[0], cam_pos[1], cam_pos[2]) - world_position);
    // transparent material_color_calc.wgsl:5-19,125-153: N, T, B from the interpolated varyings
    TBN tbn;
    tbn.N = fm::fnormalize(world_normal);
    {
        const f3 tt = tangent_xyz - tbn.N * fm::fdot(tangent_xyz, tbn.N);
        const float len_sq = fm::fdot(tt, tt);
        if (len_sq > 1e-8f) tbn.T = tt * fm::rsq(len_sq);
        else tbn.T = fm::fnormalize(cross(fabsf(tbn.N.z) > 0.999f ? mk3(0.0f, 1.0f, 0.0f) : mk3(0.0f, 0.0f, 1.0f), tbn.N));
    }
    tbn.B = cross(tbn.N, tbn.T) * handedness;
    return shade_material<GRAD, true>(sc, f, a, vy.ds0.y, f.tex_slots + (size_t)vy.draw * kCoreTextures, f.draw_mat + vy.draw, tbn, world_position, surface_to_camera, vy.ds1.w,
                                      (float)px + 0.5f, (float)py + 0.5f, mask_resolved);
}

// One workgroup per binning tile, one wavefront per 8x8 block of it (16 wavefronts).  The tile's list is read ONCE by the whole
// workgroup: each entry sets its bit in the bitmap of every block its bbox touches, and its setup record is staged in LDS at the
// position its rank has among the tile's triangles (prefix popcount of the union bitmap).  Each wavefront then walks its own bitmap
// with no global load on the path of a step — a dependent HBM/L2 round trip per triangle is what bounds a serial walk otherwise.
#ifndef AWSM_FWD_NB
#define AWSM_FWD_NB 16
#endif
constexpr int kFwdNB = AWSM_FWD_NB;          // 8x8 blocks (= wavefronts) per workgroup: 16 = the whole 32x32 tile, 8 = a 32x16 half, 4 = a 16x16 quarter
constexpr int kFwdBW = kFwdNB == 4 ? 2 : 4;  // blocks per row of the workgroup's rectangle
constexpr int kFwdSubs = 16 / kFwdNB;
constexpr uint32_t kFwdThreads = 64u * kFwdNB;
constexpr uint32_t kFwdWords = kFwdWindow / 32;
constexpr uint32_t kFwdRecCap = 320;         // setup records staged per window (25 KB); triangles beyond that are fetched from HBM by the walk
constexpr uint32_t kFragPool = 64u * kFwdNB; // fragment slots a workgroup reserves up front (one global atomic per non-empty rectangle)

template <int S, int GRAD>
__global__ __launch_bounds__(kFwdThreads) void k_forward_cover(const DevScene* __restrict__ sc, FrameDev f) {
    __shared__ uint32_t bitmap[kFwdNB][kFwdWords];
    __shared__ uint32_t ubits[kFwdWords];          // union of the block bitmaps: the rectangle's triangles of this window
    __shared__ uint32_t wprefix[kFwdWords];        // set bits of ubits below each word = staged position of the word's first triangle
    __shared__ float4 rec_q[kFwdRecCap * 5];       // TriRec, 80 bytes each
    __shared__ uint32_t rec_info[kFwdRecCap];
    __shared__ uint32_t rmin, rmax, pool_next, pool_end;

    if (frame_poisoned(f)) return;
    const uint32_t tile = f.tile_order[blockIdx.x / kFwdSubs], sub = blockIdx.x % kFwdSubs;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, blk = tid >> 6;
    constexpr int kRectW = kFwdBW * kFwdBlock, kRectH = (kFwdNB / kFwdBW) * kFwdBlock;
    const int tx0 = ((int)(tile % f.tiles_x) << kTileShift) + (kFwdNB == 4 ? (int)(sub & 1u) * kRectW : 0);
    const int ty0 = ((int)(f.tile_row0 + (tile / f.tiles_x) * f.band_n) << kTileShift) + (kFwdNB == 4 ? (int)(sub >> 1) : (int)sub) * kRectH;
    const int px = tx0 + (int)(blk % kFwdBW) * kFwdBlock + (int)(lane & 7u), py = ty0 + (int)(blk / kFwdBW) * kFwdBlock + (int)(lane >> 3);
    const bool in_frame = px < (int)f.width && py < (int)f.height && row_owned(f, py);       // a strip's edge tiles hold rows of the neighbouring shard
    const bool wave_active = __builtin_amdgcn_ballot_w64(in_frame) != 0ull;
    const size_t p = in_frame ? (size_t)py * f.width + (size_t)px : 0;

    float depth[S];                            // depth LoadOp::Load: the geometry pass's depth; out-of-frame lanes fail every test
#pragma unroll
    for (int s = 0; s < S; s++) depth[s] = in_frame ? (f.hud_pass ? 1.0f : key_depth(f.vis[p * S + s])) : -1.0f;      // hud pass: hud_depth, cleared (render.rs:490-521)
    uint32_t first = kFragNone, last = kFragNone;
    // Fragment slots: same-address device atomics cost ~6 ns each on this part and serialise, so the tile's workgroup takes kFragPool
    // slots with ONE global atomic; its wavefronts carve exact runs out of that pool with LDS atomics, and only when the pool is dry
    // does a wavefront fall back to private chunks of kFragChunk.  Unused slots are marked kFragNone for the shading kernel.
    uint32_t chunk_next = 0u, chunk_end = 0u;      // wave-uniform: the wavefront's private chunk (after the pool ran dry)
    bool pool_dry = false;

    const uint32_t off = f.tile_offset[tile];
    const uint32_t count = min(f.tile_count[tile], f.bin_capacity - min(f.bin_capacity, off));
    if (count) {
        uint32_t pool_base = 0u;
        if (tid == 0) { rmin = 0xFFFFFFFFu; rmax = 0u; pool_base = atomicAdd(&f.counters[5], kFragPool); }
        __syncthreads();
        {   // rank range of the tile's list
            uint32_t lo = 0xFFFFFFFFu, hi = 0u;
            for (uint32_t i = tid; i < count; i += kFwdThreads) { const uint32_t r = f.bin_list[off + i]; lo = min(lo, r); hi = max(hi, r); }
            if (lo <= hi) { atomicMin(&rmin, lo); atomicMax(&rmax, hi); }
        }
        if (tid == 0) { pool_next = pool_base; pool_end = pool_base + kFragPool; }
        __syncthreads();
        const uint32_t r_lo = rmin, r_hi = rmax;
        for (uint32_t wbase = r_lo - (r_lo % kFwdWindow); wbase <= r_hi; wbase += kFwdWindow) {
            const uint32_t n_words = min(kFwdWindow, r_hi - wbase + 1u + 31u) / 32u;       // words that can hold a bit
            for (uint32_t i = lane; i < n_words; i += 64u) bitmap[blk][i] = 0u;
            for (uint32_t i = tid; i < n_words; i += kFwdThreads) ubits[i] = 0u;
            __syncthreads();
            for (uint32_t i = tid; i < count; i += kFwdThreads) {
                const uint32_t r = f.bin_list[off + i], rr = r - wbase;      // unsigned: ranks below the window wrap to huge values
                if (rr >= kFwdWindow) continue;
                const uint32_t bx = f.tri_rec[r].bbox_x, by = f.tri_rec[r].bbox_y;
                const int x0 = (int)(bx & 0x7FFFu), x1 = (int)((bx >> 16) & 0x7FFFu), y0 = (int)(by & 0xFFFFu), y1 = (int)((by >> 16) & 0x7FFFu);   // bits 15 and 31 of bbox_x, 31 of bbox_y are flags
                if (x0 > x1 || x1 < tx0 || x0 > tx0 + (kRectW - 1) || y1 < ty0 || y0 > ty0 + (kRectH - 1)) continue;
                const int ba = (max(x0, tx0) - tx0) >> 3, bb = (min(x1, tx0 + (kRectW - 1)) - tx0) >> 3, bc = (max(y0, ty0) - ty0) >> 3, bd = (min(y1, ty0 + (kRectH - 1)) - ty0) >> 3;
                for (int yy = bc; yy <= bd; yy++) for (int xx = ba; xx <= bb; xx++) atomicOr(&bitmap[yy * kFwdBW + xx][rr >> 5], 1u << (rr & 31u));
                atomicOr(&ubits[rr >> 5], 1u << (rr & 31u));
            }
            __syncthreads();
            if (blk == 0u) {   // exclusive prefix of the per-word popcounts, 8 words per lane
                uint32_t c[kFwdWords / 64], sum = 0u;
#pragma unroll
                for (uint32_t k = 0; k < kFwdWords / 64; k++) { const uint32_t w = lane * (kFwdWords / 64) + k; c[k] = w < n_words ? (uint32_t)__builtin_popcount(ubits[w]) : 0u; sum += c[k]; }
                uint32_t incl = sum;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, d); if ((int)lane >= d) incl += o; }
                uint32_t run = incl - sum;
#pragma unroll
                for (uint32_t k = 0; k < kFwdWords / 64; k++) { const uint32_t w = lane * (kFwdWords / 64) + k; if (w < n_words) wprefix[w] = run; run += c[k]; }
            }
            __syncthreads();
            for (uint32_t i = tid; i < count; i += kFwdThreads) {     // stage the records of the triangles that made it into the bitmaps
                const uint32_t r = f.bin_list[off + i], rr = r - wbase;
                if (rr >= kFwdWindow) continue;
                const uint32_t word = ubits[rr >> 5], bit = 1u << (rr & 31u);
                if (!(word & bit)) continue;
                const uint32_t pos = wprefix[rr >> 5] + (uint32_t)__builtin_popcount(word & (bit - 1u));
                if (pos >= kFwdRecCap) continue;
                const float4* src = reinterpret_cast<const float4*>(f.tri_rec + r);
#pragma unroll
                for (int k = 0; k < 5; k++) rec_q[pos * 5u + k] = src[k];
                rec_info[pos] = f.tri_info[r];
            }
            __syncthreads();
            if (wave_active) {
                // The set bits of this block's bitmap in ascending order ARE its triangles in submission order (scalar iteration).
                uint32_t it_wc = 0u, it_word = 0u, it_bits = 0u, it_my = lane < n_words ? bitmap[blk][lane] : 0u;
                unsigned long long it_nz = __builtin_amdgcn_ballot_w64(it_my != 0u);
                for (;;) {
                    if (!it_bits) {
                        if (it_nz) {
                            const uint32_t wl = (uint32_t)__builtin_ctzll(it_nz);
                            it_nz &= it_nz - 1ull;
                            it_bits = (uint32_t)__builtin_amdgcn_readlane((int)it_my, (int)wl);
                            it_word = it_wc + wl;
                            continue;
                        }
                        it_wc += 64u;
                        if (it_wc >= n_words) break;
                        it_my = it_wc + lane < n_words ? bitmap[blk][it_wc + lane] : 0u;
                        it_nz = __builtin_amdgcn_ballot_w64(it_my != 0u);
                        continue;
                    }
                    const uint32_t bit = (uint32_t)__builtin_ctz(it_bits);
                    it_bits &= it_bits - 1u;
                    const uint32_t rank = wbase + it_word * 32u + bit;
                    const uint32_t pos = wprefix[it_word] + (uint32_t)__builtin_popcount(ubits[it_word] & ((1u << bit) - 1u));
                    TriRecRaw raw; uint32_t info;
                    if (pos < kFwdRecCap) {
                        const float4* q = rec_q + pos * 5u;
                        raw.q0 = q[0]; raw.q1 = q[1]; raw.q2 = q[2];
                        raw.d3 = *reinterpret_cast<const double2*>(q + 3); raw.d4 = *reinterpret_cast<const double2*>(q + 4);
                        info = rec_info[pos];
                    } else { raw = tri_rec_fetch(f.tri_rec + rank); info = f.tri_info[rank]; }
                    TriSetup t;
                    if (!tri_rec_unpack(raw, t)) continue;
                    uint32_t mask = 0u;
                    float z[S];
                    float zc = 0.0f, dz[4] = {0.0f, 0.0f, 0.0f, 0.0f};       // S == 4: the depth plane at the pixel's corner and the samples' steps (raster_setup.hpp)
                    if (S == 4) { zc = tri_plane_depth(t, tri_edges_d(t, (double)px, (double)py)) + 0.0f; msaa_depth_steps(t.a, t.b, t.zq, dz); }
#pragma unroll
                    for (int s = 0; s < S; s++) {
                        const unsigned long long k = S == 1 ? tri_sample_key_at(t, sample_coord((px << 8) + 128), sample_coord((py << 8) + 128), rank)
                                                            : tri_msaa_sample_key(t, px, py, s, zc, dz, rank);
                        z[s] = __uint_as_float((uint32_t)(k >> 32));
                        if (k != ~0ull && z[s] <= depth[s]) mask |= 1u << s;       // CompareFunction::LessEqual
                    }
                    const bool may_discard = (info >> 31) != 0u;     // ALPHA_MODE_MASK (uniform; tagged by the transform)
                    if (may_discard && __builtin_amdgcn_ballot_w64(mask != 0u) != 0ull) {
                        if (mask != 0u && !forward_alpha_test<GRAD>(sc, f, t, rank, px, py)) mask = 0u;          // discard: neither colour nor depth
                    }
                    const unsigned long long hit = __builtin_amdgcn_ballot_w64(mask != 0u);
                    if (hit == 0ull) continue;
                    const uint32_t n_hit = (uint32_t)__builtin_popcountll(hit);
                    uint32_t base = kFragNone;
                    if (chunk_next + n_hit <= chunk_end) { base = chunk_next; chunk_next += n_hit; }
                    else if (!pool_dry) {
                        uint32_t pb = 0u;
                        if (lane == 0u) pb = atomicAdd(&pool_next, n_hit);
                        pb = (uint32_t)__builtin_amdgcn_readfirstlane((int)pb);
                        const uint32_t pe = pool_end;
                        if (pb + n_hit <= pe) base = pb;
                        else {      // the pool is dry; what this wavefront took past its end, if anything is inside, stays unused
                            for (uint32_t i = pb + lane; i < pe; i += 64u) if (i < f.frag_cap) f.frag_rec[i].x = kFragNone;
                            pool_dry = true;
                        }
                    }
                    if (base == kFragNone) {
                        for (uint32_t i = chunk_next + lane; i < chunk_end; i += 64u) if (i < f.frag_cap) f.frag_rec[i].x = kFragNone;    // unused tail
                        uint32_t nb = 0u;
                        if (lane == 0u) nb = atomicAdd(&f.counters[5], kFragChunk);
                        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)nb);
                        chunk_next = base + n_hit;
                        chunk_end = base + kFragChunk;
                    }
                    if (mask != 0u) {
#pragma unroll
                        for (int s = 0; s < S; s++) if (mask & (1u << s)) depth[s] = z[s];
                        const uint32_t idx = base + (uint32_t)__builtin_popcountll(hit & ((1ull << lane) - 1ull));
                        if (idx < f.frag_cap) {
                            f.frag_rec[idx] = make_uint4(rank, (uint32_t)px | ((uint32_t)py << 16), kFragNone, mask | (may_discard ? 0x100u : 0u));
                            if (last != kFragNone) f.frag_rec[last].z = idx; else first = idx;
                            last = idx;
                        } else f.counters[6] = 1u;      // list overflow: the host grows the buffers and replays the pass
                    }
                }
            }
            __syncthreads();
        }
        for (uint32_t i = min(pool_next, pool_end) + tid; i < pool_end; i += kFwdThreads) if (i < f.frag_cap) f.frag_rec[i].x = kFragNone;      // what is left of the pool
    }
    for (uint32_t i = chunk_next + lane; i < chunk_end; i += 64u) if (i < f.frag_cap) f.frag_rec[i].x = kFragNone;    // unused tail of the last chunk
    if (in_frame) f.frag_first[p] = first;
}

template <int GRAD>
__global__ __launch_bounds__(256) void k_forward_shade(const DevScene* __restrict__ sc, FrameDev f) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (frame_poisoned(f)) return;
    if (i >= min(f.counters[5], f.frag_cap)) return;
    const uint4 rec = f.frag_rec[i];
    if (rec.x == kFragNone) return;                       // unused slot of a wavefront's chunk
    const int px = (int)(rec.y & 0xFFFFu), py = (int)(rec.y >> 16);
    TriSetup t;
    tri_rec_load(f.tri_rec + rec.x, t);
    const SurfaceOut o = forward_fragment<GRAD>(sc, f, t, rec.x, px, py, (rec.w & 0x100u) != 0u);
    const float a = o.color.w;
    f.frag_color[i] = make_float4(o.color.x * a, o.color.y * a, o.color.z * a, a);          // fragment.wgsl:283-285 premultiplied
}

template <int S>
__global__ __launch_bounds__(256) void k_forward_blend(FrameDev f) {
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= f.width * f.height || frame_poisoned(f)) return;
    if (!row_owned(f, (int)(p / f.width))) return;                               // sharded: only this shard's rows of the composite
    const uint2 c0 = reinterpret_cast<const uint2*>(f.opaque_rgba16f)[p];       // opaque -> transparent blit: every sample starts as the opaque colour
    float dst[S][4];
#pragma unroll
    for (int s = 0; s < S; s++) {
        dst[s][0] = f16_bits_to_f32((unsigned short)(c0.x & 0xFFFFu)); dst[s][1] = f16_bits_to_f32((unsigned short)(c0.x >> 16));
        dst[s][2] = f16_bits_to_f32((unsigned short)(c0.y & 0xFFFFu)); dst[s][3] = f16_bits_to_f32((unsigned short)(c0.y >> 16));
    }
    for (uint32_t i = f.frag_first[p]; i != kFragNone && i < f.frag_cap;) {
        const uint4 rec = f.frag_rec[i];
        const float4 src = f.frag_color[i];
        const float om = 1.0f - src.w;
#pragma unroll
        for (int s = 0; s < S; s++) {
            if (!(rec.w & (1u << s))) continue;
            dst[s][0] = blend_over_f16(src.x, dst[s][0], om); dst[s][1] = blend_over_f16(src.y, dst[s][1], om);
            dst[s][2] = blend_over_f16(src.z, dst[s][2], om); dst[s][3] = blend_over_f16(src.w, dst[s][3], om);
        }
        i = rec.z;
    }
    float c[4];
#pragma unroll
    for (int k = 0; k < 4; k++) c[k] = S == 1 ? dst[0][k] : resolve4_f16(dst[0][k], dst[S > 1 ? 1 : 0][k], dst[S > 2 ? 2 : 0][k], dst[S > 3 ? 3 : 0][k]);
    store_pixel(f, p, {c[0], c[1], c[2], c[3]});
}

// Block of 16x16 pixels -> pixel of this thread.  Workgroup ids are dealt round-robin over the 8 XCDs (blockIdx & 7), each
// with its own L2.  XCD x shades the block rows x, x+8, x+16, ...: coherent along a row (neighbouring blocks share
// triangles and texels in that L2) and balanced over the screen (expensive regions — minified textures far away — are
// spread over all XCDs).  Block rows are 16 consecutive rows of the shard (row mode) or the two halves of each owned
// 32-row band (band mode).  Returns false for surplus workgroup ids.
struct ShadeBlock { uint32_t blk, brow; int x0, y0; };
AWSM_DI bool shade_block(const FrameDev& f, ShadeBlock& b, uint32_t wg = blockIdx.x) {
    const uint32_t bx_n = (f.width + 15u) >> 4;
    const uint32_t by_n = f.band_n > 1u ? 2u * f.tiles_y : ((f.sy1 - f.sy0) + 15u) >> 4;
    const uint32_t nblk = bx_n * by_n;
    const uint32_t xcd = wg & 7u, k = wg >> 3;     // k-th block of this XCD
    const uint32_t rows_x = (by_n + 7u - xcd) >> 3;                 // block rows owned by this XCD
    if (k >= rows_x * bx_n) return false;
    b.blk = ((k / bx_n) * 8u + xcd) * bx_n + (k % bx_n);
    if (b.blk >= nblk) return false;
    b.brow = b.blk / bx_n;
    b.x0 = (int)((b.blk % bx_n) << 4);
    b.y0 = f.band_n > 1u ? (int)(((f.tile_row0 + (b.brow >> 1) * f.band_n) << kTileShift) + ((b.brow & 1u) << 4)) : (int)f.sy0 + (int)(b.brow << 4);
    return true;
}
// the same block from its row and column (no division: the persistent grid numbers blocks on a power-of-two pitch)
AWSM_DI void shade_block_at(const FrameDev& f, ShadeBlock& b, uint32_t brow, uint32_t bcol) {
    const uint32_t bx_n = (f.width + 15u) >> 4;
    b.blk = brow * bx_n + bcol;
    b.brow = brow;
    b.x0 = (int)(bcol << 4);
    b.y0 = f.band_n > 1u ? (int)(((f.tile_row0 + (brow >> 1) * f.band_n) << kTileShift) + ((brow & 1u) << 4)) : (int)f.sy0 + (int)(brow << 4);
}

// ------------------------------------------------------------------------------------------------
// k_shade: single-sampled opaque pass, 16x16 pixels per workgroup (compute.wgsl uses 8x8; a 64-wide wavefront covers
// 16x4 here).  5 waves/SIMD (<= 96 VGPRs): measured faster than the 4 the register allocator picks on its own, 6 spills.
// ------------------------------------------------------------------------------------------------
template <int GRAD>
AWSM_DI void shade_pixel(const DevScene* __restrict__ sc, const FrameDev& f, const ShadeBlock& b, uint32_t tid) {
    const int cx = b.x0 + (int)(tid & 15u), cy = b.y0 + (int)(tid >> 4);
    if (cx >= (int)f.width || cy >= (int)f.sy1) return;                  // compute.wgsl:111-113
    const size_t pv = (size_t)cy * f.width + (size_t)cx;                  // visibility buffer: always addressed by absolute row
    const size_t p = f.out_compact ? (size_t)(((b.brow >> 1) << kTileShift) + (uint32_t)(cy & (kTile - 1))) * f.width + (size_t)cx : pv;   // output pixel
    const unsigned long long key = f.vis[pv];
    if (f.hud_vis && f.has_opaque && f.hud_vis[pv] != ~0ull) { store_pixel(f, p, f4{0.0f, 0.0f, 0.0f, 0.0f}); return; }   // a hud mesh's triangle is what the visibility target holds here: compute.wgsl:176-179 returns, the pixel stays cleared
    if (!f.has_opaque || key == ~0ull) { store_pixel(f, p, skybox_color(sc, f, cx, cy)); return; }   // compute.wgsl:149-153 / empty.wgsl, skybox.wgsl:1-41
    const uint32_t rank = key_rank(key);
    const GBufferTexel g = reconstruct_gbuffer<(GRAD != 0)>(f, rank, cx, cy);    // STRICT
    const SurfaceOut o = shade_surface<GRAD>(sc, f, rank, cx, cy, key_depth(key), g, true);
    store_pixel(f, p, o.kind == 2u ? f4{0.0f, 0.0f, 0.0f, 0.0f} : o.color);   // hud: stays cleared (compute.wgsl:176-179)
}
template <int GRAD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(GRAD ? 4 : 5))) void k_shade(const DevScene* __restrict__ sc, FrameDev f) {
    ShadeBlock b;
    if (frame_poisoned(f) || !shade_block(f, b)) return;
    shade_pixel<GRAD>(sc, f, b, threadIdx.x);
}
// The 16x4-pixel groups k_shade_lean left behind (a draw that is not lean, or texture coordinates beyond +-32768): one wavefront
// per list entry, the general code.  A fixed small grid strides over the list; with an all-lean frame every wavefront reads the
// count and exits.
// 112 VGPRs at most (109 as compiled; the list walk sits in scalar registers for that): frame i's list is shaded while frame i + 1's k_shade_lean
// (80 VGPRs x 6 waves per SIMD) already fills the machine, and a waiting workgroup is only placed when all of it fits — 112 is what one exiting
// lean workgroup leaves free on each SIMD (32 + 80).  At 120 (114 used) this kernel waited for the whole lean kernel to drain — 228 us for an
// empty list — and held up everything ordered behind the frame.  No occupancy step of the compiler yields a 112 budget (waves_per_eu(5): 96 and
// a spill), so the count is checked after the build (awsm_renderer_amd/build.py).
// MSAA: sample 0 of the pixel by the general code, stored as k_shade_lean<.., MSAA> stores its own (the strip's masks and cells are already there: the lean
// kernel wrote them before it handed the strip over).  Hud meshes and debug views are written before the edge test and never resolved
// (compute.wgsl:118-170,303-318) — a pixel of theirs that the masks name leaves a marker in msaa_color0 that k_shade_msaa_resolve skips.
template <int GRAD>
AWSM_DI void shade_pixel_msaa0(const DevScene* __restrict__ sc, const FrameDev& f, const ShadeBlock& b, uint32_t tid) {
    const int cx = b.x0 + (int)(tid & 15u), cy = b.y0 + (int)(tid >> 4);
    if (cx >= (int)f.width || cy >= (int)f.sy1) return;
    const size_t pv = (size_t)cy * f.width + (size_t)cx;
    const size_t p = f.out_compact ? (size_t)(((b.brow >> 1) << kTileShift) + (uint32_t)(cy & (kTile - 1))) * f.width + (size_t)cx : pv;
    const unsigned long long* masks = f.msaa_edge_bits + (size_t)(b.blk * 4u + (tid >> 6)) * 2u;
    const bool want_c0 = (((masks[0] | masks[1]) >> (tid & 63u)) & 1ull) != 0ull;
    const unsigned long long key = f.vis[pv * 4];
    if (key == ~0ull) return;                                              // background: the lean kernel stored it
    const uint32_t rank = key_rank(key);
    const GBufferTexel g = reconstruct_gbuffer<(GRAD != 0)>(f, rank, cx, cy);    // STRICT
    const SurfaceOut o = shade_surface<GRAD>(sc, f, rank, cx, cy, key_depth(key), g, true);
    store_pixel(f, p, o.kind == 2u ? f4{0.0f, 0.0f, 0.0f, 0.0f} : o.color);
    if (want_c0) f.msaa_color0[pv] = o.kind != 0u ? make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(0xFFFFFFFFu)) : make_float4(o.color.x, o.color.y, o.color.z, o.color.w);
}
constexpr uint32_t kTodoBlocks = 1024;
template <int GRAD, bool MSAA>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void k_shade_todo(const DevScene* __restrict__ sc, FrameDev f) {
    // This kernel starting means this frame's k_shade_lean has ended (same stream): the next frame's opaque pass, gated on that, goes ahead
    // while the list is shaded (k_handoff_wait, kernels_geometry.hip; the two frames write different images).
    if (f.lean_done_flag && blockIdx.x == 0u && threadIdx.x == 0u) __hip_atomic_store(f.lean_done_flag, f.lean_done_serial, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (frame_poisoned(f)) return;        // (after the flag: the next frame's gate must still open)
    const uint32_t n = min(f.shade_todo[0], f.shade_todo_cap);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // the list walk in scalar registers
    for (uint32_t i = blockIdx.x * 4u + wave; i < n; i += kTodoBlocks * 4u) {
        const uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)f.shade_todo[4u + i]);
        ShadeBlock b;
        if (!shade_block(f, b, e >> 2)) continue;
        if (MSAA) shade_pixel_msaa0<GRAD>(sc, f, b, ((e & 3u) << 6) | lane);
        else shade_pixel<GRAD>(sc, f, b, ((e & 3u) << 6) | lane);
    }
}

// ------------------------------------------------------------------------------------------------
// k_shade_lean: the opaque pass for what a frame is mostly made of — PBR materials without optional blocks whose textures sit on
// TEXCOORD_0 behind a repeat / linear sampler (LeanDrawDev).  Same arithmetic as shade_pixel (STRICT G-buffer reconstruction
// through reconstruct_core, RELAXED shading with the same formulas), arranged for the machine:
//   - every pointer is global address space with a 32-bit byte offset from a wave-uniform base (SGPR pair), so a load costs no
//     64-bit address arithmetic and no flat-address dispatch; uniform data (camera, lights, environment) comes in by scalar loads;
//   - the dependent-load chain is key -> {setup record, vertex normals / tangents, tri_shade} -> {lean record, three UV pairs}
//     -> texels: four levels (the general route: key -> tri_info -> draw_shade -> indices -> UVs -> slot -> texels), and every
//     level is issued in one burst before anything waits;
//   - all texel fetches of the pixel are in flight together while the view vector and the TBN frame are computed;
//   - one 96-byte record per draw instead of 336 bytes of per-draw records per pixel.
// A wavefront (16x4 pixels) that meets a draw that is not lean, or texture coordinates beyond +-32768 (where the general sampler's
// range guard acts), writes nothing and appends itself to shade_todo; k_shade_todo shades it with the general code.
// ------------------------------------------------------------------------------------------------
#define AWSM_AS1 __attribute__((address_space(1)))
#define AWSM_AS4 __attribute__((address_space(4)))
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2a4 __attribute__((ext_vector_type(2), aligned(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2a4 __attribute__((ext_vector_type(2), aligned(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
// vector memory: uniform base + 32-bit byte offset (global_load ... v_off, s[base:base+1])
template <typename T> AWSM_DI T gload(const void* base, uint32_t byte_off) { return *(const AWSM_AS1 T*)((const AWSM_AS1 char*)base + byte_off); }
// per-lane 64-bit base + 32-bit byte offset
template <typename T> AWSM_DI T gload64(unsigned long long base, uint32_t byte_off) { return *(const AWSM_AS1 T*)((const AWSM_AS1 char*)base + byte_off); }
// scalar memory: uniform address, data that no kernel writes while this one runs
template <typename T> AWSM_DI T cload(const void* base, uint32_t byte_off) { return *(const AWSM_AS4 T*)((const AWSM_AS4 char*)base + byte_off); }
AWSM_DI float4 as_float4(f32x4 v) { return make_float4(v.x, v.y, v.z, v.w); }
AWSM_DI m4 cload_m4(const void* base, uint32_t byte_off) {
    m4 m;
#pragma unroll
    for (int i = 0; i < 4; i++) { const f32x4 v = cload<f32x4>(base, byte_off + 16u * i); m.c[i] = {v.x, v.y, v.z, v.w}; }
    return m;
}

namespace lean {
struct Tap { uint32_t t00, t10, t01, t11; float fx, fy; };
// A texture entry of the lean record, decoded: base address, log2 extents.  Decoding all five before the first texel load means the
// record's loads are waited for once, ahead of the burst (a wait inside the burst would also wait for the texels issued so far:
// vmcnt counts in order).
struct Tex { unsigned long long base; uint32_t lw, lh; };
AWSM_DI Tex decode(uint32_t lo, uint32_t hi) { return {((unsigned long long)(hi & 0xFFFFu) << 32) | lo, (hi >> 16) & 15u, (hi >> 20) & 15u}; }
// sample_level_fast's addressing for a power-of-two repeat texture whose extent comes as log2: texel (i0, j0) and its right / lower
// neighbours.  The two texels of a row are one 8-byte load; when the footprint wraps around the row end (i0 = W - 1) the second
// texel of each pair is re-fetched from the row start — a branch almost never taken, and its loads land in the pair's registers
// under the wrapped lanes' exec mask, so nothing waits.
AWSM_DI void fetch(const Tex& x, float u, float v, Tap& t) {
    const float xf = __builtin_amdgcn_ldexpf(u, (int)x.lw) - 0.5f, yf = __builtin_amdgcn_ldexpf(v, (int)x.lh) - 0.5f;     // u * W - 0.5 (W a power of two: exact product)
    const float flx = floorf(xf), fly = floorf(yf);
    t.fx = xf - flx; t.fy = yf - fly;
    const uint32_t xi = (uint32_t)(int)flx, yi = (uint32_t)(int)fly;
    const uint32_t i0 = __builtin_amdgcn_ubfe(xi, 0u, x.lw), j0 = __builtin_amdgcn_ubfe(yi, 0u, x.lh), j1 = __builtin_amdgcn_ubfe(yi + 1u, 0u, x.lh);
    const uint32_t sh = x.lw + 2u, i0b = i0 << 2;
    const u32x2a4 p0 = gload64<u32x2a4>(x.base, (j0 << sh) | i0b), p1 = gload64<u32x2a4>(x.base, (j1 << sh) | i0b);
    t.t00 = p0.x; t.t10 = p0.y; t.t01 = p1.x; t.t11 = p1.y;
    if (__builtin_amdgcn_ubfe(xi + 1u, 0u, x.lw) == 0u) {      // i1 wrapped to column 0
        t.t10 = gload64<uint32_t>(x.base, j0 << sh); t.t11 = gload64<uint32_t>(x.base, j1 << sh);
    }
}
struct Weights { float w00, w10, w01, w11; };
AWSM_DI Weights weights(const Tap& t) {      // as sample_level_fast: bilinear on the raw 0..255 values, one scale by 1/255 at the end
    const float gx = 1.0f - t.fx, gy = 1.0f - t.fy;
    return {gx * gy, t.fx * gy, gx * t.fy, t.fx * t.fy};
}
template <int BYTE> AWSM_DI float ub(uint32_t t) {
    return (float)((t >> (8 * BYTE)) & 255u);      // v_cvt_f32_ubyteN
}
template <int BYTE> AWSM_DI float channel(const Tap& t, const Weights& w) {      // the bilinear sum of the RAW 0..255 values: the 1/255 sits in the draw's factors
    return ub<BYTE>(t.t00) * w.w00 + ub<BYTE>(t.t10) * w.w10 + ub<BYTE>(t.t01) * w.w01 + ub<BYTE>(t.t11) * w.w11;
}
// brdf.wgsl:293-302 (sample_brdf_lut above, same arithmetic; global-address-space loads)
AWSM_DI f2 brdf_lut(const uint16_t* lut, uint32_t lut_w, uint32_t lut_h, float n_dot_v, float roughness) {
    const float u = saturate(n_dot_v), v = saturate(roughness);
    const int W = (int)lut_w, H = (int)lut_h;
    const float x = u * (float)W - 0.5f, y = v * (float)H - 0.5f;          // in [-0.5, extent - 0.5]: no range guard needed
    const float flx = floorf(x), fly = floorf(y);
    const float fx = x - flx, fy = y - fly;
    const int xi = (int)flx, yi = (int)fly;
    const int i0 = max(xi, 0), i1 = min(xi + 1, W - 1), j0 = max(yi, 0), j1 = min(yi + 1, H - 1);
    const uint32_t r0 = (uint32_t)(j0 * W) << 2, r1 = (uint32_t)(j1 * W) << 2;
    const uint32_t t00 = gload<uint32_t>(lut, r0 + ((uint32_t)i0 << 2)), t10 = gload<uint32_t>(lut, r0 + ((uint32_t)i1 << 2));
    const uint32_t t01 = gload<uint32_t>(lut, r1 + ((uint32_t)i0 << 2)), t11 = gload<uint32_t>(lut, r1 + ((uint32_t)i1 << 2));
    const float gx = 1.0f - fx, gy = 1.0f - fy;
    const float r_top = f16_bits_to_f32((unsigned short)(t00 & 0xFFFFu)) * gx + f16_bits_to_f32((unsigned short)(t10 & 0xFFFFu)) * fx;
    const float r_bot = f16_bits_to_f32((unsigned short)(t01 & 0xFFFFu)) * gx + f16_bits_to_f32((unsigned short)(t11 & 0xFFFFu)) * fx;
    const float g_top = f16_bits_to_f32((unsigned short)(t00 >> 16)) * gx + f16_bits_to_f32((unsigned short)(t10 >> 16)) * fx;
    const float g_bot = f16_bits_to_f32((unsigned short)(t01 >> 16)) * gx + f16_bits_to_f32((unsigned short)(t11 >> 16)) * fx;
    return {r_top * gy + r_bot * fy, g_top * gy + g_bot * fy};
}
// brdf_direct (brdf.wgsl:308-381) for a material without sheen / clearcoat, the same terms arranged for fewer instructions: one
// reciprocal for the three denominators of D * G1(l) / (4 n.l n.v), clamps as output modifiers, per-pixel factors hoisted.
// x -> sat(1 - x) equals 1 - sat(max(x, 0)) for every x, and sat(n.l) serves both as n.l >= 0 and as the saturated value (n, l unit).
// F is increasing in F0 per channel, so max(F) = F(max(F0)): k_d needs one multiply-add.  (The half vector's length and its two dot products stay as
// the WGSL forms them — v + l first: measured, |v + l|^2 = 2 + 2 v.l and n.(v + l) = n.v + n.l lose their relative accuracy to cancellation when the
// light comes from behind the viewer's side at grazing angles, and a GGX lobe multiplies that by 4 / (alpha^4): 13 pixels of a random view went out of bounds.)
struct Lit {
    f3 n, v, F0, df90, bd;          // bd = base * (1 - metallic) / pi
    float ndv_raw, ndv_dir, ndv4, a2m1, a2_g1v, gk, one_m_gk, occlusion, F0max, df90max;
};
AWSM_DI void direct(const Lit& s, f3 l, f3 radiance, f3& color) {
    const float ndl = saturate(fm::fdot(s.n, l));
#ifndef AWSM_NO_LIGHT_SKIP
    // A light behind the surface for every pixel of the wavefront (a 16x4 strip is mostly one surface: a ceiling under four lights from above, a wall
    // facing away) adds exactly zero — w = n.l * occlusion = 0 multiplies finite factors (den >= kEps^2 > 0, radiance and F finite), and 0 + c = c — so
    // the wavefront skips the term: same bits, ~45 instructions fewer per skipped light (round 5).
    if (__builtin_amdgcn_ballot_w64(ndl > 0.0f) == 0ull) return;
#endif
    const f3 sum = s.v + l;
    const float len_sq = fm::fdot(sum, sum);
    const bool has_half = len_sq > 1e-8f;
    const float inv_len = has_half ? fm::rsq(len_sq) : 0.0f;
    const float ndh = saturate(fm::fdot(s.n, sum) * inv_len);
    const float vdh = fm::fdot(s.v, sum) * inv_len;
    const float p = saturate(1.0f - (has_half ? vdh : s.ndv_dir));
    const float p2 = p * p, p5 = (p2 * p2) * p;
    const f3 F = {s.F0.x + s.df90.x * p5, s.F0.y + s.df90.y * p5, s.F0.z + s.df90.z * p5};
    const float dd = (ndh * ndh) * s.a2m1 + 1.0f;
    const float den = (((kPi * dd) * dd + kEps) * (ndl * s.one_m_gk + s.gk)) * fmaxf(s.ndv4 * ndl, kEps);
    const float spec = has_half ? (s.a2_g1v * ndl) * fm::rcp(den) : 0.0f;
    const float k_d = 1.0f - (s.F0max + s.df90max * p5);
    const float w = ndl * s.occlusion;
    color.x += (s.bd.x * k_d + F.x * spec) * (radiance.x * w);
    color.y += (s.bd.y * k_d + F.y * spec) * (radiance.y * w);
    color.z += (s.bd.z * k_d + F.z * spec) * (radiance.z * w);
}
}  // namespace lean

// Per-wavefront LDS staging of the per-triangle records (level 1 of the chain).  A 16x4 strip holds a handful of distinct triangles, and the
// 208 bytes that belong to one — setup record 80, three vertex normals 48, three tangents 48, {info word, three TEXCOORD_0} 32 — used to be
// loaded by every lane that shows it (13 vector loads of 16 bytes per lane: 60 % of what a pixel pulled through its CU's L1 / TA path).  Now the
// wavefront finds its distinct triangles (one ballot per triangle: ranks are wave-uniform scalars then), lanes 0..12 bring each record in ONCE
// with one LDS-DMA instruction (global_load_lds_dwordx4: global -> LDS without a register; 13 active lanes x 16 bytes land at slot + lane * 16),
// nothing waits inside the loop, and after one s_waitcnt every lane reads its triangle's record from LDS (ds_read_b128, lanes of one triangle
// read one address: a broadcast).  kLeanSlots distinct triangles per wavefront are staged; the lanes of a busier strip (distant, finely
// tessellated geometry) load theirs directly as before.
#ifndef AWSM_LEAN_SLOTS
#define AWSM_LEAN_SLOTS 16u
#endif
constexpr uint32_t kLeanSlots = AWSM_LEAN_SLOTS, kLeanChunks = 13u, kLeanNone = 0xFFFFFFFFu;
struct LeanStage { uint4 q[kLeanSlots][kLeanChunks]; };      // one per wavefront: 3,328 bytes
#define AWSM_AS3 __attribute__((address_space(3)))
AWSM_DI float4 bits_float4(uint4 v) { return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)); }
AWSM_DI double2 bits_double2(uint4 v) { return make_double2(__hiloint2double((int)v.y, (int)v.x), __hiloint2double((int)v.w, (int)v.z)); }
AWSM_DI uint4 as_uint4(u32x4 v) { return make_uint4(v.x, v.y, v.z, v.w); }

namespace lean {
// fetch() for a texture whose record sits in scalar registers (the draw is wave-uniform inside the draw loop): base address in an SGPR pair,
// 32-bit byte offsets per lane -> global_load ... v_off, s[base]: no 64-bit address arithmetic at all.
struct TexS { const void* base; uint32_t lw, lh; };
AWSM_DI TexS decode_s(uint32_t lo, uint32_t hi) { return {reinterpret_cast<const void*>(((unsigned long long)(hi & 0xFFFFu) << 32) | lo), (hi >> 16) & 15u, (hi >> 20) & 15u}; }
AWSM_DI void fetch_s(const TexS& x, float u, float v, Tap& t) {
    const float xf = __builtin_amdgcn_ldexpf(u, (int)x.lw) - 0.5f, yf = __builtin_amdgcn_ldexpf(v, (int)x.lh) - 0.5f;     // u * W - 0.5 (W a power of two: exact product)
    const float flx = floorf(xf), fly = floorf(yf);
    t.fx = xf - flx; t.fy = yf - fly;
    const uint32_t xi = (uint32_t)(int)flx, yi = (uint32_t)(int)fly;
    const uint32_t i0 = __builtin_amdgcn_ubfe(xi, 0u, x.lw), j0 = __builtin_amdgcn_ubfe(yi, 0u, x.lh), j1 = __builtin_amdgcn_ubfe(yi + 1u, 0u, x.lh);
    const uint32_t sh = x.lw + 2u, i0b = i0 << 2;
    const u32x2a4 p0 = gload<u32x2a4>(x.base, (j0 << sh) | i0b), p1 = gload<u32x2a4>(x.base, (j1 << sh) | i0b);
    t.t00 = p0.x; t.t10 = p0.y; t.t01 = p1.x; t.t11 = p1.y;
    if (__builtin_amdgcn_ubfe(xi + 1u, 0u, x.lw) == 0u) {      // i1 wrapped to column 0
        t.t10 = gload<uint32_t>(x.base, j0 << sh); t.t11 = gload<uint32_t>(x.base, j1 << sh);
    }
}
// MipmapMode::Gradient: textureSampleGrad by the contract of sample_slot<true> (isotropic LOD from the larger of the two gradient lengths, min / mipmap
// filters linear, repeat addressing), on a square power-of-two pool array whose record sits in scalar registers (LeanDrawDev.gtex).
// BASE: const void* — the record sits in scalar registers (a strip inside one draw: loads are 32-bit offsets from an SGPR pair); unsigned long long — per
// lane (a strip over several draws)
template <typename BASE> struct TexGT { BASE base; uint32_t lw, levels, layer, layers; };
typedef TexGT<const void*> TexG;
typedef TexGT<unsigned long long> TexGL;
AWSM_DI TexG decode_g(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) {
    return {reinterpret_cast<const void*>(((unsigned long long)(w1 & 0xFFFFu) << 32) | w0), w1 >> 24, (w1 >> 16) & 15u, w2, w3};
}
AWSM_DI TexGL decode_gl(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) {
    return {((unsigned long long)(w1 & 0xFFFFu) << 32) | w0, w1 >> 24, (w1 >> 16) & 15u, w2, w3};
}
template <typename T> AWSM_DI T gload_any(const void* base, uint32_t off) { return gload<T>(base, off); }
template <typename T> AWSM_DI T gload_any(unsigned long long base, uint32_t off) { return gload64<T>(base, off); }
struct Lod { uint32_t lo, hi; float f; };
// m2 = max(|d uv / dx|^2, |d uv / dy|^2) in uv units; scaling by the extent is exact (a power of two), so rho2 has sample_slot<true>'s bits
template <typename BASE> AWSM_DI Lod select_lod(const TexGT<BASE>& x, float m2) {
    const float rho2 = __builtin_amdgcn_ldexpf(m2, (int)(2u * x.lw));
    float lod = 0.5f * __builtin_amdgcn_logf(fmaxf(rho2, 1e-12f));      // log2(max(rho, 1e-6))
    Lod r = {0u, 0u, 0.0f};
    if (lod > 0.0f && x.levels > 1u) {
        lod = fminf(lod, (float)(x.levels - 1u));
        const float fl = floorf(lod);
        r.lo = (uint32_t)fl; r.hi = min(r.lo + 1u, x.levels - 1u); r.f = (r.hi != r.lo) ? lod - fl : 0.0f;
    }
    return r;
}
// the bilinear footprint on level `level` (per lane) of the layer
template <typename BASE> AWSM_DI void fetch_g(const TexGT<BASE>& x, uint32_t level, float u, float v, Tap& t) {
    const uint32_t lwl = x.lw - level;                                       // log2 of the level's extent
    // first texel of the level: layers * (G(lw + 1) - G(lwl + 1)); of the layer inside it: layer << 2 lwl
    const uint32_t g_all = __builtin_amdgcn_ubfe(0x55555555u, 0u, 2u * (x.lw + 1u)), g_rest = __builtin_amdgcn_ubfe(0x55555555u, 0u, 2u * (lwl + 1u));
    const uint32_t first = x.layers * (g_all - g_rest) + (x.layer << (2u * lwl));
    const float xf = __builtin_amdgcn_ldexpf(u, (int)lwl) - 0.5f, yf = __builtin_amdgcn_ldexpf(v, (int)lwl) - 0.5f;
    const float flx = floorf(xf), fly = floorf(yf);
    t.fx = xf - flx; t.fy = yf - fly;
    const uint32_t xi = (uint32_t)(int)flx, yi = (uint32_t)(int)fly;
    const uint32_t i0 = __builtin_amdgcn_ubfe(xi, 0u, lwl), j0 = __builtin_amdgcn_ubfe(yi, 0u, lwl), j1 = __builtin_amdgcn_ubfe(yi + 1u, 0u, lwl);
    const uint32_t r0 = (first + (j0 << lwl)) << 2, r1 = (first + (j1 << lwl)) << 2, i0b = i0 << 2;
    const u32x2a4 p0 = gload_any<u32x2a4>(x.base, r0 + i0b), p1 = gload_any<u32x2a4>(x.base, r1 + i0b);
    t.t00 = p0.x; t.t10 = p0.y; t.t01 = p1.x; t.t11 = p1.y;
    if (__builtin_amdgcn_ubfe(xi + 1u, 0u, lwl) == 0u) {      // i1 wrapped to column 0 (always on the 1 x 1 level)
        t.t10 = gload_any<uint32_t>(x.base, r0); t.t11 = gload_any<uint32_t>(x.base, r1);
    }
}
// textureSampleLevel on a cube through its aproned chain (CubeDev.bordered; k_cube_border filled the apron by sample_cube's seam rule, so the values and
// the weights are sample_cube's): the cube's record by scalar loads, the face by sample_cube's table, then per level two 16-byte loads (two adjacent
// RGBA16F texels of a row each) and the lerps.  `which` is a compile-time cube id.
AWSM_DI f4 cube_level_b(const void* base, const void* scene, uint32_t which, uint32_t size, uint32_t level, uint32_t face, float sn, float tn) {
    const uint32_t N = max(size >> level, 1u), P = N + 2u;
    const uint32_t lvl0 = gload<uint32_t>(scene, (uint32_t)(offsetof(DevScene, cube) + which * sizeof(CubeDev) + offsetof(CubeDev, b_level_off)) + level * 4u);
    float x = sn * (float)N - 0.5f, y = tn * (float)N - 0.5f;                 // sn, tn = 0.5 (sc / ma) + 0.5
    if (!(x >= -0.5f)) x = -0.5f;                 // also NaN (zero / non-finite direction): the face's first texel
    if (!(y >= -0.5f)) y = -0.5f;
    x = fminf(x, (float)N - 0.5f); y = fminf(y, (float)N - 0.5f);
    const float flx = floorf(x), fly = floorf(y), fx = x - flx, fy = y - fly;
    const uint32_t ib = (uint32_t)((int)flx + 1), jb = (uint32_t)((int)fly + 1);      // apron coordinates of the footprint's first texel: 0 .. N
    const uint32_t t0 = (lvl0 + (face * P + jb) * P + ib) << 3;
    typedef uint32_t u32x4a8 __attribute__((ext_vector_type(4), aligned(8)));
    const u32x4a8 r0 = gload<u32x4a8>(base, t0), r1 = gload<u32x4a8>(base, t0 + (P << 3));
    const f4 c00 = half4(make_uint2(r0.x, r0.y)), c10 = half4(make_uint2(r0.z, r0.w)), c01 = half4(make_uint2(r1.x, r1.y)), c11 = half4(make_uint2(r1.z, r1.w));
    return lerp4(lerp4(c00, c10, fx), lerp4(c01, c11, fx), fy);
}
template <uint32_t WHICH>
AWSM_DI f4 cube_sample(const DevScene* sc, f3 d, float level) {
    const uint32_t o = (uint32_t)(offsetof(DevScene, cube) + WHICH * sizeof(CubeDev));
    const u32x2 bp = cload<u32x2>(sc, o + (uint32_t)offsetof(CubeDev, bordered));
    const u32x2 sm = cload<u32x2>(sc, o + (uint32_t)offsetof(CubeDev, size));
    const void* base = reinterpret_cast<const void*>(((unsigned long long)bp.y << 32) | bp.x);
    const uint32_t size = sm.x, mips = sm.y;
    const float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    uint32_t face; float scd, tcd, ma;
    if (az >= ax && az >= ay) { face = d.z < 0.0f ? 5u : 4u; scd = d.z < 0.0f ? -d.x : d.x; tcd = -d.y; ma = az; }
    else if (ay >= ax) { face = d.y < 0.0f ? 3u : 2u; scd = d.x; tcd = d.y < 0.0f ? -d.z : d.z; ma = ay; }
    else { face = d.x < 0.0f ? 1u : 0u; scd = d.x < 0.0f ? d.z : -d.z; tcd = -d.y; ma = ax; }
    const float inv = fm::rcp(ma);
    const float sn = 0.5f * (scd * inv) + 0.5f, tn = 0.5f * (tcd * inv) + 0.5f;
    const float top = (float)(mips - 1u);
    float lod = level > 0.0f ? level : 0.0f;          // also NaN
    lod = fminf(lod, top);
    const float fl = floorf(lod), fr = lod - fl;
    const uint32_t l0 = (uint32_t)fl, l1 = min(l0 + 1u, mips - 1u);
    f4 r = cube_level_b(base, sc, WHICH, size, l0, face, sn, tn);
    if (__builtin_amdgcn_ballot_w64(fr > 0.0f && l1 != l0) != 0ull) { const f4 h = cube_level_b(base, sc, WHICH, size, l1, face, sn, tn); if (fr > 0.0f && l1 != l0) r = lerp4(r, h, fr); }
    return r;
}
AWSM_DI f3 irradiance(const DevScene* sc, f3 n) {          // brdf.wgsl:268-276
    if (!sc->cube[kCubeIrradiance].bordered) return sample_irradiance(sc, n);      // (uniform colour; scalar branch)
    const f4 c = cube_sample<kCubeIrradiance>(sc, n, 0.0f);
    return {c.x, c.y, c.z};
}
AWSM_DI f3 prefiltered(const DevScene* sc, f3 dir, float roughness) {      // brdf.wgsl:278-290
    if (!sc->cube[kCubePrefiltered].bordered) return sample_prefiltered(sc, dir, roughness);
    const uint32_t mip_count = cload<uint32_t>(sc->buf[AWSM_BUF_LIGHTS_INFO], 4u);      // IblInfo.prefiltered_env_mip_count (lights.rs:300-305)
    const f4 c = cube_sample<kCubePrefiltered>(sc, dir, roughness * (float)(mip_count - 1u));
    return {c.x, c.y, c.z};
}
struct TapG { Tap lo, hi; float f; };
// both levels of a texture: the second only when some lane of the wavefront blends (f > 0) — a magnified strip fetches level 0 once
template <typename BASE> AWSM_DI void fetch_trilinear(const TexGT<BASE>& x, float m2, float u, float v, TapG& t) {
    const Lod l = select_lod(x, m2);
    t.f = l.f;
    fetch_g(x, l.lo, u, v, t.lo);
    if (__builtin_amdgcn_ballot_w64(l.f > 0.0f) != 0ull) fetch_g(x, l.hi, u, v, t.hi);
    else t.hi = t.lo;
}
template <int BYTE> AWSM_DI float channel(const TapG& t, const Weights& wl, const Weights& wh) {
    const float a = channel<BYTE>(t.lo, wl), b = channel<BYTE>(t.hi, wh);
    return a * (1.0f - t.f) + b * t.f;                                        // sample_slot<true>: acc = c_lo (1 - f) + c_hi f
}
// ---- One footprint for all of a pixel's textures (round 5).  The isotropic rule picks, for every texture, the level whose texels match the pixel's
// footprint: lod_t = log2(rho * W_t) = L + lw_t with L = log2(rho) in uv units the same for every texture of the pixel (they share TEXCOORD_0 and its
// derivatives on this route) — so wherever the level is not clamped (0 < lod_t < levels_t - 1) the chosen level's extent, 2^(lw_t - floor(lod_t)) =
// 2^(-floor(L)), the blend factor frac(L) and with them the texel coordinates and all eight weights are THE SAME for all five textures.  A minified
// frame (the 4K atrium samples its textures two to three levels down) is entirely in that regime.  The coordinates are formed once per level
// (footprint: sample_slot<true>'s own operations, so the same bits), a texture then costs its two level bases and four loads, and a channel is one sum
// over eight taps with weights formed once.  (L + lw_t instead of one hardware log2 per texture: the same value to within an ulp of the logarithm, far
// inside what RELAXED allows; on a level boundary the blend is continuous.)  Strips with a magnified or top-clamped texture take the per-texture code.
struct Foot { uint32_t i0b, row0, row1, g_rest, sh2; bool wrap; };
AWSM_DI void footprint(uint32_t lwl, float u, float v, Foot& ft, float& fx, float& fy) {      // the level whose extent is 2^lwl: fetch_g's coordinates
    const float xf = __builtin_amdgcn_ldexpf(u, (int)lwl) - 0.5f, yf = __builtin_amdgcn_ldexpf(v, (int)lwl) - 0.5f;
    const float flx = floorf(xf), fly = floorf(yf);
    fx = xf - flx; fy = yf - fly;
    const uint32_t xi = (uint32_t)(int)flx, yi = (uint32_t)(int)fly;
    const uint32_t i0 = __builtin_amdgcn_ubfe(xi, 0u, lwl), j0 = __builtin_amdgcn_ubfe(yi, 0u, lwl), j1 = __builtin_amdgcn_ubfe(yi + 1u, 0u, lwl);
    ft.i0b = i0 << 2; ft.row0 = j0 << lwl; ft.row1 = j1 << lwl;
    ft.wrap = __builtin_amdgcn_ubfe(xi + 1u, 0u, lwl) == 0u;
    ft.g_rest = __builtin_amdgcn_ubfe(0x55555555u, 0u, 2u * (lwl + 1u)); ft.sh2 = 2u * lwl;
}
struct Quad { uint32_t t00, t10, t01, t11; };
template <typename BASE> AWSM_DI void fetch_q(const TexGT<BASE>& x, const Foot& ft, Quad& q) {
    const uint32_t g_all = __builtin_amdgcn_ubfe(0x55555555u, 0u, 2u * (x.lw + 1u));                    // (scalar when the record is)
    const uint32_t first = x.layers * (g_all - ft.g_rest) + (x.layer << ft.sh2);
    const uint32_t r0 = (first + ft.row0) << 2, r1 = (first + ft.row1) << 2;
    const u32x2a4 p0 = gload_any<u32x2a4>(x.base, r0 + ft.i0b), p1 = gload_any<u32x2a4>(x.base, r1 + ft.i0b);
    q.t00 = p0.x; q.t10 = p0.y; q.t01 = p1.x; q.t11 = p1.y;
    if (ft.wrap) { q.t10 = gload_any<uint32_t>(x.base, r0); q.t11 = gload_any<uint32_t>(x.base, r1); }
}
AWSM_DI TexG decode_any(const void*, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) { return decode_g(w0, w1, w2, w3); }
AWSM_DI TexGL decode_any(unsigned long long, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) { return decode_gl(w0, w1, w2, w3); }
struct W8 { float l00, l10, l01, l11, h00, h10, h01, h11; };
template <int BYTE> AWSM_DI float channel8(const Quad& lo, const Quad& hi, const W8& w) {      // (c_lo (1 - f) + c_hi f with the level weights folded into the eight)
    return ((ub<BYTE>(lo.t00) * w.l00 + ub<BYTE>(lo.t10) * w.l10) + (ub<BYTE>(lo.t01) * w.l01 + ub<BYTE>(lo.t11) * w.l11)) +
           ((ub<BYTE>(hi.t00) * w.h00 + ub<BYTE>(hi.t10) * w.h10) + (ub<BYTE>(hi.t01) * w.h01 + ub<BYTE>(hi.t11) * w.h11));
}
}  // namespace lean

// MSAA (x4, the reference's default AntiAliasing): the kernel shades sample 0 of every pixel — keys sit four to a pixel — and leaves what the edge
// decision (compute.wgsl:155-170,303-318, msaa.wgsl) needs behind, so that nothing is reconstructed twice:
//   * from the pixel's four keys, per strip two lane masks (FrameDev.msaa_edge_bits): "an edge whatever the neighbours show" (sample 0 is background and
//     another sample is not; or the samples differ and their view depths spread, msaa.wgsl:116-146) and "the samples differ, the neighbours decide".
//     A pixel whose four samples show ONE triangle is in neither: resolving it would average four copies of one colour (all samples a triangle covers in
//     a pixel carry the same centre-evaluated G-buffer texel and sample 0's coordinates), i.e. give the colour back to within one f32 rounding of 3c;
//   * per pixel the STRICT normal and the depth of sample 0 (FrameDev.msaa_cells) for k_msaa_detect's neighbour comparison;
//   * the colour in the image, and for a pixel in either mask also as f32 in msaa_color0, where k_shade_msaa_resolve picks it up if the pixel is resolved.
// ITEMS (k_shade_msaa_resolve): the lanes are not the pixels of a strip but (pixel, sample) items of a block's edge pixels — the triangle of sample s at
// the pixel's centre, with the pixel's shared standard coordinates (sample 0's depth; material_shading.wgsl:186-189) — handed in through LeanItem; the
// colour goes back the same way and nothing is stored.  Returns false when the wavefront has a lane this route cannot shade (wave-uniform): the caller then
// takes the general code for these lanes, as k_shade_todo does for a strip.
struct LeanItem { int cx, cy; uint32_t rank; float depth; bool live; f3 color; };
template <int GRAD, bool MSAA, bool ITEMS>
AWSM_DI bool lean_core(const DevScene* __restrict__ sc, const FrameDev& f, const ShadeBlock& b, const uint32_t wg, const uint32_t tid, LeanStage* __restrict__ st, LeanItem* __restrict__ item) {
    const uint32_t lane = tid & 63u;
    const int cx = ITEMS ? item->cx : b.x0 + (int)(tid & 15u), cy = ITEMS ? item->cy : b.y0 + (int)(tid >> 4);
    const bool inside = ITEMS ? item->live : (cx < (int)f.width && cy < (int)f.sy1);            // compute.wgsl:111-113 (no early exit: lanes 0..12 stage records for the whole wavefront)
    const uint32_t pv = (uint32_t)cy * f.width + (uint32_t)cx;
    const uint32_t p = f.out_compact ? (((b.brow >> 1) << kTileShift) + ((uint32_t)cy & (uint32_t)(kTile - 1))) * f.width + (uint32_t)cx : pv;

    u32x2 key = {0xFFFFFFFFu, 0xFFFFFFFFu};
    bool want_c0 = false;                                                 // MSAA: the pixel may be resolved — its colour also goes to msaa_color0
    if (ITEMS) { if (inside) key = {~item->rank, __float_as_uint(item->depth)}; }
    else if (!MSAA) { if (inside) key = gload<u32x2>(f.vis, pv << 3); }
    else {
        u32x4 ka = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, kb = ka;     // [pixel][4 samples]: {lo, hi} of samples 0, 1 and 2, 3
        if (inside) { ka = gload<u32x4>(f.vis, pv << 5); kb = gload<u32x4>(f.vis, (pv << 5) + 16u); }
        key = {ka.x, ka.y};
        const bool bg0 = (ka.x & ka.y) == 0xFFFFFFFFu, bg1 = (ka.z & ka.w) == 0xFFFFFFFFu, bg2 = (kb.x & kb.y) == 0xFFFFFFFFu, bg3 = (kb.z & kb.w) == 0xFFFFFFFFu;
        bool sure = false, test = false;
        if (!(bg0 && bg1 && bg2 && bg3)) {                                // (lanes outside the frame hold four background keys)
            if (bg0) sure = true;                                         // compute.wgsl:155-170: sample 0 is background, others are not
            else if (bg1 || bg2 || bg3 || ka.z != ka.x || kb.x != ka.x || kb.z != ka.x) {      // some sample shows something else than sample 0's triangle
                const m4 inv_proj = cload_m4(f.camera, 256u);
                const unsigned long long k4[4] = {((unsigned long long)ka.y << 32) | ka.x, ((unsigned long long)ka.w << 32) | ka.z, ((unsigned long long)kb.y << 32) | kb.x, ((unsigned long long)kb.w << 32) | kb.z};
                sure = edge_mask_depth_msaa_filtered(inv_proj, k4, (float)cx + 0.5f, (float)cy + 0.5f, (float)f.width, (float)f.height);      // STRICT decisions
                test = !sure;
            }
        }
        const unsigned long long m_sure = __builtin_amdgcn_ballot_w64(sure), m_test = __builtin_amdgcn_ballot_w64(test);
        if (lane == 0u) *reinterpret_cast<ulonglong2*>(f.msaa_edge_bits + (size_t)(b.blk * 4u + (tid >> 6)) * 2u) = make_ulonglong2(m_sure, m_test);
        want_c0 = sure || test;
    }
    bool hud = false;                                                     // a hud mesh covers the pixel: it stays cleared (compute.wgsl:176-179)
    if (!ITEMS && !MSAA && f.hud_vis && inside) { const u32x2 hk = gload<u32x2>(f.hud_vis, pv << 3); hud = (hk.x & hk.y) != 0xFFFFFFFFu; }
    if (hud) { store_pixel(f, p, f4{0.0f, 0.0f, 0.0f, 0.0f}); key = {0xFFFFFFFFu, 0xFFFFFFFFu}; }
    const bool hit = inside && (key.x & key.y) != 0xFFFFFFFFu;
    if (!ITEMS && inside && !hit && !hud) {                                         // compute.wgsl:149-153: no hit -> skybox (skybox.wgsl:1-41: the uniform colour or the texel cube)
        const f4 sky = skybox_color(sc, f, cx, cy);
        store_pixel(f, p, sky);
        if (MSAA) {
            f.msaa_cells[pv] = make_uint2(0u, 0xFFFFFFFFu);
            if (want_c0) f.msaa_color0[pv] = make_float4(sky.x, sky.y, sky.z, sky.w);
        }
    }
    unsigned long long rem = __builtin_amdgcn_ballot_w64(hit);
    if (rem == 0ull) return true;                                         // wave-uniform
    const uint32_t rank = hit ? ~key.x : 0xFFFFFFFFu;                     // 0xFFFFFFFF - low word; no triangle has the sentinel's rank
    const float depth = __uint_as_float(key.y);

    asm volatile("; MARK level1");
    // ---- level 1: the distinct triangles' records, once each, global -> LDS ----
    unsigned long long src_base; uint32_t src_stride;                    // lane j < 13 fetches chunk j of a record: where that chunk lives
    {
        const uint32_t j = lane;
        const unsigned long long rec = (unsigned long long)f.tri_rec, nr = (unsigned long long)f.nrm, tn = (unsigned long long)f.tan, tsh = (unsigned long long)f.tri_shade;
        src_base = j < 5u ? rec + j * 16u : (j < 8u ? nr + (j - 5u) * 16u : (j < 11u ? tn + (j - 8u) * 16u : tsh + (j - 11u) * 16u));
        src_stride = j < 5u ? (uint32_t)kTriRecBytes : (j < 11u ? 48u : 32u);
    }
    uint32_t my_slot = kLeanNone, n_slots = 0u;
    while (rem != 0ull && n_slots < kLeanSlots) {
        const uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)rank, (int)__builtin_ctzll(rem));
        if (rank == r) my_slot = n_slots;
        if (lane < kLeanChunks)
            __builtin_amdgcn_global_load_lds((const AWSM_AS1 void*)(src_base + (unsigned long long)r * src_stride), (AWSM_AS3 void*)&st->q[n_slots][0], 16, 0, 0);
        rem &= ~__builtin_amdgcn_uicmp(rank, r, 32 /* ICMP_EQ */);
        n_slots++;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // the DMA writes have landed in LDS
    const bool staged = my_slot != kLeanNone;
    const uint4* rec = &st->q[staged ? my_slot : 0u][0];
    uint4 ts0 = rec[11], ts1 = rec[12];
    TriRecRaw raw;
    raw.q0 = bits_float4(rec[0]); raw.q1 = bits_float4(rec[1]); raw.q2 = bits_float4(rec[2]); raw.d3 = bits_double2(rec[3]); raw.d4 = bits_double2(rec[4]);
    float4 n0 = bits_float4(rec[5]), n1 = bits_float4(rec[6]), n2 = bits_float4(rec[7]);
    float4 t0 = bits_float4(rec[8]), t1 = bits_float4(rec[9]), t2 = bits_float4(rec[10]);
    if (hit && !staged) {                                                  // more distinct triangles in this strip than slots: these lanes load their own
        ts0 = as_uint4(gload<u32x4>(f.tri_shade, rank << 5)); ts1 = as_uint4(gload<u32x4>(f.tri_shade, (rank << 5) + 16u));
        const uint32_t ro = rank * (uint32_t)kTriRecBytes, vo = rank * 48u;
        raw.q0 = as_float4(gload<f32x4>(f.tri_rec, ro)); raw.q1 = as_float4(gload<f32x4>(f.tri_rec, ro + 16u)); raw.q2 = as_float4(gload<f32x4>(f.tri_rec, ro + 32u));
        { const f64x2 a = gload<f64x2>(f.tri_rec, ro + 48u), c = gload<f64x2>(f.tri_rec, ro + 64u); raw.d3 = make_double2(a.x, a.y); raw.d4 = make_double2(c.x, c.y); }
        n0 = as_float4(gload<f32x4>(f.nrm, vo)); n1 = as_float4(gload<f32x4>(f.nrm, vo + 16u)); n2 = as_float4(gload<f32x4>(f.nrm, vo + 32u));
        t0 = as_float4(gload<f32x4>(f.tan, vo)); t1 = as_float4(gload<f32x4>(f.tan, vo + 16u)); t2 = as_float4(gload<f32x4>(f.tan, vo + 32u));
    }

    asm volatile("; MARK strict");
    // ---- STRICT: what fs_main wrote for this pixel ----
    TriSetup t;
    tri_rec_unpack(raw, t);
    const GBufferTexel g = reconstruct_core<(GRAD != 0)>(t, n0, n1, n2, t0, t1, t2, cx, cy);
    // the detector's operands, before the wavefront may leave for the general kernel: the cells are this kernel's
    if (MSAA && hit) f.msaa_cells[pv] = make_uint2(oct_word(mk2(g.packed_nt.x, g.packed_nt.y)), key.y);
    const float bz = (1.0f - g.bx) - g.by;                               // compute.wgsl:185-186
    float u = interp3_strict(g.bx, g.by, bz, __uint_as_float(ts0.z), __uint_as_float(ts1.x), __uint_as_float(ts1.z));
    float v = interp3_strict(g.bx, g.by, bz, __uint_as_float(ts0.w), __uint_as_float(ts1.y), __uint_as_float(ts1.w));
    const uint32_t draw = ts0.x & 0x00FFFFFFu;
    const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)draw, (int)__builtin_ctzll(__builtin_amdgcn_ballot_w64(hit)));
    const bool one_draw = __builtin_amdgcn_ballot_w64(hit && draw != d0) == 0ull;
    f2 ddx = {0.0f, 0.0f}, ddy = {0.0f, 0.0f};
    float m2 = 0.0f;      // MipmapMode::Gradient: max(|d uv / dx|^2, |d uv / dy|^2), get_uv_derivatives (helpers/mipmap.wgsl:113-205) as attr_uv<true> forms it
    float r2min = 0.0f; f2 major = {0.0f, 0.0f};      // GRAD == 2 (AWSM_CFG_ANISOTROPIC): the smaller of the two and the longer derivative (grad_footprint)
    if (GRAD) {
        const float x0 = __uint_as_float(ts0.z), y0 = __uint_as_float(ts0.w), x1 = __uint_as_float(ts1.x), y1 = __uint_as_float(ts1.y), x2 = __uint_as_float(ts1.z), y2 = __uint_as_float(ts1.w);
        const float dAlphaDx = g.bary_derivs.x, dAlphaDy = g.bary_derivs.y, dBetaDx = g.bary_derivs.z, dBetaDy = g.bary_derivs.w;
        const float dGammaDx = -dAlphaDx - dBetaDx, dGammaDy = -dAlphaDy - dBetaDy;
        ddx = {x0 * dAlphaDx + x1 * dBetaDx + x2 * dGammaDx, y0 * dAlphaDx + y1 * dBetaDx + y2 * dGammaDx};
        ddy = {x0 * dAlphaDy + x1 * dBetaDy + x2 * dGammaDy, y0 * dAlphaDy + y1 * dBetaDy + y2 * dGammaDy};
        const bool tiny = (fabsf(dAlphaDx) + fabsf(dAlphaDy) + fabsf(dBetaDx) + fabsf(dBetaDy)) < 1e-20f;
        const bool ok = (ddx.x == ddx.x) && (ddx.y == ddx.y) && (ddy.x == ddy.x) && (ddy.y == ddy.y);   // NaN guard
        if (tiny || !ok) { ddx = {0.0f, 0.0f}; ddy = {0.0f, 0.0f}; }
    }
    // The draw's shared texture transform (LeanDrawDev.tt, flags bit 3; texture_uvs.wgsl:27-35,64-84 as sample_slot applies it per texture): scalar
    // for a strip inside one draw — a draw without one costs a scalar branch — per lane otherwise.  Here for MipmapMode::Gradient, whose level
    // selection needs the transformed derivatives; MipmapMode::None applies it inside fetch_all, where the record's flags are loaded anyway.
    if (GRAD == 0) {
    } else if (one_draw) {
        const uint32_t lo = d0 * (uint32_t)sizeof(LeanDrawDev);
        if (cload<uint32_t>(f.draw_lean, lo) & 8u) {
            const f32x4 ta = cload<f32x4>(f.draw_lean, lo + 176u); const f32x2a4 tb = cload<f32x2a4>(f.draw_lean, lo + 192u);
            const float u2 = affine2_strict(ta.x, ta.y, tb.x, u, v), v2 = affine2_strict(ta.z, ta.w, tb.y, u, v);
            u = u2; v = v2;
            if (GRAD) { ddx = {ta.x * ddx.x + ta.y * ddx.y, ta.z * ddx.x + ta.w * ddx.y}; ddy = {ta.x * ddy.x + ta.y * ddy.y, ta.z * ddy.x + ta.w * ddy.y}; }
        }
    } else {
        const uint32_t lo = draw * (uint32_t)sizeof(LeanDrawDev);
        const bool xf = (gload<uint32_t>(f.draw_lean, lo) & 8u) != 0u;
        if (__builtin_amdgcn_ballot_w64(xf) != 0ull) {
            const f32x4 ta = gload<f32x4>(f.draw_lean, lo + 176u); const f32x2a4 tb = gload<f32x2a4>(f.draw_lean, lo + 192u);      // (the identity where the draw has none)
            const float u2 = affine2_strict(ta.x, ta.y, tb.x, u, v), v2 = affine2_strict(ta.z, ta.w, tb.y, u, v);
            if (xf) { u = u2; v = v2; }
            if (GRAD && xf) { ddx = {ta.x * ddx.x + ta.y * ddx.y, ta.z * ddx.x + ta.w * ddx.y}; ddy = {ta.x * ddy.x + ta.y * ddy.y, ta.z * ddy.x + ta.w * ddy.y}; }
        }
    }
    if (GRAD) {
        const float rx2 = ddx.x * ddx.x + ddx.y * ddx.y, ry2 = ddy.x * ddy.x + ddy.y * ddy.y;
        m2 = fmaxf(rx2, ry2);
        if (GRAD == 2) { r2min = fminf(rx2, ry2); major = rx2 >= ry2 ? ddx : ddy; }
    }
    // beyond +-32768 the general sampler's range guard decides (also NaN): the wavefront goes to the general kernel (the probes of an anisotropic
    // footprint stay within 1 / 2 of the major axis of the centre: a footprint that long is beyond every chain's last level anyway)
    bool todo = GRAD != 0 && __builtin_amdgcn_ballot_w64(hit && !(fabsf(u) <= 32768.0f && fabsf(v) <= 32768.0f)) != 0ull;      // (MipmapMode::None: in fetch_all)

    asm volatile("; MARK fetch");
    // ---- levels 2 and 3: the draw's 96-byte lean record and all texel fetches of the pixel.  A strip almost always lies inside ONE draw: the record
    // then comes in by scalar loads — texture bases, extents and flags sit in scalar registers, texel addresses are 32-bit offsets from an SGPR
    // base, the branches on what the material has are scalar.  A strip that straddles draws takes the per-lane form of the same loads. ----
    f3 base, emissive;
    float metallic_in, roughness_in, normal_scale, occlusion_strength, normal_bias, occlusion_bias;
    uint32_t exists;
    lean::Tap tp0, tp1, tp2, tp3, tp4;
    lean::TapG tg0, tg1, tg2, tg3, tg4;                                   // MipmapMode::Gradient: two levels per texture
    constexpr uint32_t kNeed = GRAD == 2 ? 7u : (GRAD ? 3u : 1u);        // LeanDrawDev.flags: lean, ... under MipmapMode::Gradient, ... with anisotropic probes
    // One probe of the footprint: every texel fetch of the pixel at (uu, vv) with the level chosen for m2e.  Called once (the centre) — and, on a context
    // that honours max_anisotropy, once more for every further probe (below).
    // tex_mask / factors (MipmapMode::Gradient without probes): the per-texture branch fetches and consumes its textures in two batches — base colour,
    // metallic-roughness, normal; then occlusion, emissive — so that at most three two-level footprints are live at once: that is what lets the kernel
    // run at 96 registers without scratch (round 5).  The second call leaves the draw's factors alone (the first batch has already multiplied into them).
    auto fetch_all = [&](float uu, float vv, float m2e, bool act, const uint32_t tex_mask = 31u, const bool factors = true) {      // act: this lane takes the probe (always true for the centre)
        if (one_draw) {
            const uint32_t lo = d0 * (uint32_t)sizeof(LeanDrawDev);
            const u32x4 L0 = cload<u32x4>(f.draw_lean, lo), L1 = cload<u32x4>(f.draw_lean, lo + 16u), L2 = cload<u32x4>(f.draw_lean, lo + 32u);
            const u32x4 L3 = cload<u32x4>(f.draw_lean, lo + 48u), L4 = cload<u32x4>(f.draw_lean, lo + 64u);
            const u32x2 L5 = cload<u32x2>(f.draw_lean, lo + 80u), L5s = cload<u32x2>(f.draw_lean, lo + 88u);
            if (GRAD == 0) {
                if (L0.x & 8u) {
                    const f32x4 ta = cload<f32x4>(f.draw_lean, lo + 176u); const f32x2a4 tb = cload<f32x2a4>(f.draw_lean, lo + 192u);
                    const float u2 = affine2_strict(ta.x, ta.y, tb.x, uu, vv), v2 = affine2_strict(ta.z, ta.w, tb.y, uu, vv);
                    uu = u2; vv = v2;
                }
                todo = todo || __builtin_amdgcn_ballot_w64(hit && !(fabsf(uu) <= 32768.0f && fabsf(vv) <= 32768.0f)) != 0ull;
            }
            todo = todo || (L0.x & kNeed) != kNeed;
            const uint32_t exs = todo ? 0u : L0.x >> 8;                       // scalar
            const uint32_t ex = act ? exs & tex_mask : 0u;
            if (GRAD) {
                const u32x4 G0 = cload<u32x4>(f.draw_lean, lo + 96u), G1 = cload<u32x4>(f.draw_lean, lo + 112u), G2 = cload<u32x4>(f.draw_lean, lo + 128u);
                const u32x4 G3 = cload<u32x4>(f.draw_lean, lo + 144u), G4 = cload<u32x4>(f.draw_lean, lo + 160u);
                if (ex & 1u) lean::fetch_trilinear(lean::decode_g(G0.x, G0.y, G0.z, G0.w), m2e, uu, vv, tg0);
                if (ex & 2u) lean::fetch_trilinear(lean::decode_g(G1.x, G1.y, G1.z, G1.w), m2e, uu, vv, tg1);
                if (ex & 4u) lean::fetch_trilinear(lean::decode_g(G2.x, G2.y, G2.z, G2.w), m2e, uu, vv, tg2);
                if (ex & 8u) lean::fetch_trilinear(lean::decode_g(G3.x, G3.y, G3.z, G3.w), m2e, uu, vv, tg3);
                if (ex & 16u) lean::fetch_trilinear(lean::decode_g(G4.x, G4.y, G4.z, G4.w), m2e, uu, vv, tg4);
            } else {
                if (ex & 1u) lean::fetch_s(lean::decode_s(L3.x, L3.y), uu, vv, tp0);
                if (ex & 2u) lean::fetch_s(lean::decode_s(L3.z, L3.w), uu, vv, tp1);
                if (ex & 4u) lean::fetch_s(lean::decode_s(L4.x, L4.y), uu, vv, tp2);
                if (ex & 8u) lean::fetch_s(lean::decode_s(L4.z, L4.w), uu, vv, tp3);
                if (ex & 16u) lean::fetch_s(lean::decode_s(L5.x, L5.y), uu, vv, tp4);
            }
            exists = exs;
            if (factors) {
            metallic_in = __uint_as_float(L0.y); roughness_in = __uint_as_float(L0.z); normal_scale = __uint_as_float(L0.w);
            base = {__uint_as_float(L1.x), __uint_as_float(L1.y), __uint_as_float(L1.z)}; occlusion_strength = __uint_as_float(L1.w);
            emissive = {__uint_as_float(L2.x), __uint_as_float(L2.y), __uint_as_float(L2.z)}; normal_bias = __uint_as_float(L2.w); occlusion_bias = __uint_as_float(L5s.x);
            }
        } else {
            const uint32_t lo = draw * (uint32_t)sizeof(LeanDrawDev);
            const u32x4 L0 = gload<u32x4>(f.draw_lean, lo), L1 = gload<u32x4>(f.draw_lean, lo + 16u), L2 = gload<u32x4>(f.draw_lean, lo + 32u);
            const u32x4 L3 = gload<u32x4>(f.draw_lean, lo + 48u), L4 = gload<u32x4>(f.draw_lean, lo + 64u);
            const u32x2 L5 = gload<u32x2>(f.draw_lean, lo + 80u), L5s = gload<u32x2>(f.draw_lean, lo + 88u);
            if (GRAD == 0) {
                if (__builtin_amdgcn_ballot_w64((L0.x & 8u) != 0u) != 0ull) {
                    const f32x4 ta = gload<f32x4>(f.draw_lean, lo + 176u); const f32x2a4 tb = gload<f32x2a4>(f.draw_lean, lo + 192u);      // (the identity where the draw has none)
                    const float u2 = affine2_strict(ta.x, ta.y, tb.x, uu, vv), v2 = affine2_strict(ta.z, ta.w, tb.y, uu, vv);
                    if (L0.x & 8u) { uu = u2; vv = v2; }
                }
                todo = todo || __builtin_amdgcn_ballot_w64(hit && !(fabsf(uu) <= 32768.0f && fabsf(vv) <= 32768.0f)) != 0ull;
            }
            todo = todo || __builtin_amdgcn_ballot_w64(hit && (L0.x & kNeed) != kNeed) != 0ull;
            const uint32_t exs = (todo || !hit) ? 0u : L0.x >> 8;
            const uint32_t ex = act ? exs & tex_mask : 0u;
            if (GRAD) {
                const u32x4 G0 = gload<u32x4>(f.draw_lean, lo + 96u), G1 = gload<u32x4>(f.draw_lean, lo + 112u), G2 = gload<u32x4>(f.draw_lean, lo + 128u);
                const u32x4 G3 = gload<u32x4>(f.draw_lean, lo + 144u), G4 = gload<u32x4>(f.draw_lean, lo + 160u);
                if (ex & 1u) lean::fetch_trilinear(lean::decode_gl(G0.x, G0.y, G0.z, G0.w), m2e, uu, vv, tg0);
                if (ex & 2u) lean::fetch_trilinear(lean::decode_gl(G1.x, G1.y, G1.z, G1.w), m2e, uu, vv, tg1);
                if (ex & 4u) lean::fetch_trilinear(lean::decode_gl(G2.x, G2.y, G2.z, G2.w), m2e, uu, vv, tg2);
                if (ex & 8u) lean::fetch_trilinear(lean::decode_gl(G3.x, G3.y, G3.z, G3.w), m2e, uu, vv, tg3);
                if (ex & 16u) lean::fetch_trilinear(lean::decode_gl(G4.x, G4.y, G4.z, G4.w), m2e, uu, vv, tg4);
            } else {
                const lean::Tex x0 = lean::decode(L3.x, L3.y), x1 = lean::decode(L3.z, L3.w), x2 = lean::decode(L4.x, L4.y), x3 = lean::decode(L4.z, L4.w), x4 = lean::decode(L5.x, L5.y);
                if (ex & 1u) lean::fetch(x0, uu, vv, tp0);
                if (ex & 2u) lean::fetch(x1, uu, vv, tp1);
                if (ex & 4u) lean::fetch(x2, uu, vv, tp2);
                if (ex & 8u) lean::fetch(x3, uu, vv, tp3);
                if (ex & 16u) lean::fetch(x4, uu, vv, tp4);
            }
            exists = exs;
            if (factors) {
            metallic_in = __uint_as_float(L0.y); roughness_in = __uint_as_float(L0.z); normal_scale = __uint_as_float(L0.w);
            base = {__uint_as_float(L1.x), __uint_as_float(L1.y), __uint_as_float(L1.z)}; occlusion_strength = __uint_as_float(L1.w);
            emissive = {__uint_as_float(L2.x), __uint_as_float(L2.y), __uint_as_float(L2.z)}; normal_bias = __uint_as_float(L2.w); occlusion_bias = __uint_as_float(L5s.x);
            }
        }
    };
    float nf = 1.0f;                                                       // grad_footprint's N; the level is chosen for rho_max / N
    if (GRAD == 2) {
        const uint32_t A = one_draw ? cload<uint32_t>(f.draw_lean, d0 * (uint32_t)sizeof(LeanDrawDev) + 92u) : gload<uint32_t>(f.draw_lean, draw * (uint32_t)sizeof(LeanDrawDev) + 92u);
        if (A > 1u && m2 > 0.0f) {
            const float Af = (float)min(A, 16u);
            nf = r2min * (Af * Af) <= m2 ? Af : __builtin_sqrtf(m2 / r2min);
            nf = fminf(fmaxf(nf, 1.0f), Af);
        }
    }
    const float m2c = GRAD == 2 ? m2 * fm::rcp(nf * nf) : m2;
    // what follows the texel fetches while they are in flight: the pixel's position and its tangent frame (defined below, run inside either branch)
    f3 world_position, surface_to_camera;
    TBN tbn;
    f3 normal;
    float occlusion = 1.0f;
    auto mid = [&]() {
    asm volatile("; MARK standard");
    // ---- standard.wgsl:11-62 (as shade_surface) ----
    // pixel -> NDC, inv_proj and inv_view as ONE matrix, composed on the host in f64 from the camera this frame was submitted with (FrameDev.pix2world):
    // sixteen multiply-adds and one reciprocal instead of two matrix products, two uniform reciprocals and the NDC arithmetic per pixel.
    // standard.wgsl:11-62.  pixel -> NDC and inv_proj are ONE matrix, composed on the host in f64 from the camera this frame was submitted with
    // (FrameDev.pix2view); world = view_rot * view + cam_pos.  (One matrix for all of it, inv_view folded in as well, was measured: the translation
    // then sits inside the cancellation of view_h and a far surface's view vector moves by 1e-5 rad.)
    const float pxf = (float)cx, pyf = (float)cy;
    const float* M = f.pix2view;
    const float hx = M[0] * pxf + (M[4] * pyf + (M[8] * depth + M[12])), hy = M[1] * pxf + (M[5] * pyf + (M[9] * depth + M[13]));
    const float hz = M[2] * pxf + (M[6] * pyf + (M[10] * depth + M[14])), hw = M[3] * pxf + (M[7] * pyf + (M[11] * depth + M[15]));
    const float ivw = fm::rcp(fmaxf(hw, 1e-8f));
    const f3 vp = {hx * ivw, hy * ivw, hz * ivw};                          // view_position
    const float* R = f.view_rot;
    const f3 rel = {R[0] * vp.x + (R[3] * vp.y + R[6] * vp.z), R[1] * vp.x + (R[4] * vp.y + R[7] * vp.z), R[2] * vp.x + (R[5] * vp.y + R[8] * vp.z)};   // world_position - camera
    world_position = {rel.x + f.cam_pos[0], rel.y + f.cam_pos[1], rel.z + f.cam_pos[2]};
#ifdef AWSM_STRICT_POSITION
    world_position = strict_world_position(cload_m4(f.camera, 256u), cload_m4(f.camera, 320u), cx, cy, (float)f.width, (float)f.height, depth);
#endif
    if (f.cam_ortho) {
        surface_to_camera = mk3(f.ortho_view_dir[0], f.ortho_view_dir[1], f.ortho_view_dir[2]);
    } else {
        // cam - world, as standard.wgsl:41-47 forms it (not -rel, which is the same vector without the rounding of the two camera-sized terms: the
        // oracle's result carries that rounding, and a near-mirror texel sees the difference)
        const f3 to_camera = mk3(f.cam_pos[0], f.cam_pos[1], f.cam_pos[2]) - world_position;
        surface_to_camera = fm::fdot(to_camera, to_camera) > 0.0f ? fm::fsafe_normalize(to_camera) : mk3(0.0f, 0.0f, 1.0f);
    }
    asm volatile("; MARK tbn");
    // (Not a leaner unpack without the normalisations of T and B: canonical_tb's 1 / (1 + N.z) makes (t, b) orthonormal only as far as N is a unit
    // vector to the last bit — measured: 7 pixels of a random view 15x out of bounds, all on surfaces facing -z.)
    tbn = fm::funpack_normal_tangent(g.packed_nt);
    normal = tbn.N;
    };
    // ---- MipmapMode::Gradient, every texture of the pixel in the unclamped regime for every lane: ONE footprint per pixel (lean::footprint).  The draw's
    // record by scalar loads for a strip inside one draw, per lane for a strip over several draws and for the resolve kernel's items. ----
    bool shared_fp = false;
    float Lm = 0.0f;
    auto lean_word = [&](auto one, uint32_t off) {      // one word of the lane's (or the strip's) LeanDrawDev
        if constexpr (decltype(one)::value) return cload<uint32_t>(f.draw_lean, d0 * (uint32_t)sizeof(LeanDrawDev) + off);
        else return gload<uint32_t>(f.draw_lean, draw * (uint32_t)sizeof(LeanDrawDev) + off);
    };
    auto lean_quad = [&](auto one, uint32_t off) {
        if constexpr (decltype(one)::value) return cload<u32x4>(f.draw_lean, d0 * (uint32_t)sizeof(LeanDrawDev) + off);
        else return gload<u32x4>(f.draw_lean, draw * (uint32_t)sizeof(LeanDrawDev) + off);
    };
    auto decide = [&](auto one) {
        const uint32_t fl = lean_word(one, 0u);
        const uint32_t exs = (fl & kNeed) == kNeed ? (fl >> 8) & 31u : 0u;
        // bounds over the draw's textures: every lod_t = L + lw_t in (0, levels_t - 1)  <=>  -min lw_t < L < min (levels_t - 1 - lw_t)
        int lw_min = 15, top_min = 15;
#pragma unroll
        for (uint32_t k = 0; k < (uint32_t)kCoreTextures; k++) {
            const uint32_t w1 = lean_word(one, 96u + 16u * k + 4u);
            const int lw = (int)(w1 >> 24), lv = (int)((w1 >> 16) & 15u);
            if (exs & (1u << k)) { lw_min = min(lw_min, lw); top_min = min(top_min, lv - 1 - lw); }
        }
        Lm = 0.5f * __builtin_amdgcn_logf(fmaxf(m2c, 1e-30f));
        const bool in = exs != 0u && Lm > -(float)lw_min && Lm < (float)top_min;
        return __builtin_amdgcn_ballot_w64(hit && !in) == 0ull;
    };
    if (GRAD == 1 && !todo) shared_fp = one_draw ? decide(std::true_type{}) : decide(std::false_type{});
    auto shared_path = [&](auto one) {      // -> false: no lane of the wavefront has anything more to do
        if constexpr (decltype(one)::value) asm volatile("; MARK fetch1"); else asm volatile("; MARK fetch1v");
        lean::W8 w8;
        lean::Quad ql0, qh0, ql1, qh1, ql2, qh2, ql3, qh3, ql4, qh4;
        const u32x4 L0 = lean_quad(one, 0u), L1 = lean_quad(one, 16u), L2 = lean_quad(one, 32u);
        const uint32_t L5s = lean_word(one, 88u);
        const u32x4 G0 = lean_quad(one, 96u), G1 = lean_quad(one, 112u), G2 = lean_quad(one, 128u), G3 = lean_quad(one, 144u), G4 = lean_quad(one, 160u);
        exists = hit ? (L0.x >> 8) & 31u : 0u;
        const float flL = floorf(Lm), ff = Lm - flL;
        const uint32_t cl = hit ? (uint32_t)(-(int)flL) : 1u;                // log2 of the lower level's extent, >= 1 (L < 0 here)
        lean::Foot f_lo, f_hi;
        float fxl, fyl, fxh, fyh;
        lean::footprint(cl, u, v, f_lo, fxl, fyl);
        lean::footprint(cl - 1u, u, v, f_hi, fxh, fyh);
        typedef decltype(lean::decode_any(std::conditional_t<decltype(one)::value, const void*, unsigned long long>{}, 0u, 0u, 0u, 0u)) TexT;
        const std::conditional_t<decltype(one)::value, const void*, unsigned long long> tag{};
        if (exists & 1u) { const TexT x = lean::decode_any(tag, G0.x, G0.y, G0.z, G0.w); lean::fetch_q(x, f_lo, ql0); lean::fetch_q(x, f_hi, qh0); }
        if (exists & 2u) { const TexT x = lean::decode_any(tag, G1.x, G1.y, G1.z, G1.w); lean::fetch_q(x, f_lo, ql1); lean::fetch_q(x, f_hi, qh1); }
        if (exists & 4u) { const TexT x = lean::decode_any(tag, G2.x, G2.y, G2.z, G2.w); lean::fetch_q(x, f_lo, ql2); lean::fetch_q(x, f_hi, qh2); }
        if (exists & 8u) { const TexT x = lean::decode_any(tag, G3.x, G3.y, G3.z, G3.w); lean::fetch_q(x, f_lo, ql3); lean::fetch_q(x, f_hi, qh3); }
        if (exists & 16u) { const TexT x = lean::decode_any(tag, G4.x, G4.y, G4.z, G4.w); lean::fetch_q(x, f_lo, ql4); lean::fetch_q(x, f_hi, qh4); }
        metallic_in = __uint_as_float(L0.y); roughness_in = __uint_as_float(L0.z); normal_scale = __uint_as_float(L0.w);
        base = {__uint_as_float(L1.x), __uint_as_float(L1.y), __uint_as_float(L1.z)}; occlusion_strength = __uint_as_float(L1.w);
        emissive = {__uint_as_float(L2.x), __uint_as_float(L2.y), __uint_as_float(L2.z)}; normal_bias = __uint_as_float(L2.w); occlusion_bias = __uint_as_float(L5s);
        const float gl = 1.0f - ff, gxl = 1.0f - fxl, gyl = 1.0f - fyl, gxh = 1.0f - fxh, gyh = 1.0f - fyh;
        const float a = gyl * gl, b = fyl * gl, c = gyh * ff, d = fyh * ff;
        w8 = {gxl * a, fxl * a, gxl * b, fxl * b, gxh * c, fxh * c, gxh * d, fxh * d};
        if (!hit) return false;
        mid();
        if constexpr (decltype(one)::value) asm volatile("; MARK material1"); else asm volatile("; MARK material1v");
        if (exists & 1u) base = {base.x * lean::channel8<0>(ql0, qh0, w8), base.y * lean::channel8<1>(ql0, qh0, w8), base.z * lean::channel8<2>(ql0, qh0, w8)};
        if (exists & 2u) { metallic_in = metallic_in * lean::channel8<2>(ql1, qh1, w8); roughness_in = roughness_in * lean::channel8<1>(ql1, qh1, w8); }
        if (exists & 4u) {   // material_color_calc.wgsl:301-322
            const float ntx = lean::channel8<0>(ql2, qh2, w8) * normal_scale - normal_bias, nty = lean::channel8<1>(ql2, qh2, w8) * normal_scale - normal_bias, ntz = lean::channel8<2>(ql2, qh2, w8) * (2.0f / 255.0f) - 1.0f;
            normal = fm::fnormalize(tbn.T * ntx + tbn.B * nty + tbn.N * ntz);
        }
        if (exists & 8u) occlusion = lean::channel8<0>(ql3, qh3, w8) * occlusion_strength + occlusion_bias;      // mix(1, r, s)
        if (exists & 16u) emissive = {emissive.x * lean::channel8<0>(ql4, qh4, w8), emissive.y * lean::channel8<1>(ql4, qh4, w8), emissive.z * lean::channel8<2>(ql4, qh4, w8)};
        return true;
    };
    if (GRAD == 1 && shared_fp) {
        if (!(one_draw ? shared_path(std::true_type{}) : shared_path(std::false_type{}))) return true;
    } else {
    fetch_all(u, v, m2c, true, GRAD == 1 ? 7u : 31u);
    if (todo) {      // this wavefront goes to the general kernel (k_shade_todo): nothing of a hit pixel has been written
        if (!ITEMS && lane == 0u) {
            const uint32_t slot = atomicAdd(&f.shade_todo[0], 1u);
            if (slot < f.shade_todo_cap) f.shade_todo[4u + slot] = (wg << 2) | (tid >> 6);
        }
        return false;
    }
    if (!hit) return true;

    // ---- AWSM_CFG_ANISOTROPIC: the twelve raw channel values (0..255) the material reads, averaged over grad_footprint's probes — the centre (weight 1),
    // then pairs at +-j / N along the major axis, each weighted by the part of the footprint its cell covers (zero beyond the lane's own
    // m = ceil((N - 1) / 2): the wavefront walks to its longest footprint), normalised.  Before anything else of the pixel is computed: the loop is where
    // the registers go. ----
    float ch[12] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    if (GRAD == 2) {
#define AWSM_LEAN_ACC(I, K, BYTE) ch[I] += lean::channel<BYTE>(tg##K, wl, wh) * wj
        auto accumulate = [&](float wj) {
            if (exists & 1u) { const lean::Weights wl = lean::weights(tg0.lo), wh = lean::weights(tg0.hi); AWSM_LEAN_ACC(0, 0, 0); AWSM_LEAN_ACC(1, 0, 1); AWSM_LEAN_ACC(2, 0, 2); }
            if (exists & 2u) { const lean::Weights wl = lean::weights(tg1.lo), wh = lean::weights(tg1.hi); AWSM_LEAN_ACC(3, 1, 2); AWSM_LEAN_ACC(4, 1, 1); }
            if (exists & 4u) { const lean::Weights wl = lean::weights(tg2.lo), wh = lean::weights(tg2.hi); AWSM_LEAN_ACC(5, 2, 0); AWSM_LEAN_ACC(6, 2, 1); AWSM_LEAN_ACC(7, 2, 2); }
            if (exists & 8u) { const lean::Weights wl = lean::weights(tg3.lo), wh = lean::weights(tg3.hi); AWSM_LEAN_ACC(8, 3, 0); }
            if (exists & 16u) { const lean::Weights wl = lean::weights(tg4.lo), wh = lean::weights(tg4.hi); AWSM_LEAN_ACC(9, 4, 0); AWSM_LEAN_ACC(10, 4, 1); AWSM_LEAN_ACC(11, 4, 2); }
        };
#undef AWSM_LEAN_ACC
        accumulate(1.0f);
        const float mf_ = ceilf((nf - 1.0f) * 0.5f);
        int wave_m = 0;
#pragma unroll
        for (int j = 1; j <= 8; j++) if (__builtin_amdgcn_ballot_w64(mf_ >= (float)j) != 0ull) wave_m = j;
        if (wave_m) {
            // The further probes, ONE TEXTURE AT A TIME: per texture the pair of probes at +-j / N is in flight (two trilinear footprints, 26 registers) while
            // the twelve sums wait — not the footprints of all five textures at once, as the centre probe has them (that loop held 200 registers: two
            // wavefronts per SIMD).  Every channel still receives its terms in the order centre, +1, -1, +2, -2, ...: the same bits as before.
            const float inv_n = fm::rcp(nf);
            float wsum = 1.0f;
            for (int j = 1; j <= wave_m; j++) { const float tj = (float)j * inv_n, wj = saturate((0.5f - tj) * nf + 0.5f); wsum += 2.0f * wj; }
            auto probes = [&](const auto& tex, const bool has, auto&& acc) {
                if (__builtin_amdgcn_ballot_w64(has) == 0ull) return;      // (out of the lambda)
#pragma unroll 1
                for (int j = 1; j <= wave_m; j++) {
                    const float tj = (float)j * inv_n, wj = saturate((0.5f - tj) * nf + 0.5f);      // falls with j: once no lane has a weight left, none will
                    const bool act = has && wj > 0.0f;
                    if (__builtin_amdgcn_ballot_w64(act) == 0ull) break;
                    if (act) {
                        lean::TapG ta, tb;
                        const float tn = -tj;
                        lean::fetch_trilinear(tex, m2c, u + major.x * tj, v + major.y * tj, ta);
                        lean::fetch_trilinear(tex, m2c, u + major.x * tn, v + major.y * tn, tb);
                        acc(ta, wj); acc(tb, wj);
                    }
                }
            };
#define AWSM_LEAN_ACC(I, BYTE) ch[I] += lean::channel<BYTE>(t, wl, wh) * wj
            auto acc0 = [&](const lean::TapG& t, float wj) { const lean::Weights wl = lean::weights(t.lo), wh = lean::weights(t.hi); AWSM_LEAN_ACC(0, 0); AWSM_LEAN_ACC(1, 1); AWSM_LEAN_ACC(2, 2); };
            auto acc1 = [&](const lean::TapG& t, float wj) { const lean::Weights wl = lean::weights(t.lo), wh = lean::weights(t.hi); AWSM_LEAN_ACC(3, 2); AWSM_LEAN_ACC(4, 1); };
            auto acc2 = [&](const lean::TapG& t, float wj) { const lean::Weights wl = lean::weights(t.lo), wh = lean::weights(t.hi); AWSM_LEAN_ACC(5, 0); AWSM_LEAN_ACC(6, 1); AWSM_LEAN_ACC(7, 2); };
            auto acc3 = [&](const lean::TapG& t, float wj) { const lean::Weights wl = lean::weights(t.lo), wh = lean::weights(t.hi); AWSM_LEAN_ACC(8, 0); };
            auto acc4 = [&](const lean::TapG& t, float wj) { const lean::Weights wl = lean::weights(t.lo), wh = lean::weights(t.hi); AWSM_LEAN_ACC(9, 0); AWSM_LEAN_ACC(10, 1); AWSM_LEAN_ACC(11, 2); };
#undef AWSM_LEAN_ACC
            if (one_draw) {
                const uint32_t lo = d0 * (uint32_t)sizeof(LeanDrawDev);
#define AWSM_LEAN_PROBE_TEX(K, ACC) { const u32x4 G = cload<u32x4>(f.draw_lean, lo + 96u + 16u * K); probes(lean::decode_g(G.x, G.y, G.z, G.w), (exists & (1u << K)) != 0u, ACC); }
                AWSM_LEAN_PROBE_TEX(0, acc0) AWSM_LEAN_PROBE_TEX(1, acc1) AWSM_LEAN_PROBE_TEX(2, acc2) AWSM_LEAN_PROBE_TEX(3, acc3) AWSM_LEAN_PROBE_TEX(4, acc4)
#undef AWSM_LEAN_PROBE_TEX
            } else {
                const uint32_t lo = draw * (uint32_t)sizeof(LeanDrawDev);
#define AWSM_LEAN_PROBE_TEX(K, ACC) { const u32x4 G = gload<u32x4>(f.draw_lean, lo + 96u + 16u * K); probes(lean::decode_gl(G.x, G.y, G.z, G.w), (exists & (1u << K)) != 0u, ACC); }
                AWSM_LEAN_PROBE_TEX(0, acc0) AWSM_LEAN_PROBE_TEX(1, acc1) AWSM_LEAN_PROBE_TEX(2, acc2) AWSM_LEAN_PROBE_TEX(3, acc3) AWSM_LEAN_PROBE_TEX(4, acc4)
#undef AWSM_LEAN_PROBE_TEX
            }
            const float iw = fm::rcp(wsum);
#pragma unroll
            for (int i = 0; i < 12; i++) ch[i] *= iw;
        }
    }

    mid();

    asm volatile("; MARK material");
    // ---- material_color_calc.wgsl:25-265 for a material without optional blocks ----
    // one texture's channel BYTE: bilinear on level 0 (MipmapMode::None) or the blend of two levels' bilinear values (MipmapMode::Gradient)
#define AWSM_LEAN_TEX(K, W0, W1) const lean::Weights W0 = lean::weights(GRAD ? tg##K.lo : tp##K), W1 = lean::weights(GRAD ? tg##K.hi : tp##K)
#define AWSM_LEAN_CH(K, BYTE, W0, W1) (GRAD ? lean::channel<BYTE>(tg##K, W0, W1) : lean::channel<BYTE>(tp##K, W0))
    if (GRAD != 2) {
        if (exists & 1u) { AWSM_LEAN_TEX(0, w, wh); base = {base.x * AWSM_LEAN_CH(0, 0, w, wh), base.y * AWSM_LEAN_CH(0, 1, w, wh), base.z * AWSM_LEAN_CH(0, 2, w, wh)}; }
        if (exists & 2u) { AWSM_LEAN_TEX(1, w, wh); metallic_in = metallic_in * AWSM_LEAN_CH(1, 2, w, wh); roughness_in = roughness_in * AWSM_LEAN_CH(1, 1, w, wh); }
        if (exists & 4u) {   // material_color_calc.wgsl:301-322
            AWSM_LEAN_TEX(2, w, wh);
            // (c * 2 - 1) * scale on raw texels: raw * (2 scale / 255) - scale (LeanDrawDev)
            const float ntx = AWSM_LEAN_CH(2, 0, w, wh) * normal_scale - normal_bias, nty = AWSM_LEAN_CH(2, 1, w, wh) * normal_scale - normal_bias, ntz = AWSM_LEAN_CH(2, 2, w, wh) * (2.0f / 255.0f) - 1.0f;
            normal = fm::fnormalize(tbn.T * ntx + tbn.B * nty + tbn.N * ntz);
        }
        if (GRAD == 1 && __builtin_amdgcn_ballot_w64((exists & 24u) != 0u) != 0ull) fetch_all(u, v, m2c, true, 24u, false);      // the second batch
        if (exists & 8u) { AWSM_LEAN_TEX(3, w, wh); occlusion = AWSM_LEAN_CH(3, 0, w, wh) * occlusion_strength + occlusion_bias; }      // mix(1, r, s)
        if (exists & 16u) { AWSM_LEAN_TEX(4, w, wh); emissive = {emissive.x * AWSM_LEAN_CH(4, 0, w, wh), emissive.y * AWSM_LEAN_CH(4, 1, w, wh), emissive.z * AWSM_LEAN_CH(4, 2, w, wh)}; }
    } else {
        if (exists & 1u) base = {base.x * ch[0], base.y * ch[1], base.z * ch[2]};
        if (exists & 2u) { metallic_in = metallic_in * ch[3]; roughness_in = roughness_in * ch[4]; }
        if (exists & 4u) {
            const float ntx = ch[5] * normal_scale - normal_bias, nty = ch[6] * normal_scale - normal_bias, ntz = ch[7] * (2.0f / 255.0f) - 1.0f;
            normal = fm::fnormalize(tbn.T * ntx + tbn.B * nty + tbn.N * ntz);
        }
        if (exists & 8u) occlusion = ch[8] * occlusion_strength + occlusion_bias;
        if (exists & 16u) emissive = {emissive.x * ch[9], emissive.y * ch[10], emissive.z * ch[11]};
    }
#undef AWSM_LEAN_TEX
#undef AWSM_LEAN_CH
    }      // (the per-texture branch)

    asm volatile("; MARK surface");
    // ---- lights.wgsl:121-152 / brdf.wgsl (apply_lighting, brdf_ibl, brdf_direct above, with ior 1.5, specular 1, no transmission / clearcoat / sheen) ----
    Surface sf;
    sf.n = fm::fsafe_normalize(normal);
    sf.v = fm::fsafe_normalize(surface_to_camera);
    sf.metallic = clampf(metallic_in, 0.0f, 1.0f);
    sf.roughness = fmaxf(clampf(roughness_in, 0.0f, 1.0f), 0.04f);
    sf.alpha = sf.roughness * sf.roughness;
    const float ndv = fm::fdot(sf.n, sf.v);
    sf.n_dot_v_ibl = saturate(ndv);
    sf.n_dot_v_dir = fmaxf(ndv, 1e-4f);
    const float f0b = ior_to_f0(1.5f);
    sf.F0 = mix3(splat3(fminf(f0b, 1.0f)), base, sf.metallic);
    sf.f90 = mixf(1.0f, 1.0f, sf.metallic);
    sf.sheen_scaling_dir = 1.0f;
    sf.g1_v = geometry_schlick_ggx(saturate(ndv), sf.alpha);
    sf.cc_n = sf.n;
    sf.df90 = splat3(sf.f90) - sf.F0;
    sf.base_diffuse = base * ((1.0f - sf.metallic) * (1.0f / kPi));
    const float ac = fmaxf(sf.alpha, 0.001f);
    sf.a2 = ac * ac; sf.a2m1 = sf.a2 - 1.0f;
    sf.gk = ((ac + 1.0f) * (ac + 1.0f)) * 0.125f; sf.one_m_gk = 1.0f - sf.gk;
    sf.has_sheen = false; sf.has_clearcoat = false;
    lean::Lit lit;
    lit.n = sf.n; lit.v = sf.v; lit.F0 = sf.F0; lit.df90 = sf.df90; lit.bd = sf.base_diffuse;
    lit.ndv_raw = ndv; lit.F0max = fmaxf(fmaxf(sf.F0.x, sf.F0.y), sf.F0.z); lit.df90max = sf.f90 - lit.F0max;
    lit.ndv_dir = sf.n_dot_v_dir; lit.ndv4 = 4.0f * sf.n_dot_v_dir; lit.a2m1 = sf.a2m1; lit.a2_g1v = sf.a2 * sf.g1_v; lit.gk = sf.gk; lit.one_m_gk = sf.one_m_gk;
    lit.occlusion = occlusion;
    f3 color;
    asm volatile("; MARK ibl");
    {   // brdf_ibl (brdf.wgsl:517-576): the uniform cubes of the builder default, or texel cubes (brdf.wgsl:268-290: irradiance at level 0 along N,
        // prefiltered at roughness * (mips - 1) along R) through the shared seam-aware sampler
        const f3 prefiltered = lean::prefiltered(sc, reflect3(-sf.v, sf.n), sf.roughness);
        const f3 irradiance = lean::irradiance(sc, sf.n);
        const float n_dot_v = sf.n_dot_v_ibl;
        const f3 F_view = fresnel_schlick_f90(n_dot_v, sf.F0, sf.f90);
        const float F_view_max = fmaxf(fmaxf(F_view.x, F_view.y), F_view.z);
        const f3 base_layer = (base * (1.0f / kPi)) * irradiance;
        const float k_d = (1.0f - F_view_max) * (1.0f - sf.metallic);
        const f3 base_contribution = (base_layer * k_d) * occlusion;
        const f2 lut = lean::brdf_lut(sc->lut_rg16f, sc->lut_w, sc->lut_h, n_dot_v, sf.roughness);
        const f3 spec_term = sf.F0 * lut.x + splat3(sf.f90 * lut.y);
        const f3 specular = (prefiltered * spec_term) * mixf(1.0f, occlusion, 0.5f);
        color = (base_contribution + specular) + emissive;
    }
    asm volatile("; MARK lights");
    const uint32_t n_lights = min(cload<uint32_t>(sc->buf[AWSM_BUF_LIGHTS_INFO], 0u), f.lights_cap);
    const void* lights = sc->buf[AWSM_BUF_LIGHTS];
    for (uint32_t i = 0; i < n_lights; i++) {
        const f32x4 pre0 = cload<f32x4>(f.lights_pre, i * 32u), pre1 = cload<f32x4>(f.lights_pre, i * 32u + 16u);
        const uint32_t kind = (uint32_t)pre0.w;
        f3 light_dir = {pre0.x, pre0.y, pre0.z}, radiance = {pre1.x, pre1.y, pre1.z};
        if (kind == 2u || kind == 3u) {
            const f32x4 pos_range = cload<f32x4>(lights, i * 64u);
            const f3 stl = mk3(pos_range.x, pos_range.y, pos_range.z) - world_position;
            const float d2 = fm::fdot(stl, stl);
            const float inv_d = d2 > 0.0f ? fm::rsq(d2) : 0.0f;
            const float dist = d2 * inv_d;
            float att;   // math.wgsl:12-19 inverse_square
            if (pos_range.w == 0.0f) att = fm::rcp(fmaxf(dist * dist, 0.01f));
            else { const float fo = 1.0f - fm::fdiv(dist * dist, pos_range.w * pos_range.w); att = fm::fdiv(saturate(fo * fo), dist * dist + 1.0f); }
            const f3 to_light = stl * inv_d;
            if (kind == 3u) {
                const f32x4 dir_inner = cload<f32x4>(lights, i * 64u + 16u), kind_outer = cload<f32x4>(lights, i * 64u + 48u);
                const float cos_l = fm::fdot(to_light, -light_dir);
                const float sm = saturate(fm::fdiv(cos_l - kind_outer.y, dir_inner.w - kind_outer.y));
                att = att * (sm * sm);
            }
            light_dir = to_light;
            radiance = radiance * att;
#ifndef AWSM_NO_LIGHT_SKIP
            if (__builtin_amdgcn_ballot_w64(!(att == 0.0f)) == 0ull) continue;      // out of the light's range (or cone) for the whole wavefront: the term is exactly zero
#endif
        } else if (kind != 1u) { light_dir = {0.0f, 0.0f, 0.0f}; radiance = {0.0f, 0.0f, 0.0f}; }
        lean::direct(lit, light_dir, radiance, color);
    }
    asm volatile("; MARK store");
    if (ITEMS) { item->color = color; return true; }
    store_pixel(f, p, {color.x, color.y, color.z, 1.0f});
    if (MSAA && want_c0) f.msaa_color0[pv] = make_float4(color.x, color.y, color.z, 1.0f);
    return true;
}
template <int GRAD, bool MSAA>
AWSM_DI void lean_block(const DevScene* __restrict__ sc, const FrameDev& f, const ShadeBlock& b, const uint32_t wg, const uint32_t tid, LeanStage* __restrict__ st) {
    (void)lean_core<GRAD, MSAA, false>(sc, f, b, wg, tid, st, nullptr);
}
// k_shade_lean<false>: a wavefront per 16x4-pixel strip.  k_shade_lean<true>: a persistent grid (lean_grid workgroups; workgroup w
// runs on XCD w & 7, as the hardware deals them) whose wavefronts take strips from counters until the XCD's share is used up —
// a grid of 4 workgroups per CU leaves the rest of every CU (wave slots, 160 of 512 VGPRs per lane, all LDS) to the next frame's
// geometry kernels on the other stream, which a grid of 34 k workgroups starves.
// XCD x owns the block rows x, x + 8, ...; its strips are numbered row-major on a power-of-two block pitch (strip -> block row /
// column by shifts; ids in the padding are skipped), four strips per block, so wavefronts running together shade neighbouring
// strips.  A same-address atomic costs ~25 ns at the L2 — one counter per XCD would take longer than the shading — so every XCD
// has kLeanCounters of them on separate cache lines, counter c handing out the strips c, c + kLeanCounters, ...; a wavefront
// starts on counter w % kLeanCounters and moves on when it runs dry.  The request for the next strip is issued before the current
// one is shaded.  No barrier, no LDS.
// k_shade_lean<.., 2, ..> (anisotropic probes): 117 registers since the probes beyond the centre go one texture at a time (round 4; 200 before, two
// wavefronts per SIMD): four wavefronts per SIMD like the isotropic gradient kernel.
#ifndef AWSM_ANISO_WAVES
#define AWSM_ANISO_WAVES 4
#endif
#ifndef AWSM_LEAN_COUNTERS
#define AWSM_LEAN_COUNTERS 8
#endif
constexpr uint32_t kLeanCounters = AWSM_LEAN_COUNTERS;     // <= 8 (lean_next holds 64 counter lines)
// GRAD: MipmapMode::Gradient (the reference's default): barycentric derivatives, isotropic LOD, two levels per texture — a separate instantiation, as the
// reference keeps separate pipelines; its ten footprints in flight want more registers than six waves per SIMD leave.
// Five wavefronts per SIMD (96 registers) since round 5: the strips that share one footprint (lean::footprint: the bulk of a minified frame) need 96 without
// a spill, and what the cap spills (48 bytes of scratch) sits in the per-texture branch that the other strips take.  Measured at 4K, MSAA x4 + mips: the
// kernel 330 -> 301 us, the frame 1,777 -> 1,883 frames/s (profiles/r05_lean_grad_waves.txt).  Round 4 had measured five wavefronts as a loss — with
// every strip on the per-texture code the spills sat in the texel-fetch burst of all of them.
#ifndef AWSM_LEAN_GRAD_WAVES
#define AWSM_LEAN_GRAD_WAVES 5
#endif
template <bool PERSIST, int GRAD, bool MSAA>   // the loop state costs the persistent variant 5 VGPRs: 85 (-> 88 allocated) instead of 80; under an 80 cap it spills inside the texel-fetch burst
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(GRAD == 2 ? AWSM_ANISO_WAVES : (GRAD ? AWSM_LEAN_GRAD_WAVES : (PERSIST ? 5 : AWSM_LEAN_WAVES))))) void k_shade_lean(const DevScene* __restrict__ sc, FrameDev f) {
    __shared__ LeanStage stage[4];                                         // one per wavefront (no barrier anywhere: the four are independent)
    if (frame_poisoned(f)) return;
#ifdef AWSM_LEAN_PRIO
    __builtin_amdgcn_s_setprio(AWSM_LEAN_PRIO);                            // experiment: issue priority over the geometry kernels' wavefronts on the same SIMD
#endif
    LeanStage* const st = &stage[__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))];
    const uint32_t xcd = blockIdx.x & 7u, lane = threadIdx.x & 63u;
    const uint32_t bx_n = (f.width + 15u) >> 4, by_n = f.band_n > 1u ? 2u * f.tiles_y : ((f.sy1 - f.sy0) + 15u) >> 4;
    const uint32_t lp = 32u - (uint32_t)__builtin_clz((bx_n - 1u) | 1u);  // log2 of the block pitch
    const uint32_t share = (((by_n + 7u - xcd) >> 3) << lp) * 4u;         // strip ids of this XCD (padding included)
    const uint32_t w = __builtin_amdgcn_readfirstlane((blockIdx.x >> 3) * 4u + (threadIdx.x >> 6));    // wavefront of this XCD
    uint32_t zero;                                                        // opaque to the compiler: with a provably uniform address its atomic optimiser
    asm volatile("v_mov_b32 %0, 0" : "=v"(zero));                         // broadcasts the result right behind the atomic (s_waitcnt + readfirstlane)
    uint32_t c = w % kLeanCounters, tried = 0u;
    uint32_t* ctr0 = f.lean_next + (xcd * kLeanCounters) * 16u + zero;
    uint32_t nxt = 0u;
    if (PERSIST && lane == 0u) nxt = atomicAdd(ctr0 + c * 16u, 1u);
    for (uint32_t j = w;;) {
        if (PERSIST) {
            j = __builtin_amdgcn_readfirstlane(nxt) * kLeanCounters + c;
            while (j >= share && ++tried < kLeanCounters) {               // this counter's strips are gone: the next one (once per wavefront and counter)
                c = (c + 1u) % kLeanCounters;
                if (lane == 0u) nxt = atomicAdd(ctr0 + c * 16u, 1u);
                j = __builtin_amdgcn_readfirstlane(nxt) * kLeanCounters + c;
            }
            if (j >= share) break;
            if (lane == 0u) nxt = atomicAdd(ctr0 + c * 16u, 1u);
        }
        const uint32_t kb = j >> 2, bcol = kb & ((1u << lp) - 1u), brow = (kb >> lp) * 8u + xcd;
        if (bcol < bx_n && j < share) {
            ShadeBlock b;
            shade_block_at(f, b, brow, bcol);
            lean_block<GRAD, MSAA>(sc, f, b, ((kb >> lp) * bx_n + bcol) * 8u + xcd, ((j & 3u) << 6) | lane, st);
        }
        if (!PERSIST) break;
    }
}

// ------------------------------------------------------------------------------------------------
// MSAA x4 opaque pass = k_shade_msaa + k_shade_msaa_resolve (compute.wgsl:118-170,303-318, helpers/msaa.wgsl,
// helpers/material_shading.wgsl:25-210).
//
// k_shade_msaa shades sample 0 of every pixel and runs the edge detector.  The detector compares the pixel's normal with
// its four neighbours' (sample 0): each thread publishes its strict normal + depth in an 18x18 LDS tile, the first 68
// threads fill the halo ring by reconstructing the neighbouring blocks' border pixels.  Edge pixels are not resolved in
// place — a wavefront would run the whole shading four times for a handful of its lanes — but appended to the block's
// list (LDS counter, no global atomics); k_shade_msaa_resolve then shades their remaining DISTINCT triangles densely.
// All samples a triangle covers in a pixel carry the same G-buffer texel (centre-evaluated) and the standard
// coordinates are shared (sample 0's depth), so equal triangle => equal colour and is shaded once.
// Contract choice: a neighbour outside the frame contributes nothing to the edge test (WGSL leaves out-of-bounds
// textureLoad to the implementation).
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kEdgeRecBytes = 260;      // per block: u32 count + 256 one-byte pixel slots
struct NeighbourCell { float nx, ny, nz; uint32_t depth_bits; uint32_t state; };   // state 0: outside the frame, 1: background, 2: covered

AWSM_DI void publish_cell(NeighbourCell* cells, int lx, int ly, const FrameDev& f, int px, int py) {
    NeighbourCell c = {0.0f, 0.0f, 0.0f, 0u, 0u};
    if (px >= 0 && px < (int)f.width && py >= (int)f.y0 && py < (int)f.y1) {   // rasterised rows: the shard's + one halo row each side
        unsigned long long k;
        if (f.band_n > 1u && (((uint32_t)py >> kTileShift) % f.band_n) != f.band_r) {
            // band sharding: the row belongs to another rank's band; its sample-0 keys came through the halo exchange (first / last row of a band)
            const uint32_t ty = (uint32_t)py >> kTileShift, which = ((uint32_t)py & (uint32_t)(kTile - 1)) == 0u ? 0u : 1u;
            k = f.msaa_halo ? f.msaa_halo[((((size_t)(ty % f.band_n) * f.halo_bands + ty / f.band_n) * 2u + which) * f.width) + (size_t)px] : ~0ull;
        } else k = f.vis[((size_t)py * f.width + (size_t)px) * 4];
        c.state = 1u;
        if (k != ~0ull) {
            const f3 n = strict_normal_of(f, key_rank(k), px, py);                  // strict unpack_normal_tangent(..).N
            c.nx = n.x; c.ny = n.y; c.nz = n.z; c.depth_bits = (uint32_t)(k >> 32); c.state = 2u;
        }
    }
    cells[(ly + 1) * 18 + (lx + 1)] = c;
}

// msaa.wgsl:42-112: the pixel against its four neighbours (sample 0 each) — a background neighbour, a normal that turns away, a view depth that jumps.
// STRICT.  `cell(ox, oy)` hands out the neighbour's NeighbourCell.
template <typename CellAt>
AWSM_DI bool edge_by_neighbours(const m4& inv_proj, f3 center_normal, float center_depth, float pcx, float pcy, float W, float H, CellAt cell) {
    bool is_edge = false, center_loaded = false;
    float view_depth_c = 0.0f, depth_threshold = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int ox = i == 0 ? 1 : (i == 1 ? -1 : 0), oy = i == 2 ? 1 : (i == 3 ? -1 : 0);
        if (is_edge) continue;
        const NeighbourCell nb = cell(ox, oy);
        if (nb.state == 0u) continue;
        if (nb.state == 1u) { is_edge = true; continue; }          // neighbour is background
        if (dot(center_normal, mk3(nb.nx, nb.ny, nb.nz)) < kEdgeNormalThreshold) { is_edge = true; continue; }
        if (!center_loaded) {
            view_depth_c = view_space_depth(inv_proj, center_depth, pcx, pcy, W, H);
            depth_threshold = kEdgeDepthThreshold * fabsf(view_depth_c);
            center_loaded = true;
        }
        const float nvd = view_space_depth(inv_proj, __uint_as_float(nb.depth_bits), pcx + (float)ox, pcy + (float)oy, W, H);
        if (fabsf(view_depth_c - nvd) > depth_threshold) is_edge = true;
    }
    return is_edge;
}

// Edge detection of the lean MSAA route (k_msaa_detect, after k_shade_lean<.., MSAA> and k_shade_todo<.., MSAA>): the block's
// edge list from the two masks per strip and the 8-byte cells — {octahedral normal as two f16 (the G-buffer texel's own bits), depth bits; 0xFFFFFFFF = background} — the lean
// kernel left.  Everything heavy, the reconstruction, happened where it had to happen anyway; what is left is latency, so the pixel's cell and its four
// neighbours' are requested together with the masks, before anything is known.  Decisions are msaa.wgsl:42-112's, STRICT, behind filters: the normals are
// decoded with rsq and compared with a margin of 1e-5 around the threshold (the fast decode is within 1e-6), the depths as depth_only_projection's
// comment says; whatever lands inside a margin is redone with the strict operations.  A neighbour in a halo row (the row above / below a row strip,
// another rank's band) has no cell: its normal is reconstructed from its key, as k_shade_msaa does for its ring.
AWSM_DI uint2 halo_cell(const FrameDev& f, int px, int py) {
    unsigned long long k;
    if (f.band_n > 1u && (((uint32_t)py >> kTileShift) % f.band_n) != f.band_r) {
        const uint32_t ty = (uint32_t)py >> kTileShift, which = ((uint32_t)py & (uint32_t)(kTile - 1)) == 0u ? 0u : 1u;
        k = f.msaa_halo ? f.msaa_halo[((((size_t)(ty % f.band_n) * f.halo_bands + ty / f.band_n) * 2u + which) * f.width) + (size_t)px] : ~0ull;
    } else k = f.vis[((size_t)py * f.width + (size_t)px) * 4];
    if (k == ~0ull) return make_uint2(0u, 0xFFFFFFFFu);
    return make_uint2(oct_word(strict_oct_of(f, key_rank(k), px, py)), (uint32_t)(k >> 32));
}
// One test pixel of the block (slot = ly * 16 + lx): msaa.wgsl:42-112 against the cells.
AWSM_DI bool detect_pixel(const FrameDev& f, const ShadeBlock& b, uint32_t slot, const m4& inv_proj, bool fastp) {
    const int cx = b.x0 + (int)(slot & 15u), cy = b.y0 + (int)(slot >> 4);
    // state 0: outside the frame / the rasterised rows (contributes nothing), 1: a cell, 2: a halo row (reconstructed on demand)
    uint2 nb[4];
    uint32_t state[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int px = cx + (i == 0 ? 1 : (i == 1 ? -1 : 0)), py = cy + (i == 2 ? 1 : (i == 3 ? -1 : 0));
        nb[i] = make_uint2(0u, 0xFFFFFFFFu); state[i] = 0u;
        if (px >= 0 && px < (int)f.width && py >= (int)f.y0 && py < (int)f.y1) {
            if (i < 2 || row_owned(f, py)) { nb[i] = f.msaa_cells[(size_t)py * f.width + (size_t)px]; state[i] = 1u; }
            else state[i] = 2u;
        }
    }
    const uint2 me = f.msaa_cells[(size_t)cy * f.width + (size_t)cx];
    const float W = (float)f.width, H = (float)f.height, pcx = (float)cx + 0.5f, pcy = (float)cy + 0.5f;
    const f2 oct_c = oct_of_word(me.x);
    const f3 nc = fm::fdecode_octahedral(oct_c);
    const float dc = __uint_as_float(me.y);
    const float vdc = fastp ? view_depth_approx(inv_proj, dc) : view_space_depth(inv_proj, dc, pcx, pcy, W, H);
    bool is_edge = false;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int ox = i == 0 ? 1 : (i == 1 ? -1 : 0), oy = i == 2 ? 1 : (i == 3 ? -1 : 0);
        if (is_edge || state[i] == 0u) continue;
        uint2 c = nb[i];
        if (state[i] == 2u) c = halo_cell(f, cx + ox, cy + oy);
        if (c.y == 0xFFFFFFFFu) { is_edge = true; continue; }          // neighbour is background
        const f2 oct_n = oct_of_word(c.x);
        const f3 nn = fm::fdecode_octahedral(oct_n);
        const float d = (nc.x * nn.x + nc.y * nn.y) + nc.z * nn.z;
        bool turned = d < kEdgeNormalThreshold - 1e-5f;
        if (!turned && !(d > kEdgeNormalThreshold + 1e-5f)) turned = dot(decode_octahedral(oct_c), decode_octahedral(oct_n)) < kEdgeNormalThreshold;
        if (turned) { is_edge = true; continue; }
        const float dn = __uint_as_float(c.y);
        bool decided = false, jump = false;
        if (fastp) {
            const float vdn = view_depth_approx(inv_proj, dn);
            const float lhs = fabsf(vdc - vdn), rhs = kEdgeDepthThreshold * fabsf(vdc), margin = 4e-6f * fmaxf(fabsf(vdc), fabsf(vdn));
            if (lhs > rhs + margin) { decided = true; jump = true; }
            else if (lhs < rhs - margin) decided = true;
        }
        if (!decided) {
            const float svc = view_space_depth(inv_proj, dc, pcx, pcy, W, H);
            jump = fabsf(svc - view_space_depth(inv_proj, dn, pcx + (float)ox, pcy + (float)oy, W, H)) > kEdgeDepthThreshold * fabsf(svc);
        }
        if (jump) is_edge = true;
    }
    return is_edge;
}
// The block's edge list (one-byte pixel slots, in LDS) from the strips' masks, by ONE wavefront: the sure pixels as they are, the test pixels compacted
// first — a wavefront per strip would run the test with a quarter of its lanes — then tested 64 at a time.  Returns the number of edge pixels.
AWSM_DI uint32_t detect_edges(const FrameDev& f, const ShadeBlock& b, uint8_t* eslot, uint8_t* tslot, uint32_t lane) {
    const ulonglong2* masks = reinterpret_cast<const ulonglong2*>(f.msaa_edge_bits) + (size_t)b.blk * 4u;
    const unsigned long long below = (1ull << lane) - 1ull;
    uint32_t n = 0u, nt = 0u;
#pragma unroll
    for (uint32_t sidx = 0; sidx < 4u; sidx++) {
        const ulonglong2 m = masks[sidx];          // wave-uniform
        if ((m.x >> lane) & 1ull) eslot[n + (uint32_t)__popcll(m.x & below)] = (uint8_t)(sidx * 64u + lane);
        if ((m.y >> lane) & 1ull) tslot[nt + (uint32_t)__popcll(m.y & below)] = (uint8_t)(sidx * 64u + lane);
        n += (uint32_t)__popcll(m.x); nt += (uint32_t)__popcll(m.y);
    }
    if (nt == 0u) return n;                        // wave-uniform
    __syncthreads();                               // (one wavefront: orders the LDS writes before the reads)
    const m4 inv_proj = cload_m4(f.camera, 256u);
    const bool fastp = depth_only_projection(inv_proj);
    for (uint32_t i0 = 0; i0 < nt; i0 += 64u) {
        const uint32_t i = i0 + lane;
        const uint32_t slot = i < nt ? tslot[i] : 0u;
        const bool e = i < nt && detect_pixel(f, b, slot, inv_proj, fastp);
        const unsigned long long m = __builtin_amdgcn_ballot_w64(e);
        if (e) eslot[n + (uint32_t)__popcll(m & below)] = (uint8_t)slot;
        n += (uint32_t)__popcll(m);
    }
    return n;
}

// k_shade_msaa: the general route's MSAA kernel — sample 0 of every pixel by the general code and the whole edge decision in one kernel (normals of the
// block + a halo ring in LDS).
template <int GRAD>
__global__ __launch_bounds__(256) void k_shade_msaa(const DevScene* __restrict__ sc, FrameDev f) {
    __shared__ NeighbourCell cells[18 * 18];
    __shared__ uint32_t n_edges;
    ShadeBlock b;
    if (frame_poisoned(f) || !shade_block(f, b)) return;                  // workgroup-uniform
    const uint32_t tid = threadIdx.x;
    const int lx = (int)(tid & 15u), ly = (int)(tid >> 4);
    const int cx = b.x0 + lx, cy = b.y0 + ly;
    const bool inside = cx < (int)f.width && cy < (int)f.sy1;            // rows below sy0 never occur (blocks start at sy0)
    if (tid == 0) n_edges = 0u;

    // ---- phase 1: G-buffer texel of sample 0 for every pixel of the block + the halo ring, normals into LDS ----
    unsigned long long k4[4] = {~0ull, ~0ull, ~0ull, ~0ull};
    GBufferTexel g0;
    g0.packed_nt = {0.0f, 0.0f, 0.0f, 0.0f}; g0.bx = 0.0f; g0.by = 0.0f; g0.bary_derivs = {0.0f, 0.0f, 0.0f, 0.0f};
    {
        NeighbourCell c = {0.0f, 0.0f, 0.0f, 0u, 0u};
        if (cx < (int)f.width && cy < (int)f.y1) {    // also the halo row below the shard when it falls inside this block
            const ulonglong2* kp = reinterpret_cast<const ulonglong2*>(f.vis + ((size_t)cy * f.width + (size_t)cx) * 4);
            const ulonglong2 ka = kp[0], kb = kp[1];
            k4[0] = ka.x; k4[1] = ka.y; k4[2] = kb.x; k4[3] = kb.y;
            c.state = 1u;
            if (k4[0] != ~0ull) {
                g0 = reconstruct_gbuffer<(GRAD != 0)>(f, key_rank(k4[0]), cx, cy);   // STRICT
                const f3 n = decode_octahedral(mk2(g0.packed_nt.x, g0.packed_nt.y));
                c.nx = n.x; c.ny = n.y; c.nz = n.z; c.depth_bits = (uint32_t)(k4[0] >> 32); c.state = 2u;
            }
        }
        cells[(ly + 1) * 18 + (lx + 1)] = c;
    }
    if (tid < 68u) {   // halo ring: top row (18), bottom row (18), left column (16), right column (16)
        int hx, hy;
        if (tid < 18u) { hx = (int)tid - 1; hy = -1; }
        else if (tid < 36u) { hx = (int)tid - 19; hy = 16; }
        else if (tid < 52u) { hx = -1; hy = (int)tid - 36; }
        else { hx = 16; hy = (int)tid - 52; }
        publish_cell(cells, hx, hy, f, b.x0 + hx, b.y0 + hy);
    }
    __syncthreads();

    // ---- phase 2: compute.wgsl main ----
    const size_t p = (size_t)cy * f.width + (size_t)cx;
    const size_t po = f.out_compact ? (size_t)(((b.brow >> 1) << kTileShift) + (uint32_t)(cy & (kTile - 1))) * f.width + (size_t)cx : p;   // output pixel (compact band layout)
    uint8_t* edge_rec = reinterpret_cast<uint8_t*>(f.msaa_edges) + (size_t)b.blk * kEdgeRecBytes;
    bool is_edge = false;
    if (inside) {
        const f4 sky = skybox_color(sc, f, cx, cy);
        const bool any_hit = (k4[0] & k4[1] & k4[2] & k4[3]) != ~0ull;
        if (!f.has_opaque || !any_hit) {
            store_pixel(f, po, sky);                                        // compute.wgsl:121-143
        } else if (k4[0] == ~0ull) {
            f.msaa_color0[p] = make_float4(sky.x, sky.y, sky.z, sky.w);   // compute.wgsl:155-170: sample 0 is background, others are not
            is_edge = true;
        } else {
            const SurfaceOut o = shade_surface<GRAD>(sc, f, key_rank(k4[0]), cx, cy, key_depth(k4[0]), g0, true);
            if (o.kind == 2u) store_pixel(f, po, f4{0.0f, 0.0f, 0.0f, 0.0f});   // hud
            else if (o.kind == 1u) store_pixel(f, po, o.color);                 // debug view: written before the edge test
            else {
                // compute.wgsl:303-318 + msaa.wgsl:201-237 (STRICT)
                const m4 inv_proj = load_m4(reinterpret_cast<const float*>(f.camera + 256));
                const float W = (float)f.width, H = (float)f.height, pcx = (float)cx + 0.5f, pcy = (float)cy + 0.5f;
                is_edge = edge_mask_depth_msaa(inv_proj, k4, pcx, pcy, W, H);
                if (!is_edge) {
                    const NeighbourCell me = cells[(ly + 1) * 18 + (lx + 1)];
                    is_edge = edge_by_neighbours(inv_proj, mk3(me.nx, me.ny, me.nz), key_depth(k4[0]), pcx, pcy, W, H,
                                                 [&](int ox, int oy) { return cells[(ly + 1 + oy) * 18 + (lx + 1 + ox)]; });
                }
                if (is_edge) f.msaa_color0[p] = make_float4(o.color.x, o.color.y, o.color.z, o.color.w);
                else store_pixel(f, po, o.color);
            }
        }
    }
    if (is_edge) edge_rec[4u + atomicAdd(&n_edges, 1u)] = (uint8_t)tid;
    __syncthreads();
    if (tid == 0) *reinterpret_cast<uint32_t*>(edge_rec) = n_edges;
}

// The edge pixels' samples 1..3 (material_shading.wgsl:170-210) and the average of the four.  Item-parallel: the block's edge pixels (k_shade_msaa's /
// k_msaa_detect's list) are first expanded into work items — (pixel, sample) pairs whose triangle no earlier sample of the pixel shows: all samples a
// triangle covers in a pixel carry the same G-buffer texel and the same standard coordinates (sample 0's depth), hence the same colour — then every
// thread shades ONE item (the shading call sits in the code once, and a block with 30 edge pixels keeps 40 lanes busy for one shading instead of 30
// lanes for three in a row), then the threads of the edge pixels gather their four colours.  Up to 768 items per block, in LDS.
// The lean MSAA route's edge detector: one wavefront per block writes the block's edge list (count + one-byte pixel slots) for k_shade_msaa_resolve.
// (Not the first phase of that kernel: with it inside, the register allocation of the shading code behind it came out at 155 instead of 127.)
#ifndef AWSM_DETECT_WAVES
#define AWSM_DETECT_WAVES 8
#endif
// Round 5: 72 registers instead of a 64 cap with one register in scratch — beside four wavefronts of the gradient lean kernel (now 96 each) a SIMD has 128 left.
#ifndef AWSM_DETECT_VGPRS
#define AWSM_DETECT_VGPRS 72
#endif
#if AWSM_DETECT_VGPRS
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(AWSM_DETECT_VGPRS))) void k_msaa_detect(FrameDev f) {
#else
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(AWSM_DETECT_WAVES))) void k_msaa_detect(FrameDev f) {      // 64 registers: fits beside four wavefronts of the gradient lean kernel (448 of a SIMD's 512)
#endif
    __shared__ __attribute__((aligned(16))) uint8_t eslot[256];
    __shared__ uint8_t tslot[256];
    ShadeBlock b;
    if (frame_poisoned(f) || !shade_block(f, b)) return;
    const uint32_t lane = threadIdx.x;
    const uint32_t n = detect_edges(f, b, eslot, tslot, lane);
    __syncthreads();
    uint32_t* edge_rec = f.msaa_edges + (size_t)b.blk * (kEdgeRecBytes / 4u);
    if (lane == 0u) edge_rec[0] = n;
    for (uint32_t w = lane; w * 4u < n; w += 64u) edge_rec[1u + w] = reinterpret_cast<const uint32_t*>(eslot)[w];      // (slots beyond n in the last word: never read)
}

template <int GRAD>
__global__ __launch_bounds__(64) void k_shade_msaa_resolve(const DevScene* __restrict__ sc, FrameDev f) {
    // One WAVEFRONT per block: the shading is a long dependent chain for a few dozen lanes, and what bounds the kernel is how many blocks are in flight —
    // a 256-thread workgroup parked three idle wavefronts' registers behind every busy one (177 us at 4K; 18 M VALU instructions: 83 % of the wave-cycles waiting).
    // The block's edge pixels are taken kRound at a time (most blocks have fewer): the item arrays then cost 8 KB instead of 16 and the kernel's occupancy is
    // what its registers allow (four wavefronts per SIMD) instead of what the LDS allowed (ten per CU).
    constexpr uint32_t kRound = 128u;
    __shared__ uint32_t items[3 * kRound];   // pixel slot | sample << 8
    __shared__ float4 icolor[3 * kRound];
    __shared__ uint16_t first_of[kRound];    // first item of the round's e-th edge pixel
    __shared__ uint32_t n_items;
    __shared__ LeanStage stage;              // the lean route's staged triangle records (one wavefront per workgroup here)
    ShadeBlock b;
    if (frame_poisoned(f) || !shade_block(f, b)) return;
    const uint8_t* edge_rec = reinterpret_cast<const uint8_t*>(f.msaa_edges) + (size_t)b.blk * kEdgeRecBytes;
    const uint32_t n = *reinterpret_cast<const uint32_t*>(edge_rec);
    const uint32_t lane = threadIdx.x;
    if (n == 0u) return;                                                   // wave-uniform
#pragma unroll 1
    for (uint32_t e0 = 0; e0 < n; e0 += kRound) {
    const uint32_t e1 = min(n, e0 + kRound);
    __syncthreads();
    if (lane == 0u) n_items = 0u;
    __syncthreads();
    // ---- phase 1: every edge pixel -> its items ----
    for (uint32_t e = e0 + lane; e < e1; e += 64u) {
        const uint32_t slot = edge_rec[4u + e];
        const int cx = b.x0 + (int)(slot & 15u), cy = b.y0 + (int)(slot >> 4);
        const size_t p = (size_t)cy * f.width + (size_t)cx;
        uint32_t cnt = 0u, which = 0u;
        if (__float_as_uint(f.msaa_color0[p].w) != 0xFFFFFFFFu) {          // (marker: a hud mesh or a debug view of k_shade_todo<.., MSAA> — written before the edge test, never resolved)
            const ulonglong2* kp = reinterpret_cast<const ulonglong2*>(f.vis + p * 4);
            const ulonglong2 ka = kp[0], kb = kp[1];
            const unsigned long long k4[4] = {ka.x, ka.y, kb.x, kb.y};
#pragma unroll
            for (int sidx = 1; sidx < 4; sidx++) {
                if (k4[sidx] == ~0ull) continue;
                const uint32_t r = key_rank(k4[sidx]);
                bool same = false;
#pragma unroll
                for (int t = 0; t < sidx; t++) same = same || (k4[t] != ~0ull && key_rank(k4[t]) == r);
                if (!same) { which |= (uint32_t)sidx << (2u * cnt); cnt++; }
            }
        }
        const uint32_t first = atomicAdd(&n_items, cnt);
        first_of[e - e0] = (uint16_t)first;
        for (uint32_t j = 0; j < cnt; j++) items[first + j] = slot | (((which >> (2u * j)) & 3u) << 8);
    }
    __syncthreads();
    // ---- phase 2: one item per lane and turn.  The lean route's body shades the turn when every item of it belongs to a lean draw (the record staging, the
    // scalar draw record and the per-texture probes of k_shade_lean, on lanes that are items instead of the pixels of a strip); a turn with any other item
    // takes the general code, as a strip of k_shade_lean goes to k_shade_todo. ----
    const uint32_t ni = n_items;
    const bool lean_route = f.draw_lean != nullptr && f.tri_shade != nullptr;      // (the frame's opaque pass took the lean route)
    for (uint32_t i0 = 0; i0 < ni; i0 += 64u) {
        const uint32_t i = i0 + lane;
        const bool live = i < ni;
        const uint32_t it = live ? items[i] : 0u, isl = it & 255u, sidx = it >> 8;
        const int ix = b.x0 + (int)(isl & 15u), iy = b.y0 + (int)(isl >> 4);
        const unsigned long long* kq = f.vis + ((size_t)iy * f.width + (size_t)ix) * 4;
        const unsigned long long ks = live ? kq[sidx] : ~0ull, k0 = live ? kq[0] : ~0ull;
        const uint32_t r = key_rank(ks);
        LeanItem item = {ix, iy, r, key_depth(k0), live, {0.0f, 0.0f, 0.0f}};
        bool done = false;
        if (lean_route) done = lean_core<GRAD, false, true>(sc, f, b, 0u, lane, &stage, &item);      // (wave-uniform result)
        if (done) { if (live) icolor[i] = make_float4(item.color.x, item.color.y, item.color.z, 1.0f); }
        else if (live) {
            const f4 c = shade_surface<GRAD>(sc, f, r, ix, iy, key_depth(k0), reconstruct_gbuffer<(GRAD != 0)>(f, r, ix, iy), false).color;
            icolor[i] = make_float4(c.x, c.y, c.z, c.w);
        }
    }
    __syncthreads();
    // ---- phase 3: the four colours of every edge pixel, averaged ----
    for (uint32_t e = e0 + lane; e < e1; e += 64u) {
        const uint32_t slot = edge_rec[4u + e];
        const int cx = b.x0 + (int)(slot & 15u), cy = b.y0 + (int)(slot >> 4);
        const size_t p = (size_t)cy * f.width + (size_t)cx;
        const float4 c0v = f.msaa_color0[p];
        if (__float_as_uint(c0v.w) == 0xFFFFFFFFu) continue;
        const size_t po = f.out_compact ? (size_t)(((b.brow >> 1) << kTileShift) + (uint32_t)(cy & (kTile - 1))) * f.width + (size_t)cx : p;
        const ulonglong2* kp = reinterpret_cast<const ulonglong2*>(f.vis + p * 4);
        const ulonglong2 ka = kp[0], kb = kp[1];
        const unsigned long long k4[4] = {ka.x, ka.y, kb.x, kb.y};
        const f4 sky = skybox_color(sc, f, cx, cy);
        // (no array of colours here: picking "the colour of the earlier sample with the same triangle" out of one by a run-time index sent the
        // array — and the kernel — to scratch memory, 80-96 bytes per lane; three named values and selects stay in registers)
        const f4 col0 = {c0v.x, c0v.y, c0v.z, c0v.w};
        uint32_t j = first_of[e - e0];
        auto sample_colour = [&](int sidx, const f4& a1, const f4& a2) -> f4 {
            if (k4[sidx] == ~0ull) return sky;
            const uint32_t r = key_rank(k4[sidx]);
            const bool s0 = k4[0] != ~0ull && key_rank(k4[0]) == r;
            const bool s1 = sidx > 1 && k4[1] != ~0ull && key_rank(k4[1]) == r;
            const bool s2 = sidx > 2 && k4[2] != ~0ull && key_rank(k4[2]) == r;
            if (!(s0 || s1 || s2)) { const float4 q = icolor[j]; j++; return {q.x, q.y, q.z, q.w}; }
            f4 c = col0;                     // the LAST earlier sample with this triangle (they all hold the same colour: one shading per distinct triangle)
            if (s1) c = a1;
            if (s2) c = a2;
            return c;
        };
        const f4 col1 = sample_colour(1, col0, col0);
        const f4 col2 = sample_colour(2, col1, col0);
        const f4 col3 = sample_colour(3, col1, col2);
        const f4 sum = {((col0.x + col1.x) + col2.x) + col3.x, ((col0.y + col1.y) + col2.y) + col3.y,
                        ((col0.z + col1.z) + col2.z) + col3.z, ((col0.w + col1.w) + col2.w) + col3.w};
        store_pixel(f, po, {sum.x * 0.25f, sum.y * 0.25f, sum.z * 0.25f, sum.w * 0.25f});
    }
    }
}
#pragma clang fp contract(off)

// ------------------------------------------------------------------------------------------------
// BRDF LUT (crates/renderer-core/src/brdf_lut/shader.wgsl:1-78): one thread per texel, 1024 samples.
// ------------------------------------------------------------------------------------------------
AWSM_DI float radical_inverse_vdc(uint32_t bits) { return (float)__brev(bits) * 2.3283064365386963e-10f; }
AWSM_DI float lut_g1(float ndot_v, float alpha) {
    const float a = fmaxf(alpha, 0.001f);
    const float k = ((a + 1.0f) * (a + 1.0f)) * 0.125f;
    return ndot_v / (ndot_v * (1.0f - k) + k);
}
__global__ __launch_bounds__(256) void k_brdf_lut(uint32_t* __restrict__ out_rg16f, uint32_t width, uint32_t height) {
    const uint32_t i = blockIdx.x * 16u + (threadIdx.x & 15u), j = blockIdx.y * 16u + (threadIdx.x >> 4);
    if (i >= width || j >= height) return;
    // fragment (i+0.5, j+0.5) of the full-screen triangle: uv.y is 1 at the TOP row (shader.wgsl:4-12)
    const float uvx = ((float)i + 0.5f) / (float)width, uvy = 1.0f - ((float)j + 0.5f) / (float)height;
    const float no_v = clampf(uvx, 1e-3f, 1.0f - 1e-3f);
    const float roughness = clampf(uvy, 1e-3f, 1.0f - 1e-3f);
    const f3 v = {sqrtf(fmaxf(0.0f, 1.0f - no_v * no_v)), 0.0f, no_v};
    const float alpha = roughness * roughness;
    float a = 0.0f, bsum = 0.0f;
    for (uint32_t s = 0; s < 1024u; s++) {
        const float xi_x = (float)s / 1024.0f, xi_y = radical_inverse_vdc(s);
        const float a2 = alpha * alpha;
        const float phi = 6.28318530718f * xi_x;
        const float cos_theta = sqrtf((1.0f - xi_y) / (1.0f + (a2 - 1.0f) * xi_y));
        const float sin_theta = sqrtf(fmaxf(0.0f, 1.0f - cos_theta * cos_theta));
        const f3 h = {cosf(phi) * sin_theta, sinf(phi) * sin_theta, cos_theta};
        const float vdh = dot(v, h);
        const f3 l = normalize(h * (2.0f * vdh) - v);
        const float no_l = fmaxf(l.z, 0.0f), no_h = fmaxf(h.z, 0.0f), vo_h = fmaxf(vdh, 0.0f), no_v_ = fmaxf(v.z, 0.0f);
        if (no_l > 0.0f) {
            const float g = lut_g1(no_v_, alpha) * lut_g1(no_l, alpha);
            const float g_vis = (g * vo_h) / fmaxf(no_h * no_v_, 1e-4f);
            const float fc = powf(1.0f - vo_h, 5.0f);
            a = a + (1.0f - fc) * g_vis;
            bsum = bsum + fc * g_vis;
        }
    }
    a = a / 1024.0f; bsum = bsum / 1024.0f;
    out_rg16f[(size_t)j * width + i] = (uint32_t)f16_bits(a) | ((uint32_t)f16_bits(bsum) << 16);
}

// MSAA + band sharding: the sample-0 keys of the first and the last row of each band this shard owns, [band][2][width]; bands beyond
// the shard's own count (ranks own unequal numbers) are filled with "no hit".  What the ranks all-gather between the two passes.
__global__ __launch_bounds__(256) void k_msaa_halo_export(FrameDev f, unsigned long long* __restrict__ dst, uint32_t bands_out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= bands_out * 2u * f.width) return;
    const uint32_t x = i % f.width, which = (i / f.width) & 1u, l = i / (2u * f.width);
    unsigned long long k = ~0ull;
    if (l < f.tiles_y) {
        const uint32_t ty = f.tile_row0 + l * f.band_n;
        const uint32_t y = which ? min((ty << kTileShift) + (uint32_t)(kTile - 1), f.height - 1u) : (ty << kTileShift);
        k = f.vis[((size_t)y * f.width + x) * 4];
    }
    dst[i] = k;
}

// covered-pixel count for AwsmFrameStats: runs only when the caller asks for stats (frame_end), never in the frame itself.
// (A per-wave atomicAdd on one counter inside k_shade serialised ~130k same-address atomics per 4K frame.)
__global__ __launch_bounds__(256) void k_count_covered(const unsigned long long* __restrict__ vis, uint32_t width, uint32_t y0, uint32_t y1,
                                                       uint32_t band_n, uint32_t band_r, uint32_t msaa, uint32_t* counter) {
    const size_t n = (size_t)width * (y1 - y0), base = (size_t)width * y0;
    uint32_t local = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint32_t y = y0 + (uint32_t)(i / width);
        if (band_n > 1u && ((y >> kTileShift) % band_n) != band_r) continue;      // rows of other shards hold stale keys
        if (msaa == 4u) { const unsigned long long* k = vis + (base + i) * 4; local += (k[0] & k[1] & k[2] & k[3]) != ~0ull ? 1u : 0u; }   // any sample hit
        else local += vis[base + i] != ~0ull ? 1u : 0u;
    }
    for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(counter, local);
}

// Test aid (awsm_hip_read_gbuffer): the STRICT G-buffer texel of every single-sampled pixel, as the opaque pass reconstructs it.
__global__ __launch_bounds__(256) void k_gbuffer_dump(FrameDev f, float* __restrict__ out) {
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= f.width * f.height) return;
    float* o = out + (size_t)p * 6u;
    const unsigned long long key = f.vis[p];
    if (key == ~0ull) { for (int i = 0; i < 6; i++) o[i] = 0.0f; return; }
    const GBufferTexel g = reconstruct_gbuffer<false>(f, key_rank(key), (int)(p % f.width), (int)(p / f.width));
    o[0] = g.packed_nt.x; o[1] = g.packed_nt.y; o[2] = g.packed_nt.z; o[3] = g.packed_nt.w; o[4] = g.bx; o[5] = g.by;
}
// Position-dependent 128-bit digest of the visibility keys (tests: a re-rendered frame must reproduce every key; comparing two
// digests on the device costs 16 bytes of read-back instead of 66 MB per 4K frame).  out[0] += key * (2 i + 1), out[1] ^= rotl(key, i).
__global__ __launch_bounds__(256) void k_vis_digest(const unsigned long long* __restrict__ vis, size_t n, unsigned long long* __restrict__ out) {
    unsigned long long s = 0ull, x = 0ull;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const unsigned long long k = vis[i];
        const uint32_t r = (uint32_t)i & 63u;
        s += k * (2ull * i + 1ull);
        x ^= (k << r) | (r ? k >> (64u - r) : 0ull);
    }
    for (int off = 32; off > 0; off >>= 1) { s += __shfl_down(s, off); x ^= __shfl_down(x, off); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(out, s); atomicXor(out + 1, x); }
}

// picker_wgsl/compute.wgsl: one thread; out = {valid, mesh_key_high, mesh_key_low, triangle_index}
__global__ void k_pick(const DevScene* __restrict__ sc, FrameDev f, int x, int y, uint32_t* __restrict__ out) {
    out[0] = 0u; out[1] = 0u; out[2] = 0u; out[3] = 0xFFFFFFFFu;
    if (x < 0 || y < 0 || x >= (int)f.width || y < (int)f.sy0 || y >= (int)f.sy1) return;
    if (f.band_n > 1u && (((uint32_t)y >> kTileShift) % f.band_n) != f.band_r) return;
    unsigned long long key = f.vis[((size_t)y * f.width + (size_t)x) * (f.msaa == 4u ? 4u : 1u)];   // MSAA: sample 0, as the picker's textureLoad(.., 0)
    const DrawDev* draws = f.draws; const uint32_t* tri_info = f.tri_info;
    if (f.hud_vis) {      // the HUD geometry pass drew over the visibility target (LoadOp::Load): its triangle is what the picker reads
        const unsigned long long hk = f.hud_vis[((size_t)y * f.width + (size_t)x) * (f.msaa == 4u ? 4u : 1u)];
        if (hk != ~0ull) { key = hk; draws = f.hud_draws; tri_info = f.hud_tri_info; }
    }
    if (key == ~0ull) return;
    const uint32_t rank = 0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull);
    const DrawDev dr = draws[tri_info[rank] & 0x00FFFFFFu];
    const uint32_t material_meta_offset = *reinterpret_cast<const uint32_t*>(sc->buf[AWSM_BUF_GEOM_META] + dr.geom_meta_off + 36);
    const uint32_t* mm = reinterpret_cast<const uint32_t*>(sc->buf[AWSM_BUF_MATERIAL_META] + (size_t)(material_meta_offset / 256u) * 256u);
    out[0] = 1u; out[1] = mm[0]; out[2] = mm[1]; out[3] = rank - dr.first_tri;
}

// helper kernels for readback / upload conversions
__global__ void k_rgba16f_to_rg16f(const uint16_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t n) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) out[i] = (uint32_t)in[(size_t)i * 4] | ((uint32_t)in[(size_t)i * 4 + 1] << 16);
}

}  // namespace awsm

// per-draw records (DrawShadeDev, DrawMatDev, TexSlotDev) + per-light constants, before the kernels that shade
extern "C" void awsm_launch_resolve_draws(const awsm::DevScene* sc, const awsm::FrameDev* f, hipStream_t s) {
    if (f->n_draws) hipLaunchKernelGGL(awsm::k_resolve_draws, dim3((8u * f->n_draws + 255u) / 256u), dim3(256), 0, s, sc, *f);
}
// A kernel's instantiation by mip mode: 0 MipmapMode::None, 1 MipmapMode::Gradient, 2 Gradient on a context that honours max_anisotropy
// (AWSM_CFG_ANISOTROPIC; separate instantiations: the probes cost the isotropic sampler registers it would pay for on every frame)
#define AWSM_LAUNCH_G(g, K, grid, block, ...) do { if ((g) == 2) hipLaunchKernelGGL((K<2>), grid, block, 0, s, __VA_ARGS__); else if ((g) == 1) hipLaunchKernelGGL((K<1>), grid, block, 0, s, __VA_ARGS__); \
                                                   else hipLaunchKernelGGL((K<0>), grid, block, 0, s, __VA_ARGS__); } while (0)
#define AWSM_LAUNCH_G2(g, K, B, grid, block, ...) do { if ((g) == 2) hipLaunchKernelGGL((K<2, B>), grid, block, 0, s, __VA_ARGS__); else if ((g) == 1) hipLaunchKernelGGL((K<1, B>), grid, block, 0, s, __VA_ARGS__); \
                                                       else hipLaunchKernelGGL((K<0, B>), grid, block, 0, s, __VA_ARGS__); } while (0)
#define AWSM_LAUNCH_SG(g, K, S, grid, block, ...) do { if ((g) == 2) hipLaunchKernelGGL((K<S, 2>), grid, block, 0, s, __VA_ARGS__); else if ((g) == 1) hipLaunchKernelGGL((K<S, 1>), grid, block, 0, s, __VA_ARGS__); \
                                                       else hipLaunchKernelGGL((K<S, 0>), grid, block, 0, s, __VA_ARGS__); } while (0)
static inline int mip_mode(const awsm::FrameDev* f) { return f->mipmap ? (f->aniso ? 2 : 1) : 0; }
extern "C" int awsm_shade_is_lean(const awsm::FrameDev* f);
extern "C" void awsm_launch_shade(const awsm::DevScene* sc, const awsm::FrameDev* f, hipStream_t s) {
    const uint32_t bx_n = (f->width + 15u) >> 4, by_n = f->band_n > 1u ? 2u * f->tiles_y : ((f->sy1 - f->sy0) + 15u) >> 4;
    const uint32_t nb = 8u * ((by_n + 7u) / 8u) * bx_n;   // every XCD gets ceil(by_n / 8) rows of ids; surplus ids exit
    if (!nb) return;
    const int g = mip_mode(f);      // MipmapMode::None / Gradient / Gradient + anisotropy: separate instantiations, as the reference keeps separate pipelines
    const bool msaa = f->msaa == 4u;
    if (awsm_shade_is_lean(f)) {
        // the lean kernel over the screen (a wavefront per strip id on a power-of-two block pitch: the padding exits), then the general code for the
        // wavefronts it declined (awsm_launch_shade_todo; k_deform_transform / k_resolve_draws reset the list)
        uint32_t pitch = 1; while (pitch < bx_n) pitch <<= 1;
        const uint32_t nb_ids = 8u * ((by_n + 7u) / 8u) * pitch;
        if (msaa) {
            if (g == 2) hipLaunchKernelGGL((awsm::k_shade_lean<false, 2, true>), dim3(nb_ids), dim3(256), 0, s, sc, *f);
            else if (g == 1) hipLaunchKernelGGL((awsm::k_shade_lean<false, 1, true>), dim3(nb_ids), dim3(256), 0, s, sc, *f);
            else hipLaunchKernelGGL((awsm::k_shade_lean<false, 0, true>), dim3(nb_ids), dim3(256), 0, s, sc, *f);
        } else if (g == 2) hipLaunchKernelGGL((awsm::k_shade_lean<false, 2, false>), dim3(nb_ids), dim3(256), 0, s, sc, *f);
        else if (g == 1) hipLaunchKernelGGL((awsm::k_shade_lean<false, 1, false>), dim3(nb_ids), dim3(256), 0, s, sc, *f);
        else if (f->lean_grid && f->lean_next) hipLaunchKernelGGL((awsm::k_shade_lean<true, 0, false>), dim3(min(f->lean_grid, nb_ids)), dim3(256), 0, s, sc, *f);
        else hipLaunchKernelGGL((awsm::k_shade_lean<false, 0, false>), dim3(nb_ids), dim3(256), 0, s, sc, *f);
        return;
    }
    if (msaa) {
        AWSM_LAUNCH_G(g, awsm::k_shade_msaa, dim3(nb), dim3(256), sc, *f);
        AWSM_LAUNCH_G(g, awsm::k_shade_msaa_resolve, dim3(nb), dim3(64), sc, *f);
    } else AWSM_LAUNCH_G(g, awsm::k_shade, dim3(nb), dim3(256), sc, *f);
}
// second half of the lean route; returns 0 when the frame did not take it
extern "C" int awsm_shade_is_lean(const awsm::FrameDev* f) {
    const uint32_t bx_n = (f->width + 15u) >> 4, by_n = f->band_n > 1u ? 2u * f->tiles_y : ((f->sy1 - f->sy0) + 15u) >> 4;
    return (bx_n * by_n) && (f->msaa != 4u || (f->msaa_edge_bits && f->msaa_cells)) && f->draw_lean && f->tri_shade && f->shade_todo && f->has_opaque && f->n_draws;
}
extern "C" int awsm_launch_shade_todo(const awsm::DevScene* sc, const awsm::FrameDev* f, hipStream_t s) {
    if (!awsm_shade_is_lean(f)) return 0;
    const int g = mip_mode(f);
    if (f->msaa == 4u) {      // ... and the edge pixels' remaining samples, once every sample-0 colour is in place
        const uint32_t bx_n = (f->width + 15u) >> 4, by_n = f->band_n > 1u ? 2u * f->tiles_y : ((f->sy1 - f->sy0) + 15u) >> 4;
        const uint32_t nb = 8u * ((by_n + 7u) / 8u) * bx_n;
        AWSM_LAUNCH_G2(g, awsm::k_shade_todo, true, dim3(awsm::kTodoBlocks), dim3(256), sc, *f);
        hipLaunchKernelGGL(awsm::k_msaa_detect, dim3(nb), dim3(64), 0, s, *f);
        AWSM_LAUNCH_G(g, awsm::k_shade_msaa_resolve, dim3(nb), dim3(64), sc, *f);
        return 1;
    }
    AWSM_LAUNCH_G2(g, awsm::k_shade_todo, false, dim3(awsm::kTodoBlocks), dim3(256), sc, *f);
    return 1;
}
// f: the transparent pass's frame (its own draws / vertices / bins; vis = the geometry pass's keys; opaque_rgba16f = the opaque image;
// out_rgba16f / out_rgba32f = the composite image).  Every tile of the frame is launched: a tile without transparent triangles is a copy.
extern "C" void awsm_launch_forward(const awsm::DevScene* sc, const awsm::FrameDev* f, hipStream_t s) {
    const uint32_t n_tiles = f->tiles_x * f->tiles_y;
    if (!n_tiles) return;
    const bool ms = f->msaa == 4u;
    const int g = mip_mode(f);
    const uint32_t nb_shade = (f->frag_cap + 255u) / 256u, nb_blend = (f->width * f->height + 255u) / 256u;
    if (ms) AWSM_LAUNCH_SG(g, awsm::k_forward_cover, 4, dim3(n_tiles * awsm::kFwdSubs), dim3(awsm::kFwdThreads), sc, *f);
    else AWSM_LAUNCH_SG(g, awsm::k_forward_cover, 1, dim3(n_tiles * awsm::kFwdSubs), dim3(awsm::kFwdThreads), sc, *f);
    if (nb_shade) AWSM_LAUNCH_G(g, awsm::k_forward_shade, dim3(nb_shade), dim3(256), sc, *f);
    if (ms) hipLaunchKernelGGL(awsm::k_forward_blend<4>, dim3(nb_blend), dim3(256), 0, s, *f); else hipLaunchKernelGGL(awsm::k_forward_blend<1>, dim3(nb_blend), dim3(256), 0, s, *f);
}
extern "C" void awsm_launch_msaa_halo_export(const awsm::FrameDev* f, unsigned long long* dst, uint32_t bands_out, hipStream_t s) {
    const uint32_t n = bands_out * 2u * f->width;
    if (n) hipLaunchKernelGGL(awsm::k_msaa_halo_export, dim3((n + 255u) / 256u), dim3(256), 0, s, *f, dst, bands_out);
}
extern "C" void awsm_launch_count_covered(const awsm::FrameDev* f, hipStream_t s) {
    if (f->sy1 > f->sy0) hipLaunchKernelGGL(awsm::k_count_covered, dim3(1024), dim3(256), 0, s, f->vis, f->width, f->sy0, f->sy1, f->band_n, f->band_r, f->msaa, f->counters + 3);
}
// Test aid: the STRICT G-buffer texel of every single-sampled pixel, 6 floats {packed_nt.xyzw, bx, by} (zeros = no hit)
extern "C" void awsm_launch_gbuffer_dump(const awsm::FrameDev* f, float* out, hipStream_t s) {
    hipLaunchKernelGGL(awsm::k_gbuffer_dump, dim3((f->width * f->height + 255u) / 256u), dim3(256), 0, s, *f, out);
}
extern "C" void awsm_launch_vis_digest(const unsigned long long* vis, size_t n, unsigned long long* out, hipStream_t s) {
    hipLaunchKernelGGL(awsm::k_vis_digest, dim3(1024), dim3(256), 0, s, vis, n, out);
}
extern "C" void awsm_launch_pick(const awsm::DevScene* sc, const awsm::FrameDev* f, int x, int y, uint32_t* out, hipStream_t s) {
    hipLaunchKernelGGL(awsm::k_pick, dim3(1), dim3(1), 0, s, sc, *f, x, y, out);
}
// cd: the cube with b_level_off filled and `texels` uploaded; out: room for `total` texels (the aproned chain)
extern "C" void awsm_launch_cube_border(const awsm::CubeDev* cd, uint2* out, uint32_t total, hipStream_t s) {
    if (total) hipLaunchKernelGGL(awsm::k_cube_border, dim3((total + 255u) / 256u), dim3(256), 0, s, *cd, out, total);
}
extern "C" void awsm_launch_brdf_lut(uint32_t* out_rg16f, uint32_t w, uint32_t h, hipStream_t s) {
    hipLaunchKernelGGL(awsm::k_brdf_lut, dim3((w + 15u) / 16u, (h + 15u) / 16u), dim3(256), 0, s, out_rg16f, w, h);
}
extern "C" void awsm_launch_rgba16f_to_rg16f(const uint16_t* in, uint32_t* out, uint32_t n, hipStream_t s) {
    hipLaunchKernelGGL(awsm::k_rgba16f_to_rg16f, dim3((n + 255u) / 256u), dim3(256), 0, s, in, out, n);
}
