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
// Pure gather + ALU: no MFMA.  Compulsory HBM traffic per pixel is the 8-byte key read and the 8-byte
// RGBA16F write; everything else (vertices, metas, materials, texels) is reused across neighbouring pixels
// and served by L1 / the XCD's L2 — workgroups are dealt to XCDs in contiguous screen runs for that.
#include "frame_params.hpp"
#include "raster_setup.hpp"

namespace awsm {

AWSM_DI uint32_t xcd_remap_s(uint32_t b, uint32_t n) {
    uint32_t per = (n + 7u) >> 3;
    return (b & 7u) * per + (b >> 3);
}

// ---------------- textures.wgsl ----------------
struct TexInfo {
    bool exists;
    uint32_t array_index, layer_index, uv_set_index, sampler_index, uv_transform_index;
};
AWSM_DI TexInfo tex_none() { return {false, 0u, 0u, 0u, 0u, 0u}; }
AWSM_DI TexInfo tex_load(const uint32_t* __restrict__ m, uint32_t i) {      // textures.wgsl:75-114
    TexInfo t;
    const uint32_t array_and_layer = m[i + 1], uv_and_sampler = m[i + 2], extra = m[i + 3], transform_offset = m[i + 4];
    t.array_index = array_and_layer & 0xFFFu; t.layer_index = array_and_layer >> 12;
    t.uv_set_index = uv_and_sampler & 0xFFu; t.sampler_index = uv_and_sampler >> 8;
    t.exists = (extra & 1u) != 0u;
    t.uv_transform_index = transform_offset / 32u;
    return t;
}

AWSM_DI int wrap_index(int i, int n, uint32_t mode) {
    if (mode == 1u) { int m = i % n; return m < 0 ? m + n : m; }
    if (mode == 2u) { int p = 2 * n; int m = i % p; if (m < 0) m += p; return m < n ? m : p - 1 - m; }
    return i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
}
AWSM_DI f4 texel_rgba8(const uint8_t* __restrict__ p) {
    const uint32_t u = *reinterpret_cast<const uint32_t*>(p);
    return {(float)(u & 255u) / 255.0f, (float)((u >> 8) & 255u) / 255.0f, (float)((u >> 16) & 255u) / 255.0f, (float)(u >> 24) / 255.0f};
}
AWSM_DI float safe_floor(float x, float& frac) {
    float fl = floorf(x);
    if (!(fl >= -1073741824.0f && fl <= 1073741824.0f)) { frac = 0.0f; return 0.0f; }
    frac = x - fl;
    return fl;
}
// textureSampleLevel(tex, sampler, uv, layer, 0): DESIGN.md §"Texture sampling"
AWSM_DI f4 sample_array_level0(const TexArrayDev& arr, const AwsmSampler& smp, f2 uv, uint32_t layer) {
    const int W = (int)arr.width, H = (int)arr.height;
    if (layer >= arr.layers) layer = arr.layers - 1u;
    const uint8_t* base = arr.texels + (size_t)layer * (size_t)W * (size_t)H * 4u;
    float fx, fy;
    if (smp.mag_filter == 0u) {
        const int i = wrap_index((int)safe_floor(uv.x * (float)W, fx), W, smp.address_mode_u);
        const int j = wrap_index((int)safe_floor(uv.y * (float)H, fy), H, smp.address_mode_v);
        return texel_rgba8(base + ((size_t)j * W + i) * 4u);
    }
    const float x0f = safe_floor(uv.x * (float)W - 0.5f, fx);
    const float y0f = safe_floor(uv.y * (float)H - 0.5f, fy);
    const int i0 = wrap_index((int)x0f, W, smp.address_mode_u), i1 = wrap_index((int)x0f + 1, W, smp.address_mode_u);
    const int j0 = wrap_index((int)y0f, H, smp.address_mode_v), j1 = wrap_index((int)y0f + 1, H, smp.address_mode_v);
    const f4 c00 = texel_rgba8(base + ((size_t)j0 * W + i0) * 4u), c10 = texel_rgba8(base + ((size_t)j0 * W + i1) * 4u);
    const f4 c01 = texel_rgba8(base + ((size_t)j1 * W + i0) * 4u), c11 = texel_rgba8(base + ((size_t)j1 * W + i1) * 4u);
    const float gx = 1.0f - fx, gy = 1.0f - fy;
    const f4 top = {c00.x * gx + c10.x * fx, c00.y * gx + c10.y * fx, c00.z * gx + c10.z * fx, c00.w * gx + c10.w * fx};
    const f4 bot = {c01.x * gx + c11.x * fx, c01.y * gx + c11.y * fx, c01.z * gx + c11.z * fx, c01.w * gx + c11.w * fx};
    return {top.x * gy + bot.x * fy, top.y * gy + bot.y * fy, top.z * gy + bot.z * fy, top.w * gy + bot.w * fy};
}

// ---------------- per-pixel attribute context ----------------
struct Attr {
    const DevScene* sc;
    const float* ad;          // attribute_data (f32 view)
    uint32_t v0, v1, v2;      // vertex_start of the three corners (floats)
    uint32_t uv_sets_index;
    f3 bary;
};
AWSM_DI f2 texture_uv(const Attr& a, const TexInfo& t) {                    // texture_uvs.wgsl:64-84
    const uint32_t o = a.uv_sets_index + t.uv_set_index * 2u;
    const float2 u0 = *reinterpret_cast<const float2*>(a.ad + a.v0 + o);   // strides are multiples of 8 B in practice;
    const float2 u1 = *reinterpret_cast<const float2*>(a.ad + a.v1 + o);   // see aligned8 guard in the caller
    const float2 u2 = *reinterpret_cast<const float2*>(a.ad + a.v2 + o);
    return {(a.bary.x * u0.x + a.bary.y * u1.x) + a.bary.z * u2.x, (a.bary.x * u0.y + a.bary.y * u1.y) + a.bary.z * u2.y};
}
AWSM_DI f2 texture_uv_unaligned(const Attr& a, const TexInfo& t) {
    const uint32_t o = a.uv_sets_index + t.uv_set_index * 2u;
    const float x0 = a.ad[a.v0 + o], y0 = a.ad[a.v0 + o + 1], x1 = a.ad[a.v1 + o], y1 = a.ad[a.v1 + o + 1];
    const float x2 = a.ad[a.v2 + o], y2 = a.ad[a.v2 + o + 1];
    return {(a.bary.x * x0 + a.bary.y * x1) + a.bary.z * x2, (a.bary.x * y0 + a.bary.y * y1) + a.bary.z * y2};
}
AWSM_DI f4 sample_tex(const Attr& a, const TexInfo& t) {                    // texture_uvs.wgsl:144-187, textures.wgsl:131-150
    const f2 uv = texture_uv_unaligned(a, t);
    const float* tt = reinterpret_cast<const float*>(a.sc->buf[AWSM_BUF_TEXTURE_TRANSFORMS] + (size_t)t.uv_transform_index * 32u);
    const f2 uvt = {(tt[0] * uv.x + tt[1] * uv.y) + tt[4], (tt[2] * uv.x + tt[3] * uv.y) + tt[5]};
    if (t.array_index >= a.sc->n_tex || t.sampler_index >= a.sc->n_samplers) return {0.0f, 0.0f, 0.0f, 0.0f};
    return sample_array_level0(a.sc->tex[t.array_index], a.sc->samplers[t.sampler_index], uvt, t.layer_index);
}
AWSM_DI f4 vertex_color(const Attr& a, uint32_t set_index) {               // vertex_color_attrib.wgsl:1-21
    const uint32_t o = set_index * 4u;
    float r[4];
#pragma unroll
    for (int j = 0; j < 4; j++) r[j] = (a.bary.x * a.ad[a.v0 + o + j] + a.bary.y * a.ad[a.v1 + o + j]) + a.bary.z * a.ad[a.v2 + o + j];
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

AWSM_DI f3 normal_map(const Attr& a, const TexInfo& t, float scale, const TBN& tbn) {   // material_color_calc.wgsl:301-322
    if (!t.exists) return tbn.N;
    const f4 s = sample_tex(a, t);
    const f3 tn = {(s.x * 2.0f - 1.0f) * scale, (s.y * 2.0f - 1.0f) * scale, s.z * 2.0f - 1.0f};
    m3 m; m.c[0] = tbn.T; m.c[1] = tbn.B; m.c[2] = tbn.N;
    return normalize(mul(m, tn));
}

// ---------------- brdf.wgsl ----------------
AWSM_DI float effective_ior(float ior) { return ior < 1.0f ? 1.5f : ior; }
AWSM_DI float ior_to_f0(float ior) { float v = effective_ior(ior); float r = (v - 1.0f) / (v + 1.0f); return r * r; }
AWSM_DI f3 volume_attenuation(float distance, f3 color, float att_distance) {           // brdf.wgsl:55-74
    if (distance <= 0.0f) return splat3(1.0f);
    if (att_distance <= 0.0f || att_distance > 1e10f) return splat3(1.0f);
    if (color.x >= 0.999f && color.y >= 0.999f && color.z >= 0.999f) return splat3(1.0f);
    const float e = distance / att_distance;
    return {powf(color.x, e), powf(color.y, e), powf(color.z, e)};
}
AWSM_DI bool should_apply_volume_attenuation(float thickness, float att_distance, f3 c) {
    return thickness > 0.0f && att_distance < 1e10f && (c.x < 1.0f || c.y < 1.0f || c.z < 1.0f);
}
AWSM_DI f3 safe_half_vector(f3 v, f3 l) {                                                // brdf.wgsl:94-101
    const f3 sum = v + l;
    const float len_sq = dot(sum, sum);
    if (len_sq > 1e-8f) return sum * inverse_sqrt(len_sq);
    return {0.0f, 0.0f, 0.0f};
}
AWSM_DI float pow5(float x) { return powf(x, 5.0f); }
AWSM_DI f3 fresnel_schlick_f90(float cos_theta, f3 F0, float f90) {                      // brdf.wgsl:111-115
    const float p = pow5(1.0f - saturate(cos_theta));
    return {F0.x + (f90 - F0.x) * p, F0.y + (f90 - F0.y) * p, F0.z + (f90 - F0.z) * p};
}
AWSM_DI float fresnel_schlick_scalar(float cos_theta, float F0) {                        // brdf.wgsl:104-108 (.r of a splat)
    const float p = pow5(1.0f - saturate(cos_theta));
    return F0 + (1.0f - F0) * p;
}
AWSM_DI float distribution_ggx(float n_dot_h, float alpha) {                             // brdf.wgsl:118-124
    const float a = fmaxf(alpha, 0.001f);
    const float a2 = a * a;
    const float ndh = saturate(n_dot_h);
    const float d = (ndh * ndh) * (a2 - 1.0f) + 1.0f;
    return a2 / ((kPi * d) * d + kEps);
}
AWSM_DI float geometry_schlick_ggx(float n_dot_x, float alpha) {                         // brdf.wgsl:127-132
    const float a = fmaxf(alpha, 0.001f);
    const float k = ((a + 1.0f) * (a + 1.0f)) * 0.125f;
    const float ndx = saturate(n_dot_x);
    return ndx / (ndx * (1.0f - k) + k);
}
AWSM_DI float geometry_smith(f3 n, f3 v, f3 l, float alpha) {
    return geometry_schlick_ggx(saturate(dot(n, v)), alpha) * geometry_schlick_ggx(saturate(dot(n, l)), alpha);
}
constexpr float kClearcoatF0 = 0.04f;
AWSM_DI float clearcoat_brdf_direct(float clearcoat, float cc_roughness, f3 cc_normal, f3 v, f3 l) {   // brdf.wgsl:149-181
    if (clearcoat <= 0.0f) return 0.0f;
    const f3 cc_n = safe_normalize(cc_normal);
    const f3 h = safe_half_vector(v, l);
    if (dot(h, h) == 0.0f) return 0.0f;
    const float cc_n_dot_l = fmaxf(dot(cc_n, l), 0.0f);
    const float cc_n_dot_v = fmaxf(dot(cc_n, v), 1e-4f);
    const float cc_n_dot_h = fmaxf(dot(cc_n, h), 0.0f);
    const float cc_v_dot_h = fmaxf(dot(v, h), 0.0f);
    const float cc_alpha = fmaxf(cc_roughness * cc_roughness, 0.001f);
    const float Fc = fresnel_schlick_scalar(cc_v_dot_h, kClearcoatF0);
    const float Dc = distribution_ggx(cc_n_dot_h, cc_alpha);
    const float Gc = geometry_smith(cc_n, v, l, cc_alpha);
    return (((clearcoat * Fc) * Dc) * Gc) / fmaxf((4.0f * cc_n_dot_l) * cc_n_dot_v, kEps);
}
AWSM_DI float clearcoat_fresnel(float clearcoat, float v_dot_h) {
    if (clearcoat <= 0.0f) return 0.0f;
    return clearcoat * fresnel_schlick_scalar(v_dot_h, kClearcoatF0);
}
AWSM_DI f3 sheen_brdf_direct(f3 sheen_color, float sheen_roughness, f3 n, f3 v, f3 l) {  // brdf.wgsl:198-240
    if (sheen_color.x <= 0.0f && sheen_color.y <= 0.0f && sheen_color.z <= 0.0f) return {0.0f, 0.0f, 0.0f};
    const f3 h = safe_half_vector(v, l);
    if (dot(h, h) == 0.0f) return {0.0f, 0.0f, 0.0f};
    const float n_dot_l = fmaxf(dot(n, l), 0.0f);
    const float n_dot_v = fmaxf(dot(n, v), 1e-4f);
    const float n_dot_h = fmaxf(dot(n, h), 0.0f);
    const float roughness = fmaxf(sheen_roughness, 0.07f);
    const float alpha = roughness * roughness;
    const float inv_alpha = 1.0f / alpha;
    const float sin2h = 1.0f - n_dot_h * n_dot_h;
    const float D = ((2.0f + inv_alpha) * powf(sin2h, inv_alpha * 0.5f)) / (2.0f * kPi);
    const float V = 1.0f / (4.0f * ((n_dot_l + n_dot_v) - n_dot_l * n_dot_v));
    return (sheen_color * D) * V;
}
AWSM_DI float sheen_albedo_scaling(f3 sheen_color, float sheen_roughness, float n_dot_v) {   // brdf.wgsl:245-262
    const float sheen_max = fmaxf(fmaxf(sheen_color.x, sheen_color.y), sheen_color.z);
    if (sheen_max <= 0.0f) return 1.0f;
    const float alpha = sheen_roughness * sheen_roughness;
    const float E = alpha * (0.18f + 0.06f * (1.0f - n_dot_v));
    return 1.0f - sheen_max * E;
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

// brdf.wgsl:308-381
AWSM_DI f3 brdf_direct(const PbrColor& c, f3 normal, f3 light_dir, f3 radiance, f3 surface_to_camera) {
    const f3 n = safe_normalize(normal);
    const f3 v = safe_normalize(surface_to_camera);
    const f3 l = safe_normalize(light_dir);
    const f3 h = safe_half_vector(v, l);
    const float metallic = clampf(c.mr.x, 0.0f, 1.0f);
    const float roughness = fmaxf(clampf(c.mr.y, 0.0f, 1.0f), 0.04f);
    const float alpha = roughness * roughness;
    const float n_dot_l = fmaxf(dot(n, l), 0.0f);
    const float n_dot_v = fmaxf(dot(n, v), 1e-4f);
    const bool has_half = dot(h, h) > 0.0f;
    const float n_dot_h = has_half ? fmaxf(dot(n, h), 0.0f) : 0.0f;
    const float v_dot_h = has_half ? fmaxf(dot(v, h), 0.0f) : 0.0f;
    const float f0b = ior_to_f0(c.ior);
    const f3 dielectric_f0 = min3(splat3(f0b) * c.specular_color, splat3(1.0f)) * c.specular;
    const f3 F0 = mix3(dielectric_f0, c.base, metallic);
    const float f90 = mixf(c.specular, 1.0f, metallic);
    const f3 F = has_half ? fresnel_schlick_f90(v_dot_h, F0, f90) : fresnel_schlick_f90(n_dot_v, F0, f90);
    const float D = distribution_ggx(n_dot_h, alpha);
    const float G = geometry_smith(n, v, l, alpha);
    f3 specular = {0.0f, 0.0f, 0.0f};
    if (has_half) specular = (F * (D * G)) / fmaxf((4.0f * n_dot_l) * n_dot_v, kEps);
    const float F_max = fmaxf(fmaxf(F.x, F.y), F.z);
    const float k_d = (1.0f - F_max) * (1.0f - metallic);
    const f3 diffuse = (c.base * k_d) * (1.0f / kPi);
    f3 result = (((diffuse + specular) * radiance) * n_dot_l) * c.occlusion;
    const f3 sheen = sheen_brdf_direct(c.sheen_color, c.sheen_roughness, n, v, l);
    const float sheen_scaling = sheen_albedo_scaling(c.sheen_color, c.sheen_roughness, n_dot_v);
    result = result * sheen_scaling + ((sheen * radiance) * n_dot_l) * c.occlusion;
    const float clearcoat_spec = clearcoat_brdf_direct(c.clearcoat, c.clearcoat_roughness, c.clearcoat_normal, v, l);
    const float cc_fresnel = clearcoat_fresnel(c.clearcoat, v_dot_h);
    result = result * (1.0f - cc_fresnel) + (radiance * clearcoat_spec) * n_dot_l;
    return result;
}

// brdf.wgsl:389-576 (brdf_ibl -> brdf_ibl_with_transmission); the three cubes are uniform colours
AWSM_DI f3 brdf_ibl(const DevScene* sc, const PbrColor& c, f3 normal, f3 surface_to_camera) {
    const f3 prefiltered = {sc->prefiltered_rgb[0], sc->prefiltered_rgb[1], sc->prefiltered_rgb[2]};
    const f3 irradiance = {sc->irradiance_rgb[0], sc->irradiance_rgb[1], sc->irradiance_rgb[2]};
    const f3 n = safe_normalize(normal);
    const f3 v = safe_normalize(surface_to_camera);
    const float metallic = clampf(c.mr.x, 0.0f, 1.0f);
    const float roughness = fmaxf(clampf(c.mr.y, 0.0f, 1.0f), 0.04f);
    const float n_dot_v = saturate(dot(n, v));
    const float f0b = ior_to_f0(c.ior);
    const f3 dielectric_f0 = min3(splat3(f0b) * c.specular_color, splat3(1.0f)) * c.specular;
    const f3 F0 = mix3(dielectric_f0, c.base, metallic);
    const float f90 = mixf(c.specular, 1.0f, metallic);
    const f3 F_view = fresnel_schlick_f90(n_dot_v, F0, f90);
    const float F_view_max = fmaxf(fmaxf(F_view.x, F_view.y), F_view.z);
    const float effective_transmission = c.transmission * (1.0f - metallic);
    f3 base_layer;
    if (effective_transmission > 0.0f) {
        const f3 transmission_background = prefiltered;   // brdf.wgsl:531-561, uniform cube
        const f3 diffuse_brdf = (c.base * (1.0f / kPi)) * irradiance;
        f3 attenuation = splat3(1.0f);
        if (should_apply_volume_attenuation(c.volume_thickness, c.volume_attenuation_distance, c.volume_attenuation_color))
            attenuation = volume_attenuation(c.volume_thickness, c.volume_attenuation_color, c.volume_attenuation_distance);
        const f3 transmission_btdf = (transmission_background * c.base) * attenuation;
        base_layer = mix3(diffuse_brdf, transmission_btdf, effective_transmission);
    } else {
        base_layer = (c.base * (1.0f / kPi)) * irradiance;
    }
    const float k_d = (1.0f - F_view_max) * (1.0f - metallic);
    const f3 base_contribution = (base_layer * k_d) * c.occlusion;
    const f2 lut = sample_brdf_lut(sc, n_dot_v, roughness);
    const f3 spec_term = F0 * lut.x + splat3(f90 * lut.y);
    const f3 specular = (prefiltered * spec_term) * mixf(1.0f, c.occlusion, 0.5f);
    const float sheen_scaling = sheen_albedo_scaling(c.sheen_color, c.sheen_roughness, n_dot_v);
    f3 base_with_sheen = base_contribution * sheen_scaling;
    if (c.sheen_color.x > 0.0f || c.sheen_color.y > 0.0f || c.sheen_color.z > 0.0f) {
        const float alpha = c.sheen_roughness * c.sheen_roughness;
        const float fresnel_sheen = powf(1.0f - n_dot_v, 3.0f);
        base_with_sheen = base_with_sheen + (((c.sheen_color * irradiance) * alpha) * fresnel_sheen) * c.occlusion;
    }
    f3 result = (base_with_sheen + specular) + c.emissive;
    if (c.clearcoat > 0.0f) {
        const f3 cc_n = safe_normalize(c.clearcoat_normal);
        const float cc_n_dot_v = saturate(dot(cc_n, v));
        const float cc_roughness = fmaxf(c.clearcoat_roughness, 0.04f);
        const f2 cc_lut = sample_brdf_lut(sc, cc_n_dot_v, cc_roughness);
        const f3 cc_specular = prefiltered * (kClearcoatF0 * cc_lut.x + cc_lut.y);
        const float cc_fresnel = clearcoat_fresnel(c.clearcoat, n_dot_v);
        result = result * (1.0f - cc_fresnel) + cc_specular * c.clearcoat;
    }
    return result;
}

// lights.wgsl:70-152
AWSM_DI f3 apply_lighting(const DevScene* sc, const PbrColor& mc, f3 surface_to_camera, f3 world_position, uint32_t n_lights) {
    f3 color = brdf_ibl(sc, mc, mc.normal, surface_to_camera);
    const float4* lights = reinterpret_cast<const float4*>(sc->buf[AWSM_BUF_LIGHTS]);
    for (uint32_t i = 0; i < n_lights; i++) {
        const float4 pos_range = lights[i * 4 + 0], dir_inner = lights[i * 4 + 1], color_intensity = lights[i * 4 + 2], kind_outer = lights[i * 4 + 3];
        const uint32_t kind = (uint32_t)kind_outer.x;
        const f3 lcolor = {color_intensity.x, color_intensity.y, color_intensity.z};
        f3 light_dir = {0.0f, 0.0f, 0.0f}, radiance = {0.0f, 0.0f, 0.0f};
        if (kind == 1u) {
            light_dir = normalize(-mk3(dir_inner.x, dir_inner.y, dir_inner.z));
            radiance = lcolor * color_intensity.w;
        } else if (kind == 2u || kind == 3u) {
            const f3 stl = mk3(pos_range.x, pos_range.y, pos_range.z) - world_position;
            const float dist = length(stl);
            light_dir = stl / dist;
            float att = inverse_square(pos_range.w, dist);
            if (kind == 3u) {
                const float cos_l = dot(light_dir, -normalize(mk3(dir_inner.x, dir_inner.y, dir_inner.z)));
                const float sm = saturate((cos_l - kind_outer.y) / (dir_inner.w - kind_outer.y));
                att = att * (sm * sm);
            }
            radiance = (lcolor * color_intensity.w) * att;
        }
        color = color + brdf_direct(mc, mc.normal, light_dir, radiance, surface_to_camera);
    }
    return color;
}

AWSM_DI void store_pixel(const FrameDev& f, size_t p, f4 c) {
    ushort4 h = make_ushort4(f16_bits(c.x), f16_bits(c.y), f16_bits(c.z), f16_bits(c.w));
    reinterpret_cast<ushort4*>(f.out_rgba16f)[p] = h;
    if (f.out_rgba32f) reinterpret_cast<float4*>(f.out_rgba32f)[p] = make_float4(c.x, c.y, c.z, c.w);
}

// ------------------------------------------------------------------------------------------------
// k_shade: 16x16 pixels per workgroup (compute.wgsl uses 8x8; a 64-wide wavefront covers 16x4 here).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_shade(const DevScene* __restrict__ sc, FrameDev f) {
    const uint32_t bx_n = (f.width + 15u) >> 4, by_n = ((f.y1 - f.y0) + 15u) >> 4;
    const uint32_t nblk = bx_n * by_n;
    const uint32_t blk = xcd_remap_s(blockIdx.x, nblk);
    if (blk >= nblk) return;
    const int cx = (int)((blk % bx_n) << 4) + (int)(threadIdx.x & 15u);
    const int cy = (int)f.y0 + (int)((blk / bx_n) << 4) + (int)(threadIdx.x >> 4);
    if (cx >= (int)f.width || cy >= (int)f.y1) return;                   // compute.wgsl:111-113
    const size_t p = (size_t)cy * f.width + (size_t)cx;
    const f4 sky = {sc->skybox_rgba[0], sc->skybox_rgba[1], sc->skybox_rgba[2], sc->skybox_rgba[3]};   // skybox.wgsl:1-41, uniform cube

    const unsigned long long key = f.vis[p];
    if (!f.has_opaque || key == ~0ull) { store_pixel(f, p, sky); return; }   // compute.wgsl:149-153 / empty.wgsl
    atomicAdd(&f.counters[3], 1u);   // wave-aggregated by the compiler

    const uint32_t rank = 0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull);
    const float depth_sample = __uint_as_float((uint32_t)(key >> 32));
    uint32_t lo = 0, hi = f.n_draws;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (f.draws[mid].first_tri <= rank) lo = mid; else hi = mid;
    }
    const uint32_t triangle_index = rank - f.draws[lo].first_tri;
    const uint32_t geom_meta_off = f.draws[lo].geom_meta_off;
    const uint32_t material_meta_offset = *reinterpret_cast<const uint32_t*>(sc->buf[AWSM_BUF_GEOM_META] + geom_meta_off + 36);
    const uint32_t* mm = reinterpret_cast<const uint32_t*>(sc->buf[AWSM_BUF_MATERIAL_META] + (size_t)(material_meta_offset / 256u) * 256u);
    if (mm[16] == 1u) { if (f.out_rgba32f) reinterpret_cast<float4*>(f.out_rgba32f)[p] = make_float4(0, 0, 0, 0);
                        reinterpret_cast<ushort4*>(f.out_rgba16f)[p] = make_ushort4(0, 0, 0, 0); return; }   // is_hud: stays cleared
    const uint32_t material_offset = mm[6];
    const uint32_t attr_indices_off = mm[9] / 4u, attr_data_off = mm[10] / 4u, stride = mm[11] / 4u, uv_sets_index = mm[12];

    // ---- fs_main for this pixel, rounded to the G-buffer storage formats ----
    const float4 v0 = f.clip[(size_t)rank * 3], v1 = f.clip[(size_t)rank * 3 + 1], v2 = f.clip[(size_t)rank * 3 + 2];
    TriSetup t;
    if (!tri_setup(v0, v1, v2, false, f.width, f.height, 0u, f.height, t)) { store_pixel(f, p, sky); return; }
    float e0, e1, e2;
    tri_edges(t, cx, cy, e0, e1, e2);
    const float esum = (e0 + e1) + e2;
    const float b0 = e0 / esum, b1 = e1 / esum, b2 = e2 / esum;
    const float4 n0 = f.nrm[(size_t)rank * 3], n1 = f.nrm[(size_t)rank * 3 + 1], n2 = f.nrm[(size_t)rank * 3 + 2];
    const float4 t0 = f.tan[(size_t)rank * 3], t1 = f.tan[(size_t)rank * 3 + 1], t2 = f.tan[(size_t)rank * 3 + 2];
    const f3 Ni = {(b0 * n0.x + b1 * n1.x) + b2 * n2.x, (b0 * n0.y + b1 * n1.y) + b2 * n2.y, (b0 * n0.z + b1 * n1.z) + b2 * n2.z};
    const f4 Ti = {(b0 * t0.x + b1 * t1.x) + b2 * t2.x, (b0 * t0.y + b1 * t1.y) + b2 * t2.y,
                   (b0 * t0.z + b1 * t1.z) + b2 * t2.z, (b0 * t0.w + b1 * t1.w) + b2 * t2.w};
    f4 packed = pack_normal_tangent(normalize(Ni), normalize(mk3(Ti.x, Ti.y, Ti.z)), Ti.w);
    packed = {round_f16(packed.x), round_f16(packed.y), round_f16(packed.z), round_f16(packed.w)};   // RGBA16F
    const float bx = round_f16(b0), by = round_f16(b1);                                               // RG16F

    // ---- compute.wgsl:182-211 ----
    Attr a;
    a.sc = sc;
    a.ad = reinterpret_cast<const float*>(sc->buf[AWSM_BUF_ATTR_DATA]);
    a.bary = {bx, by, (1.0f - bx) - by};
    a.uv_sets_index = uv_sets_index;
    const uint32_t* attr_idx = reinterpret_cast<const uint32_t*>(sc->buf[AWSM_BUF_ATTR_INDEX]) + attr_indices_off + triangle_index * 3u;
    a.v0 = attr_data_off + attr_idx[0] * stride;
    a.v1 = attr_data_off + attr_idx[1] * stride;
    a.v2 = attr_data_off + attr_idx[2] * stride;

    // ---- standard.wgsl:11-62 ----
    const uint8_t* cam = sc->buf[AWSM_BUF_CAMERA];
    const m4 inv_proj = load_m4(reinterpret_cast<const float*>(cam + 256));
    const m4 inv_view = load_m4(reinterpret_cast<const float*>(cam + 320));
    const float proj33 = *reinterpret_cast<const float*>(cam + 64 + 60);
    const float* cam_pos = reinterpret_cast<const float*>(cam + 384);
    const f2 uv = {((float)cx + 0.5f) / (float)f.width, ((float)cy + 0.5f) / (float)f.height};
    const f4 view_h = mul(inv_proj, {uv.x * 2.0f - 1.0f, 1.0f - uv.y * 2.0f, depth_sample, 1.0f});
    const float vw = fmaxf(view_h.w, 1e-8f);
    const f3 view_position = {view_h.x / vw, view_h.y / vw, view_h.z / vw};
    const f4 wp = mul(inv_view, {view_position.x, view_position.y, view_position.z, 1.0f});
    const f3 world_position = {wp.x, wp.y, wp.z};
    f3 surface_to_camera;
    if (proj33 > 0.9f) {
        surface_to_camera = normalize(mk3(inv_view.c[2].x, inv_view.c[2].y, inv_view.c[2].z));
    } else {
        const f3 to_camera = mk3(cam_pos[0], cam_pos[1], cam_pos[2]) - world_position;
        surface_to_camera = dot(to_camera, to_camera) > 0.0f ? safe_normalize(to_camera) : mk3(0.0f, 0.0f, 1.0f);
    }
    const TBN tbn = unpack_normal_tangent(packed);
    const uint32_t n_lights = *reinterpret_cast<const uint32_t*>(sc->buf[AWSM_BUF_LIGHTS_INFO]);

    const uint32_t* M = reinterpret_cast<const uint32_t*>(sc->buf[AWSM_BUF_MATERIALS]);
    const uint32_t shader_id = M[material_offset / 4u];
    const uint32_t b = material_offset / 4u + 1u;
    if (shader_id == 2u) {   // unlit_material.wgsl:28-73 + material_color_calc.wgsl:517-580
        const TexInfo base_tex = tex_load(M, b + 2), em_tex = tex_load(M, b + 11);
        f4 base = {mf(M, b + 7), mf(M, b + 8), mf(M, b + 9), mf(M, b + 10)};
        f3 em = {mf(M, b + 16), mf(M, b + 17), mf(M, b + 18)};
        if (base_tex.exists) { const f4 s = sample_tex(a, base_tex); base = {base.x * s.x, base.y * s.y, base.z * s.z, base.w * s.w}; }
        if (em_tex.exists) { const f4 s = sample_tex(a, em_tex); em = {em.x * s.x, em.y * s.y, em.z * s.z}; }
        store_pixel(f, p, {base.x + em.x, base.y + em.y, base.z + em.z, 1.0f});
        return;
    }

    // ---- pbr_material.wgsl:110-216 + material_color_calc.wgsl:25-265 ----
    PbrColor c;
    const uint32_t debug_bitmask = M[b + 38];
    const uint32_t fi = b + 39u;
    const uint32_t idx_vertex_color = abs_index(b, M[fi + 0]), idx_emissive_strength = abs_index(b, M[fi + 1]);
    const uint32_t idx_ior = abs_index(b, M[fi + 2]), idx_specular = abs_index(b, M[fi + 3]), idx_transmission = abs_index(b, M[fi + 4]);
    const uint32_t idx_volume = abs_index(b, M[fi + 6]), idx_clearcoat = abs_index(b, M[fi + 7]), idx_sheen = abs_index(b, M[fi + 8]);
    {
        const TexInfo tx = tex_load(M, b + 2);
        f4 base = {mf(M, b + 7), mf(M, b + 8), mf(M, b + 9), mf(M, b + 10)};
        if (tx.exists) { const f4 s = sample_tex(a, tx); base = {base.x * s.x, base.y * s.y, base.z * s.z, base.w * s.w}; }
        base.w = 1.0f;
        if (idx_vertex_color != 0u) { const f4 vc = vertex_color(a, M[idx_vertex_color]); base = {base.x * vc.x, base.y * vc.y, base.z * vc.z, base.w * vc.w}; }
        c.base = {base.x, base.y, base.z};
    }
    {
        const TexInfo tx = tex_load(M, b + 11);
        c.mr = {mf(M, b + 16), mf(M, b + 17)};
        if (tx.exists) { const f4 s = sample_tex(a, tx); c.mr = {c.mr.x * s.z, c.mr.y * s.y}; }
    }
    c.normal = normal_map(a, tex_load(M, b + 18), mf(M, b + 23), tbn);
    {
        const TexInfo tx = tex_load(M, b + 24);
        c.occlusion = 1.0f;
        if (tx.exists) { const f4 s = sample_tex(a, tx); c.occlusion = mixf(1.0f, s.x, mf(M, b + 29)); }
    }
    {
        const TexInfo tx = tex_load(M, b + 30);
        f3 em = {mf(M, b + 35), mf(M, b + 36), mf(M, b + 37)};
        if (tx.exists) { const f4 s = sample_tex(a, tx); em = {em.x * s.x, em.y * s.y, em.z * s.z}; }
        c.emissive = em * (idx_emissive_strength == 0u ? 1.0f : mf(M, idx_emissive_strength));
    }
    c.ior = idx_ior == 0u ? 1.5f : mf(M, idx_ior);
    c.specular = 1.0f; c.specular_color = {1.0f, 1.0f, 1.0f};
    if (idx_specular != 0u) {
        const uint32_t i = idx_specular;
        const TexInfo tx = tex_load(M, i), ctx = tex_load(M, i + 6);
        c.specular = mf(M, i + 5);
        if (tx.exists) c.specular = c.specular * sample_tex(a, tx).w;
        c.specular_color = {mf(M, i + 11), mf(M, i + 12), mf(M, i + 13)};
        if (ctx.exists) { const f4 s = sample_tex(a, ctx); c.specular_color = {c.specular_color.x * s.x, c.specular_color.y * s.y, c.specular_color.z * s.z}; }
    }
    c.transmission = 0.0f;
    if (idx_transmission != 0u) {
        const uint32_t i = idx_transmission;
        const TexInfo tx = tex_load(M, i);
        const float factor = mf(M, i + 5);
        if (!(!tx.exists && factor == 0.0f)) { c.transmission = factor; if (tx.exists) c.transmission = c.transmission * sample_tex(a, tx).x; }
    }
    c.volume_thickness = 0.0f; c.volume_attenuation_distance = 0.0f; c.volume_attenuation_color = {1.0f, 1.0f, 1.0f};
    if (idx_volume != 0u) {
        const uint32_t i = idx_volume;
        const TexInfo tx = tex_load(M, i);
        const float factor = mf(M, i + 5);
        if (!(!tx.exists && factor == 0.0f)) { c.volume_thickness = factor; if (tx.exists) c.volume_thickness = c.volume_thickness * sample_tex(a, tx).y; }
        c.volume_attenuation_distance = mf(M, i + 6);
        c.volume_attenuation_color = {mf(M, i + 7), mf(M, i + 8), mf(M, i + 9)};
    }
    c.clearcoat = 0.0f; c.clearcoat_roughness = 0.0f; c.clearcoat_normal = tbn.N;
    if (idx_clearcoat != 0u) {
        const uint32_t i = idx_clearcoat;
        const TexInfo tx = tex_load(M, i), rtx = tex_load(M, i + 6);
        const float factor = mf(M, i + 5);
        if (!(!tx.exists && factor == 0.0f)) { c.clearcoat = factor; if (tx.exists) c.clearcoat = c.clearcoat * sample_tex(a, tx).x; }
        c.clearcoat_roughness = mf(M, i + 11);
        if (rtx.exists) c.clearcoat_roughness = c.clearcoat_roughness * sample_tex(a, rtx).y;
        c.clearcoat_normal = normal_map(a, tex_load(M, i + 12), mf(M, i + 17), tbn);
    }
    c.sheen_color = {0.0f, 0.0f, 0.0f}; c.sheen_roughness = 0.0f;
    if (idx_sheen != 0u) {
        const uint32_t i = idx_sheen;
        const TexInfo rtx = tex_load(M, i), ctx = tex_load(M, i + 6);
        c.sheen_roughness = mf(M, i + 5);
        if (rtx.exists) c.sheen_roughness = c.sheen_roughness * sample_tex(a, rtx).w;
        c.sheen_color = {mf(M, i + 11), mf(M, i + 12), mf(M, i + 13)};
        if (ctx.exists) { const f4 s = sample_tex(a, ctx); c.sheen_color = {c.sheen_color.x * s.x, c.sheen_color.y * s.y, c.sheen_color.z * s.z}; }
    }

    if (debug_bitmask != 0u) {   // pbr_material_color.wgsl:34-60
        f3 dc = {1.0f, 0.0f, 1.0f};
        if (debug_bitmask & 1u) dc = c.base;
        else if (debug_bitmask & 2u) dc = {c.mr.x, c.mr.y, 0.0f};
        else if (debug_bitmask & 4u) dc = {c.normal.x * 0.5f + 0.5f, c.normal.y * 0.5f + 0.5f, c.normal.z * 0.5f + 0.5f};
        else if (debug_bitmask & 8u) dc = splat3(c.occlusion);
        else if (debug_bitmask & 16u) dc = c.emissive;
        else if (debug_bitmask & 32u) dc = c.specular_color * c.specular;
        store_pixel(f, p, {dc.x, dc.y, dc.z, 1.0f});
        return;
    }
    const f3 color = apply_lighting(sc, c, surface_to_camera, world_position, n_lights);
    store_pixel(f, p, {color.x, color.y, color.z, 1.0f});
}

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

// helper kernels for readback / upload conversions
__global__ void k_rgba16f_to_rg16f(const uint16_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t n) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) out[i] = (uint32_t)in[(size_t)i * 4] | ((uint32_t)in[(size_t)i * 4 + 1] << 16);
}

}  // namespace awsm

extern "C" void awsm_launch_shade(const awsm::DevScene* sc, const awsm::FrameDev* f, hipStream_t s) {
    const uint32_t bx_n = (f->width + 15u) >> 4, by_n = ((f->y1 - f->y0) + 15u) >> 4;
    const uint32_t nb = ((bx_n * by_n + 7u) / 8u) * 8u;
    if (nb) hipLaunchKernelGGL(awsm::k_shade, dim3(nb), dim3(256), 0, s, sc, *f);
}
extern "C" void awsm_launch_brdf_lut(uint32_t* out_rg16f, uint32_t w, uint32_t h, hipStream_t s) {
    hipLaunchKernelGGL(awsm::k_brdf_lut, dim3((w + 15u) / 16u, (h + 15u) / 16u), dim3(256), 0, s, out_rg16f, w, h);
}
extern "C" void awsm_launch_rgba16f_to_rg16f(const uint16_t* in, uint32_t* out, uint32_t n, hipStream_t s) {
    hipLaunchKernelGGL(awsm::k_rgba16f_to_rg16f, dim3((n + 255u) / 256u), dim3(256), 0, s, in, out, n);
}
