/*
 * oracle_brdf_lut.c — TEST INFRASTRUCTURE ONLY (parity oracle; "parity unpinned", see oracle.h).
 *
 * Split-sum BRDF LUT as the reference renders it at renderer build time:
 *   /root/reference/crates/renderer-core/src/brdf_lut/shader.wgsl:1-78   (vs_main full-screen triangle, fs_main)
 *   /root/reference/crates/renderer-core/src/brdf_lut/generate.rs:47-96  (RGBA16F target, default 1024x1024)
 *
 * Orientation: vs_main maps clip (-1,-3),(-1,1),(3,1) to uv = p*0.5+0.5, so the fragment at framebuffer
 * pixel (i,j) (row 0 = top = clip y +1) sees uv = ((i+0.5)/W, 1-(j+0.5)/H): texel ROW 0 holds
 * roughness ~ 1.  The opaque pass samples the LUT at v = roughness (brdf.wgsl:293-302), i.e. it reads the
 * row computed for 1-roughness.  That is the reference's behaviour and is kept as is.
 */
#include "oracle.h"
#include "oracle_math.h"

static float radical_inverse_vdc(uint32_t bits) {
    bits = (bits << 16) | (bits >> 16);
    bits = ((bits & 0x55555555u) << 1) | ((bits & 0xAAAAAAAAu) >> 1);
    bits = ((bits & 0x33333333u) << 2) | ((bits & 0xCCCCCCCCu) >> 2);
    bits = ((bits & 0x0F0F0F0Fu) << 4) | ((bits & 0xF0F0F0F0u) >> 4);
    bits = ((bits & 0x00FF00FFu) << 8) | ((bits & 0xFF00FF00u) >> 8);
    return (float)bits * 2.3283064365386963e-10f;
}
static float lut_geometry_schlick_ggx(float ndot_v, float alpha) {
    float a = fmaxf(alpha, 0.001f);
    float k = ((a + 1.0f) * (a + 1.0f)) * 0.125f;
    return ndot_v / (ndot_v * (1.0f - k) + k);
}

static void lut_texel(float uvx, float uvy, float* out_a, float* out_b) {
    float no_v = o_clamp(uvx, 1e-3f, 1.0f - 1e-3f);
    float roughness = o_clamp(uvy, 1e-3f, 1.0f - 1e-3f);
    ovec3 v = ov3(sqrtf(fmaxf(0.0f, 1.0f - no_v * no_v)), 0.0f, no_v);
    const uint32_t sample_count = 1024u;
    float a = 0.0f, b = 0.0f;
    float alpha = roughness * roughness;
    for (uint32_t i = 0; i < sample_count; i++) {
        float xi_x = (float)i / (float)sample_count, xi_y = radical_inverse_vdc(i);
        float a2 = alpha * alpha;
        float phi = 6.28318530718f * xi_x;
        float cos_theta = sqrtf((1.0f - xi_y) / (1.0f + (a2 - 1.0f) * xi_y));
        float sin_theta = sqrtf(fmaxf(0.0f, 1.0f - cos_theta * cos_theta));
        ovec3 h = ov3(cosf(phi) * sin_theta, sinf(phi) * sin_theta, cos_theta);
        float vdh = ov3_dot(v, h);
        ovec3 l = ov3_normalize(ov3_sub(ov3_scale(h, 2.0f * vdh), v));
        float no_l = fmaxf(l.z, 0.0f), no_h = fmaxf(h.z, 0.0f), vo_h = fmaxf(vdh, 0.0f), no_v_ = fmaxf(v.z, 0.0f);
        if (no_l > 0.0f) {
            float g = lut_geometry_schlick_ggx(no_v_, alpha) * lut_geometry_schlick_ggx(no_l, alpha);
            float g_vis = (g * vo_h) / fmaxf(no_h * no_v_, 1e-4f);
            float fc = powf(1.0f - vo_h, 5.0f);
            a = a + (1.0f - fc) * g_vis;
            b = b + fc * g_vis;
        }
    }
    *out_a = a / (float)sample_count;
    *out_b = b / (float)sample_count;
}

int oracle_brdf_lut(uint32_t width, uint32_t height, uint16_t* rg16f_out, int threads) {
    if (threads < 1) threads = 1;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
    for (int j = 0; j < (int)height; j++) {
        for (uint32_t i = 0; i < width; i++) {
            float a, b;
            lut_texel(((float)i + 0.5f) / (float)width, 1.0f - ((float)j + 0.5f) / (float)height, &a, &b);
            rg16f_out[((size_t)j * width + i) * 2 + 0] = o_f32_to_f16(a);
            rg16f_out[((size_t)j * width + i) * 2 + 1] = o_f32_to_f16(b);
        }
    }
    return 0;
}
