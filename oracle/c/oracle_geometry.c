/*
 * oracle_geometry.c — TEST INFRASTRUCTURE ONLY (parity oracle; "parity unpinned", see oracle.h).
 *
 * Geometry Pass restated on the CPU.  Paths relative to /root/reference/crates/renderer/src/ :
 *   render_passes/geometry/shader/geometry_wgsl/vertex.wgsl:36-63      vert_main
 *   render_passes/shared/shared_wgsl/vertex/apply_vertex.wgsl:24-118   apply_vertex
 *   render_passes/shared/shared_wgsl/vertex/morph.wgsl:4-168           position/normal/tangent morphs
 *   render_passes/shared/shared_wgsl/vertex/skin.wgsl:7-157            position/normal skinning
 *   render_passes/shared/shared_wgsl/vertex/transform.wgsl:3-5         get_model_transform
 *   render_passes/shared/shared_wgsl/vertex/geometry_mesh_meta.wgsl:2-15
 *   render_passes/geometry/pipeline.rs:28-73 (vertex layout, 56 B), :337-344 (TriangleList, CCW front,
 *       cull None|Back, depth write, LessEqual)
 *   render_passes/geometry/render_pass.rs:22-30,107-114 (clears: vis = 0xFFFF.. "no hit", depth = 1.0)
 *
 * The fixed-function rasteriser has no source in the reference (WebGPU leaves sub-pixel snapping,
 * interpolation order and tie handling to the implementation), so this file DEFINES the raster
 * contract both this oracle and the HIP kernels implement (DESIGN.md §"Raster contract"):
 *   homogeneous (clip-less) edge functions in f32 evaluated at pixel centres (x+0.5, y+0.5),
 *   top-left fill rule, clip to 0 <= z_ndc <= 1 per pixel, z_ndc = (e0*z0 + e1*z1 + e2*z2)/det,
 *   depth kept as f32 bits, LessEqual with draws and triangles in submission order.
 */
#include "oracle.h"
#include "oracle_math.h"
#include "oracle_raster.h"
#include <stdlib.h>

typedef struct {
    uint32_t mesh_key_high, mesh_key_low;
    uint32_t morph_len, morph_weights_off, morph_values_off;
    uint32_t skin_sets, skin_matrices_off, skin_index_weights_off;
    uint32_t transform_off, material_meta_off;
} GeomMeta;   /* geometry_mesh_meta.wgsl:2-15; meshes/meta/geometry_meta.rs:44-113 */

static inline uint32_t rd_u32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline float rd_f32(const uint8_t* p) { float v; memcpy(&v, p, 4); return v; }

static GeomMeta load_geom_meta(const OracleScene* s, uint32_t off) {
    const uint8_t* p = s->buf[AWSM_BUF_GEOM_META] + off;
    GeomMeta m;
    m.mesh_key_high = rd_u32(p + 0); m.mesh_key_low = rd_u32(p + 4);
    m.morph_len = rd_u32(p + 8); m.morph_weights_off = rd_u32(p + 12); m.morph_values_off = rd_u32(p + 16);
    m.skin_sets = rd_u32(p + 20); m.skin_matrices_off = rd_u32(p + 24); m.skin_index_weights_off = rd_u32(p + 28);
    m.transform_off = rd_u32(p + 32); m.material_meta_off = rd_u32(p + 36);
    return m;
}

uint32_t oracle_total_vertices(const OracleScene* s) {
    uint64_t t = 0;
    for (uint32_t d = 0; d < s->n_draws; d++) t += 3ull * s->draws[d].tri_count * (s->draws[d].inst_count ? s->draws[d].inst_count : 1u);   /* every instance has its own transformed vertices */
    return (uint32_t)t;
}

/* skin.wgsl:7-81 / :84-157 — the blended matrix is identical for position and normal */
static omat4 skin_matrix(const OracleScene* s, const GeomMeta* gm, uint32_t vertex_index) {
    const float* iw = (const float*)s->buf[AWSM_BUF_SKIN_INDEX_WEIGHTS];
    const float* jm = (const float*)s->buf[AWSM_BUF_SKIN_MATRICES];
    uint32_t base = gm->skin_index_weights_off / 4u + vertex_index * gm->skin_sets * 8u;
    uint32_t moff = gm->skin_matrices_off / 64u;
    omat4 skin;
    memset(&skin, 0, sizeof skin);
    for (uint32_t set = 0; set < gm->skin_sets; set++) {
        uint32_t bo = base + set * 8u;
        uint32_t ji[4]; float jw[4];
        for (int k = 0; k < 4; k++) { ji[k] = o_f32_bits(iw[bo + 2 * k]); jw[k] = iw[bo + 2 * k + 1]; }
        float acc[16];
        for (int e = 0; e < 16; e++) {
            /* w0*J0 + w1*J1 + w2*J2 + w3*J3, left to right */
            float v = jw[0] * jm[(size_t)(ji[0] + moff) * 16 + e];
            v = v + jw[1] * jm[(size_t)(ji[1] + moff) * 16 + e];
            v = v + jw[2] * jm[(size_t)(ji[2] + moff) * 16 + e];
            v = v + jw[3] * jm[(size_t)(ji[3] + moff) * 16 + e];
            acc[e] = v;
        }
        float* sk = (float*)&skin;
        if (set == 0) { for (int e = 0; e < 16; e++) sk[e] = acc[e]; }
        else { for (int e = 0; e < 16; e++) sk[e] = sk[e] + acc[e]; }
    }
    return skin;
}

/* apply_vertex.wgsl:24-118 for one vertex: position, normal, tangent and the vertex index that addresses the morph and skin
 * data (the exploded 56-byte record of the geometry pass carries original_vertex_index, pipeline.rs:28-73; the indexed
 * 40-byte vertices of the transparent pass use @builtin(vertex_index)).  wpos_out (may be NULL): world position xyz. */
static void apply_vertex(const OracleScene* s, const GeomMeta* gm, const omat4* view_proj, ovec3 pos, ovec3 normal, ovec4 tangent,
                         uint32_t vertex_index, const float* instance_mat4, float* clip_out, float* nt_out, float* wpos_out) {

    if (gm->morph_len != 0u) {                  /* morph.wgsl: weights read at [off/4 + 1 + i] (morph.wgsl:19) */
        const float* mw = (const float*)s->buf[AWSM_BUF_MORPH_WEIGHTS];
        const float* mv = (const float*)s->buf[AWSM_BUF_MORPH_VALUES];
        uint32_t wbase = gm->morph_weights_off / 4u + 1u;
        uint32_t vbase = gm->morph_values_off / 4u + vertex_index * (gm->morph_len * 10u);
        ovec3 txyz = ov3(tangent.x, tangent.y, tangent.z);
        for (uint32_t i = 0; i < gm->morph_len; i++) {
            float w = mw[wbase + i];
            const float* v = mv + vbase + i * 10u;
            pos = ov3(pos.x + w * v[0], pos.y + w * v[1], pos.z + w * v[2]);
            normal = ov3(normal.x + w * v[3], normal.y + w * v[4], normal.z + w * v[5]);
            txyz = ov3(txyz.x + w * v[6], txyz.y + w * v[7], txyz.z + w * v[8]);
        }
        tangent = ov4(txyz.x, txyz.y, txyz.z, tangent.w);
    }
    if (gm->skin_sets != 0u) {
        omat4 skin = skin_matrix(s, gm, vertex_index);
        ovec4 p = omat4_mul_v4(&skin, ov4(pos.x, pos.y, pos.z, 1.0f));
        pos = ov3(p.x, p.y, p.z);
        omat3 nm = omat3_from_mat4(&skin);       /* skin.wgsl:150-156: raw upper 3x3, no inverse-transpose */
        normal = omat3_mul_v3(&nm, normal);
        ovec3 t = omat3_mul_v3(&nm, ov3(tangent.x, tangent.y, tangent.z));
        tangent = ov4(t.x, t.y, t.z, tangent.w);
    }

    omat4 model = omat4_load((const float*)(s->buf[AWSM_BUF_TRANSFORMS] + (size_t)(gm->transform_off / 64u) * 64u));
    if (instance_mat4) {   /* apply_vertex.wgsl:47-59: model_transform = model * instance_transform (column by column) */
        omat4 inst = omat4_load(instance_mat4), mi;
        for (int j = 0; j < 4; j++) mi.c[j] = omat4_mul_v4(&model, inst.c[j]);
        model = mi;
    }
    ovec4 world_pos = omat4_mul_v4(&model, ov4(pos.x, pos.y, pos.z, 1.0f));
    ovec4 clip = omat4_mul_v4(view_proj, world_pos);

    omat3 m3 = omat3_from_mat4(&model);
    ovec3 c0 = m3.c[0], c1 = m3.c[1], c2 = m3.c[2];
    ovec3 r0 = ov3(c0.x, c1.x, c2.x), r1 = ov3(c0.y, c1.y, c2.y), r2 = ov3(c0.z, c1.z, c2.z);
    ovec3 cof0 = ov3_cross(r1, r2), cof1 = ov3_cross(r2, r0), cof2 = ov3_cross(r0, r1);
    float det_model = ov3_dot(r0, cof0);
    ovec3 wn_un;
    if (fabsf(det_model) > 1e-8f) {
        wn_un = ov3(ov3_dot(cof0, normal) / det_model, ov3_dot(cof1, normal) / det_model, ov3_dot(cof2, normal) / det_model);
    } else {
        wn_un = omat3_mul_v3(&m3, normal);
    }
    ovec3 world_normal = ov3_normalize(wn_un);

    ovec3 tangent_raw = omat3_mul_v3(&m3, ov3(tangent.x, tangent.y, tangent.z));
    ovec3 tangent_ortho = ov3_sub(tangent_raw, ov3_scale(world_normal, ov3_dot(tangent_raw, world_normal)));
    float tlen_sq = ov3_dot(tangent_ortho, tangent_ortho);
    if (tlen_sq > 1e-8f) {
        tangent_ortho = ov3_scale(tangent_ortho, o_inverse_sqrt(tlen_sq));
    } else {
        ovec3 axis = (fabsf(world_normal.z) > 0.999f) ? ov3(0.0f, 1.0f, 0.0f) : ov3(0.0f, 0.0f, 1.0f);
        tangent_ortho = ov3_normalize(ov3_cross(axis, world_normal));
    }

    clip_out[0] = clip.x; clip_out[1] = clip.y; clip_out[2] = clip.z; clip_out[3] = clip.w;
    nt_out[0] = world_normal.x; nt_out[1] = world_normal.y; nt_out[2] = world_normal.z; nt_out[3] = 0.0f;
    nt_out[4] = tangent_ortho.x; nt_out[5] = tangent_ortho.y; nt_out[6] = tangent_ortho.z; nt_out[7] = tangent.w;
    if (wpos_out) { wpos_out[0] = world_pos.x; wpos_out[1] = world_pos.y; wpos_out[2] = world_pos.z; wpos_out[3] = 1.0f; }
}

int oracle_transform(const OracleScene* s, float* clip_out, float* nt_out) {
    /* camera.wgsl:2-19: view_proj is the third mat4 of the 512-B camera UBO (byte 128) */
    omat4 view_proj = omat4_load((const float*)(s->buf[AWSM_BUF_CAMERA] + 128));
    size_t v = 0;
    for (uint32_t d = 0; d < s->n_draws; d++) {
        const AwsmDraw* dr = &s->draws[d];
        GeomMeta gm = load_geom_meta(s, dr->geom_meta_off);
        const uint8_t* base = s->buf[AWSM_BUF_VIS_GEOM_DATA] + dr->vis_data_off;
        const uint32_t copies = dr->inst_count ? dr->inst_count : 1u;       /* draw_indexed(.., instance_count): instance after instance */
        for (uint32_t k = 0; k < copies; k++) {
            const float* inst = dr->inst_count ? (const float*)(s->buf[AWSM_BUF_INSTANCES] + dr->inst_off + 64u * k) : NULL;
            for (uint32_t i = 0; i < 3u * dr->tri_count; i++, v++) {
                const uint8_t* vtx = base + (size_t)i * 56u;
                apply_vertex(s, &gm, &view_proj, ov3(rd_f32(vtx + 0), rd_f32(vtx + 4), rd_f32(vtx + 8)), ov3(rd_f32(vtx + 24), rd_f32(vtx + 28), rd_f32(vtx + 32)),
                             ov4(rd_f32(vtx + 36), rd_f32(vtx + 40), rd_f32(vtx + 44), rd_f32(vtx + 48)), rd_u32(vtx + 52), inst,
                             clip_out + v * 4, nt_out + v * 8, NULL);
            }
        }
    }
    return 0;
}

/* Transparent pass vert_main (material_transparent_wgsl/vertex.wgsl:40-72): indexed draw of the mesh's 40-byte vertices
 * (AWSM_BUF_TRANSPARENCY_GEOM_DATA at draw.vis_data_off) through the custom-attribute index buffer
 * (meshes/mesh.rs:129-200).  Output is one vertex per triangle corner, in triangle order, like oracle_transform. */
uint32_t oracle_forward_total_vertices(const AwsmDraw* draws, uint32_t n_draws) {
    uint32_t n = 0;
    for (uint32_t d = 0; d < n_draws; d++) n += 3u * draws[d].tri_count * (draws[d].inst_count ? draws[d].inst_count : 1u);
    return n;
}
int oracle_forward_transform(const OracleScene* s, const AwsmDraw* draws, uint32_t n_draws, float* clip_out, float* nt_out, float* wpos_out) {
    omat4 view_proj = omat4_load((const float*)(s->buf[AWSM_BUF_CAMERA] + 128));
    size_t v = 0;
    for (uint32_t d = 0; d < n_draws; d++) {
        const AwsmDraw* dr = &draws[d];
        GeomMeta gm = load_geom_meta(s, dr->geom_meta_off);
        const uint8_t* mm = s->buf[AWSM_BUF_MATERIAL_META] + (size_t)(gm.material_meta_off / 256u) * 256u;
        const uint32_t* index = (const uint32_t*)(s->buf[AWSM_BUF_ATTR_INDEX] + rd_u32(mm + 36));   /* custom_attribute_indices_offset */
        const uint8_t* base = s->buf[AWSM_BUF_TRANSPARENCY_GEOM_DATA] + dr->vis_data_off;
        const uint32_t copies = dr->inst_count ? dr->inst_count : 1u;
        for (uint32_t k = 0; k < copies; k++) {
            const float* inst = dr->inst_count ? (const float*)(s->buf[AWSM_BUF_INSTANCES] + dr->inst_off + 64u * k) : NULL;
            for (uint32_t i = 0; i < 3u * dr->tri_count; i++, v++) {
                const uint32_t vi = index[i];
                const uint8_t* vtx = base + (size_t)vi * 40u;
                apply_vertex(s, &gm, &view_proj, ov3(rd_f32(vtx + 0), rd_f32(vtx + 4), rd_f32(vtx + 8)), ov3(rd_f32(vtx + 12), rd_f32(vtx + 16), rd_f32(vtx + 20)),
                             ov4(rd_f32(vtx + 24), rd_f32(vtx + 28), rd_f32(vtx + 32), rd_f32(vtx + 36)), vi, inst,
                             clip_out + v * 4, nt_out + v * 8, wpos_out + v * 4);
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Raster contract (shared with awsm-renderer_amd/csrc/raster_setup.hpp — same operations, same order)
 * ---------------------------------------------------------------------------------------------- */
/* Two kinds of setup (DESIGN.md "Raster contract"):
 *  kind 0, every triangle in front of the camera (w > 0 at all three vertices, inside a +-32768-pixel guard band): the
 *    vertices are projected and SNAPPED to a 1/256-pixel grid, and everything that decides coverage — facing, edge
 *    functions, the top-left rule — is exact integer arithmetic (|values| < 2^49), as in a hardware rasteriser.  Exactness
 *    is what makes shared edges watertight and keeps small distant triangles from being lost to rounding.
 *  kind 1, triangles that touch w <= 0 (they cross the near plane; close to the camera, hence large on screen and well
 *    conditioned): homogeneous clip-less edge functions, no geometric clipping; coefficients in f32, evaluated per sample
 *    with two fused multiply-adds in f64 (the same instruction sequence the exact kind uses on the device). */
/* TriSetup: oracle_raster.h */

static inline int finite4(const float* v) {
    return isfinite(v[0]) && isfinite(v[1]) && isfinite(v[2]) && isfinite(v[3]);
}
#define SUBPIX 256            /* 8 fractional bits */
#define GUARD_BAND 8388608.0f /* |coordinate| * 256 <= 2^23  (+-32768 pixels) */

static void tri_setup(const float* v0, const float* v1, const float* v2, int cull_back,
                      uint32_t width, uint32_t height, uint32_t ry0, uint32_t ry1, TriSetup* t) {
    t->valid = 0;
    if (!finite4(v0) || !finite4(v1) || !finite4(v2)) return;
    /* frustum trivial reject: all three vertices outside the same clip plane (x,y in [-w,w], z in [0,w]) */
    if (v0[0] < -v0[3] && v1[0] < -v1[3] && v2[0] < -v2[3]) return;
    if (v0[0] > v0[3] && v1[0] > v1[3] && v2[0] > v2[3]) return;
    if (v0[1] < -v0[3] && v1[1] < -v1[3] && v2[1] < -v2[3]) return;
    if (v0[1] > v0[3] && v1[1] > v1[3] && v2[1] > v2[3]) return;
    if (v0[2] < 0.0f && v1[2] < 0.0f && v2[2] < 0.0f) return;
    if (v0[2] > v0[3] && v1[2] > v1[3] && v2[2] > v2[3]) return;

    const float hw = 0.5f * (float)width, hh = 0.5f * (float)height;
    const float* v[3] = {v0, v1, v2};
    int minx = 0, maxx = (int)width - 1, miny = (int)ry0, maxy = (int)ry1 - 1;

    int snapped = v0[3] > 0.0f && v1[3] > 0.0f && v2[3] > 0.0f;
    int64_t x[3] = {0, 0, 0}, y[3] = {0, 0, 0};
    float iw[3] = {0, 0, 0};
    if (snapped) {
        for (int i = 0; i < 3; i++) {
            iw[i] = 1.0f / v[i][3];
            const float fx = ((v[i][0] * iw[i] + 1.0f) * hw) * (float)SUBPIX;     /* screen x, y-down screen y, in 1/256 pixel */
            const float fy = ((1.0f - v[i][1] * iw[i]) * hh) * (float)SUBPIX;
            if (!(fabsf(fx) <= GUARD_BAND && fabsf(fy) <= GUARD_BAND)) { snapped = 0; break; }
            x[i] = (int64_t)rintf(fx); y[i] = (int64_t)rintf(fy);                     /* round to nearest even */
        }
    }
    if (snapped) {
        t->kind = 0;
        /* 2*area; y-down screen: negative <=> counter-clockwise in NDC <=> front facing (FrontFace::Ccw) */
        const int64_t A2 = (x[1] - x[0]) * (y[2] - y[0]) - (x[2] - x[0]) * (y[1] - y[0]);
        if (A2 == 0) return;
        if (cull_back && A2 > 0) return;
        const int64_t sgn = A2 < 0 ? -1 : 1;
        t->front = A2 < 0;
        for (int i = 0; i < 3; i++) {
            const int j = (i + 1) % 3, k = (i + 2) % 3;          /* weight of vertex i = edge j -> k */
            t->a[i] = sgn * (y[j] - y[k]);
            t->b[i] = sgn * (x[k] - x[j]);
            t->c[i] = sgn * (x[j] * y[k] - x[k] * y[j]);
        }
        const float inv_area = 1.0f / (float)(sgn * A2);
        for (int i = 0; i < 3; i++) { t->zq[i] = (v[i][2] * iw[i]) * inv_area; t->iw[i] = iw[i]; }
        /* pixels that can hold a sample strictly inside [min, max] of the snapped vertices */
        int64_t mnx = x[0] < x[1] ? x[0] : x[1], mxx = x[0] > x[1] ? x[0] : x[1], mny = y[0] < y[1] ? y[0] : y[1], mxy = y[0] > y[1] ? y[0] : y[1];
        if (x[2] < mnx) mnx = x[2];
        if (x[2] > mxx) mxx = x[2];
        if (y[2] < mny) mny = y[2];
        if (y[2] > mxy) mxy = y[2];
        const int64_t bx0 = mnx >> 8, bx1 = (mxx - 1) >> 8, by0 = mny >> 8, by1 = (mxy - 1) >> 8;      /* arithmetic shifts: floor */
        if (bx0 > minx) minx = (int)bx0;
        if (bx1 < maxx) maxx = (int)bx1;
        if (by0 > miny) miny = (int)by0;
        if (by1 < maxy) maxy = (int)by1;
    } else {
        t->kind = 1;
        float X0 = (v0[0] + v0[3]) * hw, Y0 = (v0[3] - v0[1]) * hh, w0 = v0[3];
        float X1 = (v1[0] + v1[3]) * hw, Y1 = (v1[3] - v1[1]) * hh, w1 = v1[3];
        float X2 = (v2[0] + v2[3]) * hw, Y2 = (v2[3] - v2[1]) * hh, w2 = v2[3];
        float a0 = Y1 * w2 - Y2 * w1, b0 = X2 * w1 - X1 * w2, c0 = X1 * Y2 - X2 * Y1;
        float a1 = Y2 * w0 - Y0 * w2, b1 = X0 * w2 - X2 * w0, c1 = X2 * Y0 - X0 * Y2;
        float a2 = Y0 * w1 - Y1 * w0, b2 = X1 * w0 - X0 * w1, c2 = X0 * Y1 - X1 * Y0;
        float det = (X0 * a0 + Y0 * b0) + w0 * c0;
        if (!(det != 0.0f) || !isfinite(det)) return;   /* zero area or NaN */
        if (cull_back && det > 0.0f) return;            /* det < 0 <=> front facing */
        t->front = det < 0.0f;
        if (det < 0.0f) {
            a0 = -a0; b0 = -b0; c0 = -c0; a1 = -a1; b1 = -b1; c1 = -c1; a2 = -a2; b2 = -b2; c2 = -c2;
            det = -det;
        }
        t->ha[0] = a0; t->hb[0] = b0; t->hc[0] = c0;
        t->ha[1] = a1; t->hb[1] = b1; t->hc[1] = c1;
        t->ha[2] = a2; t->hb[2] = b2; t->hc[2] = c2;
        const float inv_det = 1.0f / det;
        t->zq[0] = v0[2] * inv_det; t->zq[1] = v1[2] * inv_det; t->zq[2] = v2[2] * inv_det;
        t->iw[0] = 1.0f; t->iw[1] = 1.0f; t->iw[2] = 1.0f;
    }
    if (minx > maxx || miny > maxy) return;
    t->minx = minx; t->maxx = maxx; t->miny = miny; t->maxy = maxy;
    t->valid = 1;
}

/* WebGPU's standard 4x sample pattern (GPUMultisampleState count 4; == D3D standard pattern) in 1/256 pixel:
 * (0.375, 0.125) (0.875, 0.375) (0.125, 0.625) (0.625, 0.875); the pixel centre is (128, 128) */
const int oracle_msaa4_x[4] = {96, 224, 32, 160};
const int oracle_msaa4_y[4] = {32, 96, 160, 224};

/* Edge values at the sample (px, py) + (ox, oy)/256, as floats: exact-then-rounded for kind 0, f32-evaluated for kind 1.
 * Returns 1 if the sample is inside (top-left rule). */
static inline int tri_edges_sample(const TriSetup* t, int px, int py, int ox, int oy, float* e) {
    if (t->kind == 0) {
        const int64_t Px = (int64_t)px * SUBPIX + ox, Py = (int64_t)py * SUBPIX + oy;
        int inside = 1;
        for (int i = 0; i < 3; i++) {
            const int64_t E = t->a[i] * Px + t->b[i] * Py + t->c[i];
            if (!(E > 0 || (E == 0 && (t->a[i] > 0 || (t->a[i] == 0 && t->b[i] > 0))))) inside = 0;   /* top-left rule */
            e[i] = (float)E;
        }
        return inside;
    }
    const double X = (double)(px * SUBPIX + ox) * (1.0 / SUBPIX), Y = (double)(py * SUBPIX + oy) * (1.0 / SUBPIX);     /* exact */
    int inside = 1;
    for (int i = 0; i < 3; i++) {
        const double E = fma((double)t->ha[i], X, fma((double)t->hb[i], Y, (double)t->hc[i]));
        if (!(E > 0.0 || (E == 0.0 && (t->ha[i] > 0.0f || (t->ha[i] == 0.0f && t->hb[i] > 0.0f))))) inside = 0;
        e[i] = (float)E;
    }
    return inside;
}
/* returns 1 and the depth if the sample is covered and inside the depth clip range */
static inline int tri_sample(const TriSetup* t, int px, int py, int ox, int oy, float* depth_out) {
    float e[3];
    if (!tri_edges_sample(t, px, py, ox, oy, e)) return 0;
    float zn = (e[0] * t->zq[0] + e[1] * t->zq[1]) + e[2] * t->zq[2];
    if (!(zn >= 0.0f && zn <= 1.0f)) return 0;
    if (zn == 0.0f) zn = 0.0f;   /* canonicalise -0 so the bit pattern orders as an unsigned integer */
    *depth_out = zn;
    return 1;
}

/* One sample of a multisampled target (sample k of the standard 4x pattern).  Coverage: the edge values at the sample itself, exact, as above.
 * Depth: the depth plane's value at the pixel's CORNER — the edge values there, rounded and weighted exactly as a single sample's are — plus the
 * sample's increment along the plane's gradient, every operation rounded once in f32, nothing contracted:
 *     gx = (a0 zq0 + a1 zq1) + a2 zq2,  gy = (b0 zq0 + b1 zq1) + b2 zq2     a_i, b_i: what E_i gains per pixel in x / y, as f32
 *     dz_k = gx fx_k + gy fy_k                                              (fx_k, fy_k) = the sample's offset from the corner in pixels
 *     zn_k = zc + dz_k                                                      zc = (e0 zq0 + e1 zq1) + e2 zq2 at (px, py)
 * i.e. how a fixed-function unit steps a plane equation from a reference point (WebGPU leaves the evaluation to the implementation); until round 4
 * every sample evaluated the three-term form itself (three conversions, three products, two sums per sample for bits nobody can tell apart: both
 * forms are within 2 ulp of the plane).  The clip test 0 <= zn <= 1 stays per sample. */
static inline int tri_sample_msaa(const TriSetup* t, int px, int py, int k, float* depth_out) {
    float e[3], ec[3];
    if (!tri_edges_sample(t, px, py, oracle_msaa4_x[k], oracle_msaa4_y[k], e)) return 0;
    (void)tri_edges_sample(t, px, py, 0, 0, ec);
    float a[3], b[3];
    for (int i = 0; i < 3; i++) {
        if (t->kind == 0) { a[i] = (float)t->a[i] * (float)SUBPIX; b[i] = (float)t->b[i] * (float)SUBPIX; }      /* |a|, |b| <= 2^24: exact */
        else { a[i] = t->ha[i]; b[i] = t->hb[i]; }
    }
    const float zc = (ec[0] * t->zq[0] + ec[1] * t->zq[1]) + ec[2] * t->zq[2];
    const float gx = (a[0] * t->zq[0] + a[1] * t->zq[1]) + a[2] * t->zq[2];
    const float gy = (b[0] * t->zq[0] + b[1] * t->zq[1]) + b[2] * t->zq[2];
    const float fx = (float)oracle_msaa4_x[k] * (1.0f / (float)SUBPIX), fy = (float)oracle_msaa4_y[k] * (1.0f / (float)SUBPIX);      /* exact */
    const float dz = gx * fx + gy * fy;
    float zn = zc + dz;
    if (!(zn >= 0.0f && zn <= 1.0f)) return 0;
    if (zn == 0.0f) zn = 0.0f;
    *depth_out = zn;
    return 1;
}

/* the setup and the per-sample test, for the forward pass in oracle_shade.c */
void oracle_tri_setup(const float* v0, const float* v1, const float* v2, int cull_back, uint32_t width, uint32_t height,
                      uint32_t ry0, uint32_t ry1, TriSetup* t) { tri_setup(v0, v1, v2, cull_back, width, height, ry0, ry1, t); }
int oracle_tri_sample(const TriSetup* t, int px, int py, int ox, int oy, float* depth_out) { return tri_sample(t, px, py, ox, oy, depth_out); }
int oracle_tri_sample_msaa(const TriSetup* t, int px, int py, int k, float* depth_out) { return tri_sample_msaa(t, px, py, k, depth_out); }
/* perspective-correct barycentrics of a pixel centre from an existing setup (what oracle_tri_bary_at computes) */
void oracle_tri_bary(const TriSetup* t, int px, int py, float* b_out) {
    float e[3];
    (void)tri_edges_sample(t, px, py, 128, 128, e);
    e[0] *= t->iw[0]; e[1] *= t->iw[1]; e[2] *= t->iw[2];
    const float inv = 1.0f / ((e[0] + e[1]) + e[2]);
    b_out[0] = e[0] * inv; b_out[1] = e[1] * inv; b_out[2] = e[2] * inv;
}

static void shard_rows(const OracleScene* s, uint32_t* y0, uint32_t* y1) {
    *y0 = s->y0; *y1 = s->y1;
    if (*y1 == 0 || *y1 > s->height) { *y1 = s->height; }
    if (*y0 > *y1) *y0 = *y1;
}

int oracle_raster(const OracleScene* s, const float* clip, uint64_t* keys, int threads) {
    uint32_t W = s->width, H = s->height, sy0, sy1;
    shard_rows(s, &sy0, &sy1);
    const uint32_t S = s->msaa == 4u ? 4u : 1u;                      /* samples per pixel */
    float* depth = (float*)malloc((size_t)W * H * S * sizeof(float));
    uint32_t* rank_buf = (uint32_t*)malloc((size_t)W * H * S * sizeof(uint32_t));
    if (!depth || !rank_buf) { free(depth); free(rank_buf); return -2; }
    for (size_t i = 0; i < (size_t)W * H * S; i++) { depth[i] = 1.0f; rank_buf[i] = O_U32_MAX; }   /* render_pass.rs:107-114 */
    if (threads < 1) threads = 1;
    uint32_t rows = sy1 - sy0;
    /* bands of rows; every band walks all triangles in submission order (depth test is order dependent) */
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
    for (int band = 0; band < threads * 4; band++) {
        uint32_t by0 = sy0 + (uint32_t)(((uint64_t)rows * (uint64_t)band) / (uint64_t)(threads * 4));
        uint32_t by1 = sy0 + (uint32_t)(((uint64_t)rows * (uint64_t)(band + 1)) / (uint64_t)(threads * 4));
        if (by0 >= by1) continue;
        uint32_t rank = 0;
        for (uint32_t d = 0; d < s->n_draws; d++) {
            const AwsmDraw* dr = &s->draws[d];
            int cull_back = (dr->flags & AWSM_DRAW_CULL_BACK) != 0;
            const uint32_t copies = dr->inst_count ? dr->inst_count : 1u;
            for (uint32_t t = 0; t < dr->tri_count * copies; t++, rank++) {
                const float* v = clip + (size_t)rank * 12;
                TriSetup ts;
                /* setup uses the SHARD rect so that a shard's result equals the full frame's rows */
                tri_setup(v, v + 4, v + 8, cull_back, W, H, sy0, sy1, &ts);
                if (!ts.valid) continue;
                int y_lo = ts.miny < (int)by0 ? (int)by0 : ts.miny;
                int y_hi = ts.maxy > (int)by1 - 1 ? (int)by1 - 1 : ts.maxy;
                for (int py = y_lo; py <= y_hi; py++) {
                    for (int px = ts.minx; px <= ts.maxx; px++) {
                        float zn;
                        size_t p = (size_t)py * W + (size_t)px;
                        if (S == 1u) {
                            if (!tri_sample(&ts, px, py, 128, 128, &zn)) continue;
                            if (zn <= depth[p]) { depth[p] = zn; rank_buf[p] = rank; }   /* CompareFunction::LessEqual */
                        } else {   /* per-sample coverage + per-sample depth (multisampled depth/visibility targets) */
                            for (uint32_t k = 0; k < 4u; k++) {
                                if (!tri_sample_msaa(&ts, px, py, (int)k, &zn)) continue;
                                if (zn <= depth[p * 4 + k]) { depth[p * 4 + k] = zn; rank_buf[p * 4 + k] = rank; }
                            }
                        }
                    }
                }
            }
        }
    }
    for (size_t i = 0; i < (size_t)W * H * S; i++) {
        if (rank_buf[i] == O_U32_MAX) keys[i] = ~0ull;
        else keys[i] = ((uint64_t)o_f32_bits(depth[i]) << 32) | (uint64_t)(O_U32_MAX - rank_buf[i]);
    }
    free(depth); free(rank_buf);
    return 0;
}

/* rank -> draw via the prefix of triangle counts */
static uint32_t find_draw(const OracleScene* s, uint32_t rank, uint32_t* first_rank) {
    uint32_t acc = 0;   /* first_rank = rank of triangle 0 of the INSTANCE the rank falls in, so rank - first_rank is primitive-local */
    for (uint32_t d = 0; d < s->n_draws; d++) {
        const uint32_t tc = s->draws[d].tri_count, copies = s->draws[d].inst_count ? s->draws[d].inst_count : 1u;
        if (tc && rank < acc + tc * copies) { *first_rank = acc + ((rank - acc) / tc) * tc; return d; }
        acc += tc * copies;
    }
    *first_rank = acc;
    return s->n_draws;
}

int oracle_unpack_visibility(const OracleScene* s, const uint64_t* keys, uint32_t* tri_id, uint32_t* meta_off, float* depth) {
    size_t n = (size_t)s->width * s->height * (s->msaa == 4u ? 4u : 1u);
    for (size_t i = 0; i < n; i++) {
        if (keys[i] == ~0ull) {
            /* fragment.wgsl never ran: clear colour 0xFFFF per channel -> join32 = U32_MAX (render_pass.rs:22-30) */
            if (tri_id) tri_id[i] = O_U32_MAX;
            if (meta_off) meta_off[i] = O_U32_MAX;
            if (depth) depth[i] = 1.0f;
            continue;
        }
        uint32_t rank = O_U32_MAX - (uint32_t)(keys[i] & 0xFFFFFFFFull);
        uint32_t first;
        uint32_t d = find_draw(s, rank, &first);
        if (d >= s->n_draws) return -1;
        GeomMeta gm = load_geom_meta(s, s->draws[d].geom_meta_off);
        /* fragment.wgsl:27-35: triangle_index is the primitive-local id; m = material_mesh_meta_offset */
        if (tri_id) tri_id[i] = rank - first;
        if (meta_off) meta_off[i] = gm.material_meta_off;
        if (depth) depth[i] = o_bits_f32((uint32_t)(keys[i] >> 32));
    }
    return 0;
}

/* ---- shared with oracle_shade.c: recompute the winner's edge values at a pixel ---- */
/* Perspective-correct barycentrics of the centre of pixel (px, py) in the plane of the triangle (the pixel need not be
 * covered: MSAA extrapolates, and the derivative contract evaluates the quad neighbours).  One reciprocal, three products. */
int oracle_tri_bary_at(const float* v0, const float* v1, const float* v2, uint32_t width, uint32_t height,
                       int px, int py, float* b_out) {
    TriSetup ts;
    tri_setup(v0, v1, v2, 0, width, height, 0, height, &ts);
    if (!ts.valid) return 0;
    float e[3];
    (void)tri_edges_sample(&ts, px, py, 128, 128, e);
    e[0] *= ts.iw[0]; e[1] *= ts.iw[1]; e[2] *= ts.iw[2];   /* screen-space weights -> perspective-correct (kind 1: * 1) */
    const float inv = 1.0f / ((e[0] + e[1]) + e[2]);
    b_out[0] = e[0] * inv; b_out[1] = e[1] * inv; b_out[2] = e[2] * inv;
    return 1;
}

float oracle_det_atan2f(float y, float x) { return det_atan2f(y, x); }
uint16_t oracle_f32_to_f16(float f) { return o_f32_to_f16(f); }
float oracle_f16_to_f32(uint16_t h) { return o_f16_to_f32(h); }
