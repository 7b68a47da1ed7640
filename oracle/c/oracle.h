/*
 * oracle.h — TEST INFRASTRUCTURE ONLY.  CPU restatement (plain C, scalar f32) of awsm-renderer's
 * Geometry Pass + Opaque Pass, used as the parity oracle for the HIP kernels and as the timed
 * "port" CPU baseline in bench.py.  Nothing in the product path may include, link or call this.
 *
 * PARITY STATUS: the reference holds no golden vectors for its rendering arithmetic (SURVEY.md §4,
 * §8c) and cannot be built or run here (Rust -> wasm32 + browser WebGPU; no cargo/rustc, no WGSL
 * runtime).  For everything in this directory's C code **parity is unpinned**: the only authority
 * is the WGSL/Rust source text each function cites.  (The buffer allocators and the frustum test,
 * which the reference's own unit tests do pin, are restated in oracle/host_mirror.py.)
 */
#ifndef ORACLE_H
#define ORACLE_H

#include <stddef.h>
#include <stdint.h>
#include "../../include/awsm_hip.h"   /* ABI structs/enums only: AwsmBuf, AwsmDraw, AwsmSampler */

#ifdef __cplusplus
extern "C" {
#endif

#define ORACLE_MAX_TEX_ARRAYS 64
#define ORACLE_MAX_SAMPLERS 32

typedef struct OracleTexArray {
    const uint8_t* texels;   /* RGBA8; mips == 1: [layer][h][w]; mips > 1: the whole chain, [level][layer][h_l][w_l] */
    uint32_t width, height, layers;
    uint32_t mips;           /* 0/1 = level 0 only */
} OracleTexArray;

/* texture_cube<f32> with a mip chain: RGBA16F [level][face][y][x][4], faces +X -X +Y -Y +Z -Z; texels == NULL: the uniform colour */
typedef struct OracleCube {
    const uint16_t* texels;
    uint32_t size, mips;
} OracleCube;

typedef struct OracleScene {
    const uint8_t* buf[AWSM_BUF_COUNT];
    uint64_t buf_size[AWSM_BUF_COUNT];
    uint32_t width, height;
    uint32_t y0, y1;                 /* shard rows; y1 == 0 -> full frame */
    const AwsmDraw* draws;
    uint32_t n_draws;
    uint32_t has_opaque;             /* 0 -> "empty" pipeline (skybox only) */
    uint32_t n_tex_arrays;
    OracleTexArray tex_arrays[ORACLE_MAX_TEX_ARRAYS];
    uint32_t n_samplers;
    AwsmSampler samplers[ORACLE_MAX_SAMPLERS];
    float skybox_rgba[4];
    float prefiltered_rgb[4];
    float irradiance_rgb[4];
    const uint16_t* brdf_lut_rg16f;  /* lut_w*lut_h*2 halfs */
    uint32_t lut_width, lut_height;
    uint32_t msaa;                   /* 0 (single sample) or 4: keys hold 4 samples per pixel, [pixel][sample] */
    uint32_t mipmap;                 /* 0 = MipmapMode::None, 1 = MipmapMode::Gradient */
    OracleCube cube[3];              /* 0 skybox, 1 prefiltered environment, 2 irradiance */
} OracleScene;

/* vert_main for every exploded vertex of every draw, in draw order.
 * clip_out: 4 floats / vertex; nt_out: 8 floats / vertex {N.xyz, 0, T.xyzw}. */
int oracle_transform(const OracleScene* s, float* clip_out, float* nt_out);

/* Rasterise all draws (clip_in from oracle_transform) into keys[width*height] (keys[width*height*4] when msaa == 4,
 * sample s of pixel p at [p*4 + s], standard 4x sample positions): (depth_bits << 32) | (0xFFFFFFFF - rank);
 * all ones = no hit. */
int oracle_raster(const OracleScene* s, const float* clip_in, uint64_t* keys_out, int threads);

/* Opaque compute pass over keys; writes rgba32f (4 floats / pixel) and rgba16f (4 halfs / pixel).
 * Either output may be NULL. */
int oracle_shade(const OracleScene* s, const float* clip_in, const float* nt_in, const uint64_t* keys,
                 float* rgba32f_out, uint16_t* rgba16f_out, int threads);

/* Test aid: 6 floats / pixel {packed normal-tangent RGBA16F as f32, barycentric RG16F as f32} of the single-sampled G-buffer (zeros = no hit). */
int oracle_gbuffer(const OracleScene* s, const float* clip_in, const float* nt_in, const uint64_t* keys, float* gbuf_out, int threads);

/* World transparent pass (render.rs:224-297; material_transparent/): `draws` is the back-to-front list, draw.vis_data_off = byte
 * offset of the mesh's 40-byte vertices in AWSM_BUF_TRANSPARENCY_GEOM_DATA.  oracle_forward_transform: vert_main per triangle
 * corner (clip 4, nt 8, wpos 4 floats / vertex).  oracle_forward: blit + forward pass + resolve -> the `composite` image. */
uint32_t oracle_forward_total_vertices(const AwsmDraw* draws, uint32_t n_draws);
int oracle_forward_transform(const OracleScene* s, const AwsmDraw* draws, uint32_t n_draws, float* clip_out, float* nt_out, float* wpos_out);
int oracle_forward(const OracleScene* s, const AwsmDraw* draws, uint32_t n_draws, const float* clip, const float* nt, const float* wpos,
                   const uint64_t* keys, const uint16_t* opaque16f, float* composite32f_out, uint16_t* composite16f_out, uint8_t* touched_out, int threads);

/* key -> reference visibility texel (primitive-local triangle id, material-mesh-meta byte offset) + depth */
int oracle_unpack_visibility(const OracleScene* s, const uint64_t* keys, uint32_t* tri_id_out,
                             uint32_t* meta_off_out, float* depth_out);

/* Mip chain of a texture pool array (renderer-core/src/texture/mipmap.rs:95-250): chain holds level 0 on entry
 * ([layer][h][w] RGBA8) and every level on return ([level][layer][h_l][w_l], (w >> l).max(1)); kinds[layer] is the
 * MipmapTextureKind (0 albedo, 1 normal, 2 metallic-roughness, others = box filter).  Returns the chain size in bytes. */
size_t oracle_mip_chain_bytes(uint32_t width, uint32_t height, uint32_t layers, uint32_t levels);
uint32_t oracle_mip_levels(uint32_t width, uint32_t height);
int oracle_generate_mips(uint32_t width, uint32_t height, uint32_t layers, const uint32_t* kinds, uint32_t levels, uint8_t* chain);
/* textureSampleGrad by the sampling contract (oracle_shade.c: sample_array_grad); anisotropic = 1: max_anisotropy counts (AWSM_CFG_ANISOTROPIC's twin) */
void oracle_set_anisotropic(int on);
void oracle_sample_grad(const OracleTexArray* arr, const AwsmSampler* smp, uint32_t layer, const float* uv, const float* ddx, const float* ddy, uint32_t n, int anisotropic, float* rgba_out);

/* BRDF LUT: rg16f_out[h*w*2].  Row j, column i == fragment (i+0.5, j+0.5) of the reference's
 * full-screen triangle (renderer-core/src/brdf_lut/shader.wgsl). */
int oracle_brdf_lut(uint32_t width, uint32_t height, uint16_t* rg16f_out, int threads);

uint32_t oracle_total_vertices(const OracleScene* s);

/* exposed for unit tests */
float oracle_det_atan2f(float y, float x);
uint16_t oracle_f32_to_f16(float f);
float oracle_f16_to_f32(uint16_t h);

#ifdef __cplusplus
}
#endif
#endif
