/*
 * awsm_hip.h — C-ABI drop-in boundary for awsm-renderer's per-frame hot path on MI355X (gfx950).
 *
 * The reference (dakom/awsm-renderer, Rust -> wasm32 -> browser WebGPU) has no FFI/plugin interface;
 * its device seam is the Rust type `AwsmRendererWebGpu` (crates/renderer-core/src/renderer.rs:36-41)
 * and the ~10 methods the hot path calls on it.  Every entry point below replaces one of those call
 * sites for the Geometry Pass + Opaque Pass only.  Reference paths are relative to
 * /root/reference/.
 *
 * Conventions
 *   - plain C: opaque context pointer, plain pointers + sizes, no C++/torch types.
 *   - every function returns 0 (AWSM_OK) or a negative AwsmStatus; never aborts.  The text of the
 *     last failure on a context is available from awsm_hip_last_error().
 *     (reference: Result<_, AwsmCoreError>, crates/renderer-core/src/error.rs)
 *   - thread-compatible, not thread-safe: one ctx <-> one host thread <-> one HIP device + stream
 *     (the reference is single-threaded wasm).
 *   - no pointer handed across the boundary is retained after the call returns, except the
 *     optional externally-owned output image (awsm_hip_bind_output).
 *   - all byte offsets/sizes given to awsm_hip_buffer_write must be 4-byte aligned
 *     (crates/renderer/src/buffer/dynamic_storage.rs:196-211).
 */
#ifndef AWSM_HIP_H
#define AWSM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: AwsmFrameStats carries struct_size (the caller sets it; awsm_hip_frame_end writes no byte beyond it) and handoff_gate_timeouts;
 *    awsm_hip_frame_trace / awsm_hip_read_frame_trace; a timed-out hand-off gate drops its frame (fail closed) and is reported by
 *    awsm_hip_geometry_pass / awsm_hip_frame_flush as well.  A library and a caller of different versions refuse each other in awsm_hip_create. */
#define AWSM_HIP_ABI_VERSION 2u

typedef struct AwsmHipCtx AwsmHipCtx;

typedef enum AwsmStatus {
    AWSM_OK = 0,
    AWSM_ERR_INVALID_ARGUMENT = -1,
    AWSM_ERR_OUT_OF_MEMORY = -2,
    AWSM_ERR_DEVICE = -3,          /* a HIP runtime call failed (text in last_error) */
    AWSM_ERR_NO_DEVICE = -4,       /* no gfx950 device visible */
    AWSM_ERR_NOT_READY = -5,       /* a required buffer / size / env was never provided */
    AWSM_ERR_UNSUPPORTED = -6,     /* a combination this library does not implement; the message names it (awsm_hip_last_error) */
    AWSM_ERR_OUT_OF_RANGE = -7     /* an offset/size points outside the destination buffer */
} AwsmStatus;

/* Device buffers, one per CPU mirror the reference uploads.
 * The record layout of each is exactly the reference's; file:line of the writer is given. */
typedef enum AwsmBuf {
    AWSM_BUF_TRANSFORMS = 0,        /* mat4 col-major, stride 64          crates/renderer/src/transforms.rs:68-72,396-410 */
    AWSM_BUF_NORMAL_MATS = 1,       /* mat3, stride 36 (uploaded, unread) crates/renderer/src/transforms.rs:412-422 */
    AWSM_BUF_MATERIALS = 2,         /* u32 word stream per material       crates/renderer/src/materials/pbr.rs:258-589 */
    AWSM_BUF_LIGHTS = 3,            /* 64 B / light, dense                crates/renderer/src/lights.rs:354-473 */
    AWSM_BUF_LIGHTS_INFO = 4,       /* 16 B                                crates/renderer/src/lights.rs:293-305 */
    AWSM_BUF_CAMERA = 5,            /* 512 B                               crates/renderer/src/camera.rs:72-87,169-219 */
    AWSM_BUF_SKIN_MATRICES = 6,     /* mat4 / joint                        crates/renderer/src/meshes/skins.rs:162-194 */
    AWSM_BUF_SKIN_INDEX_WEIGHTS = 7,/* {u32 joint,f32 weight}x4 / set / vertex  crates/renderer/src/gltf/buffers/skin.rs:22-113 */
    AWSM_BUF_MORPH_WEIGHTS = 8,     /* f32 / target (+1 quirk)             crates/renderer/src/meshes/morphs.rs:148-217 */
    AWSM_BUF_MORPH_VALUES = 9,      /* 10 f32 / target / vertex            crates/renderer/src/gltf/buffers/morph.rs:31-190 */
    AWSM_BUF_GEOM_META = 10,        /* 40 B in 256-B slots                 crates/renderer/src/meshes/meta/geometry_meta.rs:44-113 */
    AWSM_BUF_MATERIAL_META = 11,    /* 68 B in 256-B slots                 crates/renderer/src/meshes/meta/material_meta.rs:96-185 */
    AWSM_BUF_VIS_GEOM_DATA = 12,    /* 56 B / exploded vertex              crates/renderer/src/gltf/buffers/mesh/visibility.rs:35-165 */
    AWSM_BUF_VIS_GEOM_INDEX = 13,   /* identity u32 (accepted, unread: redundant for a SW rasteriser) crates/renderer/src/meshes.rs:514-520 */
    AWSM_BUF_ATTR_DATA = 14,        /* interleaved f32 custom attributes   crates/renderer/src/gltf/buffers/attributes.rs:113-160 */
    AWSM_BUF_ATTR_INDEX = 15,       /* 3 x u32 / triangle                  crates/renderer/src/meshes.rs:434-446 */
    AWSM_BUF_TEXTURE_TRANSFORMS = 16,/* 32 B records                       crates/renderer/src/textures.rs:247-284 */
    AWSM_BUF_INSTANCES = 17,        /* mat4 / instance                       crates/renderer/src/instances.rs:30-57 */
    AWSM_BUF_TRANSPARENCY_GEOM_DATA = 18, /* 40 B / original vertex {pos, normal, tangent}, drawn through AWSM_BUF_ATTR_INDEX
                                       crates/renderer/src/gltf/buffers/mesh/transparency.rs:31-175, meshes.rs:1116-1125 */
    AWSM_BUF_COUNT = 19
} AwsmBuf;

/* Replaces AwsmRendererBuilder::build() (crates/renderer/src/lib.rs:213-259) for the two passes. */
typedef struct AwsmConfig {
    uint32_t struct_size;   /* sizeof(AwsmConfig), for forward compatibility */
    uint32_t abi_version;   /* AWSM_HIP_ABI_VERSION */
    int32_t  device;        /* HIP device ordinal */
    uint32_t flags;         /* AWSM_CFG_* */
    void*    stream;        /* hipStream_t to run on; NULL = the library creates its own */
} AwsmConfig;
#define AWSM_CFG_PARITY_TAP 1u   /* also keep the shaded RGBA in f32 (readable via awsm_hip_read_opaque_f32) */
#define AWSM_CFG_OVERLAP_FRAMES 4u /* pipelining across frames: the opaque pass runs on an internal stream and overlaps the NEXT frame's
                                     geometry pass (per-frame device state is double-buffered; a frame is shaded with the camera it was
                                     submitted with; any other scene write waits for the opaque passes in flight).  Work the caller
                                     enqueues on its own stream after a frame must be preceded by awsm_hip_frame_flush(); awsm_hip_frame_end
                                     and the read-back calls wait for everything. */
#define AWSM_CFG_GENERAL_SHADE_ONLY 8u /* never take the lean opaque route (k_shade_lean): every pixel through the general kernel.  For A/B
                                         measurements and for tests that compare the two routes; results must agree within the shading tolerance. */
#define AWSM_CFG_ANISOTROPIC 16u /* MipmapMode::Gradient honours AwsmSampler.max_anisotropy (gltf samplers ask for 16,
                                  * gltf/populate/material.rs:892-902): up to 17 weighted trilinear probes along the footprint's major axis,
                                  * the level chosen for rho_max / N (the contract: DESIGN.md §2, grad_footprint in kernels_shade.hip).  Off by
                                  * default: textureSampleGrad's anisotropy is implementation-defined in WebGPU, and the default here is the
                                  * isotropic rule the reference itself documents (helpers/mipmap.wgsl:419-439).  Draws whose core textures
                                  * ask for anisotropy leave the lean opaque route under this flag. */
#define AWSM_CFG_SMALL_BIN_LIST 2u /* start with a 4096-entry (triangle, tile) list instead of sizing it from the triangle count:
                                     exercises the overflow -> grow -> replay path of awsm_hip_frame_end (tests) */

/* One geometry-pass draw == Mesh::push_geometry_pass_commands (crates/renderer/src/meshes/mesh.rs:70-126):
 * set_bind_group(2, meta, [geom_meta_off]); set_vertex_buffer(0, vis_data, vis_data_off);
 * draw_indexed(3*tri_count); cull mode from the pipeline key (mesh.rs:54-67).
 * Order of the array == the reference's sorted renderable order (crates/renderer/src/renderable.rs:38-150). */
typedef struct AwsmDraw {
    uint32_t geom_meta_off;   /* byte offset of the 256-B GeometryMeshMeta slot */
    uint32_t vis_data_off;    /* byte offset of the first exploded vertex in AWSM_BUF_VIS_GEOM_DATA (geometry pass), or of the mesh's
                                 first 40-byte vertex in AWSM_BUF_TRANSPARENCY_GEOM_DATA (transparent pass) */
    uint32_t tri_count;
    uint32_t flags;           /* AWSM_DRAW_* */
    uint32_t inst_off;        /* instanced mesh: byte offset of its first mat4 in AWSM_BUF_INSTANCES (meshes/mesh.rs:91-121) */
    uint32_t inst_count;      /* instanced mesh: instance count; 0 = not instanced */
} AwsmDraw;
#define AWSM_DRAW_CULL_BACK 1u   /* CullMode::Back (single-sided); 0 = CullMode::None */

/* MaterialOpaqueRenderPass::render (crates/renderer/src/render_passes/material_opaque/render_pass.rs:47-96). */
typedef struct AwsmOpaqueParams {
    uint32_t mipmap;        /* MipmapMode: 0 = None (textureSampleLevel 0), 1 = Gradient (textureSampleGrad from the barycentric derivatives; the
                               arrays must hold their mip chain: awsm_hip_texture_array_generate_mips) */
    uint32_t has_opaque;    /* 0 -> the "empty" pipeline: skybox only (render_pass.rs:64-71) */
} AwsmOpaqueParams;

typedef enum AwsmTexFormat { AWSM_TEX_RGBA8_UNORM = 0 } AwsmTexFormat;

/* GPUSamplerDescriptor subset used by the texture pool (crates/renderer/src/materials/writer.rs:53-63). */
typedef struct AwsmSampler {
    uint32_t address_mode_u;  /* 0 clamp-to-edge, 1 repeat, 2 mirror-repeat */
    uint32_t address_mode_v;
    uint32_t mag_filter;      /* 0 nearest, 1 linear (level-0 sampling uses the mag filter) */
    uint32_t min_filter;
    uint32_t mipmap_filter;
    uint32_t max_anisotropy;  /* 1..16; counts under MipmapMode::Gradient on a context created with AWSM_CFG_ANISOTROPIC, and only with three linear filters */
} AwsmSampler;

/* Environment: skybox + IBL cubes + BRDF LUT (opaque bind group 0, bindings 14-21:
 * crates/renderer/src/render_passes/material_opaque/shader/material_opaque_wgsl/bind_groups.wgsl:22-29).
 * The three colours are the uniform cubes AwsmRendererBuilder creates by default (crates/renderer/src/lib.rs:176-207);
 * texel cubemaps go through awsm_hip_env_cube_upload. */
typedef struct AwsmEnv {
    float skybox_rgba[4];
    float prefiltered_rgb[4];
    float irradiance_rgb[4];
    uint32_t brdf_lut_width, brdf_lut_height;
    const uint16_t* brdf_lut_rgba16f;  /* width*height*4 halfs, row 0 first; NULL = keep the current LUT */
} AwsmEnv;

typedef struct AwsmFrameStats {
    uint32_t struct_size; /* IN: sizeof(AwsmFrameStats) as the caller was compiled (fields are only ever appended); OUT: bytes written */
    float ms_transform;   /* k_deform_transform */
    float ms_bin;         /* k_bin_count + scan + k_bin_fill */
    float ms_raster;      /* k_raster_tile */
    float ms_shade;       /* k_shade */
    float ms_total;       /* first kernel start -> last kernel end */
    uint32_t triangles_in;      /* sum of tri_count over draws */
    uint32_t triangles_binned;  /* survived cull */
    uint32_t bin_entries;       /* (triangle, tile) pairs */
    uint32_t covered_pixels;    /* pixels with a hit (inside the shard rect) */
    uint32_t bin_overflow_retries;
    float ms_forward;           /* transparent pass: transform + binning + k_forward_tile */
    uint32_t forward_triangles; /* sum of tri_count (x instances) over the transparent draws */
    uint32_t forward_fragment_slots; /* fragment-list slots the transparent pass used (fragments + the unused tails of the wavefronts' chunks) */
    float ms_shade_lean;          /* k_shade_lean alone when the opaque pass took the lean route (then ms_shade = this + k_shade_todo), else 0 */
    uint32_t shade_general_wavefronts; /* 16x4-pixel groups of the last opaque pass that went through the general kernel instead of the lean one */
    uint32_t frames_with_dropped_bin_entries; /* enqueue-only frames (no frame_end) whose (triangle, tile) list overflowed since the context was created:
                                         such a frame lost geometry.  The library sizes the list from the need the GPU reports for earlier
                                         frames (growing it ahead of the need), so this stays 0 unless the need jumps by more than a third
                                         between two frames; awsm_hip_frame_end additionally replays an overflowed frame. */
    uint32_t handoff_gate_timeouts;  /* AWSM_CFG_OVERLAP_FRAMES with device-side hand-off: gates that ran out of time since the context was created.  Each one
                                         dropped the frame it guarded whole (its kernels exit at once: the image is not written, nothing is shaded from
                                         half-written buffers) and was reported once with AWSM_ERR_DEVICE. */
    uint32_t geometry_cache_blocks;  /* k_deform_transform workgroups (256 exploded vertices of one draw each) of this frame's geometry pass that kept the frame
                                         slot's cached world positions / normals / tangents / per-triangle words and only formed clip = view_proj * world,
                                         because their draw sits where it sat in the slot's previous frame and nothing it reads but the camera was written
                                         since (awsm_hip_buffer_write / buffer_create ranges).  Counted only while stage timers are on. */
    uint32_t geometry_blocks;        /* ... out of this many */
} AwsmFrameStats;

/* ---- lifecycle: AwsmRendererBuilder::build() / Drop (crates/renderer/src/meshes.rs:1349-1357) ---- */
int awsm_hip_create(const AwsmConfig* cfg, AwsmHipCtx** out);
int awsm_hip_destroy(AwsmHipCtx* ctx);
const char* awsm_hip_last_error(const AwsmHipCtx* ctx);
uint32_t awsm_hip_abi_version(void);

/* ---- gpu.create_buffer(desc{size,..}) (crates/renderer-core/src/methods.rs:239; callers e.g.
 * crates/renderer/src/meshes.rs:1313-1322): the new buffer replaces the old wholesale, contents are NOT
 * preserved — the host re-uploads the full mirror right after.  Idempotent for an unchanged size. ---- */
int awsm_hip_buffer_create(AwsmHipCtx* ctx, AwsmBuf which, size_t bytes);

/* ---- gpu.write_buffer(buf, Some(offset), &raw[off..off+len]) (crates/renderer-core/src/methods.rs:339-431,
 * via write_buffer_with_dirty_ranges, crates/renderer/src/buffer/helpers.rs:170-193).  Ordered before the
 * next pass on the ctx stream; `src` is copied to a pinned staging ring before the call returns. ---- */
int awsm_hip_buffer_write(AwsmHipCtx* ctx, AwsmBuf which, size_t dst_off, const void* src, size_t len);

/* ---- render_textures.views() realloc on size/AA change (crates/renderer/src/render_textures.rs:103-147).
 * msaa: 0 (or 1) = single sample; 4 = the reference's default AntiAliasing (anti_alias.rs:28-38): the geometry pass keeps
 * four visibility samples per pixel at WebGPU's standard 4x positions (per-sample coverage and depth; the interpolants
 * are evaluated at the pixel centre, as @interpolate(perspective, center) does), and the opaque pass runs the edge
 * detector + per-sample resolve of material_opaque_wgsl/helpers/{msaa,material_shading}.wgsl.  The output image stays
 * single-sampled.  Sharding with MSAA: row strips (awsm_hip_set_shard_rows; one halo row each side is rasterised for the
 * edge detector), or bands with the halo exchange below. ---- */
int awsm_hip_resize(AwsmHipCtx* ctx, uint32_t width, uint32_t height, uint32_t msaa);

/* ---- multi-GPU screen sharding (new; no reference counterpart): this ctx rasterises and shades only
 * pixel rows [y0, y1) (full width).  y0 == y1 == 0 restores the full frame.  Any row boundary is allowed: tiles that
 * straddle it are clipped per pixel, so a shard's rows are bit-identical to the same rows of the full frame. ---- */
int awsm_hip_set_shard_rows(AwsmHipCtx* ctx, uint32_t y0, uint32_t y1);

/* Interleaved sharding for load balance (no reference counterpart): this context rasterises and shades the
 * 32-row tile rows ty with ty % n == r (n = 1 restores the full frame; replaces any set_shard_rows range).  Work per
 * shard is then proportional to 1/n wherever the expensive part of the screen lies.  With compact_output != 0 the
 * opaque image is written densely: output row = (ty / n) * 32 + (y & 31), ceil((ceil(H/32) - r) / n) * 32 rows — the
 * layout an all-gather wants; otherwise rows keep their absolute position.  The visibility buffer is always
 * addressed by absolute row.  Rows owned by a shard are bit-identical to the same rows of the unsharded frame. */
int awsm_hip_set_shard_bands(AwsmHipCtx* ctx, uint32_t n, uint32_t r, uint32_t compact_output);
/* MSAA x4 with band sharding.  The edge detector (helpers/msaa.wgsl:42-112) compares a pixel with sample 0 of its four neighbours; for the
 * first and last row of a band the vertical neighbour lies in a band another rank rasterised.  So the frame gets ONE exchange step:
 * after the geometry pass every rank exports the sample-0 keys of the first and last row of each of its bands
 * (awsm_hip_msaa_halo_export: awsm_hip_msaa_halo_bands() x 2 x width u64, enqueued on the context's stream), the ranks all-gather those
 * arrays in rank order (RCCL), and every rank binds the gathered [n][bands][2][width] array before its opaque pass
 * (awsm_hip_msaa_halo_bind; kept by reference).  Every rank holds every triangle's setup record and vertex normals (the scene is
 * replicated, vertices are transformed redundantly), so a key is all a neighbour needs.  Without a bound array the opaque pass of an
 * MSAA band shard fails with AWSM_ERR_NOT_READY. */
uint32_t awsm_hip_msaa_halo_bands(AwsmHipCtx* ctx);     /* ceil(ceil(height / 32) / n): bands per rank in both arrays */
int awsm_hip_msaa_halo_export(AwsmHipCtx* ctx, void* dst_device, size_t bytes);
int awsm_hip_msaa_halo_bind(AwsmHipCtx* ctx, const void* gathered_device, size_t bytes);

/* ---- texture pool bind (crates/renderer/src/render_passes/material_opaque/bind_group.rs:331-360):
 * array `array_idx` is a texture_2d_array of `layers` w x h images; texels = layers*h*w*4 bytes, layer-major: mip
 * level 0 (what copyExternalImageToTexture writes, renderer-core/src/texture/texture_pool.rs:252-300).  `mips` = number
 * of levels the array holds (1, or up to floor(log2(max(w,h))) + 1): the space is reserved here and levels >= 1 are
 * produced by awsm_hip_texture_array_generate_mips. ---- */
int awsm_hip_texture_array_upload(AwsmHipCtx* ctx, uint32_t array_idx, uint32_t width, uint32_t height,
                                  uint32_t layers, uint32_t mips, AwsmTexFormat fmt, const void* texels);
/* generate_mipmaps (renderer-core/src/texture/mipmap.rs:95-330): every level from the previous one, 2x2 texels, filter
 * chosen per layer by MipmapTextureKind (0 albedo, 1 normal: renormalised, 2 metallic-roughness: roughness averaged as
 * r^2, 3 occlusion, 4 emissive, 5.. = box filter); kind_per_layer = `layers` values or NULL (all albedo).  Results are
 * stored as RGBA8 (floor(clamp(v,0,1)*255 + 0.5)). */
int awsm_hip_texture_array_generate_mips(AwsmHipCtx* ctx, uint32_t array_idx, const uint32_t* kind_per_layer);
/* texels of one mip level back to the host (tests): layers*h_l*w_l*4 bytes */
int awsm_hip_texture_array_read_level(AwsmHipCtx* ctx, uint32_t array_idx, uint32_t level, void* texels_out);
int awsm_hip_sampler_set(AwsmHipCtx* ctx, uint32_t sampler_idx, const AwsmSampler* sampler);
int awsm_hip_env_upload(AwsmHipCtx* ctx, const AwsmEnv* env);
/* Texel cubemaps for the same three bindings (skybox_tex, ibl_filtered_env_tex, ibl_irradiance_tex: bind_groups.wgsl:22-27; sampled by
 * helpers/skybox.wgsl:37 and shared_wgsl/lighting/brdf.wgsl:268-290 with the linear / linear / linear clamp samplers of lights/ibl.rs:38-47).
 * texels: RGBA16F, [mip][face][y][x][4 halfs], mip m has extent max(size >> m, 1), faces in layer order +X -X +Y -Y +Z -Z (what
 * CubemapImage::create_texture_and_view uploads; decoding KTX2 / EXR files is the caller's business).  mips = levels present (>= 1).
 * texels == NULL returns the binding to the uniform colour of awsm_hip_env_upload.  The prefiltered lookup's level is
 * roughness * (IblInfo.prefiltered_env_mip_count - 1) with the count the host wrote into AWSM_BUF_LIGHTS_INFO, clamped to the chain. */
typedef enum AwsmCube { AWSM_CUBE_SKYBOX = 0, AWSM_CUBE_PREFILTERED = 1, AWSM_CUBE_IRRADIANCE = 2 } AwsmCube;
int awsm_hip_env_cube_upload(AwsmHipCtx* ctx, AwsmCube which, uint32_t size, uint32_t mips, const uint16_t* texels_rgba16f);

/* ---- BrdfLut::new (crates/renderer-core/src/brdf_lut/generate.rs:47-96 + shader.wgsl): renders the
 * split-sum LUT on the device into the ctx's LUT slot (RGBA16F semantics, RG kept). ---- */
int awsm_hip_brdf_lut_generate(AwsmHipCtx* ctx, uint32_t width, uint32_t height);
int awsm_hip_read_brdf_lut(AwsmHipCtx* ctx, uint16_t* rg16f_out /* width*height*2 halfs */);

/* ---- GeometryRenderPass::render (crates/renderer/src/render_passes/geometry/render_pass.rs:51-157):
 * clear (vis = "no hit", depth = 1.0) + one draw per entry, depth LessEqual, later primitive wins ties. ---- */
int awsm_hip_geometry_pass(AwsmHipCtx* ctx, const AwsmDraw* draws, uint32_t n_draws);

/* ---- render_textures.clear_opaque() + MaterialOpaqueRenderPass::render
 * (crates/renderer/src/render.rs:209,219-221): one dispatch over the screen. ---- */
int awsm_hip_opaque_pass(AwsmHipCtx* ctx, const AwsmOpaqueParams* params);

/* ---- opaque -> transparent blit + MaterialTransparentRenderPass::render(.., is_hud = false) (+ the MSAA resolve into `composite`)
 * (crates/renderer/src/render.rs:224-297, render_passes/material_transparent/{render_pass,pipeline}.rs,
 * material_transparent_wgsl/{vertex,fragment}.wgsl).  `draws` = the transparent renderables in the reference's order (grouped by
 * pipeline, then back to front: renderable.rs:90,131-135); each mesh's 40-byte vertices (AWSM_BUF_TRANSPARENCY_GEOM_DATA at
 * vis_data_off) are drawn through its custom-attribute indices.  Forward shading per fragment with the premultiplied "over" blend
 * (One / OneMinusSrcAlpha), depth test LessEqual against the geometry pass's depth, depth write, screen-space transmission from
 * the opaque image.  Call after awsm_hip_opaque_pass of the same frame (it uses that call's mipmap mode and the context's MSAA
 * mode).  The result is the `composite` image (awsm_hip_read_composite / awsm_hip_bind_composite); the opaque image is
 * unchanged.  n_draws = 0 is valid (composite = opaque).
 * On a sharded context (row strip or bands) the pass covers, shades and blends this shard's rows only, but screen-space transmission
 * reads the WHOLE opaque image: gather the ranks' opaque rows first and hand the full [height][width] RGBA16F image in with
 * awsm_hip_bind_opaque_source (without it: AWSM_ERR_UNSUPPORTED).  The composite is addressed by absolute row (full-size target),
 * whatever layout the opaque output has.
 * Overflow of the pass's own lists (its (triangle, tile) list, the fragment slots) is detected and replayed by awsm_hip_frame_end only: a
 * pipelined loop that never calls frame_end must call it at least once after the transparent workload changes size (the lists then stay
 * sized for it), or check AwsmFrameStats.bin_overflow_retries from time to time — an overflowed enqueue-only frame drops fragments. ---- */
int awsm_hip_transparent_pass(AwsmHipCtx* ctx, const AwsmDraw* draws, uint32_t n_draws);
/* ---- the HUD passes (crates/renderer/src/render.rs:169-178,301-312; Mesh.hud, MaterialMeshMeta.is_hud):
 *   awsm_hip_hud_geometry_pass    GeometryRenderPass::render(ctx, &renderables.hud, true) — between awsm_hip_geometry_pass and awsm_hip_opaque_pass.  The
 *                                 hud meshes' visibility geometry (draws as in awsm_hip_geometry_pass) is rasterised over the visibility targets with a
 *                                 depth buffer of its own, cleared (geometry/render_pass.rs:51-157 with is_hud): they hide the world whatever its depth.
 *                                 The opaque pass then leaves every pixel a hud mesh covers cleared, (0, 0, 0, 0) (compute.wgsl:176-179), and
 *                                 awsm_hip_pick reports the hud mesh.  The world's keys and depth are not touched (the world transparent pass tests
 *                                 against them, as the reference's does against `depth`).
 *   awsm_hip_hud_transparent_pass MaterialTransparentRenderPass::render(ctx, renderables.hud, true) — after awsm_hip_transparent_pass (call that with
 *                                 n_draws = 0 when the frame has no world-transparent mesh).  The hud meshes' transparency geometry, back to front,
 *                                 forward-shaded and blended over the composite (colour LoadOp::Load), depth-tested against hud_depth, cleared
 *                                 (render.rs:490-521).  A hud mesh carries both geometries (gltf/buffers/mesh.rs:33-39).
 * With MSAA x4 (the reference's default AntiAliasing) the reference's quirk is reproduced, because its own opaque pass reads it: the HUD geometry pass
 * draws over the multisampled visibility / barycentric / normal targets but tests and writes hud_depth, so a covered sample shows the hud triangle and
 * still the WORLD's depth.  A pixel whose sample 0 is a hud triangle stays cleared; elsewhere the edge detector sees hud normals beside world depths
 * and msaa_resolve_samples shades hud samples like any other (compute.wgsl:176-180: "this may bleed a little").  Internally the hud draws then share the
 * world pass's rank space and the opaque pass reads merged keys (hud rank under world depth); awsm_hip_read_visibility still returns the world's.
 * Sharded contexts (row strips, bands) rasterise and shade their own rows of the hud meshes; the two transparent passes need the gathered opaque
 * image as usual (awsm_hip_bind_opaque_source).  MSAA with BAND sharding: the halo keys exported after this pass are the merged ones. ---- */
int awsm_hip_hud_geometry_pass(AwsmHipCtx* ctx, const AwsmDraw* draws, uint32_t n_draws);
int awsm_hip_hud_transparent_pass(AwsmHipCtx* ctx, const AwsmDraw* draws, uint32_t n_draws);
/* the full-frame opaque image the transparent pass of a sharded context blits from and refracts through (device memory, width*height*8
 * bytes, kept by reference until replaced; NULL = this context's own opaque output, which is complete only when unsharded) */
int awsm_hip_bind_opaque_source(AwsmHipCtx* ctx, const void* device_ptr, size_t bytes);

/* ---- gpu.submit_commands(encoder.finish()) (crates/renderer/src/render.rs:370): waits for the frame,
 * fills per-kernel times.  `out` may be NULL. ---- */
int awsm_hip_frame_end(AwsmHipCtx* ctx, AwsmFrameStats* out);

/* Stage timers: by default every frame records HIP events between its stages so that awsm_hip_frame_end can report
 * AwsmFrameStats.ms_*.  Each record costs a ~5 us bubble between two kernels; a render loop that does not read the stage times
 * turns them off (the ms_* fields then read 0; counters are unaffected).  No reference counterpart (the reference's timings are
 * tracing spans around command encoding, render.rs:150-320). */
int awsm_hip_set_stage_timers(AwsmHipCtx* ctx, int enabled);

/* Frame trace (measurement aid; the reference's counterpart are the tracing spans of render.rs:150-320): with capacity > 0 every frame
 * leaves three device-clock stamps in a ring of `capacity` frames — geometry pass begins, geometry pass done, shading (+ transparent pass)
 * done — written in stream order by one-lane kernels (the hand-off's own signal kernels where the pipeline has them: no extra launch on
 * the critical path).  capacity = 0 turns it off.  read_frame_trace synchronises and returns the last n_frames frames, oldest first, as
 * ticks_out[n_frames][3] in ticks of the constant-rate device clock (ticks_per_ms_out), plus the serial number of the newest frame. */
int awsm_hip_frame_trace(AwsmHipCtx* ctx, uint32_t capacity);
int awsm_hip_read_frame_trace(AwsmHipCtx* ctx, uint64_t* ticks_out, uint32_t n_frames, uint32_t* last_serial_out, uint32_t* ticks_per_ms_out);

/* ---- frame loop without a host sync (bench / multi-frame pipelines): enqueue only. ---- */
int awsm_hip_frame_flush(AwsmHipCtx* ctx);

/* ---- output image: RGBA16F, row-major, width*height*8 bytes == the reference's `opaque` render
 * texture (crates/renderer/src/render_textures.rs:49-54).  bind_output lets the caller own the memory
 * (e.g. a torch tensor that RCCL all-gathers); NULL returns to the internal image.  `bytes` must cover what the
 * current shard layout writes: width*height*8, or bands*32*width*8 with awsm_hip_set_shard_bands(compact_output);
 * checked when the opaque pass is enqueued. ---- */
int awsm_hip_bind_output(AwsmHipCtx* ctx, void* device_ptr, size_t bytes);
/* The same for a row-strip shard (awsm_hip_set_shard_rows): device_ptr is where frame row `first_row` goes and `bytes` covers the rows from
 * there on (width*8 each); the shard's rows must lie inside.  Lets a rank hand in its [rows, width] strip of an all-gather buffer. */
int awsm_hip_bind_output_rows(AwsmHipCtx* ctx, void* device_ptr, size_t bytes, uint32_t first_row);
/* the image the last submitted frame's opaque pass writes (the bound one, else the library's; with AWSM_CFG_OVERLAP_FRAMES the library keeps
 * one image per frame slot, so the pointer alternates from frame to frame) */
void* awsm_hip_output_device_ptr(AwsmHipCtx* ctx);

/* ---- readback for parity (new).  keys: width*height u64 (x4 with MSAA: the samples of a pixel are adjacent) =
 * (depth_f32_bits << 32) | (0xFFFFFFFF - rank),
 * rank = index of the triangle in draw order over the whole draw list; all ones = no hit.
 * unpack gives the reference's visibility_data texel: triangle_index (primitive-local) and
 * material_mesh_meta_offset, plus the Depth32Float value.
 * "Bit-exact" for these keys means: bit-exact to the arithmetic contract of DESIGN.md section 3, which this repository defines where WebGPU leaves the
 * implementation free — vertex snapping to 1/256 pixel, the top-left rule, the operation order of the depth interpolation, and the depth of an MSAA sample
 * (the plane's value at the pixel's corner plus the sample's step: a rule CHANGED IN ROUND 4 FOR SPEED, oracle first, kernel second; within 1.25 f32 steps
 * of the plane in f64).  The CPU oracle (oracle/) implements the same contract without shared code and the GPU tests compare every key.  What the
 * reference's own tests pin are the buffer allocators, the dirty-range writer and the frustum — not these keys. ---- */
int awsm_hip_read_visibility(AwsmHipCtx* ctx, uint64_t* keys_out);
/* 128-bit position-dependent digest of the keys the last geometry pass left (the caller's stream; synchronous):
 * out2[0] = sum key_i * (2 i + 1) mod 2^64, out2[1] = xor rotl(key_i, i mod 64).  For tests that compare many frames
 * without reading 8 bytes per pixel back. */
int awsm_hip_visibility_digest(AwsmHipCtx* ctx, uint64_t* out2);
/* AWSM_CFG_OVERLAP_FRAMES: how the context orders its streams at the two hand-offs on a frame's critical path (geometry pass -> opaque pass of
 * the same frame; opaque pass -> the geometry pass that reuses its frame slot).  1 = device-side flags (a one-lane kernel at the end of the
 * producer stream stores the frame's serial number, a one-lane kernel at the head of the consumer stream polls it: ~2 us instead of the
 * 20-30 us a cross-stream hipEvent takes to release the waiting queue), 0 = hipEvents (contexts without overlap; AWSM_DEVICE_HANDOFF=0 in
 * the environment; the probe at create found that kernels of two streams do not run side by side, e.g. under a counter-collecting
 * profiler; or a gate timed out later, which awsm_hip_frame_end reports once with AWSM_ERR_DEVICE).  Negative = AWSM_ERR_*.
 * A gate that runs out of time fails closed: the frame it guarded is dropped whole (every kernel of it exits at its first instruction, its image
 * is not written), the streams go on, the context falls back to events, and the next awsm_hip_geometry_pass / awsm_hip_frame_flush /
 * awsm_hip_frame_end returns AWSM_ERR_DEVICE once (AwsmFrameStats.handoff_gate_timeouts counts them).
 * Environment, read at create: AWSM_DEVICE_HANDOFF=0 (events from the start), AWSM_HANDOFF_TIMEOUT_MS=x (a gate's time budget on the device clock,
 * default 4000), AWSM_TEST_HANDOFF_DROP=n (tests: withhold the first n geometry-done signals to exercise the timeout path). */
int awsm_hip_stream_handoff(AwsmHipCtx* ctx);
/* test aid: the G-buffer texel fs_main would have written for every pixel of the last geometry pass (fragment.wgsl:23-54) as the opaque pass
 * reconstructs it — 6 floats / pixel: normal_tangent RGBA16F and barycentric RG16F, each already rounded to f16; zeros where nothing was hit.
 * Single-sampled frames, no band sharding. */
int awsm_hip_read_gbuffer(AwsmHipCtx* ctx, float* out6);
int awsm_hip_read_visibility_unpacked(AwsmHipCtx* ctx, uint32_t* tri_id_out, uint32_t* meta_off_out, float* depth_out);
int awsm_hip_read_opaque(AwsmHipCtx* ctx, uint16_t* rgba16f_out);
/* the image after the transparent pass == the reference's `composite` render texture (render_textures.rs:49-54; what the
 * display pass tone-maps): RGBA16F, width*height*8 bytes.  bind_composite lets the caller own the memory (NULL = internal). */
int awsm_hip_read_composite(AwsmHipCtx* ctx, uint16_t* rgba16f_out);
int awsm_hip_read_composite_f32(AwsmHipCtx* ctx, float* rgba32f_out);  /* needs AWSM_CFG_PARITY_TAP: the same f16 values, widened */
int awsm_hip_bind_composite(AwsmHipCtx* ctx, void* device_ptr, size_t bytes);
/* transformed vertices of the last transparent pass, one per triangle corner in draw order: clip xyzw (16 B),
 * {world N xyz, pad, world T xyzw} (32 B), world position xyz1 (16 B) == vert_main outputs (material_transparent_wgsl/vertex.wgsl) */
int awsm_hip_read_transformed_forward(AwsmHipCtx* ctx, float* clip_out, float* normal_tangent_out, float* world_pos_out, uint32_t max_vertices);
int awsm_hip_read_opaque_f32(AwsmHipCtx* ctx, float* rgba32f_out);  /* needs AWSM_CFG_PARITY_TAP */
/* ---- picking (crates/renderer/src/picker.rs:55-121 + picker/shader/picker_wgsl/compute.wgsl): the mesh under pixel
 * (x, y) of the last geometry pass, read from the visibility buffer: key -> draw -> geometry meta -> material mesh meta
 * -> mesh key words.  valid = 0 for background and for coordinates outside the frame (or outside this shard).
 * triangle_index is the primitive-local triangle (the visibility texel's first component).  Synchronous. ---- */
typedef struct AwsmPick {
    uint32_t valid;
    uint32_t mesh_key_high, mesh_key_low;   /* slotmap KeyData::as_ffi >> 32, & 0xFFFFFFFF */
    uint32_t triangle_index;
} AwsmPick;
int awsm_hip_pick(AwsmHipCtx* ctx, int32_t x, int32_t y, AwsmPick* out);

/* transformed vertices of the last geometry pass: per exploded vertex clip xyzw (16 B) and
 * {world N xyz, pad, world T xyzw} (32 B) == vert_main outputs (geometry_wgsl/vertex.wgsl:36-63). */
int awsm_hip_read_transformed(AwsmHipCtx* ctx, float* clip_out, float* normal_tangent_out, uint32_t max_vertices);

/* ---- device information for the measurement harness ---- */
int awsm_hip_device_info(AwsmHipCtx* ctx, char* name_out, size_t name_cap, uint32_t* cu_count, uint64_t* hbm_bytes);

#ifdef __cplusplus
}
#endif
#endif /* AWSM_HIP_H */
