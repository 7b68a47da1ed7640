/*
 * awsm_host.h — flat C API of the C++ host layer (libawsm_host.so).
 *
 * The host layer is the MI355X build's counterpart of the reference's Rust scene state: it keeps the
 * key-based update API (TransformKey / MeshKey / MaterialKey ...) and the DynamicUniformBuffer /
 * DynamicStorageBuffer dirty-upload semantics, and drives the kernels ONLY through the C-ABI of
 * include/awsm_hip.h (loaded at run time from the library path given to awsm_host_create).
 * The reference has no C interface; each function names the Rust method it mirrors
 * (paths relative to /root/reference/crates/renderer/src/).  Rust is not available in this environment,
 * so the host is C++ and this header is how Python (ctypes) and the tests reach it.
 *
 * Keys are slotmap `KeyData::as_ffi()` values: (version << 32) | idx, never 0.  0 means "none"/"root".
 * All functions return 0 or a negative AwsmStatus (include/awsm_hip.h) unless stated otherwise.
 */
#ifndef AWSM_HOST_H
#define AWSM_HOST_H

#include <stddef.h>
#include <stdint.h>
#include "awsm_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* 2: AwsmHostMaterial carries struct_size (fields are only ever appended; a caller compiled against a shorter struct is read up to its
 *    size, the rest defaults to "block absent"); awsm_host_render passes AwsmFrameStats.struct_size through (awsm_hip.h). */
#define AWSM_HOST_ABI_VERSION 2u
uint32_t awsm_host_abi_version(void);

typedef struct AwsmHost AwsmHost;
typedef uint64_t AwsmKey;

/* AwsmRendererBuilder::build (lib.rs:213-259).  backend_path = shared library exporting the awsm_hip_* C-ABI
 * (libawsm_hip.so).  There is no built-in device fallback: a missing library or symbol fails here. */
int awsm_host_create(const char* backend_path, int device, void* stream, uint32_t cfg_flags, AwsmHost** out);
int awsm_host_destroy(AwsmHost* h);
const char* awsm_host_last_error(const AwsmHost* h);
void* awsm_host_device_ctx(AwsmHost* h);     /* the AwsmHipCtx*, for readback helpers */

/* ---- Transforms (transforms.rs:43-446) ---- */
AwsmKey awsm_host_transform_root(AwsmHost* h);
AwsmKey awsm_host_transform_insert(AwsmHost* h, const float translation[3], const float rotation_xyzw[4], const float scale[3], AwsmKey parent);
int awsm_host_transform_set_local(AwsmHost* h, AwsmKey key, const float translation[3], const float rotation_xyzw[4], const float scale[3]);
int awsm_host_transform_set_parent(AwsmHost* h, AwsmKey child, AwsmKey parent);
int awsm_host_transform_remove(AwsmHost* h, AwsmKey key);
AwsmKey awsm_host_transform_parent(AwsmHost* h, AwsmKey child);          /* 0 if none */
int awsm_host_transform_world(AwsmHost* h, AwsmKey key, float out_mat4[16]);

/* ---- Textures (textures.rs; renderer-core texture_pool): decoded RGBA8 images, one array per (w,h) ---- */
int awsm_host_texture_insert(AwsmHost* h, const uint8_t* rgba8, uint32_t width, uint32_t height);   /* returns texture id >= 0; mip kind albedo */
/* with the MipmapTextureKind the image's role implies (0 albedo, 1 normal, 2 metallic-roughness, 3 occlusion, 4 emissive, 5.. box) */
int awsm_host_texture_insert_kind(AwsmHost* h, const uint8_t* rgba8, uint32_t width, uint32_t height, uint32_t mipmap_kind);
int awsm_host_sampler_insert(AwsmHost* h, const AwsmSampler* sampler);                                /* returns sampler id >= 0 */
AwsmKey awsm_host_texture_transform_insert(AwsmHost* h, const float offset[2], const float origin[2], float rotation, const float scale[2]);

/* ---- Materials (materials.rs:60-241, materials/pbr.rs, materials/unlit.rs) ---- */
typedef struct AwsmHostTexRef {
    int32_t texture;          /* texture id, -1 = none */
    uint32_t sampler;         /* sampler id */
    uint32_t uv_index;
    uint32_t pad;
    AwsmKey transform;        /* texture-transform key, 0 = identity */
} AwsmHostTexRef;

typedef struct AwsmHostMaterial {
    uint32_t struct_size;     /* sizeof(AwsmHostMaterial) as the caller was compiled */
    uint32_t shader;          /* 1 = PBR, 2 = unlit (MaterialShaderId) */
    uint32_t double_sided;
    float base_color_factor[4];
    float metallic_factor, roughness_factor, normal_scale, occlusion_strength;
    float emissive_factor[3];
    uint32_t debug_bitmask;
    AwsmHostTexRef base_color_tex, metallic_roughness_tex, normal_tex, occlusion_tex, emissive_tex;
    /* optional features: has_* selects whether the block is written (pbr.rs:364-573) */
    uint32_t has_vertex_color, vertex_color_set;
    uint32_t has_emissive_strength; float emissive_strength;
    uint32_t has_ior; float ior;
    uint32_t has_specular; float specular_factor; float specular_color_factor[3]; AwsmHostTexRef specular_tex, specular_color_tex;
    uint32_t has_transmission; float transmission_factor; AwsmHostTexRef transmission_tex;
    uint32_t has_volume; float volume_thickness_factor, volume_attenuation_distance; float volume_attenuation_color[3]; AwsmHostTexRef volume_thickness_tex;
    uint32_t has_clearcoat; float clearcoat_factor, clearcoat_roughness_factor, clearcoat_normal_scale;
    AwsmHostTexRef clearcoat_tex, clearcoat_roughness_tex, clearcoat_normal_tex;
    uint32_t has_sheen; float sheen_roughness_factor; float sheen_color_factor[3]; AwsmHostTexRef sheen_roughness_tex, sheen_color_tex;
    /* MaterialAlphaMode (materials.rs:255-273): 0 Opaque, 1 Mask { cutoff }, 2 Blend.  Mask, Blend and any transmission route the
     * meshes that use the material to the transparent pass (pbr.rs:213-224, unlit.rs:36-38; decided when the mesh is inserted,
     * as the glTF loader does: gltf/buffers/mesh.rs:33-57). */
    uint32_t alpha_mode; float alpha_cutoff;
    /* the remaining optional blocks of the word stream (pbr.rs:418-447,529-573): the reference's shaders do not read them yet and its
     * glTF mapper leaves them unset (gltf/populate/material.rs:621-625), but a material built through the API carries them, and the
     * Materials mirror must hold the same bytes */
    uint32_t has_diffuse_transmission; float diffuse_transmission_factor; float diffuse_transmission_color_factor[3];
    AwsmHostTexRef diffuse_transmission_tex, diffuse_transmission_color_tex;
    uint32_t has_dispersion; float dispersion;
    uint32_t has_anisotropy; float anisotropy_strength, anisotropy_rotation; AwsmHostTexRef anisotropy_tex;
    uint32_t has_iridescence; float iridescence_factor, iridescence_ior, iridescence_thickness_min, iridescence_thickness_max;
    AwsmHostTexRef iridescence_tex, iridescence_thickness_tex;
} AwsmHostMaterial;

AwsmKey awsm_host_material_insert(AwsmHost* h, const AwsmHostMaterial* m);
int awsm_host_material_update(AwsmHost* h, AwsmKey key, const AwsmHostMaterial* m);   /* AwsmRenderer::update_material */
int64_t awsm_host_material_offset(AwsmHost* h, AwsmKey key);

/* ---- Skins / morphs (meshes/skins.rs:84-194, meshes/morphs.rs:121-217) ---- */
AwsmKey awsm_host_skin_insert(AwsmHost* h, const AwsmKey* joint_transforms, uint32_t n_joints, const float* inverse_bind_mat4s,
                              uint32_t set_count, const uint32_t* const* joints_per_set, const float* const* weights_per_set, uint32_t vertex_count);

/* ---- Meshes (meshes.rs:455-674; the gltf/buffers packers run inside) ---- */
typedef struct AwsmHostMorphTarget { const float* positions; const float* normals; const float* tangents; } AwsmHostMorphTarget;  /* each vertex_count*3 or NULL */
typedef struct AwsmHostPrimitive {
    uint32_t vertex_count, triangle_count;
    const float* positions;        /* vertex_count*3 */
    const float* normals;          /* vertex_count*3 */
    const float* tangents;         /* vertex_count*4 or NULL */
    const uint32_t* indices;       /* triangle_count*3 */
    uint32_t n_uv_sets; const float* uv_sets[8];        /* each vertex_count*2 */
    uint32_t n_color_sets; const float* color_sets[4];  /* each vertex_count*4 */
    uint32_t n_morph_targets; const AwsmHostMorphTarget* morph_targets;
    const float* morph_weights;            /* n_morph_targets (glTF mesh.weights) or NULL */
    const float* animated_morph_weights;   /* optional: written through update_morph_weights_with ([1..n+1)) */
    uint32_t front_face_cw;
} AwsmHostPrimitive;

AwsmKey awsm_host_mesh_insert(AwsmHost* h, const AwsmHostPrimitive* prim, AwsmKey transform, AwsmKey material, AwsmKey skin, uint32_t hidden);
/* Mesh.hud = true (meshes/mesh.rs:28; what the glTF loader's hints.hud sets): both geometries (gltf/buffers/mesh.rs:37-39), is_hud in the mesh's
 * MaterialMeshMeta, drawn by render() in the two HUD passes (render.rs:169-178,301-312) instead of the world's */
AwsmKey awsm_host_mesh_insert_hud(AwsmHost* h, const AwsmHostPrimitive* prim, AwsmKey transform, AwsmKey material, AwsmKey skin, uint32_t hidden);
int awsm_host_mesh_remove(AwsmHost* h, AwsmKey mesh);

/* ---- Lights (lights.rs:160-310) ---- */
typedef struct AwsmHostLight {
    uint32_t kind;            /* 1 directional, 2 point, 3 spot */
    float color[3]; float intensity;
    float position[3]; float range;
    float direction[3]; float inner_angle, outer_angle;
} AwsmHostLight;
AwsmKey awsm_host_light_insert(AwsmHost* h, const AwsmHostLight* l);
int awsm_host_light_remove(AwsmHost* h, AwsmKey key);
int awsm_host_set_ibl_mip_counts(AwsmHost* h, uint32_t prefiltered, uint32_t irradiance);

/* ---- Camera (camera.rs:17-28,111-227): column-major mat4s ---- */
int awsm_host_camera_update(AwsmHost* h, const float view[16], const float projection[16], const float position_world[3]);

/* ---- environment pass-through + targets ---- */
int awsm_host_env(AwsmHost* h, const AwsmEnv* env);
/* Skybox / Ibl::{prefiltered_env, irradiance} set to a texel cubemap (crates/renderer/src/environment.rs:79-140, lights/ibl.rs:13-96): RGBA16F
 * [mip][face][y][x][4], faces +X -X +Y -Y +Z -Z; NULL = back to the colour.  The mip counts the shader scales roughness by are the ones of
 * awsm_host_set_ibl_mip_counts (lights.rs:300-305). */
int awsm_host_env_cube(AwsmHost* h, AwsmCube which, uint32_t size, uint32_t mips, const uint16_t* texels_rgba16f);
int awsm_host_brdf_lut_generate(AwsmHost* h, uint32_t w, uint32_t height);
int awsm_host_resize(AwsmHost* h, uint32_t width, uint32_t height);
/* AwsmRenderer::set_anti_aliasing (anti_alias.rs:9-45): msaa_sample_count 0 (None) or 4 (recreates the render targets);
 * mipmap != 0 selects MipmapMode::Gradient in the opaque pass.  The reference's default is {Some(4), mipmap: true}. */
int awsm_host_set_anti_aliasing(AwsmHost* h, uint32_t msaa_sample_count, uint32_t mipmap);
int awsm_host_set_shard_rows(AwsmHost* h, uint32_t y0, uint32_t y1);
/* GPU instancing (meshes.rs:176-290, instances.rs): n transforms of 10 floats each (translation xyz, rotation xyzw, scale xyz);
 * the first call enables instancing for the mesh (enable_mesh_instancing), later calls replace the list (set_mesh_instances);
 * append returns the index of the first appended instance (append_mesh_instances). */
int awsm_host_mesh_set_instances(AwsmHost* h, AwsmKey mesh, const float* trs10, uint32_t n);
int awsm_host_mesh_append_instances(AwsmHost* h, AwsmKey mesh, const float* trs10, uint32_t n);
/* AwsmRenderer::pick (picker.rs:55-121): *hit = 1 and *mesh_key = the MeshKey (KeyData::as_ffi) under pixel (x, y) of the last frame, else *hit = 0 */
int awsm_host_pick(AwsmHost* h, int32_t x, int32_t y, uint32_t* hit, uint64_t* mesh_key);
int awsm_host_set_shard_bands(AwsmHost* h, uint32_t n, uint32_t r, uint32_t compact_output);   /* awsm_hip_set_shard_bands */
/* AwsmRendererLogging.render_timings (debug.rs:8-12; the spans of render.rs:150-320): per-stage times in the frame stats, on by default;
 * off = awsm_hip_set_stage_timers(ctx, 0), the ms_* fields of the stats read 0 and the frame loses its event bubbles */
int awsm_host_set_render_timings(AwsmHost* h, int enabled);

/* ---- frame: update_all (update.rs:8-18) + AwsmRenderer::render (render.rs:53-383, hot path only) ---- */
int awsm_host_update_transforms(AwsmHost* h);
/* sync != 0: ends with awsm_hip_frame_end (stats filled if non-NULL); sync == 0: enqueue only */
int awsm_host_render(AwsmHost* h, int sync, AwsmFrameStats* stats);
/* RenderHooks (crates/renderer/src/render.rs:54-63,181-190: pre_render / after_geometry_pass / ...): callbacks render() makes between
 * its passes, on the calling thread, after the pass has been enqueued.  A multi-GPU caller uses them for the exchanges the passes of a
 * sharded frame need (MSAA + bands: the halo keys after the geometry pass; a sharded transparent pass: the opaque image after the opaque
 * pass).  A hook returning non-zero aborts the frame with that status.  NULL removes a hook. */
typedef int (*AwsmHostHook)(void* user);
int awsm_host_set_render_hooks(AwsmHost* h, AwsmHostHook after_geometry_pass, void* user_geometry, AwsmHostHook after_opaque_pass, void* user_opaque);

/* ---- introspection (tests, parity, INTEGRATION) ---- */
int awsm_host_mirror(AwsmHost* h, AwsmBuf which, const uint8_t** data, size_t* len);
/* ---- glTF ingest (crates/renderer/src/gltf/{loader,buffers,populate}.rs): reads a .gltf (external / data-URI buffers and images)
 * or a .glb, decodes the images (PNG), converts every accessor, generates missing normals / tangents, and populates this host
 * through the key API above in the reference's order (transforms, skins, meshes; populate.rs:185-205).  scene_index < 0 = the
 * document's default scene.  The camera is not taken from the file.  On failure returns a negative AwsmStatus and, if err_out is
 * given, the reason (AWSM_ERR_UNSUPPORTED for arithmetic-coded / 12-bit / CMYK JPEG and KTX2 images, sparse accessors, point / line primitives, unknown required
 * extensions); objects inserted before the failure stay inserted. ---- */
typedef struct AwsmGltfInfo {
    uint32_t nodes, meshes, materials, images, samplers, skins, lights, triangles, generated_tangents, instanced_meshes, reserved[2];
} AwsmGltfInfo;
int awsm_host_load_gltf(AwsmHost* h, const char* path, int scene_index, AwsmGltfInfo* info_out, char* err_out, size_t err_cap);
/* the image decoders the reader uses (PNG: all colour types / bit depths, non-interlaced; JPEG: baseline / extended sequential Huffman,
 * 8-bit, grayscale or YCbCr) on their own: rgba_out = NULL queries the size; needs width * height * 4 bytes. */
int awsm_host_decode_image(const uint8_t* data, size_t len, uint8_t* rgba_out, size_t cap, uint32_t* width, uint32_t* height, char* err_out, size_t err_cap);

/* the world transparent pass's list (back to front), as awsm_host_draw_list gives the geometry pass's */
int awsm_host_transparent_draw_list(AwsmHost* h, AwsmDraw* out, uint32_t cap, uint32_t* n);
int awsm_host_draw_list(AwsmHost* h, AwsmDraw* out, uint32_t cap, uint32_t* n);   /* the list render() would submit */
/* the hud meshes (back to front) as the HUD geometry pass and the HUD transparent pass receive them: n entries in each array */
int awsm_host_hud_draw_lists(AwsmHost* h, AwsmDraw* geometry_out, AwsmDraw* transparent_out, uint32_t cap, uint32_t* n);
uint32_t awsm_host_texture_array_count(AwsmHost* h);
int awsm_host_texture_array_info(AwsmHost* h, uint32_t array_idx, uint32_t* width, uint32_t* height, uint32_t* layers, const uint8_t** texels);
uint64_t awsm_host_upload_bytes_last_frame(AwsmHost* h);

/* ---- raw allocators for the restated reference unit tests ---- */
typedef struct AwsmHostDub AwsmHostDub;   /* DynamicUniformBuffer */
typedef struct AwsmHostDsb AwsmHostDsb;   /* DynamicStorageBuffer */
AwsmHostDub* awsm_host_dub_new(size_t initial_capacity, size_t byte_size, size_t aligned_slice_size /*0 = byte_size*/, uint8_t zero);
void awsm_host_dub_free(AwsmHostDub* b);
int awsm_host_dub_update(AwsmHostDub* b, AwsmKey key, const uint8_t* data, size_t len);          /* -1 if oversized */
int awsm_host_dub_update_offset(AwsmHostDub* b, AwsmKey key, size_t offset, const uint8_t* data, size_t len);
int awsm_host_dub_remove(AwsmHostDub* b, AwsmKey key);                                             /* 1 removed, 0 absent */
int64_t awsm_host_dub_offset(AwsmHostDub* b, AwsmKey key);                                         /* -1 absent */
int64_t awsm_host_dub_slot(AwsmHostDub* b, AwsmKey key);
size_t awsm_host_dub_size(AwsmHostDub* b);
size_t awsm_host_dub_len(AwsmHostDub* b);
size_t awsm_host_dub_capacity(AwsmHostDub* b);
size_t awsm_host_dub_next_slot(AwsmHostDub* b);
size_t awsm_host_dub_free_slots(AwsmHostDub* b, size_t* out, size_t cap);                          /* returns count */
const uint8_t* awsm_host_dub_raw(AwsmHostDub* b);
int64_t awsm_host_dub_take_resize(AwsmHostDub* b);                                                 /* -1 = None */
size_t awsm_host_dub_take_dirty(AwsmHostDub* b, size_t* out_pairs, size_t cap_pairs);
void awsm_host_dub_force_state(AwsmHostDub* b, size_t next_slot);                                  /* free_slots.clear(); next_slot = n */

AwsmHostDsb* awsm_host_dsb_new(size_t initial_bytes, uint8_t zero);
void awsm_host_dsb_free(AwsmHostDsb* b);
size_t awsm_host_dsb_update(AwsmHostDsb* b, AwsmKey key, const uint8_t* data, size_t len);         /* returns offset */
int awsm_host_dsb_patch(AwsmHostDsb* b, AwsmKey key, size_t at, const uint8_t* data, size_t len);  /* update_with_unchecked; -1 = missing key */
void awsm_host_dsb_remove(AwsmHostDsb* b, AwsmKey key);
int64_t awsm_host_dsb_offset(AwsmHostDsb* b, AwsmKey key);
int64_t awsm_host_dsb_size_of(AwsmHostDsb* b, AwsmKey key);
size_t awsm_host_dsb_used_size(AwsmHostDsb* b);
size_t awsm_host_dsb_len(AwsmHostDsb* b);
size_t awsm_host_dsb_capacity(AwsmHostDsb* b);
size_t awsm_host_dsb_tree_root(AwsmHostDsb* b);
const uint8_t* awsm_host_dsb_raw(AwsmHostDsb* b);
int64_t awsm_host_dsb_take_resize(AwsmHostDsb* b);
size_t awsm_host_dsb_take_dirty(AwsmHostDsb* b, size_t* out_pairs, size_t cap_pairs);
size_t awsm_host_round_pow2(size_t n);
size_t awsm_host_index_to_offset(size_t idx, size_t leaves);
size_t awsm_host_offset_to_index(size_t off, size_t leaves);
/* write_buffer_with_dirty_ranges plan: pairs in, pairs out; returns number of output pairs */
size_t awsm_host_write_plan(size_t raw_len, const size_t* in_pairs, size_t n_in, size_t* out_pairs, size_t cap_pairs);
/* Frustum::from_view_projection(...).intersects_aabb (frustum.rs:42-89) */
int awsm_host_frustum_intersects(const float view_projection[16], const float aabb_min[3], const float aabb_max[3]);
/* Aabb::transformed (bounds.rs:38-61) */
void awsm_host_aabb_transformed(const float mat4[16], const float aabb_min[3], const float aabb_max[3], float out_min[3], float out_max[3]);

#ifdef __cplusplus
}
#endif
#endif /* AWSM_HOST_H */
