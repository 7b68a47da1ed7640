// frame_params.hpp — structs shared by the host context (awsm_hip.cpp) and the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/awsm_hip.h"

namespace awsm {

constexpr int kTile = 32;            // screen tile edge in pixels (one workgroup rasterises one tile out of LDS)
constexpr int kTileShift = 5;
constexpr int kMaxTexArrays = 64;
constexpr int kMaxSamplers = 32;

constexpr uint32_t kDrawInstanced = 0x80u;   // DrawDev.flags (internal; the API's AWSM_DRAW_* flags use the low bits)

// One draw as the kernels see it — an instanced API draw becomes one DrawDev per instance, in instance order, which is the
// order the hardware rasterises draw_indexed(.., instance_count) in (meshes/mesh.rs:115-121) — (AwsmDraw + prefix sums computed on the host in draw order).
struct DrawDev {
    uint32_t geom_meta_off;
    uint32_t vis_data_off;
    uint32_t tri_count;
    uint32_t flags;
    uint32_t first_tri;     // global triangle rank of this draw's triangle 0
    uint32_t first_block;   // first k_deform_transform block of this draw
    uint32_t inst_off;      // byte offset of this instance's mat4 in the instance-transform buffer (kDrawInstanced set)
    uint32_t pad1;
};

// Per-draw constants of the opaque pass, resolved once per frame by k_resolve_draws from
// geometry meta -> material mesh meta (material_mesh_meta.wgsl), so that k_shade reaches them with one dependent load
// instead of three.
struct DrawShadeDev {
    uint32_t first_tri;
    uint32_t material_word;       // material_offset / 4
    uint32_t attr_indices_word;   // attribute index offset / 4
    uint32_t attr_data_word;      // attribute data offset / 4
    uint32_t stride_words;        // attribute stride / 4
    uint32_t uv_sets_index;
    uint32_t flags;               // bit 0: hud mesh; bit 1: ALPHA_MODE_MASK material (the transparent pass may discard its fragments)
    uint32_t color_sets;          // COLOR_n sets in the mesh's vertex attributes (transparent pass)
};

// One of a material's five core textures (base colour, metallic-roughness, normal, occlusion, emissive; unlit: base, emissive),
// resolved once per frame and draw by k_resolve_draws: material words -> TextureInfo -> pool array + sampler + transform become
// one 64-byte record, so that the shading kernels reach a texel with one dependent load instead of four.
struct TexSlotDev {
    const uint32_t* base;   // level-0 texels of the layer (null when the texture is absent or names a missing array / sampler)
    uint32_t width, height;
    uint32_t flags;         // bit 0 exists, bit 1 fast path at level 0 (mag linear, repeat/repeat, power-of-two), bit 2 dangling reference (samples as
                            // zero), bit 3 repeat/repeat + power-of-two, bits 4/5/6 mag / min / mipmap filter linear, bits 13..14 / 21..22 address mode
                            // u / v, bits 24..31 uv set; slot 0 only: bits 8..12 exists mask of the five slots, bits 16..20 "uses TEXCOORD_0" mask
    float tt[6];            // texture transform m00 m01 m10 m11 bx by (the record's first six floats)
    uint32_t layer_levels;  // layer | mip levels << 24
    const uint32_t* level_off;   // the array's per-level texel offsets (TexArrayDev::level_off), MipmapMode::Gradient
    const uint32_t* array_base;  // the array's level chain
};
static_assert(sizeof(TexSlotDev) == 64, "TexSlotDev must be 64 bytes");
constexpr int kCoreTextures = 5;

// The factor half of a draw's material, gathered from the word stream (materials/pbr.rs:258-357, unlit.rs:72-105) into four
// aligned 16-byte loads.  ext_mask = 0 (no optional block: the common case) means the shading never touches the word stream.
struct DrawMatDev {
    float base_color[4];
    float metallic, roughness, normal_scale, occlusion_strength;
    float emissive[3];            // PBR: factor * emissive_strength; unlit: factor
    float ior;                    // 1.5 when the material has none
    uint32_t shader_alpha;        // shader id | alpha_mode << 8
    uint32_t debug_bitmask;
    uint32_t ext_mask;            // bit per optional block present: 0 vertex colour, 3 specular, 4 transmission, 6 volume, 7 clearcoat, 8 sheen (feature-index order)
    float alpha_cutoff;
};
static_assert(sizeof(DrawMatDev) == 64, "DrawMatDev must be 64 bytes");

// A draw the lean opaque kernel (k_shade_lean) can shade, in one record (instead of the 21 16-byte loads of DrawShadeDev +
// DrawMatDev + five TexSlotDev): a PBR material with no optional block and no debug view, not a hud mesh, every core texture it
// has on TEXCOORD_0 with the same texture transform, a repeat / repeat linear sampler and power-of-two extent.  k_resolve_draws decides
// (flags bit 0); draws that do not qualify keep the general route.
// The factors come READY FOR RAW TEXELS: where a texture exists its factor is pre-multiplied by 1/255, so the kernel multiplies the bilinear sum of
// the 0..255 texel values straight in — base colour, metallic / roughness, emissive: factor / 255; occlusion mix(1, r, s) = occlusion_bias + raw *
// occlusion_strength with bias = 1 - s, strength = s / 255 (no texture: bias 1); normal map (c * 2 - 1) * scale = raw * normal_scale - normal_bias with
// normal_scale = 2 * scale / 255, normal_bias = scale (z: raw * (2 / 255) - 1).
struct LeanDrawDev {
    uint32_t flags;               // bit 0: lean; bit 1: lean under MipmapMode::Gradient too (gtex valid); bit 2: ... with anisotropic probes too; bit 3: tt; bits 8..12: which of the five core textures exist
    float metallic, roughness, normal_scale;
    float base_color[3]; float occlusion_strength;
    float emissive[3];            // factor * emissive_strength
    float normal_bias;
    uint32_t tex[kCoreTextures][2];   // level-0 texels of the layer: address bits 0..31 | address bits 32..47, log2(width) << 16, log2(height) << 20
    float occlusion_bias;
    uint32_t pad1;
    // MipmapMode::Gradient (flags bit 1): the texture's pool array as the trilinear fetch needs it — square power-of-two layers, so the first texel of
    // level l is layers * (G(lw + 1) - G(lw + 1 - l)) with G(k) = (4^k - 1) / 3 = 0x55555555 & (4^k - 1), and level l of layer i starts i << 2 (lw - l)
    // texels further: {array address bits 0..31, bits 32..47 | levels << 16 | log2(width) << 24, layer, layers}
    uint32_t gtex[kCoreTextures][4];
    // flags bit 3: the draw's textures share ONE texture transform that is not the identity (KHR_texture_transform on every texture of a material, the
    // usual way to tile one): m00 m01 m10 m11 bx by, applied to the pixel's TEXCOORD_0 (and its derivatives) once, before all fetches
    float tt[6];
    uint32_t pad2[2];
};
static_assert(sizeof(LeanDrawDev) == 208, "LeanDrawDev must be 208 bytes");

constexpr int kMaxMipLevels = 16;
struct TexArrayDev {
    const uint8_t* texels;            // [level][layer][h_l][w_l] RGBA8, (w >> l).max(1); levels >= 1 valid after generate_mips
    uint32_t width, height, layers;
    uint32_t mips;                    // levels reserved (>= 1)
    uint32_t level_off[kMaxMipLevels];   // first texel of each level
};

// A cubemap with a mip chain (skybox / prefiltered environment / irradiance: opaque bind group 0, bindings 14-19), RGBA16F texels
// [level][face][y][x], faces in layer order +X -X +Y -Y +Z -Z.  texels == null: the uniform colour of DevScene (what
// AwsmRendererBuilder creates by default, lib.rs:176-207).
struct CubeDev {
    const uint2* texels;              // 4 halfs per texel
    uint32_t size, mips;
    uint32_t level_off[kMaxMipLevels];    // first texel of each level
    // The same chain with a one-texel apron around every face ((N + 2)^2 texels per face), filled from the faces across the edges by the seam rule of
    // sample_cube (k_cube_border): a bilinear footprint never leaves its face's array, so the lean kernels sample with two 16-byte loads per level and
    // no adjacency logic (lean::cube_sample).  Null until k_cube_border has run.
    const uint2* bordered;
    uint32_t b_level_off[kMaxMipLevels];
};
enum { kCubeSkybox = 0, kCubePrefiltered = 1, kCubeIrradiance = 2 };

struct DevScene {
    const uint8_t* buf[AWSM_BUF_COUNT];
    TexArrayDev tex[kMaxTexArrays];
    AwsmSampler samplers[kMaxSamplers];
    uint32_t n_tex, n_samplers;
    float skybox_rgba[4];
    float prefiltered_rgb[4];
    float irradiance_rgb[4];
    const uint16_t* lut_rg16f;
    uint32_t lut_w, lut_h;
    CubeDev cube[3];
};

struct TriRec;                                   // raster_setup.hpp (device side); 80 bytes
constexpr size_t kTriRecBytes = 80;
constexpr uint32_t kMaxDirtyRanges = 12;

struct FrameDev {
    uint32_t width, height;
    uint32_t y0, y1;              // rows the geometry pass rasterises: the shard's rows [sy0, sy1), plus one halo row on each side with MSAA
    uint32_t sy0, sy1;            // rows the opaque pass shades / owns
    uint32_t tiles_x, tiles_y;    // tiles covering the shard: rows [y0>>5, ceil(y1/32))
    uint32_t tile_row0;           // first tile row (32 px) of the shard
    uint32_t band_n, band_r;      // the shard owns the tile rows ty >= tile_row0 with ty % band_n == band_r (1, 0 = every row)
    uint32_t out_compact;         // band mode: output row = local band * 32 + (y & 31) instead of y
    uint32_t n_draws;
    uint32_t total_tris;          // ranks [0, total_tris) exist in the per-triangle arrays
    uint32_t total_verts;
    uint32_t rank0, block0;       // the binning kernels take the ranks [rank0, total_tris), k_deform_transform the blocks from block0 on: 0 except for the HUD geometry
                                  // pass of an MSAA frame, whose draws follow the world's in ONE rank space (awsm_hip.cpp: hud_geometry_merged)
    uint32_t bin_capacity;        // entries in the (triangle,tile) list
    uint32_t has_opaque;
    uint32_t mipmap;              // 0: MipmapMode::None (level 0 only), 1: MipmapMode::Gradient
    uint32_t aniso;               // MipmapMode::Gradient: 1 = the samplers' max_anisotropy counts (AWSM_CFG_ANISOTROPIC; grad_footprint, kernels_shade.hip)
    uint32_t msaa;                // 0: one sample per pixel (pixel centre); 4: vis holds [pixel][4 samples]
    const uint8_t* camera;        // the camera UBO this frame is shaded with (a per-frame snapshot in overlap mode)
    // The lean opaque kernel's view of that camera, composed on the host in f64 from the UBO as submitted (awsm_hip.cpp: compose_pixel_to_view):
    // view_h = pix2view * (pixel column, pixel row, depth, 1) is inv_proj * clip with the pixel -> NDC map folded in (standard.wgsl:17-27), column-major;
    // view_rot = the upper 3x3 of inv_view, column-major (world = view_rot * view + cam_pos; the direction to the camera is -view_rot * view: no
    // translation in it, so a far surface's view vector does not inherit the cancellation of two camera-sized terms).
    float pix2view[16];
    float view_rot[9];
    float cam_pos[3];             // CameraUniform.position (= inv_view's translation)
    float ortho_view_dir[3];      // orthographic camera (proj[3][3] > 0.9): the unit surface-to-camera direction, the same for every pixel
    uint32_t cam_ortho;
    const DrawDev* draws;
    DrawShadeDev* draw_shade;     // n_draws (k_resolve_draws, opaque pass)
    TexSlotDev* tex_slots;        // n_draws x kCoreTextures (k_resolve_draws)
    DrawMatDev* draw_mat;         // n_draws (k_resolve_draws)
    float4* lights_pre;           // 2 x float4 per light (k_resolve_draws): {unit direction to the light (directional) or unit spot axis, kind}, {colour * intensity, 0}
    uint32_t lights_cap;          // records lights_pre can hold
    // transformed vertices (k_deform_transform outputs)
    float4* clip;                 // total_verts
    float4* nrm;                  // total_verts  (world normal xyz, 0)
    float4* tan;                  // total_verts  (world tangent xyz, handedness)
    float4* wpos;                 // total_verts  (world position xyz, 1): transparent pass only, else null
    // Geometry cache (round 5): what of a draw's per-vertex / per-triangle outputs the camera does not touch — world position (wcache), world N / T (nrm, tan),
    // tri_shade, tri_info — stays valid in the frame slot's arrays from one frame of the slot to the next as long as the draw sits at the same place of the
    // draw list (same DrawDev, index and first_tri included: prev_draws is the list those arrays were computed for) and none of its inputs was written since
    // (dirty: the byte ranges awsm_hip_buffer_write / buffer_create received since that frame, by buffer).  Such a draw's blocks only form
    // clip = view_proj * wcache — the same operation on the same f32 values as the full path, so the same bits.  cache_on = 0: every block takes the full path.
    const uint32_t* block_draw;   // world geometry pass: the draw index of every k_deform_transform workgroup (null: the kernel searches the list)
    float4* wcache;               // total_verts  (world position as apply_vertex.wgsl forms it, model * (pos, 1)); geometry pass only, may be null
    const DrawDev* prev_draws;    // the draw list the slot's arrays were last computed for (== draws when the list did not change)
    uint32_t prev_n_draws;
    uint32_t cache_on;
    uint32_t cache_serial;        // frame serial of that computation: if its frame was dropped by a timed-out gate (poison) the arrays are older than prev_draws says
    uint32_t* cache_mark;         // null, or one word per workgroup of the launch: a workgroup that takes the cached path stores frame_serial there (AwsmFrameStats.geometry_cache_blocks)
    uint32_t n_dirty;             // ranges in dirty[] (<= kMaxDirtyRanges; more than that and the host turns the cache off for the frame)
    uint32_t dirty[12][3];        // {AwsmBuf, first byte, one past the last byte (saturating)}
    TriRec* tri_rec;              // total_tris   (k_bin<count> -> k_bin<fill>, k_raster_tile, k_shade)
    uint32_t* tri_info;           // total_tris   (draw index in bits 0..23, AWSM_DRAW_* flags of the owning draw in bits 24..30, bit 31: ALPHA_MODE_MASK draw — transparent pass)
    uint4* tri_shade;             // 2 x total_tris (geometry pass only, may be null: {info word, 0, TEXCOORD_0 of corner 0} {TEXCOORD_0 of corner 1, of corner 2}: what
                                  //               compute.wgsl:182-197 + texture_uvs.wgsl:64-84 fetch per pixel through meta -> indices -> attribute data, once per
                                  //               triangle; k_deform_transform)
    uint32_t attr_data_bytes;     // size of the attribute data buffer (k_deform_transform reads the corners' TEXCOORD_0 from it, bounds-checked)
    LeanDrawDev* draw_lean;       // n_draws (k_resolve_draws); null = the lean route is off for this frame
    uint32_t* shade_todo;         // [0] = count, [4 ..] = (block id << 2 | wavefront) of the 16x4-pixel groups k_shade_lean left to the general kernel
    uint32_t shade_todo_cap;
    uint32_t* lean_done_flag;     // overlapped pipeline: k_shade_todo's first workgroup stores lean_done_serial here as it starts (= k_shade_lean of this frame has
    uint32_t lean_done_serial;    // ended): the gate in front of the next frame's opaque pass (awsm_hip.cpp: wait_prev_pass); null = not used
    uint32_t* camera_snap;        // geometry pass, overlap mode: k_deform_transform copies camera_snap_words words of the camera UBO here (the camera the frame
    uint32_t camera_snap_words;   // is shaded with = the camera it was submitted with); null / 0 otherwise
    uint32_t* lean_next;          // 64 x 16 words: strip counter c of XCD x at [(x * kLeanCounters + c) * 16] (persistent k_shade_lean grid; zeroed by k_resolve_draws)
    uint32_t lean_grid;           // 0: one workgroup per 16x16 block; else the persistent grid's workgroup count (multiple of 8)
    // binning
    uint32_t* tile_count;         // n_tiles
    uint32_t* tile_offset;        // n_tiles + 1
    uint32_t* tile_cursor;        // n_tiles
    uint32_t* scan_tmp;           // k_bin_scan: per scan workgroup (256 tiles) kScanWords words of aggregates, then as many of bases (see there)
    uint32_t* tile_order;         // n_tiles: tile ids, heaviest first (k_bin_scan); then raster_extra_cap extra raster items
                                  // (tile | slice << 20, slice >= 1) for the tiles whose list is split over several workgroups (count in counters[7])
    uint32_t* tile_split;         // 2 * n_tiles: [2t] first scratch slot of a split tile (0xFFFFFFFF: not split), [2t+1] slices that finished
    unsigned long long* raster_scratch;   // raster_slot_cap tiles of keys (1024 * samples each): partial tiles of the split ones
    uint32_t raster_extra_cap, raster_slot_cap;
    uint32_t* bin_list;           // bin_capacity
    uint32_t* big_list;           // total_tris: ranks of the triangles covering > 16 tiles (count in counters[4])
#ifdef AWSM_STAMP
    unsigned long long* stamps;   // diagnostic builds only (tools/stamp_geometry.sh): [kernel][workgroup][8] s_memrealtime stamps
#endif
    const uint32_t* poison;       // overlapped pipeline, fail-closed hand-off: the serial of the last frame of this slot whose gate timed out (k_handoff_wait);
                                  // every kernel of that frame exits at once (frame_poisoned).  null: no device-side hand-off
    uint32_t* host_bin_status;    // pinned host memory, 2 words per frame slot: (triangle, tile) entries this frame needed, the frame's serial (k_bin_scan)
    uint32_t frame_serial;
    uint32_t* counters;           // [0] binned triangles, [1] bin entries, [2] overflow flag, [3] covered pixels, [4] big triangles, [7] extra raster items
    // targets
    unsigned long long* vis;      // width*height packed keys
    uint16_t* out_rgba16f;        // width*height*4
    float* out_rgba32f;           // optional parity tap (may be null)
    float4* msaa_color0;          // MSAA: width*height, f32 colour of sample 0 for the pixels in msaa_edges
    uint32_t* msaa_edges;         // MSAA: [0] = count, then pixel indices (y * width + x) whose four samples are resolved
    unsigned long long* msaa_edge_bits;   // MSAA, lean route: per 16x4-pixel strip (block * 4 + wavefront) TWO lane masks, written by k_shade_lean<.., MSAA> from the
                                  // pixel's four keys: [0] an edge whatever the neighbours show, [1] the neighbours decide (k_msaa_detect); a pixel in
                                  // neither is final.  null = the fused general kernel (k_shade_msaa)
    uint2* msaa_cells;            // MSAA, lean route: per pixel {octahedral normal of sample 0's G-buffer texel as two f16, depth bits; 0xFFFFFFFF: background} —
                                  // what the edge detector compares between neighbours (msaa.wgsl:42-112), left by the kernel that reconstructed it anyway
    // HUD passes (render.rs:169-178,301-312).  The HUD geometry pass rasterises the hud meshes with a depth buffer of its own (hud_depth, cleared):
    // hud_vis holds its keys (no hit = all ones), or is null when the frame has no hud geometry.  The opaque pass leaves a pixel a hud mesh covers
    // as cleared (compute.wgsl:176-179: is_hud -> return); the world's own keys and depth stay intact for the world transparent pass.
    const unsigned long long* hud_vis;
    const DrawDev* hud_draws;     // picker: the mesh under a hud-covered pixel
    const uint32_t* hud_tri_info;
    uint32_t hud_pass;            // transparent pass: 1 = MaterialTransparentRenderPass::render(.., is_hud = true): depth starts cleared (hud_depth), colours load the composite
    const unsigned long long* msaa_halo;   // MSAA + bands: sample-0 keys of the first / last row of EVERY rank's bands, [rank][band][2][width] (gathered)
    uint32_t halo_bands;          // bands per rank in that array
    const uint16_t* opaque_rgba16f;   // transparent pass: the opaque pass's image (blit source and transmission background); out_rgba16f/32f = composite
    // transparent pass fragment lists (k_forward_cover -> k_forward_shade -> k_forward_blend): counters[5] = fragments, counters[6] = overflow flag
    uint4* frag_rec;              // frag_cap records {triangle rank, pixel x | y << 16, next fragment of the pixel (0xFFFFFFFF = last), sample mask | resolved-mask flag << 8}
    float4* frag_color;           // frag_cap premultiplied colours (k_forward_shade)
    uint32_t* frag_first;         // width*height: the pixel's first fragment in submission order (0xFFFFFFFF = none)
    uint32_t frag_cap;
};

#ifdef __HIPCC__
// A hand-off gate that gave up (its signal never came within the time budget) has poisoned the frame it guarded: its kernels must not run on
// buffers that are half written, or still being read.  Wave-uniform, one scalar load.
__device__ __forceinline__ bool frame_poisoned(const FrameDev& f) {
    return f.poison != nullptr && *(const __attribute__((address_space(4))) uint32_t*)f.poison == f.frame_serial;
}
#endif

}  // namespace awsm
