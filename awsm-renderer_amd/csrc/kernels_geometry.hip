// kernels_geometry.hip — Geometry Pass on gfx950: deform/transform, tile binning, LDS tile rasteriser.
//
// Replaces (paths relative to /root/reference/crates/renderer/src/):
//   render_passes/geometry/shader/geometry_wgsl/vertex.wgsl:36-63     vert_main        -> k_deform_transform
//   render_passes/shared/shared_wgsl/vertex/apply_vertex.wgsl:24-118  apply_vertex
//   render_passes/shared/shared_wgsl/vertex/morph.wgsl:4-168, skin.wgsl:7-157
//   (fixed-function raster, render_passes/geometry/pipeline.rs:337-344)               -> k_bin + k_raster_tile
//   render_passes/geometry/render_pass.rs:51-157  (clears + one draw per renderable, in order)
//
// Design (MI355X-first): the reference issues one draw per mesh; here ALL draws of a frame go through
// three launches.  Visibility is one packed 64-bit key per pixel, (depth_bits << 32) | ~rank, resolved with
// ds_min_u64 inside a 32x32 LDS tile that exactly one workgroup owns, so no global atomics touch the
// image and the tile is written to HBM once, coalesced.
#include "frame_params.hpp"
#include "raster_setup.hpp"

// The geometry chain's wavefronts take issue priority over the opaque pass's on a SIMD they share (s_setprio; round 5).  The overlapped frame is as long as
// its longer stream; with MSAA x4 that is the geometry stream (k_raster_tile<4> runs 380 us beside the lean kernel, 172 alone: profiles/r05_b_timeline_msaa_mips.txt),
// and whatever its wavefronts do not issue the lean kernel's take anyway.  Measured, same box: MSAA x4 + mips 1,923 -> 1,943 frames/s, single-sampled
// unchanged (3,418 / 3,419); the reverse — the lean kernel at priority — costs 1.3 % and 4 % (profiles/r05_wave_priority.txt).  0 = off.
#ifndef AWSM_GEOM_PRIO
#define AWSM_GEOM_PRIO 3
#endif

namespace awsm {

// Diagnostic builds (-DAWSM_STAMP, tools/stamp_geometry.sh): where a geometry kernel's workgroups spend their time.  Stamps are the
// 100 MHz s_memrealtime counter (one time base for the whole chip), written by thread 0 to a buffer nothing else reads.
#ifdef AWSM_STAMP
#define AWSM_STAMP_AT(f, kernel, slot) do { if (threadIdx.x == 0 && (f).stamps && blockIdx.x < 16384u) (f).stamps[((size_t)(kernel) * 16384u + blockIdx.x) * 8u + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define AWSM_STAMP_AT(f, kernel, slot) do { } while (0)
#endif

struct GeomMetaDev {
    uint32_t mesh_key_high, mesh_key_low;
    uint32_t morph_len, morph_weights_off, morph_values_off;
    uint32_t skin_sets, skin_matrices_off, skin_index_weights_off;
    uint32_t transform_off, material_meta_off;
};

// Geometry cache: is draw `d` — at the same place of the list as `p`, the draw the slot's arrays were last computed for — also unchanged in everything
// k_deform_transform reads for it except the camera?  Wave-uniform (one draw per workgroup): scalar loads and branches.  Conservative where a block's extent
// is not known here (a skin's joint count, a morph's value block): any write at or behind the block's first byte counts.  Buffers whose every write
// invalidates every draw (attribute data / indices, morph values, skin index-weights) are the host's business: it turns the cache off for the frame.
AWSM_DI bool draw_unchanged(const DevScene* __restrict__ sc, const FrameDev& f, const DrawDev& d, const DrawDev& p, const uint32_t* __restrict__ gmp) {
    if (d.geom_meta_off != p.geom_meta_off || d.vis_data_off != p.vis_data_off || d.tri_count != p.tri_count || d.flags != p.flags || d.first_tri != p.first_tri ||
        d.first_block != p.first_block || d.inst_off != p.inst_off) return false;
    if (f.poison != nullptr && *(const __attribute__((address_space(4))) uint32_t*)f.poison == f.cache_serial) return false;   // that frame never ran
    if (f.n_dirty == 0u) return true;
    auto hit = [&](uint32_t buf, uint32_t lo, uint32_t hi) {      // does a dirty range of `buf` overlap [lo, hi)?
        for (uint32_t i = 0; i < f.n_dirty; i++) if (f.dirty[i][0] == buf && f.dirty[i][1] < hi && lo < f.dirty[i][2]) return true;
        return false;
    };
    if (hit(AWSM_BUF_GEOM_META, d.geom_meta_off, d.geom_meta_off + 40u)) return false;
    if (hit(AWSM_BUF_VIS_GEOM_DATA, d.vis_data_off, (uint32_t)min((unsigned long long)d.vis_data_off + 168ull * d.tri_count, 0xFFFFFFFFull))) return false;
    if ((d.flags & kDrawInstanced) && hit(AWSM_BUF_INSTANCES, d.inst_off, d.inst_off + 64u)) return false;
    const uint32_t toff = (gmp[8] / 64u) * 64u, mmoff = (gmp[9] / 256u) * 256u;
    if (hit(AWSM_BUF_TRANSFORMS, toff, toff + 64u)) return false;
    if (hit(AWSM_BUF_MATERIAL_META, mmoff, mmoff + 68u)) return false;
    if (gmp[2] != 0u && hit(AWSM_BUF_MORPH_WEIGHTS, gmp[3], gmp[3] + 4u + 4u * gmp[2])) return false;
    if (gmp[5] != 0u && hit(AWSM_BUF_SKIN_MATRICES, gmp[6], 0xFFFFFFFFu)) return false;
    return true;
}

// ------------------------------------------------------------------------------------------------
// k_deform_transform: one thread per exploded vertex, 256 vertices per workgroup.  The 56-byte vertex
// records of a workgroup are one contiguous 14 KB run: staged through LDS with 16-byte coalesced loads.
// ------------------------------------------------------------------------------------------------
// FWD = the transparent pass's vert_main (material_transparent_wgsl/vertex.wgsl:40-72): an indexed draw of the mesh's 40-byte
// vertices (AWSM_BUF_TRANSPARENCY_GEOM_DATA at draw.vis_data_off) through the custom-attribute index buffer
// (meshes/mesh.rs:129-200); one thread per triangle corner, which also leaves the world position for the fragment stage.
#ifndef AWSM_TRANSFORM_WAVES
#define AWSM_TRANSFORM_WAVES 6
#endif
template <bool FWD>
// FWD = false: capped at 80 VGPRs (122 uncapped; the spills sit in the skinning branch) so that two wavefronts per SIMD fit next to the
// previous frame's persistent opaque-pass grid — with one, the kernel ran 209 us beside it (31 alone).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(FWD ? 4 : AWSM_TRANSFORM_WAVES))) void k_deform_transform(const DevScene* __restrict__ sc, FrameDev f) {
    __shared__ __attribute__((aligned(16))) uint32_t lds_vtx[FWD ? 4 : 256 * 14];
    if (frame_poisoned(f)) return;
#if AWSM_GEOM_PRIO
    __builtin_amdgcn_s_setprio(AWSM_GEOM_PRIO);
#endif
    const uint32_t tid = threadIdx.x;
    const uint32_t b = blockIdx.x + f.block0;
    {   // per-frame clears folded into the first kernel of the frame (saves three memset launches per frame)
        const uint32_t n_tiles = f.tiles_x * f.tiles_y, gsz = gridDim.x * 256u, g0 = blockIdx.x * 256u + tid;
        for (uint32_t i = g0; i < n_tiles; i += gsz) f.tile_count[i] = 0u;
        if (g0 < 8u) f.counters[g0] = 0u;
        if (g0 == 12u || g0 == 13u) f.counters[g0] = 0u;      // k_bin_scan's arrival counter and ready flag
        // the opaque pass's lean route (same slot, after this frame's raster): its list for the general kernel and the persistent grid's strip
        // counters start empty — here and not in k_resolve_draws, which is skipped when nothing but the camera changed
        if (!FWD && f.shade_todo && g0 == 14u) f.shade_todo[0] = 0u;
        if (!FWD && f.lean_next && g0 >= 64u && g0 < 128u) f.lean_next[(g0 - 64u) * 16u] = 0u;
        if (!FWD && g0 < f.camera_snap_words) f.camera_snap[g0] = reinterpret_cast<const uint32_t*>(sc->buf[AWSM_BUF_CAMERA])[g0];   // overlap mode: the frame's camera
    }
    uint32_t lo = 0, hi = f.n_draws;
    if (!FWD && f.block_draw) lo = f.block_draw[blockIdx.x];      // (block0 = 0 whenever the map is given)
    else while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (f.draws[mid].first_block <= b) lo = mid; else hi = mid;
    }
    const DrawDev d = f.draws[lo];
    const uint32_t nverts = 3u * d.tri_count;
    const uint32_t local0 = (b - d.first_block) * 256u;
    const uint32_t count = min(256u, nverts - local0);

    const uint32_t* gmp = reinterpret_cast<const uint32_t*>(sc->buf[AWSM_BUF_GEOM_META] + d.geom_meta_off);
    if (!FWD && f.cache_on && lo < f.prev_n_draws && draw_unchanged(sc, f, d, f.prev_draws[lo], gmp)) {
        // Geometry cache hit (frame_params.hpp): the draw's world positions, normals, tangents and per-triangle words in this slot's arrays are still what the
        // full path below would write — only the camera moved.  clip = view_proj * world, the full path's own last step on the same f32 values.
        if (f.cache_mark && tid == 0u) f.cache_mark[blockIdx.x] = f.frame_serial;
        if (tid < count) {
            const size_t gv = (size_t)3u * d.first_tri + local0 + tid;
            const float4 w = f.wcache[gv];
            const m4 view_proj = load_m4(reinterpret_cast<const float*>(sc->buf[AWSM_BUF_CAMERA] + 128));
            const f4 clip = mul(view_proj, {w.x, w.y, w.z, w.w});
            f.clip[gv] = make_float4(clip.x, clip.y, clip.z, clip.w);
        }
        return;
    }
    GeomMetaDev gm;
    gm.morph_len = gmp[2]; gm.morph_weights_off = gmp[3]; gm.morph_values_off = gmp[4];
    gm.skin_sets = gmp[5]; gm.skin_matrices_off = gmp[6]; gm.skin_index_weights_off = gmp[7];
    gm.transform_off = gmp[8];

    f3 pos, normal; f4 tangent; uint32_t vertex_index;
    if (!FWD) {
        const uint8_t* src = sc->buf[AWSM_BUF_VIS_GEOM_DATA] + (size_t)d.vis_data_off + (size_t)local0 * 56u;
        const uint32_t bytes = count * 56u;
        const uint32_t n16 = bytes >> 4;
        const uint4* src16 = reinterpret_cast<const uint4*>(src);
        uint4* lds16 = reinterpret_cast<uint4*>(lds_vtx);
        for (uint32_t i = tid; i < n16; i += 256u) lds16[i] = src16[i];
        const uint32_t tail0 = n16 << 2;          // remaining dwords (0 or 2)
        if (tid < (bytes >> 2) - tail0) lds_vtx[tail0 + tid] = reinterpret_cast<const uint32_t*>(src)[tail0 + tid];
        __syncthreads();
        if (tid >= count) return;
        const uint32_t* v = lds_vtx + tid * 14u;
        pos = {__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2])};
        normal = {__uint_as_float(v[6]), __uint_as_float(v[7]), __uint_as_float(v[8])};
        tangent = {__uint_as_float(v[9]), __uint_as_float(v[10]), __uint_as_float(v[11]), __uint_as_float(v[12])};
        vertex_index = v[13];    // original_vertex_index
    } else {
        if (tid >= count) return;
        const uint32_t* mm = reinterpret_cast<const uint32_t*>(sc->buf[AWSM_BUF_MATERIAL_META] + (size_t)(gmp[9] / 256u) * 256u);
        vertex_index = reinterpret_cast<const uint32_t*>(sc->buf[AWSM_BUF_ATTR_INDEX])[mm[9] / 4u + local0 + tid];   // @builtin(vertex_index) of an indexed draw
        const float* v = reinterpret_cast<const float*>(sc->buf[AWSM_BUF_TRANSPARENCY_GEOM_DATA] + (size_t)d.vis_data_off + (size_t)vertex_index * 40u);
        pos = {v[0], v[1], v[2]};
        normal = {v[3], v[4], v[5]};
        tangent = {v[6], v[7], v[8], v[9]};
    }

    if (gm.morph_len != 0u) {               // morph.wgsl: weights at [off/4 + 1 + i]
        const float* mw = reinterpret_cast<const float*>(sc->buf[AWSM_BUF_MORPH_WEIGHTS]);
        const float* mv = reinterpret_cast<const float*>(sc->buf[AWSM_BUF_MORPH_VALUES]);
        const uint32_t wbase = gm.morph_weights_off / 4u + 1u;
        const uint32_t vbase = gm.morph_values_off / 4u + vertex_index * (gm.morph_len * 10u);
        f3 txyz = {tangent.x, tangent.y, tangent.z};
        for (uint32_t i = 0; i < gm.morph_len; i++) {
            const float w = mw[wbase + i];
            const float* dlt = mv + vbase + i * 10u;
            pos = {pos.x + w * dlt[0], pos.y + w * dlt[1], pos.z + w * dlt[2]};
            normal = {normal.x + w * dlt[3], normal.y + w * dlt[4], normal.z + w * dlt[5]};
            txyz = {txyz.x + w * dlt[6], txyz.y + w * dlt[7], txyz.z + w * dlt[8]};
        }
        tangent = {txyz.x, txyz.y, txyz.z, tangent.w};
    }
    if (gm.skin_sets != 0u) {               // skin.wgsl: M = sum_sets (w0*J0 + w1*J1 + w2*J2 + w3*J3)
        const float* iw = reinterpret_cast<const float*>(sc->buf[AWSM_BUF_SKIN_INDEX_WEIGHTS]);
        const float* jm = reinterpret_cast<const float*>(sc->buf[AWSM_BUF_SKIN_MATRICES]);
        const uint32_t base = gm.skin_index_weights_off / 4u + vertex_index * gm.skin_sets * 8u;
        const uint32_t moff = gm.skin_matrices_off / 64u;
        // Column by column: column c of the blended matrix (summed over the sets in order, as skin.wgsl does) is used at once for position,
        // normal and tangent — ((c0 x + c1 y) + c2 z) + c3 w accumulates in that very order — instead of building all sixteen elements first:
        // under the kernel's 80-VGPR cap that spilled to scratch, and a kernel with a scratch segment costs ~20 us more to dispatch.
        f3 op = {0.0f, 0.0f, 0.0f}, on = op, ot = op;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            float e0 = 0.0f, e1 = 0.0f, e2 = 0.0f;
            for (uint32_t set = 0; set < gm.skin_sets; set++) {
                const float4* p = reinterpret_cast<const float4*>(iw + base + set * 8u);   // 32-B records, 16-B aligned
                const float4 q0 = p[0], q1 = p[1];
                const uint32_t j0 = __float_as_uint(q0.x), j1 = __float_as_uint(q0.z), j2 = __float_as_uint(q1.x), j3 = __float_as_uint(q1.z);
                const float w0 = q0.y, w1 = q0.w, w2 = q1.y, w3 = q1.w;
                const float4 a = reinterpret_cast<const float4*>(jm + (size_t)(j0 + moff) * 16u)[c], bq = reinterpret_cast<const float4*>(jm + (size_t)(j1 + moff) * 16u)[c];
                const float4 cq = reinterpret_cast<const float4*>(jm + (size_t)(j2 + moff) * 16u)[c], dq = reinterpret_cast<const float4*>(jm + (size_t)(j3 + moff) * 16u)[c];
                const float s0 = ((w0 * a.x + w1 * bq.x) + w2 * cq.x) + w3 * dq.x;
                const float s1 = ((w0 * a.y + w1 * bq.y) + w2 * cq.y) + w3 * dq.y;
                const float s2 = ((w0 * a.z + w1 * bq.z) + w2 * cq.z) + w3 * dq.z;
                if (set == 0) { e0 = s0; e1 = s1; e2 = s2; } else { e0 += s0; e1 += s1; e2 += s2; }
            }
            const float px = c == 0 ? pos.x : (c == 1 ? pos.y : (c == 2 ? pos.z : 1.0f));
            if (c == 0) op = {e0 * px, e1 * px, e2 * px}; else op = {op.x + e0 * px, op.y + e1 * px, op.z + e2 * px};
            if (c < 3) {      // skin.wgsl:150-156: normal and tangent through the raw upper 3x3, no inverse-transpose
                const float nx = c == 0 ? normal.x : (c == 1 ? normal.y : normal.z), tx = c == 0 ? tangent.x : (c == 1 ? tangent.y : tangent.z);
                if (c == 0) { on = {e0 * nx, e1 * nx, e2 * nx}; ot = {e0 * tx, e1 * tx, e2 * tx}; }
                else { on = {on.x + e0 * nx, on.y + e1 * nx, on.z + e2 * nx}; ot = {ot.x + e0 * tx, ot.y + e1 * tx, ot.z + e2 * tx}; }
            }
        }
        pos = op; normal = on;
        tangent = {ot.x, ot.y, ot.z, tangent.w};
    }

    m4 model = load_m4(reinterpret_cast<const float*>(sc->buf[AWSM_BUF_TRANSFORMS] + (size_t)(gm.transform_off / 64u) * 64u));
    if (d.flags & kDrawInstanced) {   // apply_vertex.wgsl:47-59: model_transform = model * instance_transform (column by column)
        const m4 inst = load_m4(reinterpret_cast<const float*>(sc->buf[AWSM_BUF_INSTANCES] + d.inst_off));
        m4 mi;
#pragma unroll
        for (int j = 0; j < 4; j++) mi.c[j] = mul(model, inst.c[j]);
        model = mi;
    }
    const m4 view_proj = load_m4(reinterpret_cast<const float*>(sc->buf[AWSM_BUF_CAMERA] + 128));
    const f4 world_pos = mul(model, {pos.x, pos.y, pos.z, 1.0f});
    const f4 clip = mul(view_proj, world_pos);

    const m3 mm = upper3(model);
    const f3 c0 = mm.c[0], c1 = mm.c[1], c2 = mm.c[2];
    const f3 r0 = {c0.x, c1.x, c2.x}, r1 = {c0.y, c1.y, c2.y}, r2 = {c0.z, c1.z, c2.z};
    const f3 cof0 = cross(r1, r2), cof1 = cross(r2, r0), cof2 = cross(r0, r1);
    const float det_model = dot(r0, cof0);
    f3 wn_un;
    if (fabsf(det_model) > 1e-8f) wn_un = {dot(cof0, normal) / det_model, dot(cof1, normal) / det_model, dot(cof2, normal) / det_model};
    else wn_un = mul(mm, normal);
    const f3 world_normal = normalize(wn_un);

    const f3 tangent_raw = mul(mm, {tangent.x, tangent.y, tangent.z});
    f3 tangent_ortho = tangent_raw - world_normal * dot(tangent_raw, world_normal);
    const float tlen_sq = dot(tangent_ortho, tangent_ortho);
    if (tlen_sq > 1e-8f) {
        tangent_ortho = tangent_ortho * inverse_sqrt(tlen_sq);
    } else {
        const f3 axis = (fabsf(world_normal.z) > 0.999f) ? mk3(0.0f, 1.0f, 0.0f) : mk3(0.0f, 0.0f, 1.0f);
        tangent_ortho = normalize(cross(axis, world_normal));
    }

    const uint32_t lv = local0 + tid;
    const size_t gv = (size_t)3u * d.first_tri + lv;
    f.clip[gv] = make_float4(clip.x, clip.y, clip.z, clip.w);
    if (!FWD && f.wcache) f.wcache[gv] = make_float4(world_pos.x, world_pos.y, world_pos.z, world_pos.w);
    f.nrm[gv] = make_float4(world_normal.x, world_normal.y, world_normal.z, 0.0f);
    f.tan[gv] = make_float4(tangent_ortho.x, tangent_ortho.y, tangent_ortho.z, tangent.w);
    if (FWD) f.wpos[gv] = make_float4(world_pos.x, world_pos.y, world_pos.z, 1.0f);
    // transparent pass: bit 31 = the draw's material is ALPHA_MODE_MASK (k_resolve_draws ran before this kernel), so the coverage walk
    // learns it from the word it loads anyway instead of a second dependent load per triangle
    if (lv % 3u == 0u) {
        const uint32_t info = lo | ((d.flags & 0x7Fu) << 24) /* 0x80 = kDrawInstanced, internal */ | ((FWD && (f.draw_shade[lo].flags & 2u)) ? 0x80000000u : 0u);
        f.tri_info[d.first_tri + lv / 3u] = info;
        if (!FWD && f.tri_shade) {
            // compute.wgsl:182-197 + texture_uvs.wgsl:64-84, once per triangle instead of once per pixel: where each corner's TEXCOORD_0 lives.
            // Byte offsets into the attribute data (the reference's offsets are u32 as well).
            // The three corners' TEXCOORD_0 themselves (the values the opaque pass interpolates): a pixel then needs no index or attribute load at all.
            // A mesh without that set (or a stream shorter than its indices claim) leaves zeros; such a draw has no texture to look up with them.
            uint4 ts0 = make_uint4(info, 0u, 0u, 0u), ts1 = make_uint4(0u, 0u, 0u, 0u);
            if (sc->buf[AWSM_BUF_MATERIAL_META] && sc->buf[AWSM_BUF_ATTR_INDEX] && sc->buf[AWSM_BUF_ATTR_DATA]) {
                const uint32_t* mm = reinterpret_cast<const uint32_t*>(sc->buf[AWSM_BUF_MATERIAL_META] + (size_t)(gmp[9] / 256u) * 256u);
                const uint32_t* ai = reinterpret_cast<const uint32_t*>(sc->buf[AWSM_BUF_ATTR_INDEX]) + mm[9] / 4u + lv;       // 3 * triangle == lv
                const uint32_t data_word = mm[10] / 4u, stride_words = mm[11] / 4u, uv0 = mm[12];
                const uint32_t* ad = reinterpret_cast<const uint32_t*>(sc->buf[AWSM_BUF_ATTR_DATA]);
                const uint32_t w0 = data_word + ai[0] * stride_words + uv0, w1 = data_word + ai[1] * stride_words + uv0, w2 = data_word + ai[2] * stride_words + uv0;
                const uint32_t n_words = f.attr_data_bytes >> 2;
                if (w0 < n_words && w0 + 1u < n_words) { ts0.z = ad[w0]; ts0.w = ad[w0 + 1u]; }
                if (w1 < n_words && w1 + 1u < n_words) { ts1.x = ad[w1]; ts1.y = ad[w1 + 1u]; }
                if (w2 < n_words && w2 + 1u < n_words) { ts1.z = ad[w2]; ts1.w = ad[w2 + 1u]; }
            }
            f.tri_shade[2u * (d.first_tri + lv / 3u)] = ts0;
            f.tri_shade[2u * (d.first_tri + lv / 3u) + 1u] = ts1;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_bin<FILL>: one thread per triangle.  Pass 1 counts (triangle, tile) pairs per tile, pass 2 (after the scan) writes
// the triangle ranks into each tile's list.
//
// The 256 consecutive triangles of a workgroup are neighbours on screen, so they fall into a small window of tiles.
// The workgroup histograms them in LDS (window <= kBinWindow tiles) and touches global memory once per distinct tile:
// one atomicAdd of the local count (pass 1), or one atomicAdd that reserves a run of the tile's list (pass 2), instead
// of one same-address global atomic per (triangle, tile) pair.  Triangles covering more than 16 tiles are walked by the
// whole wavefront, 64 tiles per step, with direct global atomics (there are few of them).
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kBinWindow = 2048;

// (tx, l): tile column and LOCAL tile row of the shard; absolute tile row = tile_row0 + l * band_n
template <bool FILL>
AWSM_DI void bin_emit(const FrameDev& f, int tx, int l, uint32_t rank) {
    const uint32_t idx = (uint32_t)l * f.tiles_x + (uint32_t)tx;
    if (!FILL) {
        atomicAdd(&f.tile_count[idx], 1u);
    } else {
        const uint32_t slot = atomicAdd(&f.tile_cursor[idx], 1u);
        const uint32_t pos = f.tile_offset[idx] + slot;
        if (pos < f.bin_capacity) f.bin_list[pos] = rank;
    }
}

// One triangle's share of the binning: its tile rectangle in (column, local tile row) space of the shard.
struct BinTri {
    TriSetup t;
    int tx0, tx1, ty0, ty1, wdt, ntiles;
    bool ok, big, small;
};
template <bool SETUP>
AWSM_DI void bin_tri_load(const FrameDev& f, uint32_t r, BinTri& b) {
    b.ok = false;
    if (r < f.total_tris) {
        if (SETUP) {   // the first phase of the counting pass does the setup once and leaves it for everyone downstream
            const float4 v0 = f.clip[(size_t)r * 3], v1 = f.clip[(size_t)r * 3 + 1], v2 = f.clip[(size_t)r * 3 + 2];
            // A shard (bands or a row strip) sees most triangles of the frame only to drop them: decide that from a conservative
            // row range (+-1 pixel around the unsnapped vertices; any w <= 0 or non-finite value keeps the triangle) before the setup,
            // and leave just the "no fragment" marker.  The exact rows of a kept triangle come from tri_setup as before.
            bool outside = false;
            if ((f.band_n > 1u || f.y0 > 0u || f.y1 < f.height) && v0.w > 0.0f && v1.w > 0.0f && v2.w > 0.0f) {
                const float hh = 0.5f * (float)f.height;
                const float ya = (1.0f - v0.y / v0.w) * hh, yb = (1.0f - v1.y / v1.w) * hh, yc = (1.0f - v2.y / v2.w) * hh;
                const float lo = fminf(fminf(ya, yb), yc) - 1.0f, hi = fmaxf(fmaxf(ya, yb), yc) + 1.0f;
                if (hi < (float)f.y0 || lo >= (float)f.y1) outside = true;
                else if (f.band_n > 1u && lo >= (float)f.y0 && hi < (float)f.y1) {     // in range (NaN fails this): which owned tile rows can it touch?
                    const int bn = (int)f.band_n, row0 = (int)f.tile_row0;
                    const int a0 = ((int)floorf(lo) >> kTileShift) - row0, a1 = ((int)floorf(hi) >> kTileShift) - row0;
                    const int t0 = (max(a0, 0) + bn - 1) / bn, t1 = a1 >= 0 ? a1 / bn : -1;
                    outside = t0 > t1;
                }
            }
            if (outside) { tri_rec_store_invalid(f.tri_rec + r); b.ok = false; }
            else {
                const bool cull_back = ((f.tri_info[r] >> 24) & AWSM_DRAW_CULL_BACK) != 0;
                b.ok = tri_setup(v0, v1, v2, cull_back, f.width, f.height, f.y0, f.y1, b.t);
                if (!b.ok) { b.t.minx = 0; b.t.maxx = 0; b.t.miny = 0; b.t.maxy = 0; }
                tri_rec_store(f.tri_rec + r, b.t, b.ok);
            }
        } else {
            b.ok = tri_rec_load(f.tri_rec + r, b.t);
        }
    }
    // a triangle that touches none of the shard's tile rows drops out here
    b.tx0 = 0; b.tx1 = -1; b.ty0 = 0; b.ty1 = -1;
    const int bn = (int)f.band_n, row0 = (int)f.tile_row0;
    if (b.ok) {
        b.tx0 = b.t.minx >> kTileShift; b.tx1 = b.t.maxx >> kTileShift;
        const int a0 = (b.t.miny >> kTileShift) - row0, a1 = (b.t.maxy >> kTileShift) - row0;   // row mode: >= 0 (tri_setup clamps to the shard rows)
        b.ty0 = (a0 + bn - 1) / bn; b.ty1 = a1 >= 0 ? a1 / bn : -1;                               // ceil / floor; a0 > -bn
        b.ok = b.ty0 <= b.ty1;
    }
    b.wdt = b.tx1 - b.tx0 + 1;
    b.ntiles = b.ok ? b.wdt * (b.ty1 - b.ty0 + 1) : 0;
    b.big = b.ntiles > 16;
    b.small = b.ok && !b.big;
}
AWSM_DI bool bin_tile_hit(const FrameDev& f, const TriSetup& t, int ntiles, int tx, int l) {
    const int ty = (int)f.tile_row0 + l * (int)f.band_n;
    return ntiles == 1 || tile_may_overlap(t, tx << kTileShift, ty << kTileShift, (tx + 1) << kTileShift, (ty + 1) << kTileShift);
}

// kBinBatches x 256 consecutive triangles per workgroup: the more triangles share one LDS window, the fewer global
// atomics reach the hot tiles (they serialise in L2).  With more than one batch the later phases re-read the setup records
// (L2 hits); with one (the setting that measured best) the triangle stays in registers: the kernel is one wave of workgroups
// deep, so its duration is the latency chain of a single workgroup and every dependent load shows.
constexpr uint32_t kBinBatches = 1;
static_assert(kBinBatches == 1, "k_bin keeps its triangle in registers across the phases");

// The triangles that cover more than 16 tiles: one wavefront per triangle, 64 tiles per step, block-stride over the list k_bin<count> built.
template <bool FILL>
AWSM_DI void bin_big_walk(const FrameDev& f, uint32_t block, uint32_t n_blocks) {
    const uint32_t n_big = f.counters[4];
    const int lane = threadIdx.x & 63;
    const uint32_t wave = block * 4u + (threadIdx.x >> 6), n_waves = n_blocks * 4u;
    for (uint32_t i = wave; i < n_big; i += n_waves) {
        const uint32_t r = f.big_list[i];
        BinTri b;
        bin_tri_load<false>(f, r, b);
        for (int k = lane; k < b.ntiles; k += 64) {
            const int l = b.ty0 + k / b.wdt, tx = b.tx0 + k % b.wdt;
            if (bin_tile_hit(f, b.t, b.ntiles, tx, l)) bin_emit<FILL>(f, tx, l, r);
        }
    }
}
constexpr uint32_t kBinBigBlocks = 64;        // fill pass: the first workgroups of k_bin<true> walk the big triangles (no launch of their own)

template <bool FILL>
__global__ __launch_bounds__(256) void k_bin(FrameDev f) {
#if AWSM_GEOM_PRIO
    __builtin_amdgcn_s_setprio(AWSM_GEOM_PRIO);
#endif
    __shared__ int win[4];                       // tile window of the workgroup's small triangles: x0, y0, x1, y1
    __shared__ uint32_t n_ok, n_big_wg, big_base;
    __shared__ uint32_t lcount[kBinWindow];
    __shared__ uint32_t lbase[FILL ? kBinWindow : 1];

    const uint32_t tid = threadIdx.x;
    if (frame_poisoned(f)) return;
    AWSM_STAMP_AT(f, FILL ? 2 : 0, 0);
    if (FILL && blockIdx.x < kBinBigBlocks) { bin_big_walk<true>(f, blockIdx.x, kBinBigBlocks); AWSM_STAMP_AT(f, 2, 7); return; }     // workgroup-uniform
    const uint32_t r0 = f.rank0 + (blockIdx.x - (FILL ? kBinBigBlocks : 0u)) * (256u * kBinBatches) + tid;
    const int lane = tid & 63;
    if (tid == 0) { win[0] = 0x7fffffff; win[1] = 0x7fffffff; win[2] = -1; win[3] = -1; n_ok = 0; n_big_wg = 0; }
    __syncthreads();

    // ---- phase 0: setup (count pass) / load, window of the small triangles, big triangles walked by the wavefront ----
    uint32_t my_ok = 0;
    BinTri keep;                                  // kBinBatches == 1: the triangle stays in registers for the later phases
    for (uint32_t j = 0; j < kBinBatches; j++) {
        const uint32_t r = r0 + j * 256u;
        BinTri& b = keep;
        bin_tri_load<!FILL>(f, r, b);
        AWSM_STAMP_AT(f, FILL ? 2 : 0, 7);
        my_ok += b.ok ? 1u : 0u;
        {   // tile window of the workgroup's small triangles: reduced inside the wavefront first — 256 threads updating the same four LDS
            // words serialise (measured with in-kernel stamps: 5+ us of an 11 us phase), four atomics per wavefront do not
            int wx0 = b.small ? b.tx0 : 0x7fffffff, wy0 = b.small ? b.ty0 : 0x7fffffff, wx1 = b.small ? b.tx1 : -1, wy1 = b.small ? b.ty1 : -1;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                wx0 = min(wx0, __shfl_xor(wx0, off)); wy0 = min(wy0, __shfl_xor(wy0, off));
                wx1 = max(wx1, __shfl_xor(wx1, off)); wy1 = max(wy1, __shfl_xor(wy1, off));
            }
            if (lane == 0 && wx1 >= 0) { atomicMin(&win[0], wx0); atomicMin(&win[1], wy0); atomicMax(&win[2], wx1); atomicMax(&win[3], wy1); }
        }
    }
    uint32_t big_slot = 0;
    if (!FILL) {   // big triangles go to a global list that k_bin_big walks with one wavefront per triangle: a slot inside the workgroup first
        const unsigned long long mask = __ballot(keep.big);
        if (mask) {
            uint32_t base = 0;
            if (lane == __ffsll((long long)mask) - 1) base = atomicAdd(&n_big_wg, (uint32_t)__popcll(mask));
            big_slot = (uint32_t)__shfl((int)base, __ffsll((long long)mask) - 1) + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        }
    }
    if (!FILL) {
        for (int off = 32; off > 0; off >>= 1) my_ok += __shfl_down(my_ok, off);
        if (lane == 0 && my_ok) atomicAdd(&n_ok, my_ok);
    }
    AWSM_STAMP_AT(f, FILL ? 2 : 0, 1);
    __syncthreads();
    AWSM_STAMP_AT(f, FILL ? 2 : 0, 2);
    if (!FILL && n_big_wg) {   // (workgroup-uniform) ... then one returning device atomic per workgroup instead of one per wavefront
        if (tid == 0) big_base = atomicAdd(&f.counters[4], n_big_wg);
        __syncthreads();
        if (keep.big) f.big_list[big_base + big_slot] = r0;
    }
    const int wx0 = win[0], wy0 = win[1], ww = win[2] - wx0 + 1, wh = win[3] - wy0 + 1;
    const uint32_t nwin = (ww > 0 && wh > 0) ? (uint32_t)ww * (uint32_t)wh : 0u;
    const bool use_lds = nwin <= kBinWindow;                          // workgroup-uniform
    if (!FILL && tid == 0 && n_ok) atomicAdd(&f.counters[0], n_ok);
    if (use_lds) for (uint32_t i = tid; i < nwin; i += 256u) lcount[i] = 0u;
    __syncthreads();

    // ---- phase A, small triangles: histogram in LDS (or direct emission when the window is too large) ----
    for (uint32_t j = 0; j < kBinBatches; j++) {
        const uint32_t r = r0 + j * 256u;
        BinTri reload;
        if (kBinBatches > 1) bin_tri_load<false>(f, r, reload);
        const BinTri& b = kBinBatches > 1 ? reload : keep;
        if (b.small)
            for (int l = b.ty0; l <= b.ty1; l++)
                for (int tx = b.tx0; tx <= b.tx1; tx++)
                    if (bin_tile_hit(f, b.t, b.ntiles, tx, l)) {
                        if (use_lds) atomicAdd(&lcount[(l - wy0) * ww + (tx - wx0)], 1u);
                        else bin_emit<FILL>(f, tx, l, r);
                    }
    }
    if (!use_lds) return;
    __syncthreads();
    AWSM_STAMP_AT(f, FILL ? 2 : 0, 3);
    for (uint32_t i = tid; i < nwin; i += 256u) {   // one global atomic per distinct tile of the workgroup
        const uint32_t c = lcount[i];
        if (c) {
            const uint32_t gidx = (uint32_t)(wy0 + (int)(i / (uint32_t)ww)) * f.tiles_x + (uint32_t)(wx0 + (int)(i % (uint32_t)ww));
            if (!FILL) atomicAdd(&f.tile_count[gidx], c);
            else { lbase[i] = f.tile_offset[gidx] + atomicAdd(&f.tile_cursor[gidx], c); lcount[i] = 0u; }
        }
    }
    AWSM_STAMP_AT(f, FILL ? 2 : 0, 4);
    if (!FILL) return;
    __syncthreads();
    AWSM_STAMP_AT(f, 2, 5);
    // ---- phase B: write the ranks into the reserved runs (order inside a tile's list is irrelevant: the raster kernel
    // resolves visibility with a min over packed keys) ----
    for (uint32_t j = 0; j < kBinBatches; j++) {
        const uint32_t r = r0 + j * 256u;
        BinTri reload;
        if (kBinBatches > 1) bin_tri_load<false>(f, r, reload);
        const BinTri& b = kBinBatches > 1 ? reload : keep;
        if (b.small)
            for (int l = b.ty0; l <= b.ty1; l++)
                for (int tx = b.tx0; tx <= b.tx1; tx++)
                    if (bin_tile_hit(f, b.t, b.ntiles, tx, l)) {
                        const uint32_t li = (uint32_t)((l - wy0) * ww + (tx - wx0));
                        const uint32_t pos = lbase[li] + atomicAdd(&lcount[li], 1u);
                        if (pos < f.bin_capacity) f.bin_list[pos] = r;
                    }
    }
    AWSM_STAMP_AT(f, 2, 6);
}

// k_bin_big<FILL>: the triangles that cover more than 16 tiles (near the camera: few, but each a long walk, and they come
// in runs — whole workgroups of k_bin were nothing but such triangles and ran 3x longer than the rest of the grid).
// One wavefront per triangle, 64 tiles per step, grid-stride over the list k_bin<count> built.
template <bool FILL>
__global__ __launch_bounds__(256) void k_bin_big(FrameDev f) { if (frame_poisoned(f)) return; bin_big_walk<FILL>(f, blockIdx.x, gridDim.x); }

// Exclusive scan of tile_count -> tile_offset (single workgroup; n_tiles is a few thousand), plus tile_order: the tile
// ids sorted by log2(count), heaviest first.  Workgroups start in blockIdx order, so k_raster_tile begins with the
// fullest tiles and the light ones fill in behind them (longest-processing-time-first; the fullest tile of a frame holds
// 20-40x the median number of triangles and would otherwise be the tail of the kernel).
constexpr uint32_t kRasterSlice = 256;     // triangles of a tile's list one raster workgroup takes (= one batch of k_raster_tile)
// Multi-workgroup, one launch: workgroup k owns the tiles [256 k, 256 k + 256), one per thread.
//   1. local: the thread's count c, its bucket (0: empty, b: 2^(b-1) <= c < 2^b), the split quantities (a tile with more than kRasterSlice
//      triangles is rasterised by ns = ceil(c / kRasterSlice) workgroups: ns - 1 extra raster items, ns scratch tiles); workgroup-exclusive
//      prefixes of c / extras / slots (wavefront shuffles + 4 partials), a 33-bucket LDS histogram that also hands each thread its rank;
//      the workgroup's aggregates (3 totals + 33 bucket counts) go to scan_tmp[k] with stores that reach memory (sc1), every wavefront drains
//      them (s_waitcnt vmcnt(0)) before the barrier in front of the arrival counter.
//   2. the workgroup that arrives last turns the aggregates into bases — per quantity the exclusive prefix over workgroups (lane = workgroup,
//      shuffles) — and per bucket the sizes of all heavier buckets (tile_order is heaviest first), writes both behind the aggregates (sc1),
//      writes the frame totals, drains, raises the ready flag.  The others poll the flag (s_sleep); the last arriver never waits, so no order of
//      placement can deadlock, and every workgroup is small (256 threads, < 64 VGPRs): it is placed on the first exit of an opaque-pass workgroup.
//   3. every workgroup: tile_offset, tile_cursor, tile_order position, split records from its bases.
// The single-workgroup scan this replaces spent 35 us of its 40 in 32 serial steps per thread of same-address LDS atomics (19 us with 1,024
// threads, which then had to wait for room beside the previous frame's opaque pass: 40-190 us).
constexpr uint32_t kScanThreads = 256, kScanWords = 40;     // per workgroup: [0] sum c, [1] sum extras, [2] sum slots, [3 + b] bucket b (b = 0..32), pad
AWSM_DI uint32_t ld_sc1(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
AWSM_DI void st_sc1(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__global__ __launch_bounds__(kScanThreads) void k_bin_scan(FrameDev f, uint32_t n_tiles) {
#if AWSM_GEOM_PRIO
    __builtin_amdgcn_s_setprio(AWSM_GEOM_PRIO);
#endif
    __shared__ uint32_t hist[36], part[3][4], base[kScanWords], last_flag;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, k = blockIdx.x, G = gridDim.x;
    const uint32_t i = k * kScanThreads + tid;
    const bool live = i < n_tiles;
    if (frame_poisoned(f)) return;
    AWSM_STAMP_AT(f, 1, 0);
    if (tid < 36u) hist[tid] = 0u;
    const uint32_t c = live ? f.tile_count[i] : 0u;
    const uint32_t b = 32u - (uint32_t)__clz(c);
    const uint32_t ns = (live && c > kRasterSlice && f.raster_scratch) ? (c + kRasterSlice - 1u) / kRasterSlice : 0u;
    const uint32_t v[3] = {c, ns ? ns - 1u : 0u, ns};
    __syncthreads();
    const uint32_t rank = live ? atomicAdd(&hist[b], 1u) : 0u;
    uint32_t incl[3] = {v[0], v[1], v[2]};
#pragma unroll
    for (int d = 1; d < 64; d <<= 1)
#pragma unroll
        for (int q = 0; q < 3; q++) { const uint32_t o = (uint32_t)__shfl_up((int)incl[q], d); if ((int)lane >= d) incl[q] += o; }
    if (lane == 63u) { part[0][wave] = incl[0]; part[1][wave] = incl[1]; part[2][wave] = incl[2]; }
    __syncthreads();
    uint32_t excl[3], total[3];
#pragma unroll
    for (int q = 0; q < 3; q++) {
        uint32_t before = 0u, all = 0u;
#pragma unroll
        for (uint32_t w = 0; w < 4u; w++) { const uint32_t p = part[q][w]; all += p; if (w < wave) before += p; }
        excl[q] = before + incl[q] - v[q]; total[q] = all;
    }
    uint32_t* agg = f.scan_tmp + (size_t)k * kScanWords;
    uint32_t* bases = f.scan_tmp + (size_t)(G + k) * kScanWords;
    if (tid < 3u) st_sc1(agg + tid, total[tid]);
    else if (tid < 36u) st_sc1(agg + tid, hist[tid - 3u]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0u) last_flag = atomicAdd(&f.counters[12], 1u) == G - 1u ? 1u : 0u;
    __syncthreads();
    AWSM_STAMP_AT(f, 1, 1);
    if (last_flag) {      // workgroup-uniform
        // quantity q (36 of them): exclusive prefix over the workgroups, 64 workgroups per pass with lane = workgroup; wavefront w takes q = w, w + 4, ...
        __shared__ uint32_t q_total[36];
        for (uint32_t q = wave; q < 36u; q += 4u) {
            uint32_t carry = 0u;
            for (uint32_t k0 = 0; k0 < G; k0 += 64u) {
                const uint32_t kk = k0 + lane;
                const uint32_t a = kk < G ? ld_sc1(f.scan_tmp + (size_t)kk * kScanWords + q) : 0u;
                uint32_t in = a;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)in, d); if ((int)lane >= d) in += o; }
                if (kk < G) st_sc1(f.scan_tmp + (size_t)(G + kk) * kScanWords + q, carry + in - a);
                carry += (uint32_t)__shfl((int)in, 63);
            }
            if (lane == 0u) q_total[q] = carry;
        }
        __syncthreads();
        if (tid < 33u) {      // bucket b's run of tile_order starts behind all heavier buckets
            uint32_t heavier = 0u;
            for (uint32_t bb = tid + 1u; bb < 33u; bb++) heavier += q_total[3u + bb];
            st_sc1(f.scan_tmp + (size_t)2u * G * kScanWords + tid, heavier);
        }
        if (tid == 0u) {
            const uint32_t total_entries = q_total[0];
            f.tile_offset[n_tiles] = total_entries;
            f.counters[1] = total_entries;
            if (total_entries > f.bin_capacity) f.counters[2] = 1u;
            f.counters[7] = min(q_total[1], f.raster_extra_cap);
            if (f.host_bin_status) {   // for frames nobody waits for: the host sizes the list of later frames from this (awsm_hip_geometry_pass)
                __hip_atomic_store(f.host_bin_status, total_entries, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                // bit 31: THIS frame ran with a list too short for it (the host may have grown the list since, so it cannot tell from the size it sees)
                __hip_atomic_store(f.host_bin_status + 1, (f.frame_serial & 0x7FFFFFFFu) | (total_entries > f.bin_capacity ? 0x80000000u : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // bases and run starts went out with sc1 stores: in memory once every wavefront has drained
        __syncthreads();
        if (tid == 0u) st_sc1(&f.counters[13], 1u);
    } else {
        if (tid == 0u) {      // bounded (~1 s): a grid whose counters were not reset must end, not hang the device; the frame is then flagged as overflowed
            uint32_t spins = 0u;
            while (ld_sc1(&f.counters[13]) == 0u && ++spins < (1u << 22)) __builtin_amdgcn_s_sleep(8);
            if (spins >= (1u << 22)) f.counters[2] = 1u;
        }
        __syncthreads();
    }
    AWSM_STAMP_AT(f, 1, 2);
    if (tid < 36u) base[tid] = ld_sc1(bases + tid) + (tid >= 3u ? ld_sc1(f.scan_tmp + (size_t)2u * G * kScanWords + tid - 3u) : 0u);
    __syncthreads();
    if (live) {
        f.tile_offset[i] = base[0] + excl[0];
        f.tile_cursor[i] = 0u;
        f.tile_order[base[3u + b] + rank] = i;
        if (ns) {     // tile_split is written (and read, k_raster_tile) for these tiles only.  The caps hold whenever the bin list itself does not overflow.
            const uint32_t eb = base[1] + excl[1], sb = base[2] + excl[2];
            const bool fits = eb + ns - 1u <= f.raster_extra_cap && sb + ns <= f.raster_slot_cap && ns <= 4095u;
            for (uint32_t s = 1; s < ns; s++) if (eb + s - 1u < f.raster_extra_cap) f.tile_order[n_tiles + eb + s - 1u] = fits ? (i | (s << 20)) : 0xFFFFFFFFu;
            reinterpret_cast<uint2*>(f.tile_split)[i] = make_uint2(fits ? sb : 0xFFFFFFFFu, 0u);
        }
    }
    AWSM_STAMP_AT(f, 1, 4);
}

// ------------------------------------------------------------------------------------------------
// k_raster_tile: one 256-thread workgroup owns one 32x32 tile whose packed keys live in LDS (8 KB).
// Each thread sets up one binned triangle and classifies it by the pixel area of (bbox ∩ tile):
//   - <= 4 pixels:  the thread samples them itself;
//   - <= 256 pixels ("mid", the bulk of a 260k-triangle 4K frame: median bbox ~11x11): LDS work list drained by
//     16-lane groups, one triangle per group at a time, 16 lanes = one 4x4 pixel block per step;
//   - larger ("big"): LDS work list drained by whole wavefronts, 64 lanes = one 8x8 pixel block per step.
// All paths resolve depth + submission order with ds_min_u64 on the LDS tile.
// ------------------------------------------------------------------------------------------------
struct WorkTri {
    double c[3];
    float a[3], b[3];
    float zq[3];
    uint32_t rank;
    uint32_t bbox;   // x0 | x1<<8 | y0<<16 | y1<<24, tile-local, inclusive
    uint32_t exact;  // 0: edge values by FMA per sample; 1: TriSetup::exact (stepped in f64); 2: TriSetup::small (stepped in 32-bit integers)
};
static_assert(sizeof(WorkTri) == 72, "WorkTri");
// A `small` mid triangle of a single-sampled frame, set up for the integer walk by the thread that classified it (one pass over the batch,
// instead of once per 16-lane group inside the walk phase): e = E - bias at the centre of the box's first pixel (where its first 4x4 block starts), the
// per-pixel steps.  Same size as WorkTri, rank / bbox / exact (= 2) at the same offsets.
struct WorkSmall {
    int e[3], a[3], b[3];
    float zq[3];
    uint32_t bias;   // bit i: edge i does not own its zero line (accepts E >= 1)
    uint32_t pad[2];
    uint32_t rank, bbox, exact;
};
static_assert(sizeof(WorkSmall) == sizeof(WorkTri) && offsetof(WorkSmall, rank) == offsetof(WorkTri, rank) && offsetof(WorkSmall, exact) == offsetof(WorkTri, exact), "WorkSmall");

AWSM_DI void load_work_tri(const WorkTri& g, TriSetup& t) {
#pragma unroll
    for (int i = 0; i < 3; i++) { t.a[i] = g.a[i]; t.b[i] = g.b[i]; t.c[i] = g.c[i]; t.zq[i] = g.zq[i]; }
    t.exact = g.exact != 0u; t.small = g.exact == 2u;
}

// One pixel of the tile against one triangle: S = 1 samples the pixel centre, S = 4 the four standard MSAA positions
// (per-sample coverage and per-sample depth, as the multisampled visibility / depth targets of the reference receive them).
template <int S>
AWSM_DI void raster_pixel(unsigned long long* keys, const TriSetup& t, int tpx, int tpy, int px, int py, uint32_t r) {
    if (S == 1) {
        const int sx = (tpx + px) << 8, sy = (tpy + py) << 8;
        const unsigned long long k = tri_sample_key_at(t, sample_coord(sx + 128), sample_coord(sy + 128), r);
        if (k != ~0ull) atomicMin(&keys[py * kTile + px], k);
    } else {
        const float zc = tri_plane_depth(t, tri_edges_d(t, (double)(tpx + px), (double)(tpy + py))) + 0.0f;      // the depth plane at the pixel's corner (-0 -> +0: depth_key_bits_sum)
        float dz[4];
        msaa_depth_steps(t.a, t.b, t.zq, dz);
#pragma unroll
        for (int s = 0; s < S; s++) {
            const unsigned long long k = tri_msaa_sample_key(t, tpx + px, tpy + py, s, zc, dz, r);
            if (k != ~0ull) atomicMin(&keys[(py * kTile + px) * S + s], k);
        }
    }
}

// The pixels of (bbox ∩ tile) in STEP x STEP blocks, this lane at (lx, ly) inside every block; the first block sits in the box's own corner (blocks
// aligned to multiples of STEP cover w / STEP + 0.75 columns of a box w wide on average, these ceil(w / STEP): a fifth fewer steps at the median
// box of the 4K frame; what a pixel's key is does not depend on where the blocks lie).  Single-sampled exact triangles (all but the
// near-plane crossers): E_i at the lane's pixel of the first block of a row by the two FMAs, then E_i += STEP * a_i from block to block —
// integers below 2^49 throughout, so the sums are the values the FMAs would give, at one f64 add per edge and pixel instead of two FMAs and
// the coordinate conversions.
// The same walk for a small exact triangle (TriSetup::small) in 32-bit integers: every E it meets is an integer below 2^30, so E at the lane's
// first pixel comes from the two FMAs once, converted, and rows and columns are integer adds.  The top-left rule is a bias: an edge that owns
// its zero line accepts E >= 0, the others E >= 1, so with e = E - bias the pixel is covered when no e is negative — one OR3 and one compare
// for the three f64 comparisons.  The depth is computed from (float)E as before: an integer converts to the same f32 from i32 as from f64.
// With four samples per pixel the stepped value is E at the pixel's corner and each sample adds its own constant, (a ox + b oy) / 256 — an integer,
// a and b being multiples of 256 — so a sample costs three adds, the OR3 and the compare where it cost six f64 FMAs and three f64 comparisons.
template <int S, int STEP>
AWSM_DI void raster_walk_i32(unsigned long long* keys, const TriSetup& t, int tpx, int tpy, int x0, int x1, int y0, int y1, int lx, int ly, uint32_t r) {
    if (S == 4) {
        const double X0 = (double)(tpx + x0 + lx), Y0 = (double)(tpy + y0 + ly);      // blocks start at the box's corner, not at multiples of STEP (see raster_walk_small)
        int e[3], bias[3], sx[3], sy[3], d[4][3];
        float dz[4];
        msaa_depth_steps(t.a, t.b, t.zq, dz);
#pragma unroll
        for (int i = 0; i < 3; i++) {
            bias[i] = (t.a[i] > 0.0f || (t.a[i] == 0.0f && t.b[i] > 0.0f)) ? 0 : 1;
            e[i] = (int)fma((double)t.a[i], X0, fma((double)t.b[i], Y0, t.c[i])) - bias[i];
            sx[i] = (int)t.a[i] * STEP; sy[i] = (int)t.b[i] * STEP;
            const int ai = (int)t.a[i] >> 8, bi = (int)t.b[i] >> 8;      // exact: multiples of 256
#pragma unroll
            for (int k = 0; k < 4; k++) d[k][i] = ai * msaa4_x(k) + bi * msaa4_y(k);
        }
        for (int by = y0; by <= y1; by += STEP) {
            const int py = by + ly;
            int r0 = e[0], r1 = e[1], r2 = e[2];
            for (int bx = x0; bx <= x1; bx += STEP) {
                const int px = bx + lx;
                if (px <= x1 && py <= y1) {
                    const float e0 = (float)(r0 + bias[0]), e1 = (float)(r1 + bias[1]), e2 = (float)(r2 + bias[2]);
                    const float zc = ((e0 * t.zq[0] + e1 * t.zq[1]) + e2 * t.zq[2]) + 0.0f;      // tri_plane_depth at the pixel's corner (-0 -> +0: depth_key_bits_sum)
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const int q0 = r0 + d[k][0], q1 = r1 + d[k][1], q2 = r2 + d[k][2];
                        uint32_t zbits;
                        if ((q0 | q1 | q2) >= 0 && depth_key_bits_sum(zc, dz[k], zbits))
                            atomicMin(&keys[(py * kTile + px) * 4 + k], ((unsigned long long)zbits << 32) | (unsigned long long)(0xFFFFFFFFu - r));
                    }
                }
                r0 += sx[0]; r1 += sx[1]; r2 += sx[2];
            }
            e[0] += sy[0]; e[1] += sy[1]; e[2] += sy[2];
        }
        return;
    }
    const double X0 = (double)(tpx + x0 + lx) + 0.5, Y0 = (double)(tpy + y0 + ly) + 0.5;
    int e[3], bias[3], sx[3], sy[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        bias[i] = (t.a[i] > 0.0f || (t.a[i] == 0.0f && t.b[i] > 0.0f)) ? 0 : 1;
        e[i] = (int)fma((double)t.a[i], X0, fma((double)t.b[i], Y0, t.c[i])) - bias[i];
        sx[i] = (int)t.a[i] * STEP; sy[i] = (int)t.b[i] * STEP;      // a, b: integers times 256, below 2^23
    }
    // a lane outside the box in y sits out the whole row of blocks: the row test folds into the sign test as an all-ones word
    const uint32_t dx = (uint32_t)(x1 - x0);
    const unsigned long long key_lo = (unsigned long long)(0xFFFFFFFFu - r);
    for (int by = y0; by <= y1; by += STEP) {
        const int py = by + ly;
        const int row_out = (py <= y1) ? 0 : -1;
        int r0 = e[0], r1 = e[1], r2 = e[2];
        uint32_t ux = (uint32_t)lx;      // px - x0
        unsigned long long* row = keys + py * kTile + x0;
        for (int bx = x0; bx <= x1; bx += STEP) {
            if (ux <= dx && (r0 | r1 | r2 | row_out) >= 0) {
                const float e0 = (float)(r0 + bias[0]), e1 = (float)(r1 + bias[1]), e2 = (float)(r2 + bias[2]);
                const float zn = (e0 * t.zq[0] + e1 * t.zq[1]) + e2 * t.zq[2];      // tri_key_from_edges
                uint32_t zbits;
                if (depth_key_bits(zn, zbits)) atomicMin(&row[ux], ((unsigned long long)zbits << 32) | key_lo);
            }
            r0 += sx[0]; r1 += sx[1]; r2 += sx[2]; ux += (uint32_t)STEP;
        }
        e[0] += sy[0]; e[1] += sy[1]; e[2] += sy[2];
    }
}

// raster_walk_i32<1, 4> from a WorkSmall record
AWSM_DI void raster_walk_small(unsigned long long* keys, const WorkSmall& g, int lx, int ly) {
    const uint32_t bb = g.bbox, r = g.rank, bw = g.bias;
    const int x0 = (int)(bb & 255u), x1 = (int)((bb >> 8) & 255u), y0 = (int)((bb >> 16) & 255u), y1 = (int)(bb >> 24);
    int e[3], bias[3], sx[3], sy[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const int a = g.a[i], b = g.b[i];      // |.| < 2^23
        bias[i] = (int)((bw >> i) & 1u);
        e[i] = g.e[i] + __mul24(lx, a) + __mul24(ly, b);
        sx[i] = a << 2; sy[i] = b << 2;
    }
    const float zq0 = g.zq[0], zq1 = g.zq[1], zq2 = g.zq[2];
    const uint32_t dx = (uint32_t)(x1 - x0);
    const unsigned long long key_lo = (unsigned long long)(0xFFFFFFFFu - r);
    for (int by = y0; by <= y1; by += 4) {
        const int py = by + ly;
        const int row_out = (py <= y1) ? 0 : -1;
        int r0 = e[0], r1 = e[1], r2 = e[2];
        uint32_t ux = (uint32_t)lx;      // px - x0
        unsigned long long* row = keys + py * kTile + x0;
        for (int bx = x0; bx <= x1; bx += 4) {
            if (ux <= dx && (r0 | r1 | r2 | row_out) >= 0) {
                const float e0 = (float)(r0 + bias[0]), e1 = (float)(r1 + bias[1]), e2 = (float)(r2 + bias[2]);
                const float zn = (e0 * zq0 + e1 * zq1) + e2 * zq2;      // tri_key_from_edges
                uint32_t zbits;
                if (depth_key_bits(zn, zbits)) atomicMin(&row[ux], ((unsigned long long)zbits << 32) | key_lo);
            }
            r0 += sx[0]; r1 += sx[1]; r2 += sx[2]; ux += 4u;
        }
        e[0] += sy[0]; e[1] += sy[1]; e[2] += sy[2];
    }
}

// raster_walk_i32<4, 4> from a WorkSmall record whose e is E - bias at the CORNER of the box's first pixel: a sample adds its own
// constant (a ox + b oy) / 256 (a, b: multiples of 256), three adds, one OR3 and one compare per sample.
AWSM_DI void raster_walk_small4(unsigned long long* keys, const WorkSmall& g, int lx, int ly) {
    const uint32_t bb = g.bbox, r = g.rank, bw = g.bias;
    const int x0 = (int)(bb & 255u), x1 = (int)((bb >> 8) & 255u), y0 = (int)((bb >> 16) & 255u), y1 = (int)(bb >> 24);
    int e[3], bias[3], sx[3], sy[3], d[4][3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const int a = g.a[i], b = g.b[i];      // |.| < 2^23
        bias[i] = (int)((bw >> i) & 1u);
        e[i] = g.e[i] + __mul24(lx, a) + __mul24(ly, b);
        sx[i] = a << 2; sy[i] = b << 2;
        const int ai = a >> 8, bi = b >> 8;    // exact: multiples of 256
#pragma unroll
        for (int k = 0; k < 4; k++) d[k][i] = __mul24(ai, msaa4_x(k)) + __mul24(bi, msaa4_y(k));
    }
    const float zq0 = g.zq[0], zq1 = g.zq[1], zq2 = g.zq[2];
    float dz[4];
    {
        const float af[3] = {(float)g.a[0], (float)g.a[1], (float)g.a[2]}, bf[3] = {(float)g.b[0], (float)g.b[1], (float)g.b[2]}, zq[3] = {zq0, zq1, zq2};      // TriSetup's a, b: integers
        msaa_depth_steps(af, bf, zq, dz);
    }
    const uint32_t dx = (uint32_t)(x1 - x0);
    const unsigned long long key_lo = (unsigned long long)(0xFFFFFFFFu - r);
    for (int by = y0; by <= y1; by += 4) {
        const int py = by + ly;
        const bool row_in = py <= y1;
        int r0 = e[0], r1 = e[1], r2 = e[2];
        uint32_t ux = (uint32_t)lx;      // px - x0
        unsigned long long* row = keys + (py * kTile + x0) * 4;
        for (int bx = x0; bx <= x1; bx += 4) {
            if (ux <= dx && row_in) {
                const float e0 = (float)(r0 + bias[0]), e1 = (float)(r1 + bias[1]), e2 = (float)(r2 + bias[2]);
                const float zc = ((e0 * zq0 + e1 * zq1) + e2 * zq2) + 0.0f;      // tri_plane_depth at the pixel's corner (-0 -> +0: depth_key_bits_sum)
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int q0 = r0 + d[k][0], q1 = r1 + d[k][1], q2 = r2 + d[k][2];
                    uint32_t zbits;
                    if ((q0 | q1 | q2) >= 0 && depth_key_bits_sum(zc, dz[k], zbits)) atomicMin(&row[ux * 4u + (uint32_t)k], ((unsigned long long)zbits << 32) | key_lo);
                }
            }
            r0 += sx[0]; r1 += sx[1]; r2 += sx[2]; ux += 4u;
        }
        e[0] += sy[0]; e[1] += sy[1]; e[2] += sy[2];
    }
}

template <int S, int STEP>
AWSM_DI void raster_walk(unsigned long long* keys, const TriSetup& t, int tpx, int tpy, int x0, int x1, int y0, int y1, int lx, int ly, uint32_t r) {
    if (t.small) { raster_walk_i32<S, STEP>(keys, t, tpx, tpy, x0, x1, y0, y1, lx, ly, r); return; }
    if (S == 1 && t.exact) {
        const double step[3] = {(double)t.a[0] * (double)STEP, (double)t.a[1] * (double)STEP, (double)t.a[2] * (double)STEP};
        const double X0 = (double)(tpx + x0 + lx) + 0.5;
        for (int by = y0; by <= y1; by += STEP) {
            const int py = by + ly;
            const double Y = (double)(tpy + py) + 0.5;
            EdgeVals ev;
#pragma unroll
            for (int i = 0; i < 3; i++) ev.E[i] = fma((double)t.a[i], X0, fma((double)t.b[i], Y, t.c[i]));
            for (int bx = x0; bx <= x1; bx += STEP) {
                const int px = bx + lx;
                if (px <= x1 && py <= y1) {
                    const unsigned long long k = tri_key_from_edges(t, ev, r);
                    if (k != ~0ull) atomicMin(&keys[py * kTile + px], k);
                }
#pragma unroll
                for (int i = 0; i < 3; i++) ev.E[i] += step[i];
            }
        }
        return;
    }
    for (int by = y0; by <= y1; by += STEP)
        for (int bx = x0; bx <= x1; bx += STEP) {
            const int px = bx + lx, py = by + ly;
            if (px <= x1 && py <= y1) raster_pixel<S>(keys, t, tpx, tpy, px, py, r);
        }
}

#ifndef AWSM_RASTER_BATCH4
#define AWSM_RASTER_BATCH4 96u
#endif
#ifndef AWSM_RASTER_BATCH1
#define AWSM_RASTER_BATCH1 128u
#endif
#ifndef AWSM_RASTER_WAVES1
#define AWSM_RASTER_WAVES1 7
#endif
#ifndef AWSM_RASTER_WAVES4
#define AWSM_RASTER_WAVES4 5
#endif
template <int S>
AWSM_DI void raster_tile_body(const FrameDev& f, unsigned long long* keys) {      // keys: 8 KB of LDS, or 32 KB with 4 samples per pixel: [pixel][sample]
    // Triangles per batch: fewer than threads, for occupancy — most tiles hold fewer than a hundred anyway.  Four samples per pixel: 96, so that keys +
    // list fit a CU's LDS four times instead of three (32 KB + 6.75 KB against 32 KB + 18 KB; k_raster_tile<4> 281 -> 240 us at 4K).  One sample:
    // 128 and a 72-register budget, seven workgroups per CU instead of six (75 -> 72 us; an eighth needs 64 registers and spills).
    constexpr uint32_t kBatch = S == 4 ? (AWSM_RASTER_BATCH4 ? AWSM_RASTER_BATCH4 : 256u) : AWSM_RASTER_BATCH1;
    __shared__ WorkTri work[kBatch];   // mid triangles from the front, big triangles from the back
    __shared__ uint32_t n_mid, n_big, next_mid, next_big;

    // Heaviest tiles first (tile_order, k_bin_scan).  Consecutive ids go to different XCDs (blockIdx & 7), which also
    // spreads the dense band of the screen over all eight of them.
    // The first counters[7] workgroups take the extra slices of the split tiles (the heaviest work of the frame), the rest one tile each.
    if (frame_poisoned(f)) return;
#if AWSM_GEOM_PRIO
    __builtin_amdgcn_s_setprio(AWSM_GEOM_PRIO);      // experiment: the geometry chain's wavefronts ahead of the opaque pass's on a shared SIMD
#endif
    const uint32_t n_tiles = f.tiles_x * f.tiles_y, n_extra = min(f.counters[7], f.raster_extra_cap);
    uint32_t item;
    if (blockIdx.x < n_extra) item = f.tile_order[n_tiles + blockIdx.x];
    else if (blockIdx.x - n_extra < n_tiles) item = f.tile_order[blockIdx.x - n_extra];
    else return;
    if (item == 0xFFFFFFFFu) return;
    AWSM_STAMP_AT(f, 3, 0);
    const uint32_t tile = item & 0xFFFFFu, slice = item >> 20;
    const uint32_t tid = threadIdx.x;
    const int tpx = (int)(tile % f.tiles_x) << kTileShift;
    const int tpy = (int)(f.tile_row0 + (tile / f.tiles_x) * f.band_n) << kTileShift;

#pragma unroll
    for (int i = 0; i < 4 * S; i++) keys[tid + i * 256] = ~0ull;   // render_pass.rs:22-30,107-114: "no hit", depth 1.0
    const uint32_t off = f.tile_offset[tile];
    const uint32_t count_all = f.tile_count[tile];
    const uint32_t slot0 = (count_all > kRasterSlice && f.raster_scratch) ? f.tile_split[2u * tile] : 0xFFFFFFFFu;
    const bool split = slot0 != 0xFFFFFFFFu;
    const uint32_t n_slices = split ? (count_all + kRasterSlice - 1u) / kRasterSlice : 1u;
    const uint32_t count_fit = min(count_all, f.bin_capacity - min(f.bin_capacity, off));
    const uint32_t first = split ? min(slice * kRasterSlice, count_fit) : 0u;
    const uint32_t count = split ? min((slice + 1u) * kRasterSlice, count_fit) : count_fit;
    const int lane = tid & 63, wave = tid >> 6;

    for (uint32_t base = first; base < count; base += kBatch) {
        if (tid == 0) { n_mid = 0; n_big = 0; next_mid = 16u; next_big = 4u; }
        __syncthreads();
        const uint32_t idx = base + tid;
        if (base == first) AWSM_STAMP_AT(f, 3, 1);
        if (idx < count && tid < kBatch) {
            const uint32_t r = f.bin_list[off + idx];
            TriSetup t;
            if (tri_rec_load(f.tri_rec + r, t)) {       // setup done once per frame by k_bin<count>
                const int x0 = max(t.minx, tpx), x1 = min(t.maxx, tpx + kTile - 1);
                const int y0 = max(t.miny, tpy), y1 = min(t.maxy, tpy + kTile - 1);
                if (x0 <= x1 && y0 <= y1) {
                    const int area = (x1 - x0 + 1) * (y1 - y0 + 1);
                    if (area <= 4) {
                        for (int py = y0; py <= y1; py++)
                            for (int px = x0; px <= x1; px++) raster_pixel<S>(keys, t, tpx, tpy, px - tpx, py - tpy, r);
                    } else {
                        const uint32_t slot = (area <= 256) ? atomicAdd(&n_mid, 1u) : (kBatch - 1u) - atomicAdd(&n_big, 1u);
                        const uint32_t bbox = (uint32_t)(x0 - tpx) | ((uint32_t)(x1 - tpx) << 8) | ((uint32_t)(y0 - tpy) << 16) | ((uint32_t)(y1 - tpy) << 24);
                        if (area <= 256 && t.small) {
                            WorkSmall& g = reinterpret_cast<WorkSmall&>(work[slot]);
                            const double half = S == 1 ? 0.5 : 0.0;      // one sample: the pixel's centre; four: its corner, the samples add their own offsets
                            const double X = (double)x0 + half, Y = (double)y0 + half;      // the first block starts at the box's corner
                            uint32_t bw = 0u;
#pragma unroll
                            for (int i = 0; i < 3; i++) {
                                const int bias = (t.a[i] > 0.0f || (t.a[i] == 0.0f && t.b[i] > 0.0f)) ? 0 : 1;
                                g.e[i] = (int)fma((double)t.a[i], X, fma((double)t.b[i], Y, t.c[i])) - bias;
                                g.a[i] = (int)t.a[i]; g.b[i] = (int)t.b[i]; g.zq[i] = t.zq[i];
                                bw |= (uint32_t)bias << i;
                            }
                            g.bias = bw; g.rank = r; g.bbox = bbox; g.exact = 2u;
                        } else {
                            WorkTri& g = work[slot];
#pragma unroll
                            for (int i = 0; i < 3; i++) { g.a[i] = t.a[i]; g.b[i] = t.b[i]; g.c[i] = t.c[i]; g.zq[i] = t.zq[i]; }
                            g.rank = r; g.exact = (t.small && !(area <= 256)) ? 2u : (t.exact ? 1u : 0u);
                            g.bbox = bbox;
                        }
                    }
                }
            }
        }
        __syncthreads();
        if (base == first) AWSM_STAMP_AT(f, 3, 2);
        const uint32_t nm = n_mid, nb = n_big;
        {   // mid: 16 groups of 16 lanes, 4x4 pixel blocks.  The first triangle of a group is its own number; the next ones come from a
            // shared counter, so a group that drew small triangles takes more of them (they differ 60x in area: a static deal leaves
            // most of the workgroup waiting at the barrier for the group that drew the large ones).
            const uint32_t group = tid >> 4;
            const int lx = (int)(tid & 3u), ly = (int)((tid >> 2) & 3u);
            for (uint32_t j = group; j < nm; ) {
                const WorkTri& g = work[j];
                if (g.exact == 2u) { if (S == 1) raster_walk_small(keys, reinterpret_cast<const WorkSmall&>(g), lx, ly); else raster_walk_small4(keys, reinterpret_cast<const WorkSmall&>(g), lx, ly); }      // (every mid entry marked 2 is a WorkSmall)
                else {
                TriSetup t;
                load_work_tri(g, t);
                const uint32_t bb = g.bbox, r = g.rank;
                const int x0 = (int)(bb & 255u), x1 = (int)((bb >> 8) & 255u), y0 = (int)((bb >> 16) & 255u), y1 = (int)(bb >> 24);
                raster_walk<S, 4>(keys, t, tpx, tpy, x0, x1, y0, y1, lx, ly, r);
                }
                uint32_t nx = 0u;
                if ((tid & 15u) == 0u) nx = atomicAdd(&next_mid, 1u);
                j = (uint32_t)__shfl((int)nx, 0, 16);
            }
        }
        for (uint32_t j = wave; j < nb; ) {   // big: one wavefront per triangle, 8x8 pixel blocks
            const WorkTri& g = work[(kBatch - 1u) - j];
            TriSetup t;
            load_work_tri(g, t);
            const uint32_t bb = g.bbox, r = g.rank;
            const int x0 = (int)(bb & 255u), x1 = (int)((bb >> 8) & 255u), y0 = (int)((bb >> 16) & 255u), y1 = (int)(bb >> 24);
            const int lx = lane & 7, ly = lane >> 3;
            raster_walk<S, 8>(keys, t, tpx, tpy, x0, x1, y0, y1, lx, ly, r);
            uint32_t nx = 0u;
            if (lane == 0) nx = atomicAdd(&next_big, 1u);
            j = (uint32_t)__shfl((int)nx, 0, 64);
        }
        __syncthreads();
    }
    __syncthreads();
    AWSM_STAMP_AT(f, 3, 3);
    if (split) {
        // Partial tile of a split list: park it in its scratch slot; the slice that finishes last folds the others into its own
        // (min over packed keys, the same resolve as inside a tile) and writes the tile.  The slices run on different XCDs, whose L2s
        // are not coherent for plain stores, and a device-scope release / acquire fence costs a write-back + invalidate of the whole
        // L2 per wavefront (measured: the raster kernel 40 % slower).  So the scratch traffic itself is device-scope atomic — relaxed
        // stores and loads that go through to memory (sc1).  Ordering against the arrival counter: EVERY storing wavefront drains its
        // own stores (s_waitcnt vmcnt(0)) before the workgroup barrier that precedes lane 0's counter add.  s_barrier waits for no
        // memory counter and a workgroup-scope release fence emits no wait for vector stores on gfx950, so the wait is written out
        // (inline asm: invisible to the pass that would otherwise drop a wait it believes redundant).  The slice whose add returns
        // n_slices - 1 knows every other slice's stores have completed; its sc1 loads bypass its CU's L1.
        __shared__ uint32_t arrived;
        unsigned long long* mine = f.raster_scratch + (size_t)(slot0 + slice) * (kTile * kTile * S);
#pragma unroll
        for (int i = 0; i < 4 * S; i++) __hip_atomic_store(mine + tid + i * 256, keys[tid + i * 256], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) arrived = __hip_atomic_fetch_add(&f.tile_split[2u * tile + 1u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (arrived != n_slices - 1u) return;
        for (uint32_t k = 0; k < n_slices; k++) {
            if (k == slice) continue;
            const unsigned long long* other = f.raster_scratch + (size_t)(slot0 + k) * (kTile * kTile * S);
            unsigned long long v[4 * S];      // all loads of a slice in flight together, then the min
#pragma unroll
            for (int i = 0; i < 4 * S; i++) v[i] = __hip_atomic_load(other + tid + i * 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int i = 0; i < 4 * S; i++) if (v[i] < keys[tid + i * 256]) keys[tid + i * 256] = v[i];
        }
        __syncthreads();
    }
    // tile -> HBM, once, row-major image with the samples of a pixel adjacent: each wavefront writes 256 B (S = 1: two
    // 32-pixel rows) or 512 B runs per step
#pragma unroll
    for (int i = 0; i < 4 * S; i++) {
        const int e = (int)tid + i * 256;
        const int p = e / S, s = e % S;
        const int px = tpx + (p & (kTile - 1)), py = tpy + (p >> kTileShift);
        if (px < (int)f.width && py >= (int)f.y0 && py < (int)f.y1) __builtin_nontemporal_store(keys[e], &f.vis[((size_t)py * f.width + px) * S + s]);      // written once, read by the next kernel
    }
    AWSM_STAMP_AT(f, 3, 4);
}
// The two instantiations as explicit specialisations, because their register budgets differ in kind: one sample — seven workgroups per CU (72 registers);
// four samples — the 32-KB tile allows four workgroups per CU whatever the registers, but at 104 registers four of its wavefronts leave a SIMD 96,
// less than one wavefront of the gradient lean kernel needs (112): the two kernels then take turns on a CU instead of sharing it, and
// k_raster_tile<4> adds its whole duration to the frame (tools/knockout.sh).  96: four of its wavefronts and one of the lean kernel's fit a SIMD's 512.
// (The compiler drops a waves-per-SIMD request that the kernel's static LDS makes unreachable, and with it the register cap it implies; so <4>'s tile is
// dynamic LDS — the launcher passes its 32 KB — and the request, 5 = at most 96 registers, stands.)
template <int S> __global__ void k_raster_tile(FrameDev f);
template <> __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(AWSM_RASTER_WAVES1))) void k_raster_tile<1>(FrameDev f) {
    __shared__ unsigned long long keys[kTile * kTile];
    raster_tile_body<1>(f, keys);
}
// Round 5: with the gradient lean kernel down to 96 registers, four wavefronts of this kernel could hold 104 each beside one of its (4 x 104 + 96 = 512) —
// what the kernel uses uncapped, and the 96 cap's 12 bytes of scratch (two registers) would be gone.  Measured (profiles/r05_ab_register_caps.txt): the
// kernel alone the same 174 us, the MSAA x4 + mips frame 1,845 frames/s against 1,908 with the 96 cap — every wavefront of the two kernels the same size
// packs a SIMD better than sizes that only add up in one combination.  The cap stays (AWSM_RASTER_VGPRS4 = 104 is the experiment).
#ifndef AWSM_RASTER_VGPRS4
#define AWSM_RASTER_VGPRS4 0
#endif
#if AWSM_RASTER_VGPRS4
template <> __global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(AWSM_RASTER_VGPRS4))) void k_raster_tile<4>(FrameDev f) {
#else
template <> __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(AWSM_RASTER_WAVES4))) void k_raster_tile<4>(FrameDev f) {
#endif
    extern __shared__ unsigned long long keys_dynamic[];
    raster_tile_body<4>(f, keys_dynamic);
}

// MSAA frames with hud meshes (render.rs:169-178 with render_passes/geometry/render_pass.rs:55-57,107-114): the HUD geometry pass draws over the four
// visibility targets (LoadOp::Load) but tests and writes `hud_depth`, not `depth` — so after it a sample covered by a hud mesh shows the hud triangle in
// the visibility, barycentric and normal targets and still the WORLD's depth in the depth target (1.0 where the world left none), and that mix is what
// the opaque pass's edge detector and per-sample resolve read (helpers/msaa.wgsl:42-146, helpers/material_shading.wgsl:170-210).  The same thing as one key
// per sample: the hud triangle's rank under the world's depth bits.  One thread per sample.
__global__ __launch_bounds__(256) void k_hud_merge(const unsigned long long* __restrict__ world, const unsigned long long* __restrict__ hud, unsigned long long* __restrict__ out,
                                                   size_t first, size_t n) {
    const size_t i = first + (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= first + n) return;
    const unsigned long long w = world[i], h = hud[i];
    const unsigned long long depth = w == ~0ull ? 0x3F800000ull : (w >> 32);      // render_pass.rs:107-114: depth cleared to 1.0
    out[i] = h == ~0ull ? w : ((depth << 32) | (h & 0xFFFFFFFFull));
}

// Small host->device uploads (dirty ranges of the scene mirrors, the draw list) read the pinned staging ring directly
// from a kernel: the copy stays in the compute queue, where an SDMA copy would stall the in-order stream on a
// cross-engine signal for longer than the whole transform kernel runs.
__global__ __launch_bounds__(256) void k_upload_words(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src_pinned, uint32_t n_words) {
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n_words; i += gridDim.x * 256u) dst[i] = __builtin_nontemporal_load(src_pinned + i);
}

// generate_mipmaps, one level (renderer-core/src/texture/mipmap.rs:140-250), STRICT f32 so that the RGBA8 results are
// bit-identical everywhere: 2x2 texel loads clamped to 2 x the destination extent (and to the real source extent), filter
// by MipmapTextureKind, store as unorm8 = floor(clamp(v,0,1)*255 + 0.5).  One thread per destination texel and layer.
AWSM_DI uint32_t to_unorm8(float v) {
    if (!(v > 0.0f)) return 0u;          // also NaN
    if (v > 1.0f) v = 1.0f;
    return (uint32_t)floorf(v * 255.0f + 0.5f);
}
__global__ __launch_bounds__(256) void k_gen_mip_level(uint32_t* __restrict__ chain, uint32_t src_off, uint32_t dst_off, uint32_t sw, uint32_t sh,
                                                       uint32_t dw, uint32_t dh, uint32_t layers, const uint32_t* __restrict__ kinds) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= dw * dh * layers) return;
    const uint32_t x = i % dw, y = (i / dw) % dh, layer = i / (dw * dh);
    const uint32_t kind = kinds[layer];
    const uint32_t* src = chain + src_off + (size_t)layer * sw * sh;
    float r[4][4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint32_t sx = min(min(x * 2u + (uint32_t)(k & 1), dw * 2u - 1u), sw - 1u);
        uint32_t sy = min(min(y * 2u + (uint32_t)(k >> 1), dh * 2u - 1u), sh - 1u);
        const uint32_t t = src[(size_t)sy * sw + sx];
        r[k][0] = (float)(t & 255u) / 255.0f; r[k][1] = (float)((t >> 8) & 255u) / 255.0f;
        r[k][2] = (float)((t >> 16) & 255u) / 255.0f; r[k][3] = (float)(t >> 24) / 255.0f;
    }
    float o0, o1, o2, o3;
    if (kind == 2u) {            // filter_metallic_roughness
        float m = 0.0f, r2 = 0.0f, b = 0.0f, al = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; k++) { m += r[k][0]; r2 += r[k][1] * r[k][1]; b += r[k][2]; al += r[k][3]; }
        o0 = m * 0.25f; o1 = sqrtf(r2 * 0.25f); o2 = b * 0.25f; o3 = al * 0.25f;
    } else {
        float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; k++) { s0 += r[k][0]; s1 += r[k][1]; s2 += r[k][2]; s3 += r[k][3]; }
        o0 = s0 * 0.25f; o1 = s1 * 0.25f; o2 = s2 * 0.25f; o3 = s3 * 0.25f;          // filter_simple
        if (kind == 1u) {        // filter_normal: renormalise
            const f3 n = normalize(mk3(o0 * 2.0f - 1.0f, o1 * 2.0f - 1.0f, o2 * 2.0f - 1.0f));
            o0 = n.x * 0.5f + 0.5f; o1 = n.y * 0.5f + 0.5f; o2 = n.z * 0.5f + 0.5f;
        }
    }
    chain[dst_off + ((size_t)layer * dh + y) * dw + x] = to_unorm8(o0) | (to_unorm8(o1) << 8) | (to_unorm8(o2) << 16) | (to_unorm8(o3) << 24);
}

// Device-side hand-off between the streams of an overlapped frame pipeline (awsm_hip.cpp: enqueue_opaque).  A cross-stream hipEvent costs the
// waiting queue 20-30 us after the event has fired (the command processor resolves barrier packets on a timer: both queues of a frame boundary
// were seen to resume at the same instant, 19.6 us after the last kernel ended, the GPU idle in between).  Here the producer stream ends with a
// one-lane kernel that stores the frame's serial number, and the consumer stream begins with a one-lane kernel that polls it: the stream's own
// in-order execution does the rest, and the hand-off costs a memory round trip.  Visibility of the producer's data is what the kernel boundaries
// give (release at the end of the producer's last kernel, which the signalling kernel follows in order; acquire at the start of the consumer's
// next kernel); the flag itself is read and written past the non-coherent L2s (sc1).  The poll is bounded: a gate that is never opened — kernels
// serialised by a counter-collecting profiler, or two streams folded onto one hardware queue — ends, counts itself in a pinned word, and the host
// falls back to events (awsm_hip_frame_end reports the frame).  Gates enqueued before the host noticed see the count differ from the one they
// were enqueued with and give up after a few polls, so a run-ahead host costs one timeout, not one per queued frame.  awsm_hip_create probes the
// mechanism on the context's own streams first and leaves it off when a gate and its signal do not run side by side.
// `stamp` (frame trace, awsm_hip_frame_trace): the constant-rate device clock at the moment the producer stream reached this point.
__global__ __launch_bounds__(64) void k_handoff_signal(uint32_t* flag, uint32_t serial, unsigned long long* stamp) {
    if (threadIdx.x != 0) return;
    if (stamp) *stamp = wall_clock64();
    if (flag) st_sc1(flag, serial);
}
// The budget is time (ticks of the constant-rate device clock, 100 MHz on this part), not a poll count: a count means whatever the memory system's
// latency makes of it.  A gate that ends unopened FAILS CLOSED: it stores the serial of the frame it guarded in that frame slot's poison word,
// and every kernel of the frame exits at its first instruction (frame_poisoned) — the frame is dropped whole, its image untouched, instead of
// being shaded from half-written buffers.  The stream goes on, its own signal kernels included, so the frames behind it are not held up.
__global__ __launch_bounds__(64) void k_handoff_wait(const uint32_t* flag, uint32_t serial, unsigned long long budget_ticks, uint32_t* timeouts_host, uint32_t timeouts_known,
                                                     uint32_t* poison, uint32_t poison_serial) {
    if (threadIdx.x != 0) return;
    if (__hip_atomic_load(timeouts_host, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != timeouts_known) budget_ticks = min(budget_ticks, 2000ull);   // 20 us: the pipeline is already broken
    const unsigned long long t0 = wall_clock64();
    while ((int32_t)(ld_sc1(flag) - serial) < 0) {
        if (wall_clock64() - t0 >= budget_ticks) {
            if (poison) st_sc1(poison, poison_serial);
            __hip_atomic_fetch_add(timeouts_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
        __builtin_amdgcn_s_sleep(8);
    }
}

}  // namespace awsm

extern "C" void awsm_launch_gen_mip_level(uint8_t* chain, uint32_t src_off, uint32_t dst_off, uint32_t sw, uint32_t sh, uint32_t dw, uint32_t dh, uint32_t layers,
                                          const uint32_t* kinds, hipStream_t s) {
    const uint32_t n = dw * dh * layers;
    if (n) hipLaunchKernelGGL(awsm::k_gen_mip_level, dim3((n + 255u) / 256u), dim3(256), 0, s, (uint32_t*)chain, src_off, dst_off, sw, sh, dw, dh, layers, kinds);
}
extern "C" void awsm_launch_upload_words(void* dst, const void* src_pinned, uint32_t n_words, hipStream_t s) {
    const uint32_t nb = (n_words + 255u) / 256u;
    if (nb) hipLaunchKernelGGL(awsm::k_upload_words, dim3(nb < 64u ? nb : 64u), dim3(256), 0, s, (uint32_t*)dst, (const uint32_t*)src_pinned, n_words);
}

// ---- launch wrappers (called from awsm_hip.cpp) ----
extern "C" void awsm_launch_handoff_signal(uint32_t* flag, uint32_t serial, unsigned long long* stamp, hipStream_t s) {
    hipLaunchKernelGGL(awsm::k_handoff_signal, dim3(1), dim3(64), 0, s, flag, serial, stamp);
}
extern "C" void awsm_launch_handoff_wait(const uint32_t* flag, uint32_t serial, unsigned long long budget_ticks, uint32_t* timeouts_host, uint32_t timeouts_known,
                                         uint32_t* poison, uint32_t poison_serial, hipStream_t s) {
    hipLaunchKernelGGL(awsm::k_handoff_wait, dim3(1), dim3(64), 0, s, flag, serial, budget_ticks, timeouts_host, timeouts_known, poison, poison_serial);
}
extern "C" void awsm_launch_hud_merge(const unsigned long long* world, const unsigned long long* hud, unsigned long long* out, size_t first, size_t n, hipStream_t s) {
    if (n) hipLaunchKernelGGL(awsm::k_hud_merge, dim3((unsigned)((n + 255u) / 256u)), dim3(256), 0, s, world, hud, out, first, n);
}
extern "C" void awsm_launch_transform(const awsm::DevScene* sc, const awsm::FrameDev* f, uint32_t n_blocks, hipStream_t s) {
    if (n_blocks) hipLaunchKernelGGL(awsm::k_deform_transform<false>, dim3(n_blocks), dim3(256), 0, s, sc, *f);
}
extern "C" void awsm_launch_transform_forward(const awsm::DevScene* sc, const awsm::FrameDev* f, uint32_t n_blocks, hipStream_t s) {
    if (n_blocks) hipLaunchKernelGGL(awsm::k_deform_transform<true>, dim3(n_blocks), dim3(256), 0, s, sc, *f);
}
extern "C" void awsm_launch_bin_count(const awsm::FrameDev* f, hipStream_t s) {
    const uint32_t per = 256u * awsm::kBinBatches, nb = (f->total_tris - f->rank0 + per - 1u) / per;
    if (nb) hipLaunchKernelGGL(awsm::k_bin<false>, dim3(nb), dim3(256), 0, s, *f);
}
extern "C" void awsm_launch_bin_big(const awsm::FrameDev* f, int fill, hipStream_t s) {
    if (f->total_tris <= f->rank0) return;
    if (fill) return;       // the fill pass walks them inside k_bin<true> (awsm_launch_bin_fill)
    hipLaunchKernelGGL(awsm::k_bin_big<false>, dim3(512), dim3(256), 0, s, *f);
}
// experiment (AWSM_DEBUG_CHAIN_PAD_US): one idle wavefront that holds its stream for so many microseconds — is the frame bound by the length of the
// geometry stream's chain of launches, or by what the kernels of both streams need of the machine?
__global__ void k_debug_pad(unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
extern "C" void awsm_launch_bin_scan(const awsm::FrameDev* f, hipStream_t s) {
#ifdef AWSM_DEBUG_SWITCHES      // tools/build_variants.sh g_debug "-DAWSM_DEBUG_SWITCHES": the product library reads no debug switch
    static const long pad_us = getenv("AWSM_DEBUG_CHAIN_PAD_US") ? atol(getenv("AWSM_DEBUG_CHAIN_PAD_US")) : 0;
    if (pad_us > 0) hipLaunchKernelGGL(k_debug_pad, dim3(1), dim3(64), 0, s, (unsigned long long)pad_us * 100ull);      // wall_clock64: 100 MHz
#endif
    const uint32_t n_tiles = f->tiles_x * f->tiles_y;
    if (n_tiles) hipLaunchKernelGGL(awsm::k_bin_scan, dim3((n_tiles + awsm::kScanThreads - 1u) / awsm::kScanThreads), dim3(awsm::kScanThreads), 0, s, *f, n_tiles);
}
extern "C" void awsm_launch_bin_fill(const awsm::FrameDev* f, hipStream_t s) {
    const uint32_t per = 256u * awsm::kBinBatches, nb = (f->total_tris - f->rank0 + per - 1u) / per;
    if (nb) hipLaunchKernelGGL(awsm::k_bin<true>, dim3(nb + awsm::kBinBigBlocks), dim3(256), 0, s, *f);
}
extern "C" void awsm_launch_raster(const awsm::FrameDev* f, hipStream_t s) {
    const uint32_t n_tiles = f->tiles_x * f->tiles_y;
    if (!n_tiles) return;
    const uint32_t nb = n_tiles + (f->raster_scratch ? f->raster_extra_cap : 0u);     // surplus ids exit at once
    if (f->msaa == 4u) hipLaunchKernelGGL(awsm::k_raster_tile<4>, dim3(nb), dim3(256), awsm::kTile * awsm::kTile * 4 * sizeof(unsigned long long), s, *f);      // the tile: dynamic LDS
    else hipLaunchKernelGGL(awsm::k_raster_tile<1>, dim3(nb), dim3(256), 0, s, *f);
}
