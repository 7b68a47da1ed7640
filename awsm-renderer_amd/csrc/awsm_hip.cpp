// awsm_hip.cpp — context, device-memory management and the C-ABI entry points of include/awsm_hip.h.
//
// Sits where the reference's `AwsmRendererWebGpu` wrapper sits (crates/renderer-core/src/methods.rs:
// create_buffer :239, write_buffer :339-431, submit_commands :283-287) but knows the two passes of the hot
// path (crates/renderer/src/render.rs:144-221).  No torch, no WebGPU, no CPU fallback: every entry point
// either drives the HIP kernels or fails with a status code.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>

#include "frame_params.hpp"
#include "dirty_log.hpp"

using namespace awsm;

extern "C" {
void awsm_launch_bin_big(const FrameDev* f, int fill, hipStream_t s);
void awsm_launch_gen_mip_level(uint8_t* chain, uint32_t src_off, uint32_t dst_off, uint32_t sw, uint32_t sh, uint32_t dw, uint32_t dh, uint32_t layers,
                               const uint32_t* kinds, hipStream_t s);
void awsm_launch_pick(const DevScene* sc, const FrameDev* f, int x, int y, uint32_t* out, hipStream_t s);
void awsm_launch_transform(const DevScene* sc, const FrameDev* f, uint32_t n_blocks, hipStream_t s);
void awsm_launch_hud_merge(const unsigned long long* world, const unsigned long long* hud, unsigned long long* out, size_t first, size_t n, hipStream_t s);
void awsm_launch_transform_forward(const DevScene* sc, const FrameDev* f, uint32_t n_blocks, hipStream_t s);
void awsm_launch_forward(const DevScene* sc, const FrameDev* f, hipStream_t s);
void awsm_launch_upload_words(void* dst, const void* src_pinned, uint32_t n_words, hipStream_t s);
void awsm_launch_bin_count(const FrameDev* f, hipStream_t s);
void awsm_launch_bin_scan(const FrameDev* f, hipStream_t s);
void awsm_launch_bin_fill(const FrameDev* f, hipStream_t s);
void awsm_launch_raster(const FrameDev* f, hipStream_t s);
void awsm_launch_shade(const DevScene* sc, const FrameDev* f, hipStream_t s);
int awsm_shade_is_lean(const FrameDev* f);
int awsm_launch_shade_todo(const DevScene* sc, const FrameDev* f, hipStream_t s);
void awsm_launch_gbuffer_dump(const FrameDev* f, float* out, hipStream_t s);
void awsm_launch_handoff_signal(uint32_t* flag, uint32_t serial, unsigned long long* stamp, hipStream_t s);
void awsm_launch_handoff_wait(const uint32_t* flag, uint32_t serial, unsigned long long budget_ticks, uint32_t* timeouts_host, uint32_t timeouts_known,
                              uint32_t* poison, uint32_t poison_serial, hipStream_t s);
void awsm_launch_resolve_draws(const DevScene* sc, const FrameDev* f, hipStream_t s);
void awsm_launch_count_covered(const FrameDev* f, hipStream_t s);
void awsm_launch_msaa_halo_export(const FrameDev* f, unsigned long long* dst, uint32_t bands_out, hipStream_t s);
void awsm_launch_vis_digest(const unsigned long long* vis, size_t n, unsigned long long* out, hipStream_t s);
void awsm_launch_brdf_lut(uint32_t* out_rg16f, uint32_t w, uint32_t h, hipStream_t s);
void awsm_launch_cube_border(const awsm::CubeDev* cd, uint2* out, uint32_t total, hipStream_t s);
void awsm_launch_rgba16f_to_rg16f(const uint16_t* in, uint32_t* out, uint32_t n, hipStream_t s);
}

namespace {

// Frame slots of the overlap mode: frame i's opaque pass reads slot i % kSlots while the geometry pass of frame i + 1 fills the next.  (Three
// slots — the geometry pass of frame i + 2 no longer waiting for frame i's opaque pass — measured no different, and cost a fifth stream.)
#ifndef AWSM_SLOTS
#define AWSM_SLOTS 2
#endif
constexpr int kSlots = AWSM_SLOTS;
constexpr int kLeanWgsPerCu = 0;   // measured: the one-wavefront-per-strip grid wins once the geometry kernels fit beside it
struct DevBuf {
    void* ptr = nullptr;
    size_t size = 0;
};

enum { EV_START = 0, EV_TRANSFORM, EV_BIN, EV_RASTER, EV_SHADE_BEGIN, EV_SHADE_LEAN, EV_SHADE, EV_FWD_BEGIN, EV_FWD, EV_COUNT };

// Everything the geometry pass produces for one frame and the opaque pass consumes.
struct FrameBufs {
    DevBuf wpos;                           // transparent pass only
    DevBuf frag_rec, frag_color, frag_first;   // transparent pass: per-pixel fragment lists
    uint32_t frag_cap = 0;
    DevBuf tex_slots;                      // n_draws x kCoreTextures TexSlotDev (k_resolve_draws)
    DevBuf draw_mat;                       // n_draws DrawMatDev (k_resolve_draws)
    DevBuf tri_shade, draw_lean;           // geometry pass: per-triangle attribute offsets (k_deform_transform), per-draw lean records (k_resolve_draws)
    DevBuf clip, nrm, tan, tri_rec, tri_flags, draws_dev, draw_shade, tile_count, tile_offset, tile_cursor, tile_order, bin_list, big_list, counters, vis;
    DevBuf camera;                         // snapshot of the camera UBO taken by the geometry pass (overlap mode)
    DevBuf scan_tmp;                       // k_bin_scan's cross-workgroup tables
    DevBuf tile_split, raster_scratch;     // split raster tiles: per-tile {first scratch slot, slices done}; partial tiles (geometry pass only)
    uint32_t raster_extra_cap = 0, raster_slot_cap = 0;
    uint32_t bin_capacity = 0;
    float pix2view[16] = {}, view_rot[9] = {}, cam_pos[3] = {}, ortho_view_dir[3] = {};   // the lean opaque kernel's view of the camera this slot's frame was submitted with
    uint32_t cam_ortho = 0;
    size_t tri_cap = 0, draw_cap = 0;      // triangles / draws the per-pass buffers below are sized for (reserve_pass_buffers)
    uint32_t sized_w = 0, sized_h = 0;     // ... and the frame size their tile tables are sized for
    std::vector<DrawDev> draws_uploaded;   // what draws_dev currently holds
    void* draws_uploaded_ptr = nullptr;
    bool draws_uploaded_valid = false;
    uint64_t draws_version = 0;            // bumped whenever draws_dev receives a new list
    // MSAA frames with hud meshes: the hud draws behind the world's in draws_dev (awsm_hip_hud_geometry_pass), cached by themselves so that the world
    // list's own cache keeps hitting
    std::vector<DrawDev> tail_uploaded;
    void* tail_uploaded_ptr = nullptr;
    size_t tail_uploaded_at = 0;
    bool tail_uploaded_valid = false;
    // Geometry cache (frame_params.hpp; the world geometry pass only): the draw list this slot's wcache / nrm / tan / tri_shade / tri_info were last
    // computed for lives on in draws_prev when a new list arrives (the two buffers swap), so that k_deform_transform can compare draw by draw.
    DevBuf wcache, draws_prev;
    DevBuf block_map;                      // world geometry pass: draw index of every k_deform_transform workgroup (uploaded with the draw list; spares each workgroup a binary search over the list = log2(n) dependent loads in front of everything else)
    bool block_map_valid = false;
    DevBuf cache_mark;                     // one word per k_deform_transform workgroup: the serial of the last frame in which it took the cached path (statistics only)
    uint32_t cached_n_draws = 0;           // draws of the list the arrays were computed for
    bool cache_valid = false;              // ... and whether they were (false after any re-allocation, a frame without geometry, a dropped frame)
    bool cached_is_prev = false;           // that list is in draws_prev (a new list has been uploaded since), else it is draws_dev's
    bool cache_had_tri_shade = false;
    uint64_t cache_seq = 0;                // write_seq when that geometry pass was enqueued: dirty ranges logged after it apply
    uint32_t cache_serial = 0;             // its frame serial
};

}  // namespace

struct AwsmHipCtx {
    int device = 0;
    uint32_t flags = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    bool stage_timers = true;      // hipEventRecord between the stages of a frame (AwsmFrameStats.ms_*); awsm_hip_set_stage_timers
    std::string last_error;

    DevBuf bufs[AWSM_BUF_COUNT];
    bool camera_written_since_snapshot = false;      // awsm_hip_buffer_write(AWSM_BUF_CAMERA) since the last geometry pass took its snapshot
    uint8_t camera_host[512] = {};   // what the caller last wrote to AWSM_BUF_CAMERA (awsm_hip_buffer_write): compose_pixel_to_view reads it
    DevScene scene{};            // host copy
    DevScene* scene_dev = nullptr;
    bool scene_dirty = true;
    DevBuf tex[kMaxTexArrays];
    DevBuf lut;
    DevBuf cube_tex[3], cube_bordered[3];      // the uploaded chains; the same with a one-texel apron per face (CubeDev.bordered)

    // frame targets
    uint32_t width = 0, height = 0;
    uint32_t y0 = 0, y1 = 0;     // shard rows (0,0 = full)
    uint32_t band_n = 1, band_r = 0, band_compact = 0;   // shard bands (awsm_hip_set_shard_bands)
    uint32_t msaa = 0;           // 0 or 4 (awsm_hip_resize)
    DevBuf mip_kinds;                 // scratch for awsm_hip_texture_array_generate_mips
    // MSAA scratch, per frame slot (frame i + 1's k_shade_lean may run beside frame i's k_msaa_detect / k_shade_msaa_resolve):
    DevBuf msaa_color0[kSlots], msaa_edges[kSlots];   // f32 colour of sample 0 per pixel; per block [count, one-byte pixel slots...] of edge pixels
    DevBuf msaa_edge_bits[kSlots];    // lean route: two u64 per 16x4 strip (FrameDev.msaa_edge_bits)
    DevBuf msaa_cells[kSlots];        // lean route: normal + depth of sample 0 per pixel (FrameDev.msaa_cells)
    DevBuf out16[kSlots], out32[kSlots];        // the opaque image (+ f32 parity tap) per frame slot: with two images two frames' opaque passes need no order between them
    DevBuf digest;                    // 2 x u64 (awsm_hip_visibility_digest)
    uint32_t lean_grid = 0;           // persistent k_shade_lean grid (workgroups), 0 = one workgroup per block
    DevBuf shade_todo[kSlots];             // per frame slot (the per-draw resolve of frame i + 1 resets one while frame i's opaque pass appends to the other); [0] count + entries: the 16x4-pixel groups the lean opaque kernel leaves to the general one
    void* bound_out = nullptr;
    size_t bound_out_bytes = 0;
    const uint8_t* slot_out_lo[kSlots] = {}; const uint8_t* slot_out_hi[kSlots] = {};     // the bound image each slot's last opaque pass wrote (null: the library's own)

    // geometry-pass resources
    // Frame overlap (AWSM_CFG_OVERLAP_FRAMES): the opaque pass of frame i runs on shade_stream while the caller's stream
    // already runs the geometry pass of frame i+1 into the other slot.
    bool overlap = false;
    hipStream_t shade_streams[kSlots] = {};     // one per frame slot: the waits in front of frame i + 1's opaque pass are consumed while frame i's still runs (enqueue_opaque)
    bool shade_recorded[kSlots] = {};
    bool uploads_recorded[kSlots] = {};  // ev_uploads[slot] was recorded by this frame's geometry pass
    // What a slot's per-draw records (k_resolve_draws: draw_shade / draw_mat / tex_slots / draw_lean / lights_pre) were computed from: they are
    // recomputed only when that changed — a frame that moved nothing but the camera reuses them (no kernel, no event).
    struct ResolveKey { uint64_t write_seq, draws_version; const void* ptrs[5]; uint32_t n_draws, mipmap, has_opaque, lights_cap; } resolved[kSlots] = {};   // ev_shade_done[slot] has been recorded since the last full synchronisation
    hipEvent_t ev_geom_done[kSlots] = {}, ev_shade_done[kSlots] = {}, ev_uploads[kSlots] = {} ;
    hipEvent_t ev_flush = nullptr;             // awsm_hip_frame_flush: system-scope release in front of what the CALLER enqueues next (a collective, a peer / host copy)
    uint64_t write_seq = 0, geom_write_seq[kSlots] = {};     // scene writes so far / at the time the slot's geometry pass was enqueued
    // Geometry cache: the byte ranges written (awsm_hip_buffer_write / buffer_create) to the buffers k_deform_transform reads, each with the write_seq it got;
    // a slot's geometry pass hands the kernel those newer than the slot's cache_seq.  all_dirty_seq: everything up to that write_seq counts as "all
    // written" (the log overflowed, or a buffer was written whose every write invalidates every draw).
    DirtyLog dirty_log;                         // (dirty_log.hpp)
    bool geometry_cache = true;                 // AWSM_GEOMETRY_CACHE=0 turns it off (A/B measurements, tests)
    uint32_t cache_blocks_last = 0;             // AwsmFrameStats.geometry_cache_blocks of the frame frame_end last looked at
    bool shade_pending[kSlots] = {};
    // Device-side hand-off between the streams (kernels_geometry.hip: k_handoff_signal / k_handoff_wait) in place of the two cross-stream events
    // on a frame's critical path: geometry pass -> opaque pass of the same frame, opaque pass of frame i -> geometry pass of frame i + kSlots.
    bool handoff = false;
    uint32_t* handoff_flags = nullptr;          // device: [slot] geometry done, [kSlots + slot] shading done, [2 kSlots + slot] k_shade_lean done — the serial number last signalled
    uint32_t geom_sig[kSlots] = {}, shade_sig[kSlots] = {}, lean_sig[kSlots] = {};   // flags [2 kSlots + slot]: the slot's k_shade_lean has ended (stored by k_shade_todo as it starts)
    bool lean_flagged[kSlots] = {};             // the slot's last opaque pass took the lean route with the flag: the next frame's pass may start on lean_sig
    unsigned long long handoff_budget_ticks = 400000000ull;   // a gate's time budget in device-clock ticks (AWSM_HANDOFF_TIMEOUT_MS; default 4 s at 100 MHz):
                                                // far longer than any frame or host stall, short enough that a gate nobody opens ends
    uint32_t clock_khz = 100000;                // hipDeviceAttributeWallClockRate
    uint32_t* handoff_poison = nullptr;         // device: [slot] serial of the slot's last frame whose gate timed out (FrameDev.poison)
    uint32_t handoff_timeouts_seen = 0;
    uint32_t handoff_dropped_frames = 0;        // gates that timed out = frames dropped whole (AwsmFrameStats.frames_dropped_by_handoff)
    bool handoff_error_pending = false;         // a time-out has been counted but not yet reported to the caller
    uint32_t handoff_test_drop = 0;             // AWSM_TEST_HANDOFF_DROP: that many geometry-done signals are withheld (tests of the timeout path)
    FrameBufs fb[kSlots];             // per-frame device state; two slots when frames overlap (AWSM_CFG_OVERLAP_FRAMES), else slot 0 only
    FrameBufs tr[kSlots];             // the same for the transparent pass's own draws (vertices, setup records, bins); no visibility buffer
    FrameBufs hud[kSlots];            // ... and for the HUD geometry pass (render.rs:169-178): its own vertices, bins and visibility keys (= hud_depth + the hud triangles)
    std::vector<DrawDev> hud_draws_host;
    uint32_t hud_total_tris = 0, hud_n_blocks = 0;
    bool hud_geometry_done = false;   // this frame has hud geometry: the opaque pass leaves its pixels cleared
    // MSAA frames: the hud draws follow the world's in ONE rank space (transformed into the world pass's vertex / setup arrays behind the world's, binned and
    // rasterised by themselves into hud[slot].vis), and the opaque pass reads the merged keys (k_hud_merge: hud rank under world depth) — what the
    // reference's targets hold after its HUD geometry pass.  hud_merged: this frame's hud pass went that way.
    bool hud_merged = false;
    DevBuf merged_vis[kSlots];
    std::vector<DrawDev> hud_combined;   // world draws, then the hud draws with first_tri / first_block continued
    bool hud_transparent = false;     // the transparent pass being enqueued is the HUD one (depth cleared, colours loaded from the composite)
    // [0] the world transparent pass, [1] the HUD transparent pass: each with its own per-slot device state (tr / htr) — the HUD pass of a frame is enqueued
    // while the world pass of the same frame may not have started on the shade stream, so they share neither a draw list nor counters
    std::vector<DrawDev> tr_draws_host[2];
    uint32_t tr_total_tris[2] = {0, 0}, tr_n_blocks[2] = {0, 0};
    bool transparent_done = false, hud_transparent_done = false;
    FrameBufs htr[kSlots];
    DevBuf comp16, comp32;       // composite image (after the transparent pass) + parity tap
    DevBuf lights_pre[kSlots];        // per frame slot: per-light constants (k_resolve_draws), sized with the lights buffer
    void* bound_comp = nullptr;
    size_t bound_comp_bytes = 0;
    const void* msaa_halo = nullptr;       // awsm_hip_msaa_halo_bind
    size_t msaa_halo_bytes = 0;
    const void* opaque_src = nullptr;      // awsm_hip_bind_opaque_source: the gathered full-frame opaque image (sharded transparent pass)
    size_t opaque_src_bytes = 0;
    int slot = 0;
    std::vector<DrawDev> draws_host;
    std::vector<AwsmDraw> draws_api;
    uint32_t total_tris = 0, total_verts = 0, n_blocks = 0;
    bool geometry_done = false, opaque_done = false;
    AwsmOpaqueParams last_opaque{};
    uint32_t overflow_retries = 0;
    bool has_opaque_for_stats() const { return !opaque_done || last_opaque.has_opaque != 0; }

    // pinned staging ring for buffer_write / small uploads
    uint8_t* stage = nullptr;
    size_t stage_cap = 0, stage_head = 0;
    uint32_t* counters_host = nullptr;   // pinned, 16 u32: [0..8) geometry pass, [8..16) transparent pass; then kSlots x {entries needed, frame serial} (k_bin_scan)
    uint32_t frame_serial = 0, status_seen[kSlots] = {};
    uint32_t dropped_frames = 0;         // enqueue-only frames that overflowed their bin list
    uint32_t out_first_row = 0;          // awsm_hip_bind_output_rows
    bool out_rows_mode = false;

    hipEvent_t ev[EV_COUNT] = {};
    bool ev_valid[EV_COUNT] = {};

    // frame trace (awsm_hip_frame_trace): device-clock stamps [3][trace_cap] — geometry pass begins / geometry pass done / shading done —
    // of frame `serial` at index serial % trace_cap, written by one-lane kernels in stream order (the hand-off's signal kernels where they exist)
    unsigned long long* trace_dev = nullptr;
    uint32_t trace_cap = 0;
};

namespace {

inline FrameBufs& FB(AwsmHipCtx* c) { return c->fb[c->slot]; }
inline hipStream_t shade_stream_of(AwsmHipCtx* c) { return c->overlap ? c->shade_streams[c->slot] : c->stream; }
inline int n_slots(const AwsmHipCtx* c) { return c->overlap ? kSlots : 1; }
inline int prev_slot(const AwsmHipCtx* c) { return (c->slot + kSlots - 1) % kSlots; }
inline uint32_t* handoff_timeouts(AwsmHipCtx* c) { return c->counters_host + 16 + 2 * kSlots; }
inline unsigned long long* trace_slot(AwsmHipCtx* c, int which) { return c->trace_dev ? c->trace_dev + (size_t)which * c->trace_cap + c->frame_serial % c->trace_cap : nullptr; }
// the slot's shading is finished: for the host and the rarely taken waits an event, for the next user of the slot's buffers the flag
inline hipError_t mark_shade_done(AwsmHipCtx* c, hipStream_t ss) {
    if (c->handoff) awsm_launch_handoff_signal(c->handoff_flags + kSlots + c->slot, ++c->shade_sig[c->slot], trace_slot(c, 2), ss);
    else if (c->trace_dev) awsm_launch_handoff_signal(nullptr, 0u, trace_slot(c, 2), ss);
    const hipError_t e = hipEventRecord(c->ev_shade_done[c->slot], ss);
    c->shade_pending[c->slot] = true; c->shade_recorded[c->slot] = true;
    return e;
}
// a shade stream goes on once the OTHER slot's passes have finished (one bound image for both frames, bound halo keys, the composite) —
// or, main_kernel_only, once its k_shade_lean has (its k_shade_todo may still run: per-slot images and lists)
inline hipError_t wait_prev_slot(AwsmHipCtx* c, hipStream_t ss, bool main_kernel_only = false) {
    const int p = prev_slot(c);
    if (c->handoff && c->shade_sig[p]) {
        const bool lean = main_kernel_only && c->lean_flagged[p];
        awsm_launch_handoff_wait(c->handoff_flags + (lean ? 2 * kSlots : kSlots) + p, lean ? c->lean_sig[p] : c->shade_sig[p], c->handoff_budget_ticks, handoff_timeouts(c), c->handoff_timeouts_seen,
                                 c->handoff_poison + c->slot, c->frame_serial, ss);
        return hipSuccess;
    }
    return hipStreamWaitEvent(ss, c->ev_shade_done[p], 0);
}
// the caller's stream goes on once the slot's last opaque (or transparent) pass has finished
inline hipError_t wait_slot_free(AwsmHipCtx* c) {
    if (c->handoff) { awsm_launch_handoff_wait(c->handoff_flags + kSlots + c->slot, c->shade_sig[c->slot], c->handoff_budget_ticks, handoff_timeouts(c), c->handoff_timeouts_seen,
                                               c->handoff_poison + c->slot, c->frame_serial, c->stream); return hipSuccess; }
    return hipStreamWaitEvent(c->stream, c->ev_shade_done[c->slot], 0);
}
inline hipError_t sync_shade_streams(AwsmHipCtx* c) {
    for (hipStream_t s : c->shade_streams) if (s) { const hipError_t e = hipStreamSynchronize(s); if (e != hipSuccess) return e; }
    for (bool& b : c->shade_recorded) b = false;
    return hipSuccess;
}

// AWSM_HOST_TRACE=<microseconds>: report (stderr) every section of an enqueue call that kept the host longer than that — the HIP runtime
// growing a pool, a full queue, a synchronisation nobody asked for.  Diagnostic; off by default.
struct HostTrace {
    static long threshold() { static long t = [] { const char* e = getenv("AWSM_HOST_TRACE"); return e ? atol(e) : 0L; }(); return t; }
    timespec last{};
    uint32_t serial;
    explicit HostTrace(uint32_t frame_serial) : serial(frame_serial) { if (threshold()) clock_gettime(CLOCK_MONOTONIC, &last); }
    void mark(const char* what) {
        if (!threshold()) return;
        timespec now; clock_gettime(CLOCK_MONOTONIC, &now);
        const long us = (now.tv_sec - last.tv_sec) * 1000000L + (now.tv_nsec - last.tv_nsec) / 1000L;
        if (us >= threshold()) fprintf(stderr, "awsm_hip host trace: frame %u: %s took %ld us\n", serial, what, us);
        last = now;
    }
};

int fail(AwsmHipCtx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->last_error = buf;
    return code;
}

// A hand-off gate that ran out of time has dropped the frame it guarded (fail closed: k_handoff_wait poisons it, its kernels exit).  The host
// learns of it from a pinned counter: count the gates, order the streams with events from here on, and tell the caller once — at the next
// awsm_hip_geometry_pass, awsm_hip_frame_flush or awsm_hip_frame_end, whichever comes first, so that an enqueue-only loop hears of it too.
int handoff_check(AwsmHipCtx* c) {
    if (!c->overlap) return AWSM_OK;
    const uint32_t now = *(volatile uint32_t*)handoff_timeouts(c);
    if (now != c->handoff_timeouts_seen) {
        c->handoff_dropped_frames += now - c->handoff_timeouts_seen;
        c->handoff_timeouts_seen = now;
        c->handoff = false;
        c->handoff_error_pending = true;
        for (FrameBufs& b : c->fb) b.cache_valid = false;      // a dropped frame's kernels wrote nothing: the geometry cache's book-keeping no longer describes the arrays
    }
    if (!c->handoff_error_pending) return AWSM_OK;
    c->handoff_error_pending = false;
    return fail(c, AWSM_ERR_DEVICE, "a device-side stream hand-off timed out (kernels serialised by a profiler, the streams folded onto one hardware queue, or a stall longer than "
                                    "AWSM_HANDOFF_TIMEOUT_MS): %u gate(s) so far, each dropped the frame it guarded whole — that frame's image was not written; this context orders "
                                    "its streams with events from here on (AWSM_DEVICE_HANDOFF=0 selects them from the start)", c->handoff_dropped_frames);
}

#define HIPCHK(c, call)                                                                            \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail((c), e_ == hipErrorOutOfMemory ? AWSM_ERR_OUT_OF_MEMORY : AWSM_ERR_DEVICE, \
                        "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

int dev_realloc(AwsmHipCtx* c, DevBuf& b, size_t bytes, bool zero) {
    if (b.ptr && b.size == bytes) {
        if (zero) HIPCHK(c, hipMemsetAsync(b.ptr, 0, bytes, c->stream));
        return AWSM_OK;
    }
    if (b.ptr) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, sync_shade_streams(c));
        HIPCHK(c, hipFree(b.ptr));
        b.ptr = nullptr; b.size = 0;
    }
    if (bytes == 0) return AWSM_OK;
    HIPCHK(c, hipMalloc(&b.ptr, bytes));
    b.size = bytes;
    if (zero) HIPCHK(c, hipMemsetAsync(b.ptr, 0, bytes, c->stream));
    return AWSM_OK;
}

int dev_reserve(AwsmHipCtx* c, DevBuf& b, size_t bytes) {   // grow-only, contents not preserved
    if (b.size >= bytes && b.ptr) return AWSM_OK;
    size_t want = std::max(bytes, b.size + b.size / 2);
    return dev_realloc(c, b, want, false);
}

constexpr size_t kKernelUploadMax = 256u << 10;   // larger uploads go to the copy engine

// returns a pinned pointer valid until the copy enqueued from it has executed
int stage_alloc(AwsmHipCtx* c, size_t len, uint8_t** out) {
    len = (len + 255) & ~size_t(255);
    if (len > c->stage_cap) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->stage) HIPCHK(c, hipHostFree(c->stage));
        c->stage = nullptr;
        size_t cap = std::max<size_t>(len * 2, 8u << 20);
        HIPCHK(c, hipHostMalloc((void**)&c->stage, cap, hipHostMallocDefault));
        c->stage_cap = cap; c->stage_head = 0;
    }
    if (c->stage_head + len > c->stage_cap) {
        HIPCHK(c, hipStreamSynchronize(c->stream));   // everything enqueued from the ring has been consumed
        c->stage_head = 0;
    }
    *out = c->stage + c->stage_head;
    c->stage_head += len;
    return AWSM_OK;
}

int upload_small(AwsmHipCtx* c, void* dst, const void* src, size_t len) {
    uint8_t* st;
    int rc = stage_alloc(c, len, &st);
    if (rc) return rc;
    memcpy(st, src, len);
    if (len <= kKernelUploadMax && (len & 3u) == 0 && ((uintptr_t)dst & 3u) == 0 && ((uintptr_t)st & 3u) == 0) {
        awsm_launch_upload_words(dst, st, (uint32_t)(len >> 2), c->stream);   // pinned memory is device-visible at the same address
        HIPCHK(c, hipGetLastError());
    } else {
        HIPCHK(c, hipMemcpyAsync(dst, st, len, hipMemcpyHostToDevice, c->stream));
    }
    return AWSM_OK;
}

void shard(const AwsmHipCtx* c, uint32_t* y0, uint32_t* y1) {
    *y0 = c->y0; *y1 = c->y1;
    if (*y1 == 0 || *y1 > c->height) *y1 = c->height;
    if (*y0 > *y1) *y0 = *y1;
}

// Geometry cache: remember what was written to the buffers k_deform_transform reads (call after the write got its write_seq).
void log_dirty(AwsmHipCtx* c, AwsmBuf which, size_t lo, size_t hi) {
    if (c->geometry_cache) c->dirty_log.log((uint32_t)which, lo, hi, c->write_seq);
}

// Overlap mode: anything that writes scene state the opaque pass reads (every buffer but the camera, whose snapshot the
// geometry pass takes; textures; samplers; environment; the DevScene table) is ordered after the opaque passes still in
// flight on the shade stream.  A frame therefore always shades the scene as it was when it was submitted.
int scene_write_barrier(AwsmHipCtx* c, bool is_write = true) {
    if (is_write) c->write_seq++;
    if (!c->overlap) return AWSM_OK;
    for (int s = 0; s < kSlots; s++)
        if (c->shade_pending[s]) { HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_shade_done[s], 0)); c->shade_pending[s] = false; }
    return AWSM_OK;
}
int sync_all(AwsmHipCtx* c) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, sync_shade_streams(c));
    for (bool& b : c->shade_pending) b = false;
    return AWSM_OK;
}

int sync_scene(AwsmHipCtx* c) {
    if (!c->scene_dirty) return AWSM_OK;
    c->write_seq++;
    { int rcb = scene_write_barrier(c, false); if (rcb) return rcb; }
    for (int i = 0; i < AWSM_BUF_COUNT; i++) c->scene.buf[i] = (const uint8_t*)c->bufs[i].ptr;
    c->scene.lut_rg16f = (const uint16_t*)c->lut.ptr;
    int rc = upload_small(c, c->scene_dev, &c->scene, sizeof(DevScene));
    if (rc) return rc;
    c->scene_dirty = false;
    return AWSM_OK;
}

void fill_frame(AwsmHipCtx* c, FrameDev* f) {
    memset(f, 0, sizeof *f);
    uint32_t y0, y1;
    shard(c, &y0, &y1);
    f->width = c->width; f->height = c->height; f->y0 = y0; f->y1 = y1; f->sy0 = y0; f->sy1 = y1;
    if (c->msaa && (y0 > 0 || y1 < c->height)) {   // MSAA edge detector reads the rows next to the shard: rasterise one halo row each side
        f->y0 = y0 > 0 ? y0 - 1 : 0; f->y1 = std::min(c->height, y1 + 1);
    }
    f->tiles_x = (c->width + kTile - 1) / kTile;
    f->band_n = 1; f->band_r = 0; f->out_compact = 0;
    f->tile_row0 = f->y0 >> kTileShift;
    f->tiles_y = (f->y1 > f->y0) ? ((f->y1 + kTile - 1) / kTile - f->tile_row0) : 0;
    if (c->band_n > 1) {   // band mode: every band_n-th 32-row tile row, starting at band_r (row range is the full frame)
        const uint32_t rows_full = (c->height + kTile - 1) / kTile;
        f->band_n = c->band_n; f->band_r = c->band_r; f->out_compact = c->band_compact;
        f->tile_row0 = c->band_r;
        f->tiles_y = c->band_r < rows_full ? (rows_full - c->band_r + c->band_n - 1) / c->band_n : 0;
    }
    f->n_draws = (uint32_t)c->draws_host.size();
    f->total_tris = c->total_tris; f->total_verts = c->total_verts;
    f->bin_capacity = FB(c).bin_capacity;
    f->draws = (const DrawDev*)FB(c).draws_dev.ptr;
    f->draw_shade = (DrawShadeDev*)FB(c).draw_shade.ptr;
    f->tex_slots = (TexSlotDev*)FB(c).tex_slots.ptr;
    f->draw_mat = (DrawMatDev*)FB(c).draw_mat.ptr;
    f->clip = (float4*)FB(c).clip.ptr; f->nrm = (float4*)FB(c).nrm.ptr; f->tan = (float4*)FB(c).tan.ptr;
    f->tri_info = (uint32_t*)FB(c).tri_flags.ptr;
    // The lean opaque route (k_shade_lean): frames whose per-pixel key / per-triangle / per-vertex / attribute byte offsets fit 32 bits
    const bool lean_ok = !(c->flags & AWSM_CFG_GENERAL_SHADE_ONLY) && (uint64_t)c->width * c->height * (c->msaa == 4 ? 32u : 8u) < (1ull << 32) && (c->msaa == 0 || (c->msaa_edge_bits[c->slot].ptr && c->msaa_cells[c->slot].ptr)) && FB(c).tri_shade.ptr && FB(c).draw_lean.ptr && c->shade_todo[c->slot].ptr && ((uint64_t)c->total_tris + (c->hud_merged ? c->hud_total_tris : 0u)) * kTriRecBytes < (1ull << 32) &&
                         c->bufs[AWSM_BUF_ATTR_DATA].size < (1ull << 32) - 64;
    f->tri_shade = lean_ok ? (uint4*)FB(c).tri_shade.ptr : nullptr;
    f->attr_data_bytes = (uint32_t)std::min<size_t>(c->bufs[AWSM_BUF_ATTR_DATA].size, 0xFFFFFFFFu);
    f->draw_lean = lean_ok ? (LeanDrawDev*)FB(c).draw_lean.ptr : nullptr;
    f->shade_todo = (uint32_t*)c->shade_todo[c->slot].ptr;
    f->shade_todo_cap = f->shade_todo ? (uint32_t)(c->shade_todo[c->slot].size / 4 - 4 - 1024) : 0u;
    f->lean_next = f->shade_todo ? f->shade_todo + 4 + f->shade_todo_cap : nullptr;
    f->lean_grid = c->lean_grid;
    f->tri_rec = (TriRec*)FB(c).tri_rec.ptr;
    f->tile_count = (uint32_t*)FB(c).tile_count.ptr; f->tile_offset = (uint32_t*)FB(c).tile_offset.ptr;
    f->tile_cursor = (uint32_t*)FB(c).tile_cursor.ptr; f->bin_list = (uint32_t*)FB(c).bin_list.ptr;
    f->tile_order = (uint32_t*)FB(c).tile_order.ptr;
    f->scan_tmp = (uint32_t*)FB(c).scan_tmp.ptr;
    f->tile_split = (uint32_t*)FB(c).tile_split.ptr; f->raster_scratch = (unsigned long long*)FB(c).raster_scratch.ptr;
    f->raster_extra_cap = FB(c).raster_extra_cap; f->raster_slot_cap = FB(c).raster_slot_cap;
    f->big_list = (uint32_t*)FB(c).big_list.ptr;
    f->counters = (uint32_t*)FB(c).counters.ptr;
#ifdef AWSM_STAMP
    {
        static unsigned long long* stamps = nullptr;
        if (!stamps && hipMalloc((void**)&stamps, 4ull * 16384 * 8 * 8) == hipSuccess) (void)hipMemset(stamps, 0, 4ull * 16384 * 8 * 8);
        f->stamps = stamps;
        static bool told = false;
        if (!told) { told = true; FILE* fp = fopen("/tmp/awsm_stamps_ptr", "w"); if (fp) { fprintf(fp, "%llu\n", (unsigned long long)(uintptr_t)stamps); fclose(fp); } }
    }
#endif
    f->poison = c->handoff_poison ? c->handoff_poison + c->slot : nullptr;
    f->host_bin_status = c->counters_host + 16 + 2 * c->slot;
    f->frame_serial = c->frame_serial;
    f->vis = (unsigned long long*)FB(c).vis.ptr;
    f->camera = (const uint8_t*)FB(c).camera.ptr;      // the snapshot the geometry pass took: every kernel of a frame, lean or general, sees the camera the frame was submitted with
    f->camera_snap = nullptr; f->camera_snap_words = 0;
    memcpy(f->pix2view, FB(c).pix2view, sizeof f->pix2view); memcpy(f->view_rot, FB(c).view_rot, sizeof f->view_rot); memcpy(f->cam_pos, FB(c).cam_pos, sizeof f->cam_pos);
    memcpy(f->ortho_view_dir, FB(c).ortho_view_dir, sizeof f->ortho_view_dir); f->cam_ortho = FB(c).cam_ortho;
    f->msaa = c->msaa;
    f->msaa_color0 = (float4*)c->msaa_color0[c->slot].ptr;
    f->msaa_edges = (uint32_t*)c->msaa_edges[c->slot].ptr;
    f->msaa_edge_bits = (lean_ok && c->msaa == 4) ? (unsigned long long*)c->msaa_edge_bits[c->slot].ptr : nullptr;
    f->msaa_cells = (uint2*)c->msaa_cells[c->slot].ptr;
    f->hud_vis = c->hud_geometry_done ? (const unsigned long long*)c->hud[c->slot].vis.ptr : nullptr;
    f->hud_draws = (const DrawDev*)c->hud[c->slot].draws_dev.ptr; f->hud_tri_info = (const uint32_t*)c->hud[c->slot].tri_flags.ptr;
    if (c->hud_geometry_done && c->hud_merged) {      // one rank space: the hud draws behind the world's; the opaque pass sees the merged keys
        f->n_draws = (uint32_t)c->hud_combined.size();
        f->total_tris = c->total_tris + c->hud_total_tris; f->total_verts = 3u * f->total_tris;
        f->vis = (unsigned long long*)c->merged_vis[c->slot].ptr;
        f->hud_draws = f->draws; f->hud_tri_info = f->tri_info;      // (the picker: hud keys name global ranks)
    }
    f->hud_pass = 0;
    f->msaa_halo = (const unsigned long long*)c->msaa_halo;
    f->halo_bands = c->band_n > 1 ? ((c->height + kTile - 1) / kTile + c->band_n - 1) / c->band_n : 0u;
    f->out_rgba16f = (uint16_t*)(c->bound_out ? (uint8_t*)c->bound_out - (size_t)c->out_first_row * c->width * 8 : c->out16[c->slot].ptr);   // kernels address by absolute row
    f->out_rgba32f = (float*)c->out32[c->slot].ptr;
    f->lights_pre = (float4*)c->lights_pre[c->slot].ptr;
    f->lights_cap = (uint32_t)(c->lights_pre[c->slot].size / 32);
}

// standard.wgsl:17-27 as one matrix: view_h = inv_proj * clip with clip = (2 (px + 0.5) / W - 1, 1 - 2 (py + 0.5) / H, depth, 1).  Composed in f64 from the f32
// matrices of the camera UBO as the caller wrote it (camera.rs:169-219: inv_proj at 256, inv_view at 320, position at 384, proj at 64), once per frame at
// awsm_hip_geometry_pass — the camera the frame is shaded with is the camera it was submitted with.
void compose_pixel_to_view(AwsmHipCtx* c, FrameBufs& b) {
    float cam[128];
    memcpy(cam, c->camera_host, 512);
    const float* proj = cam + 16; const float* inv_proj = cam + 64; const float* inv_view = cam + 80;
    const double W = (double)c->width, H = (double)c->height;
    // A: (px, py, depth, 1) -> clip, column-major
    const double A[16] = {2.0 / W, 0, 0, 0,   0, -2.0 / H, 0, 0,   0, 0, 1, 0,   1.0 / W - 1.0, 1.0 - 1.0 / H, 0, 1};
    for (int col = 0; col < 4; col++) for (int row = 0; row < 4; row++) {
        double s = 0.0;
        for (int k = 0; k < 4; k++) s += (double)inv_proj[k * 4 + row] * A[col * 4 + k];
        b.pix2view[col * 4 + row] = (float)s;
    }
    for (int col = 0; col < 3; col++) for (int row = 0; row < 3; row++) b.view_rot[col * 3 + row] = inv_view[col * 4 + row];
    for (int i = 0; i < 3; i++) b.cam_pos[i] = cam[96 + i];
    b.cam_ortho = proj[15] > 0.9f ? 1u : 0u;      // standard.wgsl:38: proj[3][3]
    const double vx = inv_view[8], vy = inv_view[9], vz = inv_view[10], l = std::sqrt(vx * vx + vy * vy + vz * vz);
    b.ortho_view_dir[0] = l > 0.0 ? (float)(vx / l) : 0.0f; b.ortho_view_dir[1] = l > 0.0 ? (float)(vy / l) : 0.0f; b.ortho_view_dir[2] = l > 0.0 ? (float)(vz / l) : 1.0f;
}

int record(AwsmHipCtx* c, int which, hipStream_t s = nullptr) {
    if (!c->stage_timers) { c->ev_valid[which] = false; return AWSM_OK; }     // each record is a ~5 us bubble between two kernels
    HIPCHK(c, hipEventRecord(c->ev[which], s ? s : c->stream));
    c->ev_valid[which] = true;
    return AWSM_OK;
}

AwsmHipCtx::ResolveKey resolve_key(AwsmHipCtx* c, const FrameDev& f) {
    AwsmHipCtx::ResolveKey k;
    memset(&k, 0, sizeof k);      // padding too: the keys are compared bytewise
    k.write_seq = c->write_seq + (c->scene_dirty ? 1u : 0u); k.draws_version = FB(c).draws_version;
    k.ptrs[0] = f.draw_shade; k.ptrs[1] = f.draw_mat; k.ptrs[2] = f.tex_slots; k.ptrs[3] = f.draw_lean; k.ptrs[4] = f.lights_pre;
    k.n_draws = f.n_draws; k.mipmap = c->last_opaque.mipmap; k.has_opaque = c->last_opaque.has_opaque; k.lights_cap = f.lights_cap;
    return k;
}
bool resolve_stale(AwsmHipCtx* c, const FrameDev& f) {
    const AwsmHipCtx::ResolveKey k = resolve_key(c, f);
    return memcmp(&k, &c->resolved[c->slot], sizeof k) != 0;
}

// The geometry cache's part of the world pass's frame: which draws may keep what the slot's arrays hold (frame_params.hpp).
void fill_geometry_cache(AwsmHipCtx* c, FrameDev* f, bool replay) {
    FrameBufs& b = FB(c);
    f->wcache = (float4*)b.wcache.ptr;
    f->cache_on = 0; f->prev_draws = nullptr; f->prev_n_draws = 0; f->n_dirty = 0; f->cache_serial = b.cache_serial;
    if (!c->geometry_cache || replay || !b.cache_valid || !b.wcache.ptr || b.cache_had_tri_shade != (f->tri_shade != nullptr)) return;
    // the ranges written since the slot's arrays were computed, merged per buffer; too many: one bounding range per buffer; still too many (or a write that
    // invalidates every draw): no cache this frame
    std::vector<DirtyRange> r;
    if (!c->dirty_log.ranges_since(b.cache_seq, kMaxDirtyRanges, r)) return;
    for (size_t i = 0; i < r.size(); i++) { f->dirty[i][0] = r[i].buf; f->dirty[i][1] = r[i].lo; f->dirty[i][2] = r[i].hi; }
    f->n_dirty = (uint32_t)r.size();
    f->prev_draws = (const DrawDev*)(b.cached_is_prev ? b.draws_prev.ptr : b.draws_dev.ptr);
    f->prev_n_draws = b.cached_n_draws;
    f->cache_on = f->prev_draws ? 1u : 0u;
}


// replay: the frame's geometry pass is enqueued again (a bin list that overflowed, world arrays that moved under the hud pass): the camera snapshot the first
// enqueue took stays — pix2view / cam_pos were composed from that camera at awsm_hip_geometry_pass — and the geometry cache is not consulted.
int enqueue_geometry(AwsmHipCtx* c, bool replay = false) {
    HostTrace ht(c->frame_serial);
    FrameDev f;
    { const bool hd = c->hud_geometry_done, hm = c->hud_merged; c->hud_geometry_done = false; c->hud_merged = false; fill_frame(c, &f); c->hud_geometry_done = hd; c->hud_merged = hm; }      // the world pass's own view (no merged hud keys, no hud ranks)
    fill_geometry_cache(c, &f, replay);
    f.block_draw = FB(c).block_map_valid ? (const uint32_t*)FB(c).block_map.ptr : nullptr;
    // AwsmFrameStats.geometry_cache_blocks: marked only when stage times are asked for too (a plain store per workgroup, counted by frame_end on the host)
    f.cache_mark = (c->stage_timers && FB(c).cache_mark.size >= (size_t)c->n_blocks * 4) ? (uint32_t*)FB(c).cache_mark.ptr : nullptr;
    const uint32_t n_tiles = f.tiles_x * f.tiles_y;
    int rc = sync_scene(c);
    if (rc) return rc;
    ht.mark("geometry: sync_scene");
    // this slot's buffers were last read by the opaque pass kSlots frames ago (already ordered by awsm_hip_geometry_pass; a replay comes here directly)
    if (c->overlap && c->shade_pending[c->slot]) { HIPCHK(c, wait_slot_free(c)); c->shade_pending[c->slot] = false; }
    // the camera the frame is shaded with = the camera it was submitted with, in every mode: the lean kernel's pix2view was composed from the caller's
    // bytes at awsm_hip_geometry_pass, the general kernels and the MSAA detector read this snapshot — a camera write between the two passes of a frame
    // cannot give its strips two cameras
    // (copied by k_deform_transform when the frame has geometry: a 512-byte hipMemcpyAsync costs the stream 18 us of gap + copy)
    if (c->bufs[AWSM_BUF_CAMERA].ptr && !replay) {
        const size_t cam_bytes = std::min<size_t>(512, c->bufs[AWSM_BUF_CAMERA].size);
        if (c->total_tris && n_tiles) { f.camera_snap = (uint32_t*)FB(c).camera.ptr; f.camera_snap_words = (uint32_t)(cam_bytes / 4); }
        else HIPCHK(c, hipMemcpyAsync(FB(c).camera.ptr, c->bufs[AWSM_BUF_CAMERA].ptr, cam_bytes, hipMemcpyDeviceToDevice, c->stream));
    }
    if (c->overlap) {
        // everything the per-draw resolve of this frame reads (draw list, scene buffers) is on the stream by now — an event only when a resolve
        // will run (resolve_key): a record packet costs the caller's stream ~19 us between the camera upload and the transform kernel
        c->uploads_recorded[c->slot] = false;
        if (resolve_stale(c, f)) { HIPCHK(c, hipEventRecord(c->ev_uploads[c->slot], c->stream)); c->uploads_recorded[c->slot] = true; }
        c->geom_write_seq[c->slot] = c->write_seq;
    }
    ht.mark("geometry: slot wait + upload event");
    const bool has_geometry = c->total_tris && n_tiles;
    if (!has_geometry) {   // otherwise k_deform_transform clears counters + tile_count and k_bin_scan clears tile_cursor
        HIPCHK(c, hipMemsetAsync(FB(c).counters.ptr, 0, 8 * sizeof(uint32_t), c->stream));
        HIPCHK(c, hipMemsetAsync((uint32_t*)FB(c).counters.ptr + 12, 0, 2 * sizeof(uint32_t), c->stream));
        if (n_tiles) HIPCHK(c, hipMemsetAsync(FB(c).tile_count.ptr, 0, n_tiles * sizeof(uint32_t), c->stream));
    }
    if (c->trace_dev) awsm_launch_handoff_signal(nullptr, 0u, trace_slot(c, 0), c->stream);
    if ((rc = record(c, EV_START))) return rc;
    // Measurement aid (tools/knockout.sh; static camera only): from the ninth frame on leave out the raster (4), the binning launches as well (6) or
    // the whole pass (7) — the slot's buffers then keep the previous frames' keys, the opaque pass does the same work, and the frame rate
    // says what each stage's kernels take from the kernels they run beside.
#ifdef AWSM_DEBUG_SWITCHES      // tools/build_variants.sh h_debug "-DAWSM_DEBUG_SWITCHES": not in the product library
    static const int knockout_env = getenv("AWSM_DEBUG_KNOCKOUT") ? atoi(getenv("AWSM_DEBUG_KNOCKOUT")) : 0;
    const int knockout = c->frame_serial > 8 ? knockout_env : 0;
#else
    const int knockout = 0;
#endif
    if (c->total_tris && n_tiles && !(knockout & 1)) awsm_launch_transform(c->scene_dev, &f, c->n_blocks, c->stream);
    if ((rc = record(c, EV_TRANSFORM))) return rc;
    ht.mark("geometry: transform launch");
    if (n_tiles && !(knockout & 2)) {
        if (c->total_tris) { awsm_launch_bin_count(&f, c->stream); awsm_launch_bin_big(&f, 0, c->stream); }
        awsm_launch_bin_scan(&f, c->stream);
        if (c->total_tris) { awsm_launch_bin_fill(&f, c->stream); awsm_launch_bin_big(&f, 1, c->stream); }
    }
    if ((rc = record(c, EV_BIN))) return rc;
    ht.mark("geometry: bin launches");
    if (n_tiles && !(knockout & 4)) awsm_launch_raster(&f, c->stream);
    if ((rc = record(c, EV_RASTER))) return rc;
    ht.mark("geometry: raster launch");
    HIPCHK(c, hipGetLastError());
    {   // the slot's arrays now hold this list's world positions / N / T / per-triangle words (all of them: hits kept theirs, misses were recomputed)
        FrameBufs& b = FB(c);
        b.cache_valid = has_geometry && !(knockout & 1) && b.wcache.ptr != nullptr;
        b.cached_n_draws = (uint32_t)c->draws_host.size(); b.cached_is_prev = false;
        b.cache_had_tri_shade = f.tri_shade != nullptr;
        b.cache_seq = c->write_seq; b.cache_serial = c->frame_serial;
        // entries every slot has seen are of no further use
        uint64_t oldest = ~0ull;
        for (int sl = 0; sl < n_slots(c); sl++) oldest = std::min(oldest, c->fb[sl].cache_valid ? c->fb[sl].cache_seq : c->write_seq);
        c->dirty_log.prune(oldest);
    }
    return AWSM_OK;
}

int enqueue_opaque(AwsmHipCtx* c) {
    HostTrace ht(c->frame_serial);
    FrameDev f;
    fill_frame(c, &f);
    f.has_opaque = c->last_opaque.has_opaque;
    f.mipmap = c->last_opaque.mipmap;
    f.aniso = (c->flags & AWSM_CFG_ANISOTROPIC) ? 1u : 0u;
    if (c->bound_out) {
        if (c->out_rows_mode) {   // a row strip's own buffer: rows [first_row, first_row + bytes / (width * 8))
            if (f.out_compact || f.sy0 < c->out_first_row || (size_t)(f.sy1 - c->out_first_row) * c->width * 8 > c->bound_out_bytes)
                return fail(c, AWSM_ERR_INVALID_ARGUMENT, "opaque_pass: bound output starts at row %u and holds %zu bytes, the shard shades rows [%u, %u)", c->out_first_row, c->bound_out_bytes, f.sy0, f.sy1);
        } else {
            const size_t need = (f.out_compact ? (size_t)f.tiles_y * kTile : (size_t)c->height) * c->width * 8;
            if (c->bound_out_bytes < need) return fail(c, AWSM_ERR_INVALID_ARGUMENT, "opaque_pass: bound output holds %zu bytes, this shard layout writes %zu", c->bound_out_bytes, need);
        }
    }
    int rc = sync_scene(c);
    if (rc) return rc;
    hipStream_t ss = shade_stream_of(c);
    // The per-draw resolve only needs the uploads, not the geometry pass, and nothing of the previous frame: its outputs are per frame
    // slot.  In overlap mode it goes first onto this slot's shade stream — idle since the frame before last — and so runs beside the
    // previous frame's opaque pass (on the single shade stream of round 1 it sat between two frames' shading kernels: todo -> 14 us gap ->
    // resolve 30 us -> gap -> lean; on the caller's stream it would lengthen the geometry chain).  No stream of its own: the process has
    // four hardware queues (GPU_MAX_HW_QUEUES), and a fifth stream shares one of them with another stream — seen as 25-30 us of idle GPU
    // at every frame boundary.  A scene write between the two passes falls back to the late order.
    // (with stage timers on it always runs: EV_SHADE_BEGIN must sit behind a kernel of this stream, not right behind the cross-stream wait — see below)
    const bool want_resolve = f.sy1 > f.sy0 && f.has_opaque && (resolve_stale(c, f) || c->stage_timers);
    if (want_resolve) c->resolved[c->slot] = resolve_key(c, f);
    // (With stage timers on the late order stays: an event recorded right behind a cross-stream wait is stamped when the wait is
    // consumed, not when it is satisfied, and ms_shade would include the tail of the raster kernel — seen: 0.40 instead of 0.36 ms.)
    const bool early = c->uploads_recorded[c->slot] && c->overlap && !c->stage_timers && want_resolve && c->geometry_done && c->geom_write_seq[c->slot] == c->write_seq;
    if (early) {      // on this slot's shade stream, ahead of the wait for the geometry pass
        HIPCHK(c, hipStreamWaitEvent(ss, c->ev_uploads[c->slot], 0));
        awsm_launch_resolve_draws(c->scene_dev, &f, ss);
    }
    ht.mark("opaque: sync_scene + early resolve");
    if (c->overlap) {   // the shade stream picks up where the caller's stream is now (geometry pass + uploads of this frame)
        if (c->handoff) {
            ++c->geom_sig[c->slot];
            if (c->handoff_test_drop) c->handoff_test_drop--;
            else awsm_launch_handoff_signal(c->handoff_flags + c->slot, c->geom_sig[c->slot], trace_slot(c, 1), c->stream);
            awsm_launch_handoff_wait(c->handoff_flags + c->slot, c->geom_sig[c->slot], c->handoff_budget_ticks, handoff_timeouts(c), c->handoff_timeouts_seen, c->handoff_poison + c->slot, c->frame_serial, ss);
        } else {
            if (c->trace_dev) awsm_launch_handoff_signal(nullptr, 0u, trace_slot(c, 1), c->stream);
            HIPCHK(c, hipEventRecord(c->ev_geom_done[c->slot], c->stream));
            HIPCHK(c, hipStreamWaitEvent(ss, c->ev_geom_done[c->slot], 0));
        }
        // ... and behind the previous frame's opaque pass on the other slot's shade stream.  Where the two share something — one bound output image
        // for both, the halo keys a caller bound for MSAA + bands — behind all of it.  Otherwise (the library's own images, or a caller alternating
        // between two bound ones: images, per-draw records and todo lists are per slot) behind its main kernel only: frame i + 1's
        // k_shade_lean starts when frame i's has ended, beside frame i's k_shade_todo.  Not earlier, although nothing shared would forbid it:
        // with two frame slots the geometry pass of frame i + 2 waits for frame i's shading, and two opaque passes running into each other
        // push that out — measured 2,780 frames/s free-running against 2,980 in step.  (Last of the waits on purpose: the first two are
        // consumed while the previous frame still shades.)
        const uint8_t* lo = (const uint8_t*)c->bound_out; const uint8_t* hi = lo ? lo + c->bound_out_bytes : nullptr;
        const int p = prev_slot(c);
        const bool same_image = lo && c->slot_out_lo[p] && lo < c->slot_out_hi[p] && c->slot_out_lo[p] < hi;
        if (c->shade_recorded[p]) HIPCHK(c, wait_prev_slot(c, ss, !(same_image || (c->msaa != 0 && c->msaa_halo))));
        c->slot_out_lo[c->slot] = lo; c->slot_out_hi[c->slot] = hi;
    }
    ht.mark("opaque: hand-off launches");
    if (want_resolve && !early) awsm_launch_resolve_draws(c->scene_dev, &f, ss);
    if ((rc = record(c, EV_SHADE_BEGIN, ss))) return rc;      // after the resolve: ms_shade is the shading kernels alone
    c->ev_valid[EV_SHADE_LEAN] = false;
    c->lean_flagged[c->slot] = false;
    if (c->overlap && c->handoff && f.sy1 > f.sy0 && awsm_shade_is_lean(&f)) {
        f.lean_done_flag = c->handoff_flags + 2 * kSlots + c->slot; f.lean_done_serial = ++c->lean_sig[c->slot];
        c->lean_flagged[c->slot] = true;
    }
    if (f.sy1 > f.sy0) {
        awsm_launch_shade(c->scene_dev, &f, ss);
        // the lean kernel alone (AwsmFrameStats.ms_shade_lean): one more event, only when stage times are asked for
        if (c->stage_timers && awsm_shade_is_lean(&f) && (rc = record(c, EV_SHADE_LEAN, ss))) return rc;
        (void)awsm_launch_shade_todo(c->scene_dev, &f, ss);
    }
    if ((rc = record(c, EV_SHADE, ss))) return rc;
    ht.mark("opaque: shade launches");
    if (c->overlap) HIPCHK(c, mark_shade_done(c, ss));
    else if (c->trace_dev) awsm_launch_handoff_signal(nullptr, 0u, trace_slot(c, 2), ss);
    ht.mark("opaque: shade-done signal + event");
    HIPCHK(c, hipGetLastError());
    return AWSM_OK;
}

inline FrameBufs& TR(AwsmHipCtx* c) { return (c->hud_transparent ? c->htr : c->tr)[c->slot]; }

// The transparent pass's frame: its own draws / vertices / setup records / bins; the geometry pass's visibility keys for the depth
// test; the opaque image as blit source and transmission background; the composite image as target.
void fill_frame_forward(AwsmHipCtx* c, FrameDev* f) {
    fill_frame(c, f);
    FrameBufs& t = TR(c);
    const int which = c->hud_transparent ? 1 : 0;
    f->vis = (unsigned long long*)FB(c).vis.ptr;      // the WORLD's depth (render.rs:224-297), whatever the opaque pass read
    f->n_draws = (uint32_t)c->tr_draws_host[which].size();
    f->total_tris = c->tr_total_tris[which]; f->total_verts = 3u * c->tr_total_tris[which];
    f->bin_capacity = t.bin_capacity;
    f->draws = (const DrawDev*)t.draws_dev.ptr;
    f->draw_shade = (DrawShadeDev*)t.draw_shade.ptr;
    f->tex_slots = (TexSlotDev*)t.tex_slots.ptr;
    f->draw_mat = (DrawMatDev*)t.draw_mat.ptr;
    f->clip = (float4*)t.clip.ptr; f->nrm = (float4*)t.nrm.ptr; f->tan = (float4*)t.tan.ptr; f->wpos = (float4*)t.wpos.ptr;
    f->tri_info = (uint32_t*)t.tri_flags.ptr;
    f->tri_shade = nullptr; f->draw_lean = nullptr; f->shade_todo = nullptr; f->shade_todo_cap = 0; f->lean_next = nullptr; f->lean_grid = 0;     // the opaque pass's lean route only
    f->tri_rec = (TriRec*)t.tri_rec.ptr;
    f->tile_count = (uint32_t*)t.tile_count.ptr; f->tile_offset = (uint32_t*)t.tile_offset.ptr;
    f->tile_cursor = (uint32_t*)t.tile_cursor.ptr; f->bin_list = (uint32_t*)t.bin_list.ptr;
    f->tile_order = (uint32_t*)t.tile_order.ptr;
    f->scan_tmp = (uint32_t*)t.scan_tmp.ptr;
    f->tile_split = (uint32_t*)t.tile_split.ptr; f->raster_scratch = nullptr;      // the transparent pass has its own tile kernel: nothing is split
    f->raster_extra_cap = 0; f->raster_slot_cap = 0;
    f->big_list = (uint32_t*)t.big_list.ptr;
    f->counters = (uint32_t*)t.counters.ptr;
    f->opaque_rgba16f = c->opaque_src ? (const uint16_t*)c->opaque_src : f->out_rgba16f;
    f->out_compact = 0;                    // the composite is addressed by absolute row
    f->frag_rec = (uint4*)t.frag_rec.ptr; f->frag_color = (float4*)t.frag_color.ptr; f->frag_first = (uint32_t*)t.frag_first.ptr; f->frag_cap = t.frag_cap;
    f->out_rgba16f = (uint16_t*)(c->bound_comp ? c->bound_comp : c->comp16.ptr);
    f->out_rgba32f = (float*)c->comp32.ptr;
    f->has_opaque = c->last_opaque.has_opaque;
    f->mipmap = c->last_opaque.mipmap;
    f->aniso = (c->flags & AWSM_CFG_ANISOTROPIC) ? 1u : 0u;
    if (c->hud_transparent) {      // the HUD pass: depth starts cleared, the colours it blends over are the composite's own (in place, a pixel per thread)
        f->hud_pass = 1;
        f->opaque_rgba16f = f->out_rgba16f;
    }
}

int enqueue_transparent(AwsmHipCtx* c) {
    FrameDev f;
    fill_frame_forward(c, &f);
    const uint32_t n_tiles = f.tiles_x * f.tiles_y;
    int rc = sync_scene(c);
    if (rc) return rc;
    hipStream_t ss = shade_stream_of(c);          // in order after the opaque pass
    // the composite image and the fragment lists' bookkeeping are per context: behind the previous frame's passes on the other slot's stream
    if (c->overlap && c->shade_recorded[prev_slot(c)]) HIPCHK(c, wait_prev_slot(c, ss));
    if ((rc = record(c, EV_FWD_BEGIN, ss))) return rc;
    const bool has_geometry = f.total_tris && n_tiles;
    if (!has_geometry) {
        HIPCHK(c, hipMemsetAsync(TR(c).counters.ptr, 0, 8 * sizeof(uint32_t), ss));
        HIPCHK(c, hipMemsetAsync((uint32_t*)TR(c).counters.ptr + 12, 0, 2 * sizeof(uint32_t), ss));
        if (n_tiles) HIPCHK(c, hipMemsetAsync(TR(c).tile_count.ptr, 0, n_tiles * sizeof(uint32_t), ss));
    } else {
        awsm_launch_resolve_draws(c->scene_dev, &f, ss);      // first: the transform tags each triangle with its draw's alpha mode
        awsm_launch_transform_forward(c->scene_dev, &f, c->tr_n_blocks[c->hud_transparent ? 1 : 0], ss);
    }
    if (n_tiles) {
        if (f.total_tris) { awsm_launch_bin_count(&f, ss); awsm_launch_bin_big(&f, 0, ss); }
        awsm_launch_bin_scan(&f, ss);
        if (f.total_tris) { awsm_launch_bin_fill(&f, ss); awsm_launch_bin_big(&f, 1, ss); }
        awsm_launch_forward(c->scene_dev, &f, ss);
    }
    if ((rc = record(c, EV_FWD, ss))) return rc;
    if (c->overlap) HIPCHK(c, mark_shade_done(c, ss));
    HIPCHK(c, hipGetLastError());
    return AWSM_OK;
}

int ensure_bin_capacity_of(AwsmHipCtx* c, FrameBufs& b, uint32_t entries) {
    if (entries <= b.bin_capacity && b.bin_list.ptr) return AWSM_OK;
    uint32_t cap = std::max(entries, b.bin_capacity + b.bin_capacity / 2);
    int rc = dev_realloc(c, b.bin_list, (size_t)cap * 4, false);
    if (rc) return rc;
    b.bin_capacity = cap;
    return AWSM_OK;
}

int ensure_fragment_capacity(AwsmHipCtx* c, FrameBufs& b, uint32_t fragments) {
    if (fragments <= b.frag_cap && b.frag_rec.ptr) return AWSM_OK;
    const uint32_t cap = std::max(fragments, b.frag_cap + b.frag_cap / 2);
    int rc = dev_realloc(c, b.frag_rec, (size_t)cap * 16, false);
    if (!rc) rc = dev_realloc(c, b.frag_color, (size_t)cap * 16, false);
    if (rc) return rc;
    b.frag_cap = cap;
    return AWSM_OK;
}

// API draws -> one DrawDev per (draw, instance) with the running triangle / block prefix sums; `data` = the vertex buffer the
// draws index, `bytes_per_tri` the extent check per triangle (0 = the vertices are indexed: not checkable here).
int build_draw_list(AwsmHipCtx* c, const char* pass, const AwsmDraw* draws, uint32_t n, AwsmBuf data, uint32_t bytes_per_tri,
                    std::vector<DrawDev>& out, uint64_t* tris_out, uint64_t* blocks_out) {
    out.clear();
    uint64_t tris = 0, blocks = 0;
    for (uint32_t i = 0; i < n; i++) {
        const AwsmDraw& d = draws[i];
        if (d.inst_count != 0) {   // instanced draw: inst_count mat4s at inst_off of the instance-transform buffer (instances.rs)
            const DevBuf& ib = c->bufs[AWSM_BUF_INSTANCES];
            if (!ib.ptr) return fail(c, AWSM_ERR_NOT_READY, "%s: draw %u: instanced, but the instance-transform buffer was never created", pass, i);
            if ((d.inst_off & 15u) || (uint64_t)d.inst_off + 64ull * d.inst_count > ib.size)
                return fail(c, AWSM_ERR_OUT_OF_RANGE, "%s: draw %u: %u instances at %u exceed the instance-transform buffer", pass, i, d.inst_count, d.inst_off);
        }
        if (d.vis_data_off & (bytes_per_tri ? 15u : 3u)) return fail(c, AWSM_ERR_INVALID_ARGUMENT, "%s: draw %u: vertex data offset %u misaligned", pass, i, d.vis_data_off);
        if ((uint64_t)d.vis_data_off + (uint64_t)bytes_per_tri * d.tri_count > c->bufs[data].size || (!bytes_per_tri && d.tri_count && (uint64_t)d.vis_data_off + 40 > c->bufs[data].size))
            return fail(c, AWSM_ERR_OUT_OF_RANGE, "%s: draw %u: %u triangles at %u exceed the vertex buffer", pass, i, d.tri_count, d.vis_data_off);
        if ((uint64_t)d.geom_meta_off + 40 > c->bufs[AWSM_BUF_GEOM_META].size || (d.geom_meta_off & 3u))
            return fail(c, AWSM_ERR_OUT_OF_RANGE, "%s: draw %u: geometry meta offset %u out of range", pass, i, d.geom_meta_off);
        if (d.tri_count == 0) continue;   // draw_indexed(0) draws nothing; keeps first_block strictly increasing
        const uint32_t copies = d.inst_count ? d.inst_count : 1u;
        for (uint32_t k = 0; k < copies; k++) {
            DrawDev dd{};
            dd.geom_meta_off = d.geom_meta_off; dd.vis_data_off = d.vis_data_off; dd.tri_count = d.tri_count; dd.flags = d.flags & 0x7Fu;
            if (d.inst_count) { dd.flags |= kDrawInstanced; dd.inst_off = d.inst_off + 64u * k; }
            dd.first_tri = (uint32_t)tris; dd.first_block = (uint32_t)blocks;
            out.push_back(dd);
            tris += d.tri_count;
            blocks += (3ull * d.tri_count + 255) / 256;
            if (tris > 0x55555555ull) return fail(c, AWSM_ERR_UNSUPPORTED, "%s: more than 2^32/3 triangles in one pass", pass);
            if (out.size() >= (1u << 24)) return fail(c, AWSM_ERR_UNSUPPORTED, "%s: more than 2^24 non-empty draws (instances included) in one pass", pass);
        }
    }
    *tris_out = tris; *blocks_out = blocks;
    return AWSM_OK;
}

// per-pass device state sized for `draws_host` / total_tris; uploads the draw list when it changed
// raster items: one per tile + one per extra slice of a split tile (<= entries / 256); scratch tiles: one per slice of a split tile
int reserve_raster_items(AwsmHipCtx* c, FrameBufs& b, bool forward) {
    const uint32_t tiles_x = (c->width + kTile - 1) / kTile, tiles_y_full = (c->height + kTile - 1) / kTile;
    const size_t n_tiles_full = (size_t)tiles_x * tiles_y_full;
    const uint32_t extra_cap = forward ? 0u : b.bin_capacity / 256u + 1u, slot_cap = forward ? 0u : b.bin_capacity / 128u + 2u;
    int rc;
    if ((rc = dev_reserve(c, b.tile_order, (n_tiles_full + extra_cap) * 4))) return rc;
    if (!forward && (rc = dev_reserve(c, b.raster_scratch, (size_t)slot_cap * kTile * kTile * 8 * (c->msaa == 4u ? 4u : 1u)))) return rc;
    b.raster_extra_cap = extra_cap; b.raster_slot_cap = slot_cap;
    return AWSM_OK;
}

// Device state of one pass in one frame slot, sized for `tri_cap` triangles and `draw_cap` draws.
int size_pass_buffers(AwsmHipCtx* c, FrameBufs& b, size_t tri_cap, size_t draw_cap, bool forward) {
    int rc;
    const size_t nd = std::max<size_t>(draw_cap, 1), nv = std::max<size_t>(3 * tri_cap, 1), nt = std::max<size_t>(tri_cap, 1);
    if ((rc = dev_reserve(c, b.draws_dev, nd * sizeof(DrawDev)))) return rc;
    const bool world_pass = &b >= c->fb && &b < c->fb + kSlots;
    if (world_pass && c->geometry_cache) {      // geometry cache: the previous list beside the current one, world positions per vertex
        if ((rc = dev_reserve(c, b.draws_prev, nd * sizeof(DrawDev)))) return rc;
        if ((rc = dev_reserve(c, b.wcache, nv * 16))) return rc;
        const size_t blocks_bound = nv / 256 + nd + 1;
        if ((rc = dev_reserve(c, b.block_map, blocks_bound * 4))) return rc;
        b.block_map_valid = false;
        if (b.cache_mark.size < blocks_bound * 4) { if ((rc = dev_realloc(c, b.cache_mark, (blocks_bound + blocks_bound / 2) * 4, true))) return rc; }
    }
    b.cache_valid = false;      // the arrays may have moved
    if ((rc = dev_reserve(c, b.draw_shade, nd * sizeof(DrawShadeDev)))) return rc;
    if ((rc = dev_reserve(c, b.tex_slots, nd * kCoreTextures * sizeof(TexSlotDev)))) return rc;
    if ((rc = dev_reserve(c, b.draw_mat, nd * sizeof(DrawMatDev)))) return rc;
    if ((rc = dev_reserve(c, b.clip, nv * 16))) return rc;
    if ((rc = dev_reserve(c, b.nrm, nv * 16))) return rc;
    if ((rc = dev_reserve(c, b.tan, nv * 16))) return rc;
    if (forward && (rc = dev_reserve(c, b.wpos, nv * 16))) return rc;
    if ((rc = dev_reserve(c, b.tri_flags, nt * 4))) return rc;
    if (!forward && (rc = dev_reserve(c, b.tri_shade, nt * 32))) return rc;
    if (!forward && (rc = dev_reserve(c, b.draw_lean, nd * sizeof(LeanDrawDev)))) return rc;
    if ((rc = dev_reserve(c, b.big_list, nt * 4))) return rc;
    if ((rc = dev_reserve(c, b.tri_rec, nt * kTriRecBytes))) return rc;
    const uint32_t tiles_x = (c->width + kTile - 1) / kTile, tiles_y_full = (c->height + kTile - 1) / kTile;
    const size_t n_tiles_full = (size_t)tiles_x * tiles_y_full;
    if ((rc = dev_reserve(c, b.tile_count, n_tiles_full * 4))) return rc;
    if ((rc = dev_reserve(c, b.tile_offset, (n_tiles_full + 1) * 4))) return rc;
    if ((rc = dev_reserve(c, b.tile_cursor, n_tiles_full * 4))) return rc;
    if ((rc = dev_reserve(c, b.tile_split, n_tiles_full * 8))) return rc;
    if ((rc = dev_reserve(c, b.scan_tmp, (((n_tiles_full + 255) / 256) * 2 + 1) * 40 * 4))) return rc;      // kScanWords = 40: aggregates, bases, run starts (kernels_geometry.hip)
    if ((rc = ensure_bin_capacity_of(c, b, (c->flags & AWSM_CFG_SMALL_BIN_LIST) ? 4096u : (uint32_t)std::min<size_t>(std::max<size_t>(4 * tri_cap + 65536u, 1u << 18), 0xFFFFFFF0u)))) return rc;
    if ((rc = reserve_raster_items(c, b, forward))) return rc;
    b.tri_cap = tri_cap; b.draw_cap = draw_cap; b.sized_w = c->width; b.sized_h = c->height;
    return AWSM_OK;
}

// Sizes the pass's device state for `draws_host` / total_tris (see reserve_pass_buffers).
int ensure_pass_capacity(AwsmHipCtx* c, FrameBufs& b, const std::vector<DrawDev>& draws_host, uint32_t total_tris, bool forward) {
    int rc;
    if (total_tris > b.tri_cap || draws_host.size() > b.draw_cap || b.sized_w != c->width || b.sized_h != c->height || !b.bin_list.ptr) {
        bool instanced = false;
        for (const DrawDev& d : draws_host) instanced |= (d.flags & kDrawInstanced) != 0;
        const size_t tri_bound = forward || instanced ? ~size_t(0) : c->bufs[AWSM_BUF_VIS_GEOM_DATA].size / 168u;
        const size_t tri_cap = std::max<size_t>(std::min<size_t>((size_t)total_tris + total_tris / 4 + 4096, std::max<size_t>(tri_bound, total_tris)), b.tri_cap);
        const size_t draw_cap = std::max<size_t>(draws_host.size() + draws_host.size() / 4 + 64, b.draw_cap);
        FrameBufs* set = forward ? ((&b >= c->htr && &b < c->htr + kSlots) ? c->htr : c->tr) : ((&b >= c->hud && &b < c->hud + kSlots) ? c->hud : c->fb);
        for (int sl = 0; sl < n_slots(c); sl++) {
            // (the slot in use first: if memory runs out half-way the current frame still has its buffers)
            FrameBufs& t = set[(c->slot + sl) % kSlots];
            if ((rc = size_pass_buffers(c, t, std::max(tri_cap, t.tri_cap), std::max(draw_cap, t.draw_cap), forward))) return rc;
        }
    }
    return AWSM_OK;
}

int upload_draws_at(AwsmHipCtx* c, FrameBufs& b, size_t at, const std::vector<DrawDev>& list) {
    const size_t bytes = list.size() * sizeof(DrawDev);
    uint8_t* dst = (uint8_t*)b.draws_dev.ptr + at * sizeof(DrawDev);
    if (bytes <= (1u << 20)) return upload_small(c, dst, list.data(), bytes);
    HIPCHK(c, hipMemcpyAsync(dst, list.data(), bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return AWSM_OK;
}

// The sorted draw list of a static or slowly moving scene repeats frame after frame: upload it only when it changed
// (an in-stream host-to-device copy costs more GPU idle time than the transform kernel takes).
// keep_prev (the world geometry pass): a new list goes into the other one of the slot's two list buffers, so that the list the slot's arrays were
// computed for stays readable beside it (geometry cache: k_deform_transform compares the two draw by draw).
int upload_draw_list(AwsmHipCtx* c, FrameBufs& b, const std::vector<DrawDev>& draws_host, bool keep_prev) {
    const bool same_draws = b.draws_uploaded_valid && b.draws_uploaded_ptr == b.draws_dev.ptr && b.draws_uploaded.size() == draws_host.size() &&
                            (draws_host.empty() || memcmp(b.draws_uploaded.data(), draws_host.data(), draws_host.size() * sizeof(DrawDev)) == 0);
    auto upload_block_map = [&]() -> int {      // the world pass: which draw each transform workgroup belongs to
        b.block_map_valid = false;
        if (!keep_prev || !b.block_map.ptr) return AWSM_OK;
        std::vector<uint32_t> map;
        for (size_t i = 0; i < draws_host.size(); i++) map.insert(map.end(), (3ull * draws_host[i].tri_count + 255) / 256, (uint32_t)i);
        if (map.size() * 4 > b.block_map.size || map.size() * 4 > (1u << 20)) return AWSM_OK;
        const int rc = upload_small(c, b.block_map.ptr, map.data(), map.size() * 4);
        if (!rc) b.block_map_valid = true;
        return rc;
    };
    if (draws_host.empty()) return AWSM_OK;
    if (same_draws) return (keep_prev && !b.block_map_valid) ? upload_block_map() : AWSM_OK;      // (the map's buffer was re-sized under an unchanged list)
    if (keep_prev && b.cache_valid && !b.cached_is_prev && b.draws_prev.ptr && b.draws_prev.size == b.draws_dev.size) {
        std::swap(b.draws_dev, b.draws_prev);
        b.cached_is_prev = true;
    } else if (keep_prev && !b.cached_is_prev) b.cache_valid = false;      // the cached list is about to be overwritten
    b.draws_uploaded = draws_host; b.draws_uploaded_ptr = b.draws_dev.ptr; b.draws_uploaded_valid = true; b.draws_version++;
    b.tail_uploaded_valid = false;
    { const int rc = upload_block_map(); if (rc) return rc; }
    return upload_draws_at(c, b, 0, draws_host);
}

// Per-pass device state for `draws_host` / total_tris; uploads the draw list when it changed.  A frame that fits what the slot was sized for
// touches nothing (the common case: one comparison).  When it does not fit, EVERY frame slot of the pass is re-sized at once and with
// headroom — a re-allocation synchronises all streams, so a camera move that un-culls a few more triangles every frame must not pay it
// per frame and per slot (seen in round 2's driver run: frames 9, 10 and 13 of the process stalled the host for 2.1 / 1.8 / 0.9 ms, a third
// of a 20-frame measurement).  Triangles: need + 25 % + 4096, but no more than the vertex buffer can hold when no draw is instanced
// (then the pass can never outgrow its buffers again); draws: need + 25 % + 64.
int reserve_pass_buffers(AwsmHipCtx* c, FrameBufs& b, const std::vector<DrawDev>& draws_host, uint32_t total_tris, bool forward) {
    int rc = ensure_pass_capacity(c, b, draws_host, total_tris, forward);
    if (rc) return rc;
    return upload_draw_list(c, b, draws_host, !forward && &b >= c->fb && &b < c->fb + kSlots);
}

uint32_t mip_levels_full(uint32_t w, uint32_t h) {   // calculate_mipmap_levels (renderer-core/src/texture/mipmap.rs:60-62)
    uint32_t m = std::max(w, h), n = 0;
    while (m > 1u) { m >>= 1; n++; }
    return n + 1u;
}

int ensure_bin_capacity(AwsmHipCtx* c, uint32_t entries) { return ensure_bin_capacity_of(c, FB(c), entries); }

}  // namespace

extern "C" {

uint32_t awsm_hip_abi_version(void) { return AWSM_HIP_ABI_VERSION; }

const char* awsm_hip_last_error(const AwsmHipCtx* ctx) { return ctx ? ctx->last_error.c_str() : "null context"; }

int awsm_hip_create(const AwsmConfig* cfg, AwsmHipCtx** out) {
    if (!cfg || !out || cfg->struct_size < sizeof(AwsmConfig) || cfg->abi_version != AWSM_HIP_ABI_VERSION)
        return AWSM_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0 || cfg->device < 0 || cfg->device >= n_dev) return AWSM_ERR_NO_DEVICE;
    AwsmHipCtx* c = new (std::nothrow) AwsmHipCtx();
    if (!c) return AWSM_ERR_OUT_OF_MEMORY;
    c->device = cfg->device;
    c->flags = cfg->flags;
    auto bail = [&](int code) { awsm_hip_destroy(c); return code; };
    if (hipSetDevice(c->device) != hipSuccess) return bail(AWSM_ERR_DEVICE);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, c->device) != hipSuccess) return bail(AWSM_ERR_DEVICE);
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        fprintf(stderr, "awsm_hip: device %d is %s; this library carries gfx950 code objects only\n", c->device, prop.gcnArchName);
        return bail(AWSM_ERR_NO_DEVICE);
    }
    if (cfg->stream) { c->stream = (hipStream_t)cfg->stream; c->own_stream = false; }
    else { if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return bail(AWSM_ERR_DEVICE); c->own_stream = true; }
    for (int i = 0; i < EV_COUNT; i++) if (hipEventCreate(&c->ev[i]) != hipSuccess) return bail(AWSM_ERR_DEVICE);
    if (hipMalloc((void**)&c->scene_dev, sizeof(DevScene)) != hipSuccess) return bail(AWSM_ERR_OUT_OF_MEMORY);
    c->overlap = (cfg->flags & AWSM_CFG_OVERLAP_FRAMES) != 0;
    { const char* e = getenv("AWSM_GEOMETRY_CACHE"); c->geometry_cache = !(e && e[0] == '0'); }      // "0": k_deform_transform recomputes every draw every frame (A/B measurements, tests)
    {   // k_shade_lean as a persistent grid of N workgroups per CU (it is VALU-bound from 4 waves/SIMD up): the rest of each CU's
        // wave slots, registers and LDS stays free for the next frame's geometry kernels on the other stream
        const char* e = getenv("AWSM_LEAN_WGS_PER_CU");
        const int per_cu = e ? atoi(e) : (c->overlap ? kLeanWgsPerCu : 0);
        c->lean_grid = per_cu > 0 ? (uint32_t)(per_cu * prop.multiProcessorCount) & ~7u : 0u;
    }
    for (int s = 0; s < n_slots(c); s++) {
        if (hipMalloc(&c->fb[s].counters.ptr, 16 * sizeof(uint32_t)) != hipSuccess || hipMemset(c->fb[s].counters.ptr, 0, 16 * sizeof(uint32_t)) != hipSuccess) return bail(AWSM_ERR_OUT_OF_MEMORY);   // 8 frame counters + 4 pick words + k_bin_scan's arrival counter [12] and ready flag [13]
        c->fb[s].counters.size = 16 * sizeof(uint32_t);
        if (hipMalloc(&c->tr[s].counters.ptr, 16 * sizeof(uint32_t)) != hipSuccess) return bail(AWSM_ERR_OUT_OF_MEMORY);
        c->tr[s].counters.size = 16 * sizeof(uint32_t);
        if (hipMalloc(&c->htr[s].counters.ptr, 16 * sizeof(uint32_t)) != hipSuccess) return bail(AWSM_ERR_OUT_OF_MEMORY);
        c->htr[s].counters.size = 16 * sizeof(uint32_t);
        if (hipMalloc(&c->hud[s].counters.ptr, 16 * sizeof(uint32_t)) != hipSuccess) return bail(AWSM_ERR_OUT_OF_MEMORY);
        c->hud[s].counters.size = 16 * sizeof(uint32_t);
        if (hipMalloc(&c->fb[s].camera.ptr, 512) != hipSuccess || hipMemset(c->fb[s].camera.ptr, 0, 512) != hipSuccess) return bail(AWSM_ERR_OUT_OF_MEMORY);      // per-frame camera snapshot (every mode)
        c->fb[s].camera.size = 512;
    }

    if (c->overlap) {
        for (int s = 0; s < kSlots; s++) {
#ifdef AWSM_DEBUG_SWITCHES
            if (const char* m = getenv("AWSM_SHADE_CU_MASK")) {      // experiment (tools/ab_cu_mask.sh): hex words, least significant first, comma separated
                std::vector<uint32_t> words;
                for (const char* p = m; *p; ) { char* end = nullptr; words.push_back((uint32_t)strtoul(p, &end, 16)); p = (*end == ',') ? end + 1 : end; if (end == p && *p) break; }
                if (words.empty() || hipExtStreamCreateWithCUMask(&c->shade_streams[s], (uint32_t)words.size(), words.data()) != hipSuccess) return bail(AWSM_ERR_DEVICE);
            } else
#endif
            if (hipStreamCreateWithFlags(&c->shade_streams[s], hipStreamNonBlocking) != hipSuccess) return bail(AWSM_ERR_DEVICE);
            // These events order streams of this device among themselves.  Recorded with the default (system-scope) release they write the
            // L2s back and invalidate them each time — behind a raster or shading kernel that is 66 MB of dirty lines: 8-28 us per record on
            // the path between two frames (tools/frame_timeline.sh) — so: device-scope release.
            const unsigned ev_flags = hipEventDisableTiming | hipEventReleaseToDevice;
            if (hipEventCreateWithFlags(&c->ev_geom_done[s], ev_flags) != hipSuccess || hipEventCreateWithFlags(&c->ev_shade_done[s], ev_flags) != hipSuccess ||
                hipEventCreateWithFlags(&c->ev_uploads[s], ev_flags) != hipSuccess) return bail(AWSM_ERR_DEVICE);
            if (s == 0 && hipEventCreateWithFlags(&c->ev_flush, hipEventDisableTiming) != hipSuccess) return bail(AWSM_ERR_DEVICE);
        }
    }
    if (hipHostMalloc((void**)&c->counters_host, (16 + 2 * kSlots + 1) * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) return bail(AWSM_ERR_OUT_OF_MEMORY);   // last word: hand-off gates that timed out
    memset(c->counters_host, 0, (16 + 2 * kSlots + 1) * sizeof(uint32_t));
    if (c->overlap) {
        const char* e = getenv("AWSM_DEVICE_HANDOFF");      // "0": cross-stream events instead (for a profiler that serialises kernels: tools/pmc_*.sh)
        c->handoff = !(e && e[0] == '0');
        { int khz = 0; if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device) == hipSuccess && khz > 0) c->clock_khz = (uint32_t)khz; }
        double budget_ms = 4000.0;
        if (const char* p = getenv("AWSM_HANDOFF_TIMEOUT_MS")) { const double v = atof(p); if (v > 0.0) budget_ms = v; }
        c->handoff_budget_ticks = (unsigned long long)(budget_ms * (double)c->clock_khz) + 1ull;
        if (hipMalloc((void**)&c->handoff_flags, 3 * kSlots * sizeof(uint32_t)) != hipSuccess || hipMalloc((void**)&c->handoff_poison, kSlots * sizeof(uint32_t)) != hipSuccess) return bail(AWSM_ERR_OUT_OF_MEMORY);
        if (hipMemset(c->handoff_flags, 0, 3 * kSlots * sizeof(uint32_t)) != hipSuccess || hipMemset(c->handoff_poison, 0xFF, kSlots * sizeof(uint32_t)) != hipSuccess ||
            hipDeviceSynchronize() != hipSuccess) return bail(AWSM_ERR_DEVICE);
        if (const char* d = getenv("AWSM_TEST_HANDOFF_DROP")) c->handoff_test_drop = (uint32_t)atoi(d);
        // Probe: a gate on one stream, its signal on the other, both directions of every pair the frames will use.  Where the two do not run
        // side by side (kernels serialised by a counter-collecting profiler; two streams on one hardware queue) the gate gives up after ~10 ms
        // and the context orders its streams with events.
        for (int k = 0; c->handoff && k < 2 * kSlots; k++) {
            hipStream_t ss = c->shade_streams[k % kSlots], waiter = k < kSlots ? ss : c->stream, setter = k < kSlots ? c->stream : ss;
            const uint32_t serial = ++c->geom_sig[0];
            awsm_launch_handoff_wait(c->handoff_flags, serial, 10ull * c->clock_khz /* 10 ms */, handoff_timeouts(c), 0u, nullptr, 0u, waiter);
            awsm_launch_handoff_signal(c->handoff_flags, serial, nullptr, setter);
            if (hipStreamSynchronize(waiter) != hipSuccess || hipStreamSynchronize(setter) != hipSuccess) return bail(AWSM_ERR_DEVICE);
            if (*(volatile uint32_t*)handoff_timeouts(c) != 0u) { c->handoff = false; c->handoff_timeouts_seen = *(volatile uint32_t*)handoff_timeouts(c); }
        }
    }
    memset(&c->scene, 0, sizeof c->scene);
    // defaults == AwsmRendererBuilder::new (crates/renderer/src/lib.rs:168-207): black skybox, white IBL
    c->scene.skybox_rgba[3] = 1.0f;
    for (int i = 0; i < 3; i++) { c->scene.prefiltered_rgb[i] = 1.0f; c->scene.irradiance_rgb[i] = 1.0f; }
    *out = c;
    return AWSM_OK;
}

int awsm_hip_destroy(AwsmHipCtx* c) {
    if (!c) return AWSM_ERR_INVALID_ARGUMENT;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    (void)sync_shade_streams(c);
    auto fr = [](DevBuf& b) { if (b.ptr) (void)hipFree(b.ptr); b.ptr = nullptr; b.size = 0; };
    for (auto& b : c->bufs) fr(b);
    for (auto& b : c->tex) fr(b);
    for (auto& b : c->merged_vis) fr(b);
    fr(c->lut); for (auto& b : c->cube_tex) fr(b); for (auto& b : c->cube_bordered) fr(b); fr(c->digest); for (auto& b : c->shade_todo) fr(b); for (int sl = 0; sl < kSlots; sl++) { fr(c->msaa_color0[sl]); fr(c->msaa_edges[sl]); fr(c->msaa_edge_bits[sl]); fr(c->msaa_cells[sl]); } fr(c->mip_kinds); for (auto& b : c->out16) fr(b); for (auto& b : c->out32) fr(b); fr(c->comp16); fr(c->comp32); for (auto& b : c->lights_pre) fr(b);
    for (int k = 0; k < 4 * kSlots; k++) {
        FrameBufs& b = k < kSlots ? c->fb[k] : (k < 2 * kSlots ? c->tr[k - kSlots] : (k < 3 * kSlots ? c->hud[k - 2 * kSlots] : c->htr[k - 3 * kSlots]));
        fr(b.vis); fr(b.wpos); fr(b.frag_rec); fr(b.frag_color); fr(b.frag_first); fr(b.tex_slots); fr(b.draw_mat); fr(b.clip); fr(b.nrm); fr(b.tan); fr(b.tri_rec); fr(b.tri_flags); fr(b.tri_shade); fr(b.draw_lean); fr(b.draws_dev); fr(b.draw_shade); fr(b.tile_count); fr(b.tile_offset);
        fr(b.wcache); fr(b.draws_prev); fr(b.cache_mark); fr(b.block_map); fr(b.tile_cursor); fr(b.tile_order); fr(b.scan_tmp); fr(b.tile_split); fr(b.raster_scratch); fr(b.bin_list); fr(b.big_list); fr(b.counters); fr(b.camera);
    }
    for (hipStream_t st : c->shade_streams) if (st) (void)hipStreamDestroy(st);
    for (int i = 0; i < kSlots; i++) { if (c->ev_geom_done[i]) (void)hipEventDestroy(c->ev_geom_done[i]); if (c->ev_shade_done[i]) (void)hipEventDestroy(c->ev_shade_done[i]); if (c->ev_uploads[i]) (void)hipEventDestroy(c->ev_uploads[i]); }
    if (c->ev_flush) (void)hipEventDestroy(c->ev_flush);
    if (c->scene_dev) (void)hipFree(c->scene_dev);
    if (c->stage) (void)hipHostFree(c->stage);
    if (c->counters_host) (void)hipHostFree(c->counters_host);
    if (c->handoff_flags) (void)hipFree(c->handoff_flags);
    if (c->handoff_poison) (void)hipFree(c->handoff_poison);
    if (c->trace_dev) (void)hipFree(c->trace_dev);
    for (int i = 0; i < EV_COUNT; i++) if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return AWSM_OK;
}

int awsm_hip_device_info(AwsmHipCtx* c, char* name_out, size_t name_cap, uint32_t* cu_count, uint64_t* hbm_bytes) {
    if (!c) return AWSM_ERR_INVALID_ARGUMENT;
    hipDeviceProp_t prop;
    HIPCHK(c, hipGetDeviceProperties(&prop, c->device));
    if (name_out && name_cap) snprintf(name_out, name_cap, "%s (%s)", prop.name, prop.gcnArchName);
    if (cu_count) *cu_count = (uint32_t)prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (uint64_t)prop.totalGlobalMem;
    return AWSM_OK;
}

int awsm_hip_buffer_create(AwsmHipCtx* c, AwsmBuf which, size_t bytes) {
    if (!c || (int)which < 0 || which >= AWSM_BUF_COUNT) return fail(c, AWSM_ERR_INVALID_ARGUMENT, "buffer_create: bad buffer id %d", (int)which);
    HIPCHK(c, hipSetDevice(c->device));
    { int rcb = scene_write_barrier(c); if (rcb) return rcb; }
    // +16 bytes of slack so 16-byte vector loads of the last record never leave the allocation (size = the logical size)
    log_dirty(c, which, 0, ~size_t(0));
    if (bytes && c->bufs[which].ptr && c->bufs[which].size == bytes) {     // unchanged size: same allocation, cleared ("contents are NOT preserved")
        if (which == AWSM_BUF_CAMERA) memset(c->camera_host, 0, sizeof c->camera_host);
        HIPCHK(c, hipMemsetAsync(c->bufs[which].ptr, 0, bytes + 16, c->stream));
        return AWSM_OK;
    }
    int rc = dev_realloc(c, c->bufs[which], bytes ? bytes + 16 : 0, true);
    if (rc) return rc;
    if (bytes) c->bufs[which].size = bytes;
    if (which == AWSM_BUF_CAMERA) memset(c->camera_host, 0, sizeof c->camera_host);      // "contents are NOT preserved": the host shadow neither
    if (which == AWSM_BUF_LIGHTS) for (int sl = 0; sl < n_slots(c); sl++) if ((rc = dev_realloc(c, c->lights_pre[sl], std::max<size_t>(bytes / 64, 1) * 32, true))) return rc;   // 2 x float4 per 64-byte light
    c->scene_dirty = true;
    return AWSM_OK;
}

int awsm_hip_buffer_write(AwsmHipCtx* c, AwsmBuf which, size_t dst_off, const void* src, size_t len) {
    if (!c || (int)which < 0 || which >= AWSM_BUF_COUNT || (!src && len)) return fail(c, AWSM_ERR_INVALID_ARGUMENT, "buffer_write: bad argument");
    if ((dst_off & 3) || (len & 3)) return fail(c, AWSM_ERR_INVALID_ARGUMENT, "buffer_write: offset %zu / size %zu not 4-byte aligned", dst_off, len);
    DevBuf& b = c->bufs[which];
    if (!b.ptr) return fail(c, AWSM_ERR_NOT_READY, "buffer_write: buffer %d was never created", (int)which);
    if (dst_off > b.size || len > b.size - dst_off) return fail(c, AWSM_ERR_OUT_OF_RANGE, "buffer_write: [%zu,+%zu) outside buffer %d of %zu bytes", dst_off, len, (int)which, b.size);
    if (len == 0) return AWSM_OK;
    HIPCHK(c, hipSetDevice(c->device));
    if (which != AWSM_BUF_CAMERA) { int rcb = scene_write_barrier(c); if (rcb) return rcb; log_dirty(c, which, dst_off, dst_off + len); }   // the opaque pass reads a per-frame camera snapshot
    else { if (dst_off < 512) memcpy(c->camera_host + dst_off, src, std::min<size_t>(len, 512 - dst_off)); c->camera_written_since_snapshot = true; }
    uint8_t* dst = (uint8_t*)b.ptr + dst_off;
    if (len <= (1u << 20)) return upload_small(c, dst, src, len);
    // large (resize-time) uploads: the runtime stages pageable memory itself; wait so `src` is not retained
    HIPCHK(c, hipMemcpyAsync(dst, src, len, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return AWSM_OK;
}

int awsm_hip_resize(AwsmHipCtx* c, uint32_t width, uint32_t height, uint32_t msaa) {
    if (!c || width == 0 || height == 0 || width > 16384 || height > 16384) return fail(c, AWSM_ERR_INVALID_ARGUMENT, "resize: bad size %ux%u", width, height);
    if (msaa == 1) msaa = 0;
    if (msaa != 0 && msaa != 4) return fail(c, AWSM_ERR_UNSUPPORTED, "resize: MSAA x%u (the reference supports None or 4, anti_alias.rs:28-38)", msaa);
    HIPCHK(c, hipSetDevice(c->device));
    const size_t px = (size_t)width * height, samples = msaa == 4 ? 4 : 1;
    int rc;
    if ((rc = sync_all(c))) return rc;
    for (int s = 0; s < n_slots(c); s++) {
        if ((rc = dev_realloc(c, c->fb[s].vis, px * samples * 8, false))) return rc;
        HIPCHK(c, hipMemsetAsync(c->fb[s].vis.ptr, 0xFF, px * samples * 8, c->stream));
    }
    if (msaa == 4) for (int sl = 0; sl < n_slots(c); sl++) {
        const size_t blocks = (size_t)((width + 15) / 16) * ((height + 15) / 16);
        if ((rc = dev_realloc(c, c->msaa_edge_bits[sl], blocks * 4 * 16, true))) return rc;
        if ((rc = dev_realloc(c, c->msaa_cells[sl], px * 8, false))) return rc;
        if ((rc = dev_realloc(c, c->msaa_color0[sl], px * 16, false))) return rc;
        if ((rc = dev_realloc(c, c->msaa_edges[sl], blocks * 260, false))) return rc;      // per 16x16 block: count + 256 one-byte slots
    }
    c->msaa = msaa;
    for (int sl = 0; sl < n_slots(c); sl++) if ((rc = dev_realloc(c, c->shade_todo[sl], ((size_t)((width + 15) / 16) * ((height + 15) / 16) * 4 + 4 + 1024) * 4, true))) return rc;   // one entry per wavefront of the opaque grid
    for (int sl = 0; sl < n_slots(c); sl++) {
        if ((rc = dev_realloc(c, c->out16[sl], px * 8, true))) return rc;
        if ((c->flags & AWSM_CFG_PARITY_TAP) && (rc = dev_realloc(c, c->out32[sl], px * 16, true))) return rc;
    }
    c->width = width; c->height = height;
    c->y0 = c->y1 = 0;
    c->band_n = 1; c->band_r = 0; c->band_compact = 0;
    c->geometry_done = c->opaque_done = c->transparent_done = c->hud_transparent_done = false;
    return AWSM_OK;
}

int awsm_hip_set_shard_bands(AwsmHipCtx* c, uint32_t n, uint32_t r, uint32_t compact_output) {
    if (!c) return AWSM_ERR_INVALID_ARGUMENT;
    if (c->width == 0) return fail(c, AWSM_ERR_NOT_READY, "set_shard_bands before resize");
    if (n == 0 || r >= n) return fail(c, AWSM_ERR_INVALID_ARGUMENT, "set_shard_bands: need r < n (got n=%u r=%u)", n, r);
    { int rcs = sync_all(c); if (rcs) return rcs; }
    c->y0 = c->y1 = 0;                                   // bands and row ranges are alternatives
    c->band_n = n; c->band_r = n > 1 ? r : 0; c->band_compact = (n > 1 && compact_output) ? 1u : 0u;
    return AWSM_OK;
}

uint32_t awsm_hip_msaa_halo_bands(AwsmHipCtx* c) {
    return (c && c->band_n > 1 && c->height) ? ((c->height + kTile - 1) / kTile + c->band_n - 1) / c->band_n : 0u;
}

int awsm_hip_msaa_halo_export(AwsmHipCtx* c, void* dst, size_t bytes) {
    if (!c || !dst) return AWSM_ERR_INVALID_ARGUMENT;
    if (c->msaa != 4 || c->band_n <= 1) return fail(c, AWSM_ERR_INVALID_ARGUMENT, "msaa_halo_export: only for MSAA x4 with band sharding");
    if (!c->geometry_done) return fail(c, AWSM_ERR_NOT_READY, "msaa_halo_export before geometry_pass");
    const uint32_t bands = awsm_hip_msaa_halo_bands(c);
    if (bytes < (size_t)bands * 2 * c->width * 8) return fail(c, AWSM_ERR_INVALID_ARGUMENT, "msaa_halo_export: %zu bytes, %zu needed (%u bands x 2 rows x %u keys)", bytes, (size_t)bands * 2 * c->width * 8, bands, c->width);
    HIPCHK(c, hipSetDevice(c->device));
    FrameDev f;
    fill_frame(c, &f);
    awsm_launch_msaa_halo_export(&f, (unsigned long long*)dst, bands, c->stream);
    HIPCHK(c, hipGetLastError());
    return AWSM_OK;
}

int awsm_hip_msaa_halo_bind(AwsmHipCtx* c, const void* gathered, size_t bytes) {
    if (!c) return AWSM_ERR_INVALID_ARGUMENT;
    if (gathered && bytes == 0) return fail(c, AWSM_ERR_INVALID_ARGUMENT, "msaa_halo_bind: zero-sized array");
    c->msaa_halo = gathered; c->msaa_halo_bytes = bytes;
    return AWSM_OK;
}

int awsm_hip_set_shard_rows(AwsmHipCtx* c, uint32_t y0, uint32_t y1) {
    if (!c) return AWSM_ERR_INVALID_ARGUMENT;
    if (c->width == 0) return fail(c, AWSM_ERR_NOT_READY, "set_shard_rows before resize");
    { int rcs = sync_all(c); if (rcs) return rcs; }
    c->band_n = 1; c->band_r = 0; c->band_compact = 0;   // bands and row ranges are alternatives
    if (y0 == 0 && y1 == 0) { c->y0 = c->y1 = 0; return AWSM_OK; }
    if (y0 >= y1 || y1 > c->height) return fail(c, AWSM_ERR_INVALID_ARGUMENT, "set_shard_rows: need y0 < y1 <= height (got %u,%u)", y0, y1);

    c->y0 = y0; c->y1 = y1;
    return AWSM_OK;
}

int awsm_hip_texture_array_upload(AwsmHipCtx* c, uint32_t array_idx, uint32_t width, uint32_t height, uint32_t layers,
                                  uint32_t mips, AwsmTexFormat fmt, const void* texels) {
    if (!c || array_idx >= (uint32_t)kMaxTexArrays || !texels || width == 0 || height == 0 || layers == 0)
        return fail(c, AWSM_ERR_INVALID_ARGUMENT, "texture_array_upload: bad argument");
    if (fmt != AWSM_TEX_RGBA8_UNORM) return fail(c, AWSM_ERR_UNSUPPORTED, "texture_array_upload: only RGBA8_UNORM");
    if (layers > 65536u) return fail(c, AWSM_ERR_UNSUPPORTED, "texture_array_upload: %u layers (the per-draw texture records hold a 16-bit layer; WebGPU's maxTextureArrayLayers is 256..2048)", layers);
    const uint32_t full = mip_levels_full(width, height);
    if (mips == 0) mips = 1;
    if (mips > full || mips > (uint32_t)kMaxMipLevels) return fail(c, AWSM_ERR_INVALID_ARGUMENT, "texture_array_upload: %u mip levels, a %ux%u texture has at most %u", mips, width, height, full);
    HIPCHK(c, hipSetDevice(c->device));
    { int rcb = scene_write_barrier(c); if (rcb) return rcb; }
    TexArrayDev t{};
    size_t texels_total = 0;
    for (uint32_t l = 0; l < mips; l++) { t.level_off[l] = (uint32_t)texels_total; texels_total += (size_t)layers * std::max(1u, width >> l) * std::max(1u, height >> l); }
    if (texels_total > 0xFFFFFFFFull) return fail(c, AWSM_ERR_UNSUPPORTED, "texture_array_upload: array larger than 2^32 texels");
    const size_t bytes0 = (size_t)width * height * layers * 4;
    int rc = dev_realloc(c, c->tex[array_idx], texels_total * 4 + 16, false);   // +16: the shade kernel's paired row loads may read one texel past the end
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->tex[array_idx].ptr, texels, bytes0, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    t.texels = (const uint8_t*)c->tex[array_idx].ptr; t.width = width; t.height = height; t.layers = layers; t.mips = mips;
    c->scene.tex[array_idx] = t;
    c->scene.n_tex = std::max(c->scene.n_tex, array_idx + 1);
    c->scene_dirty = true;
    return AWSM_OK;
}

int awsm_hip_texture_array_generate_mips(AwsmHipCtx* c, uint32_t array_idx, const uint32_t* kind_per_layer) {
    if (!c || array_idx >= (uint32_t)kMaxTexArrays) return AWSM_ERR_INVALID_ARGUMENT;
    const TexArrayDev& t = c->scene.tex[array_idx];
    if (!t.texels) return fail(c, AWSM_ERR_NOT_READY, "generate_mips: array %u was never uploaded", array_idx);
    if (t.mips < 2) return AWSM_OK;
    HIPCHK(c, hipSetDevice(c->device));
    { int rcb = scene_write_barrier(c); if (rcb) return rcb; }
    std::vector<uint32_t> kinds(t.layers, 0u);
    if (kind_per_layer) kinds.assign(kind_per_layer, kind_per_layer + t.layers);
    int rc = dev_reserve(c, c->mip_kinds, kinds.size() * 4);
    if (rc) return rc;
    if ((rc = upload_small(c, c->mip_kinds.ptr, kinds.data(), kinds.size() * 4))) return rc;
    for (uint32_t l = 1; l < t.mips; l++)
        awsm_launch_gen_mip_level((uint8_t*)c->tex[array_idx].ptr, t.level_off[l - 1], t.level_off[l], std::max(1u, t.width >> (l - 1)), std::max(1u, t.height >> (l - 1)),
                                  std::max(1u, t.width >> l), std::max(1u, t.height >> l), t.layers, (const uint32_t*)c->mip_kinds.ptr, c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));   // mip_kinds may be reused by the next call
    return AWSM_OK;
}

int awsm_hip_texture_array_read_level(AwsmHipCtx* c, uint32_t array_idx, uint32_t level, void* out) {
    if (!c || !out || array_idx >= (uint32_t)kMaxTexArrays) return AWSM_ERR_INVALID_ARGUMENT;
    const TexArrayDev& t = c->scene.tex[array_idx];
    if (!t.texels || level >= t.mips) return fail(c, AWSM_ERR_OUT_OF_RANGE, "texture_array_read_level: array %u has %u levels", array_idx, t.mips);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const size_t n = (size_t)t.layers * std::max(1u, t.width >> level) * std::max(1u, t.height >> level) * 4;
    HIPCHK(c, hipMemcpy(out, t.texels + (size_t)t.level_off[level] * 4, n, hipMemcpyDeviceToHost));
    return AWSM_OK;
}

int awsm_hip_sampler_set(AwsmHipCtx* c, uint32_t idx, const AwsmSampler* s) {
    if (!c || !s || idx >= (uint32_t)kMaxSamplers) return fail(c, AWSM_ERR_INVALID_ARGUMENT, "sampler_set: bad argument");
    if (s->address_mode_u > 2 || s->address_mode_v > 2 || s->mag_filter > 1) return fail(c, AWSM_ERR_INVALID_ARGUMENT, "sampler_set: bad enum value");
    c->scene.samplers[idx] = *s;
    c->scene.n_samplers = std::max(c->scene.n_samplers, idx + 1);
    c->scene_dirty = true;
    return AWSM_OK;
}

int awsm_hip_env_upload(AwsmHipCtx* c, const AwsmEnv* env) {
    if (!c || !env) return AWSM_ERR_INVALID_ARGUMENT;
    HIPCHK(c, hipSetDevice(c->device));
    { int rcb = scene_write_barrier(c); if (rcb) return rcb; }
    memcpy(c->scene.skybox_rgba, env->skybox_rgba, 16);
    memcpy(c->scene.prefiltered_rgb, env->prefiltered_rgb, 16);
    memcpy(c->scene.irradiance_rgb, env->irradiance_rgb, 16);
    if (env->brdf_lut_rgba16f) {
        if (env->brdf_lut_width == 0 || env->brdf_lut_height == 0 || env->brdf_lut_width > 8192 || env->brdf_lut_height > 8192)
            return fail(c, AWSM_ERR_INVALID_ARGUMENT, "env_upload: LUT size %ux%u (1..8192 per side)", env->brdf_lut_width, env->brdf_lut_height);
        const size_t n = (size_t)env->brdf_lut_width * env->brdf_lut_height;
        DevBuf tmp;
        int rc = dev_realloc(c, tmp, n * 8, false);
        if (rc) return rc;
        if ((rc = dev_realloc(c, c->lut, n * 4, false))) { (void)hipFree(tmp.ptr); return rc; }
        hipError_t e = hipMemcpyAsync(tmp.ptr, env->brdf_lut_rgba16f, n * 8, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) { awsm_launch_rgba16f_to_rg16f((const uint16_t*)tmp.ptr, (uint32_t*)c->lut.ptr, (uint32_t)n, c->stream); e = hipStreamSynchronize(c->stream); }   // only .rg is sampled (brdf.wgsl:301)
        (void)hipFree(tmp.ptr);
        if (e != hipSuccess) return fail(c, AWSM_ERR_DEVICE, "env_upload: LUT upload failed: %s", hipGetErrorString(e));
        c->scene.lut_w = env->brdf_lut_width; c->scene.lut_h = env->brdf_lut_height;
    }
    c->scene_dirty = true;
    return AWSM_OK;
}

int awsm_hip_env_cube_upload(AwsmHipCtx* c, AwsmCube which, uint32_t size, uint32_t mips, const uint16_t* texels) {
    if (!c || (int)which < 0 || (int)which > 2) return fail(c, AWSM_ERR_INVALID_ARGUMENT, "env_cube_upload: bad cube id %d", (int)which);
    HIPCHK(c, hipSetDevice(c->device));
    { int rcb = scene_write_barrier(c); if (rcb) return rcb; }
    CubeDev cd{};
    if (!texels) {   // back to the uniform colour
        int rc = sync_all(c);
        if (rc) return rc;
        rc = dev_realloc(c, c->cube_tex[which], 0, false);
        if (!rc) rc = dev_realloc(c, c->cube_bordered[which], 0, false);
        if (rc) return rc;
        c->scene.cube[which] = cd;
        c->scene_dirty = true;
        return AWSM_OK;
    }
    if (size == 0 || size > 8192 || mips == 0 || mips > (uint32_t)kMaxMipLevels || mips > mip_levels_full(size, size))
        return fail(c, AWSM_ERR_INVALID_ARGUMENT, "env_cube_upload: %u mip levels of a %u^2 cube (1..8192 per side, at most %u levels)", mips, size, size ? mip_levels_full(size, size) : 0u);
    size_t total = 0;
    for (uint32_t l = 0; l < mips; l++) { cd.level_off[l] = (uint32_t)total; const size_t n = std::max(1u, size >> l); total += 6 * n * n; }
    int rc = dev_realloc(c, c->cube_tex[which], total * 8, false);
    if (rc) return rc;
    size_t b_total = 0;
    for (uint32_t l = 0; l < mips; l++) { cd.b_level_off[l] = (uint32_t)b_total; const size_t n = std::max(1u, size >> l) + 2; b_total += 6 * n * n; }
    const bool aproned = b_total < (1ull << 29);      // byte offsets into the aproned chain are 32-bit; a larger cube keeps the general sampler on the lean route too
    rc = dev_realloc(c, c->cube_bordered[which], aproned ? b_total * 8 : 0, false);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->cube_tex[which].ptr, texels, total * 8, hipMemcpyHostToDevice, c->stream));
    cd.texels = (const uint2*)c->cube_tex[which].ptr; cd.size = size; cd.mips = mips;
    if (aproned) awsm_launch_cube_border(&cd, (uint2*)c->cube_bordered[which].ptr, (uint32_t)b_total, c->stream);      // the apron, from the faces across the edges
    HIPCHK(c, hipStreamSynchronize(c->stream));      // `texels` is not retained
    cd.bordered = aproned ? (const uint2*)c->cube_bordered[which].ptr : nullptr;
    c->scene.cube[which] = cd;
    c->scene_dirty = true;
    return AWSM_OK;
}

int awsm_hip_brdf_lut_generate(AwsmHipCtx* c, uint32_t width, uint32_t height) {
    if (!c || width == 0 || height == 0 || width > 8192 || height > 8192) return fail(c, AWSM_ERR_INVALID_ARGUMENT, "brdf_lut_generate: bad size");
    HIPCHK(c, hipSetDevice(c->device));
    { int rcb = scene_write_barrier(c); if (rcb) return rcb; }
    int rc = dev_realloc(c, c->lut, (size_t)width * height * 4, false);
    if (rc) return rc;
    awsm_launch_brdf_lut((uint32_t*)c->lut.ptr, width, height, c->stream);
    HIPCHK(c, hipGetLastError());
    c->scene.lut_w = width; c->scene.lut_h = height;
    c->scene_dirty = true;
    return AWSM_OK;
}

int awsm_hip_read_brdf_lut(AwsmHipCtx* c, uint16_t* out) {
    if (!c || !out) return AWSM_ERR_INVALID_ARGUMENT;
    if (!c->lut.ptr) return fail(c, AWSM_ERR_NOT_READY, "no BRDF LUT");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out, c->lut.ptr, (size_t)c->scene.lut_w * c->scene.lut_h * 4, hipMemcpyDeviceToHost));
    return AWSM_OK;
}

int awsm_hip_geometry_pass(AwsmHipCtx* c, const AwsmDraw* draws, uint32_t n) {
    if (!c || (!draws && n)) return AWSM_ERR_INVALID_ARGUMENT;
    if (c->width == 0) return fail(c, AWSM_ERR_NOT_READY, "geometry_pass before resize");
    static const AwsmBuf need[] = {AWSM_BUF_TRANSFORMS, AWSM_BUF_CAMERA, AWSM_BUF_GEOM_META, AWSM_BUF_VIS_GEOM_DATA};
    if (n) for (AwsmBuf b : need) if (!c->bufs[b].ptr) return fail(c, AWSM_ERR_NOT_READY, "geometry_pass: buffer %d missing", (int)b);
    HIPCHK(c, hipSetDevice(c->device));

    // validate before touching any per-frame state: a failed call leaves the previous frame (and its slot) current
    HostTrace ht(c->frame_serial + 1u);
    std::vector<DrawDev> new_draws;
    uint64_t tris = 0, blocks = 0;
    int rc = build_draw_list(c, "geometry_pass", draws, n, AWSM_BUF_VIS_GEOM_DATA, 168u, new_draws, &tris, &blocks);
    if (rc) return rc;
    ht.mark("geometry_pass: draw list");
    if ((rc = handoff_check(c))) return rc;      // an earlier frame was dropped by a timed-out gate: said once, here or in frame_flush / frame_end
    c->hud_geometry_done = false; c->hud_merged = false;      // (a new frame: no hud pass yet)
    c->frame_serial++;
    if (c->overlap) {
        c->slot = (c->slot + 1) % kSlots;    // the opaque passes of the previous frames may still be reading the other slots
        // ... and the opaque pass of kSlots frames ago may still be reading THIS slot (its draw list, per-draw records): order everything
        // this call puts on the caller's stream — the draw-list upload included — after it
        // (asked on the host first: kSlots frames back it has usually finished, and a barrier packet on the caller's stream costs the frame 6-9 us)
        if (c->shade_pending[c->slot]) {
            if (hipEventQuery(c->ev_shade_done[c->slot]) != hipSuccess) HIPCHK(c, wait_slot_free(c));
            c->shade_pending[c->slot] = false;
        }
    }
    ht.mark("geometry_pass: slot-free wait");
    c->draws_api.assign(draws, draws + n);
    c->draws_host.swap(new_draws);
    {   // What earlier frames needed in their (triangle, tile) lists, as far as the GPU has reported it (pinned words written by k_bin_scan;
        // no wait).  A frame rendered without frame_end cannot be replayed: size the list ahead of the need instead, and count the
        // frames that did overflow (AwsmFrameStats.frames_with_dropped_bin_entries).
        uint32_t need = 0;
        for (int s = 0; s < kSlots; s++) {
            const volatile uint32_t* st = c->counters_host + 16 + 2 * s;
            const uint32_t entries = st[0], serial = st[1];      // serial: bit 31 = that frame overflowed the list it ran with (k_bin_scan's own verdict)
            if (serial == c->status_seen[s]) continue;
            c->status_seen[s] = serial;
            if (serial >> 31) c->dropped_frames++;
            need = std::max(need, entries);
        }
        for (int s = 0; s < n_slots(c); s++)
            if (need && (uint64_t)need * 4 > (uint64_t)c->fb[s].bin_capacity * 3 && c->fb[s].bin_list.ptr && !(c->flags & AWSM_CFG_SMALL_BIN_LIST)) {
                if ((rc = ensure_bin_capacity_of(c, c->fb[s], need + need / 2 + 1024))) return rc;
                if ((rc = reserve_raster_items(c, c->fb[s], false))) return rc;
            }
    }
    c->total_tris = (uint32_t)tris; c->total_verts = (uint32_t)(3 * tris); c->n_blocks = (uint32_t)blocks;
    ht.mark("geometry_pass: bin capacity");
    compose_pixel_to_view(c, FB(c));
    if ((rc = reserve_pass_buffers(c, FB(c), c->draws_host, c->total_tris, false))) return rc;
    ht.mark("geometry_pass: reserve + draw-list upload");
    if ((rc = enqueue_geometry(c))) return rc;
    c->camera_written_since_snapshot = false;
    c->geometry_done = true; c->opaque_done = false; c->transparent_done = false; c->hud_transparent_done = false; c->hud_geometry_done = false;
    return AWSM_OK;
}

// GeometryRenderPass::render(ctx, &renderables.hud, true) (render.rs:169-178, geometry/render_pass.rs:51-157 with is_hud): the hud meshes are drawn
// over the visibility targets (LoadOp::Load) with a depth buffer of their own, cleared — they hide the world whatever its depth, and depth-test among
// themselves.  Here: the same kernels into the slot's hud key buffer; the world's keys and depth stay as they are (the world transparent pass tests
// against them), and the opaque pass, the picker and the HUD transparent pass look at the hud keys.
static int enqueue_hud_geometry(AwsmHipCtx* c) {
    FrameBufs& hb = c->hud[c->slot];
    FrameDev f;
    { const bool hd = c->hud_geometry_done; c->hud_geometry_done = false; fill_frame(c, &f); c->hud_geometry_done = hd; }
    f.bin_capacity = hb.bin_capacity;
    f.shade_todo = nullptr; f.shade_todo_cap = 0; f.lean_next = nullptr; f.lean_grid = 0;      // (k_deform_transform's per-frame resets of the opaque pass's lists: the world pass's)
    if (c->hud_merged) {
        // MSAA: the hud draws in the world pass's rank space — vertices, setup records and per-triangle words go behind the world's in the SAME arrays
        // (f keeps the world's pointers), only the tile tables, the lists and the keys are this pass's own
        f.n_draws = (uint32_t)c->hud_combined.size();
        f.total_tris = c->total_tris + c->hud_total_tris; f.total_verts = 3u * f.total_tris;
        f.rank0 = c->total_tris; f.block0 = c->n_blocks;
    } else {
    f.n_draws = (uint32_t)c->hud_draws_host.size();
    f.total_tris = c->hud_total_tris; f.total_verts = 3u * c->hud_total_tris;
    f.draws = (const DrawDev*)hb.draws_dev.ptr;
    f.clip = (float4*)hb.clip.ptr; f.nrm = (float4*)hb.nrm.ptr; f.tan = (float4*)hb.tan.ptr;
    f.tri_info = (uint32_t*)hb.tri_flags.ptr;
    f.tri_shade = nullptr; f.draw_lean = nullptr;
    f.tri_rec = (TriRec*)hb.tri_rec.ptr;
    }
    f.tile_count = (uint32_t*)hb.tile_count.ptr; f.tile_offset = (uint32_t*)hb.tile_offset.ptr;
    f.tile_cursor = (uint32_t*)hb.tile_cursor.ptr; f.bin_list = (uint32_t*)hb.bin_list.ptr;
    f.tile_order = (uint32_t*)hb.tile_order.ptr; f.scan_tmp = (uint32_t*)hb.scan_tmp.ptr;
    f.tile_split = (uint32_t*)hb.tile_split.ptr; f.raster_scratch = (unsigned long long*)hb.raster_scratch.ptr;
    f.raster_extra_cap = hb.raster_extra_cap; f.raster_slot_cap = hb.raster_slot_cap;
    f.big_list = (uint32_t*)hb.big_list.ptr;
    f.counters = (uint32_t*)hb.counters.ptr;
    f.host_bin_status = nullptr;                 // the world pass's status words are not this pass's to write
    f.camera_snap = nullptr; f.camera_snap_words = 0;
    f.vis = (unsigned long long*)hb.vis.ptr;     // every tile of the frame is written: a tile without hud triangles becomes "no hit"
    f.hud_vis = nullptr;
    int rc;
    if ((rc = sync_scene(c))) return rc;
    const uint32_t n_tiles = f.tiles_x * f.tiles_y;
    awsm_launch_transform(c->scene_dev, &f, c->hud_n_blocks, c->stream);
    if (n_tiles) {
        awsm_launch_bin_count(&f, c->stream); awsm_launch_bin_big(&f, 0, c->stream);
        awsm_launch_bin_scan(&f, c->stream);
        awsm_launch_bin_fill(&f, c->stream); awsm_launch_bin_big(&f, 1, c->stream);
        awsm_launch_raster(&f, c->stream);
    }
    if (c->hud_merged) {      // what the opaque pass reads: per sample the hud triangle (where one was drawn) under the world's depth
        const size_t spp = c->msaa == 4 ? 4 : 1, first = (size_t)f.y0 * c->width * spp, n = (size_t)(f.y1 - f.y0) * c->width * spp;
        awsm_launch_hud_merge((const unsigned long long*)FB(c).vis.ptr, (const unsigned long long*)hb.vis.ptr, (unsigned long long*)c->merged_vis[c->slot].ptr, first, n, c->stream);
    }
    HIPCHK(c, hipGetLastError());
    return AWSM_OK;
}

int awsm_hip_hud_geometry_pass(AwsmHipCtx* c, const AwsmDraw* draws, uint32_t n) {
    if (!c || (!draws && n)) return AWSM_ERR_INVALID_ARGUMENT;
    if (!c->geometry_done || c->opaque_done) return fail(c, AWSM_ERR_NOT_READY, "hud_geometry_pass goes between the geometry pass and the opaque pass of a frame");
    HIPCHK(c, hipSetDevice(c->device));
    uint64_t tris = 0, blocks = 0;
    int rc = build_draw_list(c, "hud_geometry_pass", draws, n, AWSM_BUF_VIS_GEOM_DATA, 168u, c->hud_draws_host, &tris, &blocks);
    if (rc) return rc;
    c->hud_total_tris = (uint32_t)tris; c->hud_n_blocks = (uint32_t)blocks;
    c->hud_merged = false;
    if (!tris) { c->hud_geometry_done = false; return AWSM_OK; }
    FrameBufs& hb = c->hud[c->slot];
    const size_t px = (size_t)c->width * c->height, spp = c->msaa == 4 ? 4 : 1;
    if ((rc = dev_reserve(c, hb.vis, px * 8 * spp))) return rc;
    if ((rc = reserve_pass_buffers(c, hb, c->hud_draws_host, c->hud_total_tris, false))) return rc;      // tile tables, lists (and, single-sampled, the pass's own vertex arrays)
    if (c->msaa == 4) {
        // one rank space with the world pass (see AwsmHipCtx::hud_merged)
        if ((uint64_t)c->total_tris + tris > 0x55555555ull) return fail(c, AWSM_ERR_UNSUPPORTED, "hud_geometry_pass: more than 2^32/3 triangles in the frame");
        c->hud_combined = c->draws_host;
        for (DrawDev d : c->hud_draws_host) { d.first_tri += c->total_tris; d.first_block += c->n_blocks; c->hud_combined.push_back(d); }
        if (c->hud_combined.size() >= (1u << 24)) return fail(c, AWSM_ERR_UNSUPPORTED, "hud_geometry_pass: more than 2^24 draws in the frame");
        FrameBufs& wb = FB(c);
        const void* before[3] = {wb.clip.ptr, wb.tri_rec.ptr, wb.tri_flags.ptr};
        if ((rc = ensure_pass_capacity(c, wb, c->hud_combined, c->total_tris + c->hud_total_tris, false))) return rc;      // room for the hud draws behind the world's
        // The world's list stays cached as the world pass uploaded it (re-uploaded only if the list buffer just moved); the hud draws go behind it with a
        // cache of their own — a static scene with a hud mesh uploads nothing per frame and keeps its per-draw records (the combined list used to
        // replace the world list's cache, so both were uploaded every frame and k_resolve_draws never rested).
        if ((rc = upload_draw_list(c, wb, c->draws_host, true))) return rc;
        {
            const size_t at = c->draws_host.size();
            const std::vector<DrawDev> tail(c->hud_combined.begin() + (long)at, c->hud_combined.end());
            const bool same = wb.tail_uploaded_valid && wb.tail_uploaded_ptr == wb.draws_dev.ptr && wb.tail_uploaded_at == at && wb.tail_uploaded.size() == tail.size() &&
                              memcmp(wb.tail_uploaded.data(), tail.data(), tail.size() * sizeof(DrawDev)) == 0;
            if (!same) {
                if ((rc = upload_draws_at(c, wb, at, tail))) return rc;
                wb.tail_uploaded = tail; wb.tail_uploaded_ptr = wb.draws_dev.ptr; wb.tail_uploaded_at = at; wb.tail_uploaded_valid = true; wb.draws_version++;
                // The upload sits on the caller's stream BEHIND the world geometry pass, after ev_uploads[slot] was recorded: a per-draw resolve that went
                // ahead on that event would read hud entries that are not there yet (out-of-bounds meta offsets on a slot's first frame, the
                // list of two frames ago afterwards).  The resolve of this frame takes the late order, behind the geometry pass's hand-off.
                c->uploads_recorded[c->slot] = false;
            }
        }
        if ((rc = dev_reserve(c, c->merged_vis[c->slot], px * 8 * spp))) return rc;
        if (before[0] != wb.clip.ptr || before[1] != wb.tri_rec.ptr || before[2] != wb.tri_flags.ptr) {
            // the world pass's arrays moved (first frame with this much hud geometry): its results went with them — run it again into the new ones
            if ((rc = enqueue_geometry(c, true))) return rc;
        }
        c->hud_merged = true;
    }
    if ((rc = enqueue_hud_geometry(c))) return rc;
    c->hud_geometry_done = true;
    return AWSM_OK;
}

int awsm_hip_opaque_pass(AwsmHipCtx* c, const AwsmOpaqueParams* p) {
    if (!c || !p) return AWSM_ERR_INVALID_ARGUMENT;
    if (p->mipmap > 1) return fail(c, AWSM_ERR_INVALID_ARGUMENT, "opaque_pass: mipmap must be 0 (MipmapMode::None) or 1 (MipmapMode::Gradient)");
    if (c->width == 0) return fail(c, AWSM_ERR_NOT_READY, "opaque_pass before resize");
    if (p->has_opaque) {
        if (!c->geometry_done) return fail(c, AWSM_ERR_NOT_READY, "opaque_pass before geometry_pass");
        static const AwsmBuf need[] = {AWSM_BUF_CAMERA, AWSM_BUF_GEOM_META, AWSM_BUF_MATERIAL_META, AWSM_BUF_MATERIALS, AWSM_BUF_ATTR_INDEX,
                                       AWSM_BUF_ATTR_DATA, AWSM_BUF_TEXTURE_TRANSFORMS, AWSM_BUF_LIGHTS_INFO, AWSM_BUF_LIGHTS};
        for (AwsmBuf b : need) if (!c->bufs[b].ptr) return fail(c, AWSM_ERR_NOT_READY, "opaque_pass: buffer %d missing", (int)b);
        if (!c->lut.ptr) return fail(c, AWSM_ERR_NOT_READY, "opaque_pass: no BRDF LUT (env_upload or brdf_lut_generate)");
    }
    if (c->msaa == 4 && c->band_n > 1 && p->has_opaque) {
        const size_t need = (size_t)c->band_n * awsm_hip_msaa_halo_bands(c) * 2 * c->width * 8;
        if (!c->msaa_halo || c->msaa_halo_bytes < need)
            return fail(c, AWSM_ERR_NOT_READY, "opaque_pass: MSAA with band sharding needs the gathered halo keys (awsm_hip_msaa_halo_export -> all-gather -> awsm_hip_msaa_halo_bind, %zu bytes)", need);
    }
    HIPCHK(c, hipSetDevice(c->device));
    c->last_opaque = *p;
    if (!p->has_opaque && c->camera_written_since_snapshot && c->bufs[AWSM_BUF_CAMERA].ptr) {
        // the "empty" pipeline (skybox only) may run without a geometry pass of its own: it then has no snapshot of the camera it was given.  Rare
        // and cheap to make right: drain, copy, go on.
        int rcs = sync_all(c);
        if (rcs) return rcs;
        compose_pixel_to_view(c, FB(c));
        HIPCHK(c, hipMemcpy(FB(c).camera.ptr, c->bufs[AWSM_BUF_CAMERA].ptr, std::min<size_t>(512, c->bufs[AWSM_BUF_CAMERA].size), hipMemcpyDeviceToDevice));
        c->camera_written_since_snapshot = false;
    }
    int rc = enqueue_opaque(c);
    if (rc) return rc;
    c->opaque_done = true; c->transparent_done = false; c->hud_transparent_done = false;
    return AWSM_OK;
}

static int transparent_pass_body(AwsmHipCtx* c, const AwsmDraw* draws, uint32_t n, bool hud);
static int transparent_pass_impl(AwsmHipCtx* c, const AwsmDraw* draws, uint32_t n, bool hud) {
    const int rc = transparent_pass_body(c, draws, n, hud);
    if (c) c->hud_transparent = false;      // (whatever the outcome: TR(c) names the world pass's buffers outside a HUD pass's own enqueue)
    return rc;
}
static int transparent_pass_body(AwsmHipCtx* c, const AwsmDraw* draws, uint32_t n, bool hud) {
    if (!c || (!draws && n)) return AWSM_ERR_INVALID_ARGUMENT;
    if (!c->geometry_done || !c->opaque_done) return fail(c, AWSM_ERR_NOT_READY, "transparent_pass needs the geometry and opaque passes of the same frame first");
    if (hud && !c->transparent_done) return fail(c, AWSM_ERR_NOT_READY, "hud_transparent_pass draws over the composite: call transparent_pass first (n_draws = 0 is valid)");
    c->hud_transparent = hud;
    const bool sharded = c->band_n > 1 || (c->y1 != 0 && !(c->y0 == 0 && c->y1 >= c->height));
    if (sharded && !c->opaque_src)
        return fail(c, AWSM_ERR_UNSUPPORTED, "transparent_pass on a sharded context: screen-space transmission reads the whole opaque image — gather it and bind it with awsm_hip_bind_opaque_source first");
    if (c->opaque_src && c->opaque_src_bytes < (size_t)c->width * c->height * 8)
        return fail(c, AWSM_ERR_INVALID_ARGUMENT, "transparent_pass: the bound opaque source holds %zu bytes, the frame needs %zu", c->opaque_src_bytes, (size_t)c->width * c->height * 8);
    static const AwsmBuf need[] = {AWSM_BUF_TRANSFORMS, AWSM_BUF_CAMERA, AWSM_BUF_GEOM_META, AWSM_BUF_MATERIAL_META, AWSM_BUF_MATERIALS, AWSM_BUF_ATTR_INDEX,
                                   AWSM_BUF_ATTR_DATA, AWSM_BUF_TEXTURE_TRANSFORMS, AWSM_BUF_LIGHTS_INFO, AWSM_BUF_LIGHTS, AWSM_BUF_TRANSPARENCY_GEOM_DATA};
    if (n) {
        for (AwsmBuf b : need) if (!c->bufs[b].ptr) return fail(c, AWSM_ERR_NOT_READY, "transparent_pass: buffer %d missing", (int)b);
        if (!c->lut.ptr) return fail(c, AWSM_ERR_NOT_READY, "transparent_pass: no BRDF LUT (env_upload or brdf_lut_generate)");
    }
    HIPCHK(c, hipSetDevice(c->device));
    uint64_t tris = 0, blocks = 0;
    const int which = hud ? 1 : 0;
    int rc = build_draw_list(c, hud ? "hud_transparent_pass" : "transparent_pass", draws, n, AWSM_BUF_TRANSPARENCY_GEOM_DATA, 0u, c->tr_draws_host[which], &tris, &blocks);
    if (rc) return rc;
    c->tr_total_tris[which] = (uint32_t)tris; c->tr_n_blocks[which] = (uint32_t)blocks;
    const size_t px = (size_t)c->width * c->height;
    if (!c->bound_comp && (rc = dev_reserve(c, c->comp16, px * 8))) return rc;
    if (c->bound_comp && c->bound_comp_bytes < px * 8) return fail(c, AWSM_ERR_INVALID_ARGUMENT, "transparent_pass: bound composite holds %zu bytes, the frame needs %zu", c->bound_comp_bytes, px * 8);
    if ((c->flags & AWSM_CFG_PARITY_TAP) && (rc = dev_reserve(c, c->comp32, px * 16))) return rc;
    if ((rc = reserve_pass_buffers(c, TR(c), c->tr_draws_host[which], c->tr_total_tris[which], true))) return rc;
    if ((rc = dev_reserve(c, TR(c).frag_first, px * 4))) return rc;
    if ((rc = ensure_fragment_capacity(c, TR(c), (c->flags & AWSM_CFG_SMALL_BIN_LIST) ? 4096u : (uint32_t)std::max<size_t>(px / 2, 1u << 20)))) return rc;
    if (c->overlap) {   // the draw-list upload went to the caller's stream; the pass runs on the shade stream
        HIPCHK(c, hipEventRecord(c->ev_geom_done[c->slot], c->stream));
        HIPCHK(c, hipStreamWaitEvent(shade_stream_of(c), c->ev_geom_done[c->slot], 0));
    }
    if ((rc = enqueue_transparent(c))) return rc;
    if (hud) c->hud_transparent_done = true; else { c->transparent_done = true; c->hud_transparent_done = false; }
    return AWSM_OK;
}
int awsm_hip_transparent_pass(AwsmHipCtx* c, const AwsmDraw* draws, uint32_t n) { return transparent_pass_impl(c, draws, n, false); }
// MaterialTransparentRenderPass::render(ctx, renderables.hud, true) (render.rs:301-312, begin_hud_transparent_pass :490-521): the hud meshes drawn
// forward over what the frame holds so far (colours LoadOp::Load -> the composite, in place), depth-tested among themselves against hud_depth, cleared.
int awsm_hip_hud_transparent_pass(AwsmHipCtx* c, const AwsmDraw* draws, uint32_t n) { return transparent_pass_impl(c, draws, n, true); }

int awsm_hip_set_stage_timers(AwsmHipCtx* c, int enabled) {
    if (!c) return AWSM_ERR_INVALID_ARGUMENT;
    c->stage_timers = enabled != 0;
    return AWSM_OK;
}

int awsm_hip_frame_trace(AwsmHipCtx* c, uint32_t capacity) {
    if (!c || capacity > (1u << 20)) return AWSM_ERR_INVALID_ARGUMENT;
    HIPCHK(c, hipSetDevice(c->device));
    { int rcs = sync_all(c); if (rcs) return rcs; }
    if (c->trace_dev) { HIPCHK(c, hipFree(c->trace_dev)); c->trace_dev = nullptr; c->trace_cap = 0; }
    if (!capacity) return AWSM_OK;
    HIPCHK(c, hipMalloc((void**)&c->trace_dev, (size_t)3 * capacity * 8));
    HIPCHK(c, hipMemset(c->trace_dev, 0, (size_t)3 * capacity * 8));
    c->trace_cap = capacity;
    return AWSM_OK;
}

int awsm_hip_read_frame_trace(AwsmHipCtx* c, uint64_t* ticks_out, uint32_t n_frames, uint32_t* last_serial_out, uint32_t* ticks_per_ms_out) {
    if (!c || !ticks_out) return AWSM_ERR_INVALID_ARGUMENT;
    if (!c->trace_dev) return fail(c, AWSM_ERR_NOT_READY, "read_frame_trace: tracing is off (awsm_hip_frame_trace)");
    if (n_frames == 0 || n_frames > c->trace_cap) return fail(c, AWSM_ERR_OUT_OF_RANGE, "read_frame_trace: %u frames asked, the ring holds %u", n_frames, c->trace_cap);
    HIPCHK(c, hipSetDevice(c->device));
    { int rcs = sync_all(c); if (rcs) return rcs; }
    std::vector<uint64_t> ring((size_t)3 * c->trace_cap);
    HIPCHK(c, hipMemcpy(ring.data(), c->trace_dev, ring.size() * 8, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n_frames; i++) {      // oldest first: frames serial - n + 1 .. serial
        const uint32_t serial = c->frame_serial - (n_frames - 1u - i);
        for (int w = 0; w < 3; w++) ticks_out[(size_t)i * 3 + w] = ring[(size_t)w * c->trace_cap + serial % c->trace_cap];
    }
    if (last_serial_out) *last_serial_out = c->frame_serial;
    if (ticks_per_ms_out) { int khz = 0; *ticks_per_ms_out = hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device) == hipSuccess && khz > 0 ? (uint32_t)khz : 100000u; }
    return AWSM_OK;
}

int awsm_hip_frame_flush(AwsmHipCtx* c) {
    if (!c) return AWSM_ERR_INVALID_ARGUMENT;
    // Nothing is batched host-side.  In overlap mode the opaque passes run on an internal stream: order them before whatever the
    // caller enqueues next on its own stream (a collective over the image, a copy, ...).
    { int rch = handoff_check(c); if (rch) return rch; }
    int rc = scene_write_barrier(c, false);
    if (rc || !c->overlap) return rc;
    // The streams' internal events release to device scope only (no cache write-back per record: see awsm_hip_create).  What follows a flush on the
    // caller's stream may be a DMA engine, a peer GPU or the host: one system-scope release here, where the caller asked for it.
    HIPCHK(c, hipEventRecord(c->ev_flush, c->stream));
    return AWSM_OK;
}

int awsm_hip_frame_end(AwsmHipCtx* c, AwsmFrameStats* out) {
    if (!c) return AWSM_ERR_INVALID_ARGUMENT;
    HIPCHK(c, hipSetDevice(c->device));
    bool still_over = false;
    for (int attempt = 0;; attempt++) {
        if (out && c->geometry_done && c->has_opaque_for_stats()) {   // stats only: count covered pixels from the visibility buffer
            FrameDev f;
            fill_frame(c, &f);
            HIPCHK(c, hipMemsetAsync((uint32_t*)FB(c).counters.ptr + 3, 0, sizeof(uint32_t), c->stream));
            awsm_launch_count_covered(&f, c->stream);
        }
        HIPCHK(c, hipMemcpyAsync(c->counters_host, FB(c).counters.ptr, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->cache_blocks_last = 0;
        if (out && c->geometry_done && c->stage_timers && c->n_blocks && FB(c).cache_mark.size >= (size_t)c->n_blocks * 4) {
            std::vector<uint32_t> marks(c->n_blocks);
            HIPCHK(c, hipMemcpy(marks.data(), FB(c).cache_mark.ptr, marks.size() * 4, hipMemcpyDeviceToHost));
            for (uint32_t m : marks) c->cache_blocks_last += m == c->frame_serial ? 1u : 0u;
        }
        if (c->overlap) { HIPCHK(c, sync_shade_streams(c)); for (bool& b : c->shade_pending) b = false; }
        { int rch = handoff_check(c); if (rch) return rch; }      // a gate ended unopened: the frame it guarded was dropped
        memset(c->counters_host + 8, 0, 8 * sizeof(uint32_t));
        uint32_t hud_fwd[8] = {}, hud_geo[8] = {};      // the HUD transparent pass's and the HUD geometry pass's own counters (same layout)
        if (c->transparent_done) HIPCHK(c, hipMemcpy(c->counters_host + 8, c->tr[c->slot].counters.ptr, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost));
        if (c->hud_transparent_done) HIPCHK(c, hipMemcpy(hud_fwd, c->htr[c->slot].counters.ptr, sizeof hud_fwd, hipMemcpyDeviceToHost));
        if (c->hud_geometry_done) HIPCHK(c, hipMemcpy(hud_geo, c->hud[c->slot].counters.ptr, sizeof hud_geo, hipMemcpyDeviceToHost));
        const bool geom_over = c->geometry_done && c->counters_host[2] != 0, fwd_over = c->transparent_done && c->counters_host[10] != 0;
        const bool frag_over = c->transparent_done && c->counters_host[14] != 0;      // a pixel's fragment list did not fit
        const bool hud_geo_over = c->hud_geometry_done && hud_geo[2] != 0, hud_fwd_over = c->hud_transparent_done && hud_fwd[2] != 0, hud_frag_over = c->hud_transparent_done && hud_fwd[6] != 0;
        still_over = geom_over || fwd_over || frag_over || hud_geo_over || hud_fwd_over || hud_frag_over;
        if (!still_over || attempt >= 4) break;
        // a list overflowed: grow it to the measured need and replay from the first pass that lost entries.  The passes of a frame are replayed in their
        // order — the HUD transparent pass blends into the composite in place, so it is never replayed by itself: the world transparent pass (which
        // starts from the opaque image) always goes first.
        int rc;
        c->overflow_retries++;
        if (geom_over) {
            if ((rc = ensure_bin_capacity(c, c->counters_host[1] + c->counters_host[1] / 4 + 1024))) return rc;
            if ((rc = reserve_raster_items(c, FB(c), false))) return rc;
            if ((rc = enqueue_geometry(c, true))) return rc;
        }
        if (hud_geo_over || (geom_over && c->hud_geometry_done && c->hud_merged)) {      // (merged keys: a new world pass needs a new merge)
            FrameBufs& hb = c->hud[c->slot];
            if (hud_geo_over && (rc = ensure_bin_capacity_of(c, hb, hud_geo[1] + hud_geo[1] / 4 + 1024))) return rc;
            if (hud_geo_over && (rc = reserve_raster_items(c, hb, false))) return rc;
            if ((rc = enqueue_hud_geometry(c))) return rc;
        }
        if ((geom_over || hud_geo_over) && c->opaque_done && (rc = enqueue_opaque(c))) return rc;
        if (fwd_over && (rc = ensure_bin_capacity_of(c, c->tr[c->slot], c->counters_host[9] + c->counters_host[9] / 4 + 1024))) return rc;
        if (frag_over && (rc = ensure_fragment_capacity(c, c->tr[c->slot], c->counters_host[13] + c->counters_host[13] / 4 + 1024))) return rc;
        if (hud_fwd_over && (rc = ensure_bin_capacity_of(c, c->htr[c->slot], hud_fwd[1] + hud_fwd[1] / 4 + 1024))) return rc;
        if (hud_frag_over && (rc = ensure_fragment_capacity(c, c->htr[c->slot], hud_fwd[5] + hud_fwd[5] / 4 + 1024))) return rc;
        if (c->transparent_done) {
            c->hud_transparent = false;
            if ((rc = enqueue_transparent(c))) return rc;
            if (c->hud_transparent_done) {
                c->hud_transparent = true;
                rc = enqueue_transparent(c);
                c->hud_transparent = false;
                if (rc) return rc;
            }
        }
    }
    if (still_over) return fail(c, AWSM_ERR_DEVICE, "bin / fragment list overflow persisted after retries");
    if (out) {
        // the caller says how much of the struct it knows (struct_size, ABI 2): nothing beyond that is written
        const uint32_t caller_size = out->struct_size;
        if (caller_size < 8u || caller_size > 4096u) return fail(c, AWSM_ERR_INVALID_ARGUMENT, "frame_end: AwsmFrameStats.struct_size = %u (set it to sizeof(AwsmFrameStats))", caller_size);
        AwsmFrameStats st_local;
        AwsmFrameStats* const caller_out = out;
        out = &st_local;
        memset(out, 0, sizeof *out);
        auto ms = [&](int a, int b) { float t = 0.0f; if (c->ev_valid[a] && c->ev_valid[b] && hipEventElapsedTime(&t, c->ev[a], c->ev[b]) == hipSuccess) return t; return 0.0f; };
        if (c->geometry_done) {
            out->ms_transform = ms(EV_START, EV_TRANSFORM);
            out->ms_bin = ms(EV_TRANSFORM, EV_BIN);
            out->ms_raster = ms(EV_BIN, EV_RASTER);
        }
        if (c->opaque_done) { out->ms_shade = ms(EV_SHADE_BEGIN, EV_SHADE); out->ms_shade_lean = ms(EV_SHADE_BEGIN, EV_SHADE_LEAN); }
        if (c->transparent_done) { out->ms_forward = ms(EV_FWD_BEGIN, EV_FWD); out->forward_triangles = c->tr_total_tris[0] + (c->hud_transparent_done ? c->tr_total_tris[1] : 0u); out->forward_fragment_slots = c->counters_host[13]; }
        out->ms_total = (c->geometry_done ? ms(EV_START, EV_RASTER) : 0.0f) + out->ms_shade + out->ms_forward;   // the two passes may run on different streams
        out->triangles_in = c->total_tris;
        out->triangles_binned = c->counters_host[0];
        out->bin_entries = c->counters_host[1];
        out->covered_pixels = c->counters_host[3];
        out->bin_overflow_retries = c->overflow_retries;
        out->frames_with_dropped_bin_entries = c->dropped_frames;
        out->handoff_gate_timeouts = c->handoff_dropped_frames;
        out->geometry_cache_blocks = (c->geometry_done && c->stage_timers) ? c->cache_blocks_last : 0u;
        out->geometry_blocks = c->geometry_done ? c->n_blocks : 0u;
        if (c->opaque_done && c->shade_todo[c->slot].ptr && c->last_opaque.has_opaque && !(c->flags & AWSM_CFG_GENERAL_SHADE_ONLY) && !c->draws_host.empty())
            HIPCHK(c, hipMemcpy(&out->shade_general_wavefronts, c->shade_todo[c->slot].ptr, 4, hipMemcpyDeviceToHost));
        out->struct_size = (uint32_t)std::min<size_t>(caller_size, sizeof(AwsmFrameStats));
        memcpy(caller_out, out, out->struct_size);
    }
    return AWSM_OK;
}

int awsm_hip_bind_output(AwsmHipCtx* c, void* device_ptr, size_t bytes) {
    if (!c) return AWSM_ERR_INVALID_ARGUMENT;
    if (device_ptr && bytes == 0) return fail(c, AWSM_ERR_INVALID_ARGUMENT, "bind_output: zero-sized image");
    c->bound_out = device_ptr; c->bound_out_bytes = bytes; c->out_first_row = 0; c->out_rows_mode = false;   // checked against the shard layout when the opaque pass is enqueued
    return AWSM_OK;
}

int awsm_hip_bind_output_rows(AwsmHipCtx* c, void* device_ptr, size_t bytes, uint32_t first_row) {
    if (!c) return AWSM_ERR_INVALID_ARGUMENT;
    if (!device_ptr || bytes == 0) return fail(c, AWSM_ERR_INVALID_ARGUMENT, "bind_output_rows: null or zero-sized image");
    c->bound_out = device_ptr; c->bound_out_bytes = bytes; c->out_first_row = first_row; c->out_rows_mode = true;
    return AWSM_OK;
}

void* awsm_hip_output_device_ptr(AwsmHipCtx* c) { return c ? (c->bound_out ? c->bound_out : c->out16[c->slot].ptr) : nullptr; }

int awsm_hip_read_visibility(AwsmHipCtx* c, uint64_t* keys_out) {
    if (!c || !keys_out) return AWSM_ERR_INVALID_ARGUMENT;
    if (!FB(c).vis.ptr) return fail(c, AWSM_ERR_NOT_READY, "read_visibility before resize");
    HIPCHK(c, hipSetDevice(c->device));
    { int rcs = sync_all(c); if (rcs) return rcs; }
    HIPCHK(c, hipMemcpy(keys_out, FB(c).vis.ptr, (size_t)c->width * c->height * (c->msaa == 4 ? 4 : 1) * 8, hipMemcpyDeviceToHost));
    return AWSM_OK;
}

#ifdef AWSM_STAMP
// diagnostic builds only: the geometry kernels' stamps of the last frame, [4 kernels][16384 workgroups][8] u64
extern "C" int awsm_hip_debug_read_stamps(AwsmHipCtx* c, unsigned long long* out) {
    FrameDev f;
    fill_frame(c, &f);
    if (sync_all(c)) return AWSM_ERR_DEVICE;
    return hipMemcpy(out, f.stamps, 4ull * 16384 * 8 * 8, hipMemcpyDeviceToHost) == hipSuccess ? AWSM_OK : AWSM_ERR_DEVICE;
}
#endif

int awsm_hip_read_gbuffer(AwsmHipCtx* c, float* out6) {
    if (!c || !out6) return AWSM_ERR_INVALID_ARGUMENT;
    if (!c->geometry_done) return fail(c, AWSM_ERR_NOT_READY, "read_gbuffer before geometry_pass");
    if (c->msaa != 0 || c->band_n > 1) return fail(c, AWSM_ERR_UNSUPPORTED, "read_gbuffer: single-sampled, unsharded-by-bands frames only");
    HIPCHK(c, hipSetDevice(c->device));
    { int rcs = sync_all(c); if (rcs) return rcs; }
    FrameDev f;
    fill_frame(c, &f);
    const size_t bytes = (size_t)c->width * c->height * 6 * sizeof(float);
    float* dev = nullptr;
    HIPCHK(c, hipMalloc((void**)&dev, bytes));
    awsm_launch_gbuffer_dump(&f, dev, c->stream);
    hipError_t e = hipMemcpyAsync(out6, dev, bytes, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(dev);
    HIPCHK(c, e);
    return AWSM_OK;
}

int awsm_hip_stream_handoff(AwsmHipCtx* c) {
    if (!c) return AWSM_ERR_INVALID_ARGUMENT;
    return c->overlap && c->handoff ? 1 : 0;
}

int awsm_hip_visibility_digest(AwsmHipCtx* c, uint64_t* out2) {
    if (!c || !out2) return AWSM_ERR_INVALID_ARGUMENT;
    if (!FB(c).vis.ptr) return fail(c, AWSM_ERR_NOT_READY, "visibility_digest before resize");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = dev_reserve(c, c->digest, 16);
    if (rc) return rc;
    HIPCHK(c, hipMemsetAsync(c->digest.ptr, 0, 16, c->stream));
    awsm_launch_vis_digest((const unsigned long long*)FB(c).vis.ptr, (size_t)c->width * c->height * (c->msaa == 4 ? 4 : 1), (unsigned long long*)c->digest.ptr, c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(out2, c->digest.ptr, 16, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return AWSM_OK;
}

int awsm_hip_read_visibility_unpacked(AwsmHipCtx* c, uint32_t* tri_id, uint32_t* meta_off, float* depth) {
    if (!c) return AWSM_ERR_INVALID_ARGUMENT;
    if (!FB(c).vis.ptr) return fail(c, AWSM_ERR_NOT_READY, "read_visibility before resize");
    const size_t n = (size_t)c->width * c->height * (c->msaa == 4 ? 4 : 1);
    std::vector<uint64_t> keys(n);
    int rc = awsm_hip_read_visibility(c, keys.data());
    if (rc) return rc;
    std::vector<uint32_t> meta(c->draws_host.size());
    for (size_t d = 0; d < c->draws_host.size(); d++)
        HIPCHK(c, hipMemcpy(&meta[d], (const uint8_t*)c->bufs[AWSM_BUF_GEOM_META].ptr + c->draws_host[d].geom_meta_off + 36, 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; i++) {
        if (keys[i] == ~0ull) {   // cleared texel: 0xFFFF per channel -> join32 == U32_MAX (geometry/render_pass.rs:22-30)
            if (tri_id) tri_id[i] = 0xFFFFFFFFu;
            if (meta_off) meta_off[i] = 0xFFFFFFFFu;
            if (depth) depth[i] = 1.0f;
            continue;
        }
        const uint32_t rank = 0xFFFFFFFFu - (uint32_t)(keys[i] & 0xFFFFFFFFull);
        size_t lo = 0, hi = c->draws_host.size();
        while (hi - lo > 1) { size_t mid = (lo + hi) / 2; if (c->draws_host[mid].first_tri <= rank) lo = mid; else hi = mid; }
        if (tri_id) tri_id[i] = rank - c->draws_host[lo].first_tri;
        if (meta_off) meta_off[i] = meta[lo];
        if (depth) { uint32_t b = (uint32_t)(keys[i] >> 32); memcpy(&depth[i], &b, 4); }
    }
    return AWSM_OK;
}

int awsm_hip_pick(AwsmHipCtx* c, int32_t x, int32_t y, AwsmPick* out) {
    if (!c || !out) return AWSM_ERR_INVALID_ARGUMENT;
    if (!c->geometry_done) return fail(c, AWSM_ERR_NOT_READY, "pick before geometry_pass");
    if (!c->bufs[AWSM_BUF_MATERIAL_META].ptr || !c->bufs[AWSM_BUF_GEOM_META].ptr) return fail(c, AWSM_ERR_NOT_READY, "pick: meta buffers missing");
    HIPCHK(c, hipSetDevice(c->device));
    FrameDev f;
    fill_frame(c, &f);
    int rc = sync_scene(c);
    if (rc) return rc;
    uint32_t* dev_out = (uint32_t*)FB(c).counters.ptr + 8;          // 4 words after the frame counters
    awsm_launch_pick(c->scene_dev, &f, x, y, dev_out, c->stream);
    HIPCHK(c, hipGetLastError());
    uint32_t host_out[4] = {0, 0, 0, 0};
    HIPCHK(c, hipMemcpyAsync(host_out, dev_out, sizeof host_out, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    out->valid = host_out[0]; out->mesh_key_high = host_out[1]; out->mesh_key_low = host_out[2]; out->triangle_index = host_out[3];
    return AWSM_OK;
}

int awsm_hip_read_opaque(AwsmHipCtx* c, uint16_t* out) {
    if (!c || !out) return AWSM_ERR_INVALID_ARGUMENT;
    void* src = awsm_hip_output_device_ptr(c);
    if (!src) return fail(c, AWSM_ERR_NOT_READY, "read_opaque before resize");
    if (c->bound_out && c->out_rows_mode) return fail(c, AWSM_ERR_INVALID_ARGUMENT, "read_opaque: the bound output is a row strip (bind_output_rows); read it from its owner");
    HIPCHK(c, hipSetDevice(c->device));
    { int rcs = sync_all(c); if (rcs) return rcs; }
    HIPCHK(c, hipMemcpy(out, src, (size_t)c->width * c->height * 8, hipMemcpyDeviceToHost));
    return AWSM_OK;
}

int awsm_hip_read_opaque_f32(AwsmHipCtx* c, float* out) {
    if (!c || !out) return AWSM_ERR_INVALID_ARGUMENT;
    if (!c->out32[c->slot].ptr) return fail(c, AWSM_ERR_NOT_READY, "read_opaque_f32 needs AWSM_CFG_PARITY_TAP and a resize");
    HIPCHK(c, hipSetDevice(c->device));
    { int rcs = sync_all(c); if (rcs) return rcs; }
    HIPCHK(c, hipMemcpy(out, c->out32[c->slot].ptr, (size_t)c->width * c->height * 16, hipMemcpyDeviceToHost));
    return AWSM_OK;
}

int awsm_hip_bind_opaque_source(AwsmHipCtx* c, const void* device_ptr, size_t bytes) {
    if (!c) return AWSM_ERR_INVALID_ARGUMENT;
    if (device_ptr && bytes == 0) return fail(c, AWSM_ERR_INVALID_ARGUMENT, "bind_opaque_source: zero-sized image");
    c->opaque_src = device_ptr; c->opaque_src_bytes = bytes;
    return AWSM_OK;
}

int awsm_hip_bind_composite(AwsmHipCtx* c, void* device_ptr, size_t bytes) {
    if (!c) return AWSM_ERR_INVALID_ARGUMENT;
    if (device_ptr && bytes == 0) return fail(c, AWSM_ERR_INVALID_ARGUMENT, "bind_composite: zero-sized image");
    c->bound_comp = device_ptr; c->bound_comp_bytes = bytes;
    return AWSM_OK;
}

int awsm_hip_read_composite(AwsmHipCtx* c, uint16_t* out) {
    if (!c || !out) return AWSM_ERR_INVALID_ARGUMENT;
    if (!c->transparent_done) return fail(c, AWSM_ERR_NOT_READY, "read_composite before transparent_pass");
    HIPCHK(c, hipSetDevice(c->device));
    { int rcs = sync_all(c); if (rcs) return rcs; }
    HIPCHK(c, hipMemcpy(out, c->bound_comp ? c->bound_comp : c->comp16.ptr, (size_t)c->width * c->height * 8, hipMemcpyDeviceToHost));
    return AWSM_OK;
}

int awsm_hip_read_composite_f32(AwsmHipCtx* c, float* out) {
    if (!c || !out) return AWSM_ERR_INVALID_ARGUMENT;
    if (!c->transparent_done || !c->comp32.ptr) return fail(c, AWSM_ERR_NOT_READY, "read_composite_f32 needs AWSM_CFG_PARITY_TAP and a transparent_pass");
    HIPCHK(c, hipSetDevice(c->device));
    { int rcs = sync_all(c); if (rcs) return rcs; }
    HIPCHK(c, hipMemcpy(out, c->comp32.ptr, (size_t)c->width * c->height * 16, hipMemcpyDeviceToHost));
    return AWSM_OK;
}

int awsm_hip_read_transformed_forward(AwsmHipCtx* c, float* clip_out, float* nt_out, float* wpos_out, uint32_t max_vertices) {
    if (!c) return AWSM_ERR_INVALID_ARGUMENT;
    if (!c->transparent_done) return fail(c, AWSM_ERR_NOT_READY, "read_transformed_forward before transparent_pass");
    const uint32_t n = std::min(max_vertices, 3u * c->tr_total_tris[0]);
    HIPCHK(c, hipSetDevice(c->device));
    { int rcs = sync_all(c); if (rcs) return rcs; }
    if (n == 0) return AWSM_OK;
    if (clip_out) HIPCHK(c, hipMemcpy(clip_out, TR(c).clip.ptr, (size_t)n * 16, hipMemcpyDeviceToHost));
    if (wpos_out) HIPCHK(c, hipMemcpy(wpos_out, TR(c).wpos.ptr, (size_t)n * 16, hipMemcpyDeviceToHost));
    if (nt_out) {
        std::vector<float> nn((size_t)n * 4), tt((size_t)n * 4);
        HIPCHK(c, hipMemcpy(nn.data(), TR(c).nrm.ptr, (size_t)n * 16, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(tt.data(), TR(c).tan.ptr, (size_t)n * 16, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; i++) { memcpy(nt_out + i * 8, &nn[i * 4], 16); memcpy(nt_out + i * 8 + 4, &tt[i * 4], 16); }
    }
    return AWSM_OK;
}

int awsm_hip_read_transformed(AwsmHipCtx* c, float* clip_out, float* nt_out, uint32_t max_vertices) {
    if (!c) return AWSM_ERR_INVALID_ARGUMENT;
    if (!c->geometry_done) return fail(c, AWSM_ERR_NOT_READY, "read_transformed before geometry_pass");
    const uint32_t n = std::min(max_vertices, c->total_verts);
    HIPCHK(c, hipSetDevice(c->device));
    { int rcs = sync_all(c); if (rcs) return rcs; }
    if (n == 0) return AWSM_OK;
    if (clip_out) HIPCHK(c, hipMemcpy(clip_out, FB(c).clip.ptr, (size_t)n * 16, hipMemcpyDeviceToHost));
    if (nt_out) {
        std::vector<float> nn((size_t)n * 4), tt((size_t)n * 4);
        HIPCHK(c, hipMemcpy(nn.data(), FB(c).nrm.ptr, (size_t)n * 16, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(tt.data(), FB(c).tan.ptr, (size_t)n * 16, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; i++) { memcpy(nt_out + i * 8, &nn[i * 4], 16); memcpy(nt_out + i * 8 + 4, &tt[i * 4], 16); }
    }
    return AWSM_OK;
}

}  // extern "C"
