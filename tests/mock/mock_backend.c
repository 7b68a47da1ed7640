/*
 * mock_backend.c — TEST INFRASTRUCTURE ONLY.  A recording stand-in for libawsm_hip.so so that the C++ host layer's
 * dirty-upload behaviour can be tested on a machine without a GPU: it implements the awsm_hip_* entry points the host
 * resolves, keeps a byte-exact shadow of every device buffer and logs every call.  It renders nothing.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/awsm_hip.h"

typedef struct MockCall { int op; int which; uint64_t a, b; } MockCall;   /* op: 1 create, 2 write, 3 resize, 4 texture, 5 sampler, 6 env, 7 geometry, 8 opaque, 9 frame_end, 10 lut, 11 shard, 15 transparent */
struct AwsmHipCtx {
    uint8_t* buf[AWSM_BUF_COUNT]; size_t size[AWSM_BUF_COUNT];
    MockCall* log; size_t n_log, cap_log;
    AwsmDraw* draws; uint32_t n_draws; uint32_t has_opaque;
    uint32_t width, height;
};
static void logc(AwsmHipCtx* c, int op, int which, uint64_t a, uint64_t b) {
    if (c->n_log == c->cap_log) { c->cap_log = c->cap_log ? c->cap_log * 2 : 256; c->log = (MockCall*)realloc(c->log, c->cap_log * sizeof(MockCall)); }
    c->log[c->n_log++] = (MockCall){op, which, a, b};
}
uint32_t awsm_hip_abi_version(void) { return AWSM_HIP_ABI_VERSION; }
const char* awsm_hip_last_error(const AwsmHipCtx* c) { (void)c; return "mock"; }
int awsm_hip_create(const AwsmConfig* cfg, AwsmHipCtx** out) { (void)cfg; *out = (AwsmHipCtx*)calloc(1, sizeof(AwsmHipCtx)); return *out ? 0 : -2; }
int awsm_hip_destroy(AwsmHipCtx* c) { for (int i = 0; i < AWSM_BUF_COUNT; i++) free(c->buf[i]); free(c->log); free(c->draws); free(c); return 0; }
int awsm_hip_buffer_create(AwsmHipCtx* c, AwsmBuf w, size_t bytes) {
    free(c->buf[w]); c->buf[w] = (uint8_t*)calloc(bytes ? bytes : 1, 1); c->size[w] = bytes;   /* contents NOT preserved */
    logc(c, 1, w, bytes, 0); return 0;
}
int awsm_hip_buffer_write(AwsmHipCtx* c, AwsmBuf w, size_t off, const void* src, size_t len) {
    if ((off & 3) || (len & 3)) return AWSM_ERR_INVALID_ARGUMENT;
    if (!c->buf[w]) return AWSM_ERR_NOT_READY;
    if (off + len > c->size[w]) return AWSM_ERR_OUT_OF_RANGE;
    memcpy(c->buf[w] + off, src, len); logc(c, 2, w, off, len); return 0;
}
int awsm_hip_resize(AwsmHipCtx* c, uint32_t w, uint32_t h, uint32_t msaa) { if (msaa) return AWSM_ERR_UNSUPPORTED; c->width = w; c->height = h; logc(c, 3, 0, w, h); return 0; }
int awsm_hip_set_shard_rows(AwsmHipCtx* c, uint32_t y0, uint32_t y1) { logc(c, 11, 0, y0, y1); return 0; }
int awsm_hip_texture_array_generate_mips(AwsmHipCtx* c, uint32_t idx, const uint32_t* kinds) { logc(c, 14, (int)idx, kinds ? kinds[0] : 0, 0); return 0; }
int awsm_hip_texture_array_read_level(AwsmHipCtx* c, uint32_t idx, uint32_t level, void* out) { (void)c; (void)idx; (void)level; (void)out; return AWSM_ERR_UNSUPPORTED; }
int awsm_hip_pick(AwsmHipCtx* c, int32_t x, int32_t y, AwsmPick* out) { logc(c, 13, 0, (uint32_t)x, (uint32_t)y); out->valid = 0; out->mesh_key_high = 0; out->mesh_key_low = 0; out->triangle_index = 0xFFFFFFFFu; return 0; }
int awsm_hip_set_shard_bands(AwsmHipCtx* c, uint32_t n, uint32_t r, uint32_t compact) { logc(c, 12, (int)compact, n, r); return 0; }
int awsm_hip_set_stage_timers(AwsmHipCtx* c, int enabled) { logc(c, 16, 0, (uint64_t)(enabled != 0), 0); return 0; }
int awsm_hip_texture_array_upload(AwsmHipCtx* c, uint32_t idx, uint32_t w, uint32_t h, uint32_t layers, uint32_t mips, AwsmTexFormat fmt, const void* t) {
    (void)mips; (void)fmt; (void)t; logc(c, 4, (int)idx, ((uint64_t)w << 32) | h, layers); return 0;
}
int awsm_hip_sampler_set(AwsmHipCtx* c, uint32_t idx, const AwsmSampler* s) { (void)s; logc(c, 5, (int)idx, 0, 0); return 0; }
int awsm_hip_env_upload(AwsmHipCtx* c, const AwsmEnv* e) { (void)e; logc(c, 6, 0, 0, 0); return 0; }
int awsm_hip_env_cube_upload(AwsmHipCtx* c, AwsmCube which, uint32_t size, uint32_t mips, const uint16_t* t) { (void)t; logc(c, 6, (int)which + 1, size, mips); return 0; }
int awsm_hip_brdf_lut_generate(AwsmHipCtx* c, uint32_t w, uint32_t h) { logc(c, 10, 0, w, h); return 0; }
int awsm_hip_geometry_pass(AwsmHipCtx* c, const AwsmDraw* d, uint32_t n) {
    free(c->draws); c->draws = (AwsmDraw*)malloc((n ? n : 1) * sizeof(AwsmDraw)); if (n) memcpy(c->draws, d, n * sizeof(AwsmDraw)); c->n_draws = n;
    logc(c, 7, 0, n, 0); return 0;
}
int awsm_hip_opaque_pass(AwsmHipCtx* c, const AwsmOpaqueParams* p) { c->has_opaque = p->has_opaque; logc(c, 8, 0, p->has_opaque, p->mipmap); return 0; }
int awsm_hip_transparent_pass(AwsmHipCtx* c, const AwsmDraw* d, uint32_t n) { (void)d; logc(c, 15, 0, n, 0); return 0; }
int awsm_hip_hud_geometry_pass(AwsmHipCtx* c, const AwsmDraw* d, uint32_t n) { (void)d; logc(c, 17, 0, n, 0); return 0; }
int awsm_hip_hud_transparent_pass(AwsmHipCtx* c, const AwsmDraw* d, uint32_t n) { (void)d; logc(c, 18, 0, n, 0); return 0; }
int awsm_hip_frame_end(AwsmHipCtx* c, AwsmFrameStats* s) { if (s) { uint32_t n = s->struct_size < sizeof *s ? s->struct_size : (uint32_t)sizeof *s; memset(s, 0, n); s->struct_size = n; } logc(c, 9, 0, 0, 0); return 0; }
/* ---- inspection ---- */
size_t mock_log_count(AwsmHipCtx* c) { return c->n_log; }
void mock_log_get(AwsmHipCtx* c, size_t i, int* op, int* which, uint64_t* a, uint64_t* b) { *op = c->log[i].op; *which = c->log[i].which; *a = c->log[i].a; *b = c->log[i].b; }
void mock_log_clear(AwsmHipCtx* c) { c->n_log = 0; }
const uint8_t* mock_buffer(AwsmHipCtx* c, int which, size_t* size) { *size = c->size[which]; return c->buf[which]; }
uint32_t mock_draws(AwsmHipCtx* c, AwsmDraw* out, uint32_t cap) { uint32_t n = c->n_draws < cap ? c->n_draws : cap; if (n) memcpy(out, c->draws, n * sizeof(AwsmDraw)); return c->n_draws; }
