/* oracle_raster.h — TEST INFRASTRUCTURE ONLY (see oracle.h).  The triangle setup of the raster contract, shared by the
 * geometry pass (oracle_geometry.c) and the forward transparent pass (oracle_shade.c). */
#ifndef ORACLE_RASTER_H
#define ORACLE_RASTER_H
#include <stdint.h>

typedef struct {
    int kind;
    int64_t a[3], b[3], c[3];  /* kind 0: E_i(P) = a*Px + b*Py + c in 1/256-pixel units, sign-normalised (>= 0 inside), weight of vertex i */
    float ha[3], hb[3], hc[3]; /* kind 1: e_i(X,Y) = fma(a, X, fma(b, Y, c)) in f64, X/Y in pixels */
    float zq[3];               /* kind 0: (z_i / w_i) / |2*area| ; kind 1: z_i / det */
    float iw[3];               /* kind 0: 1 / w_i (perspective correction of the attribute interpolation) ; kind 1: 1 (e_i already is) */
    int minx, maxx, miny, maxy;  /* inclusive, conservative, clamped to the target rect */
    int front;                 /* @builtin(front_facing): counter-clockwise in NDC (FrontFace::Ccw) */
    int valid;
} TriSetup;

extern const int oracle_msaa4_x[4], oracle_msaa4_y[4];
void oracle_tri_setup(const float* v0, const float* v1, const float* v2, int cull_back, uint32_t width, uint32_t height,
                      uint32_t ry0, uint32_t ry1, TriSetup* t);
int oracle_tri_sample(const TriSetup* t, int px, int py, int ox, int oy, float* depth_out);
int oracle_tri_sample_msaa(const TriSetup* t, int px, int py, int k, float* depth_out);      /* sample k of the 4x pattern: oracle_geometry.c */
void oracle_tri_bary(const TriSetup* t, int px, int py, float* b_out);
#endif
