/*
 * dp_env_oracle.c -- CPU restatement of tpl's occupancy pipeline:
 * triangle soup -> Cartesian time-bit image -> Frenet occupancy -> longitudinal
 * distance map.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under tpl_amd/ may include, link or call
 * this file; it is the checker for the HIP path (tests/, bench.py's cpu_baseline
 * leg), never the thing shipped or measured.
 *
 * PARITY UNPINNED: the reference renders with OpenGL (EGL context, fragment
 * pipeline with GL_OR logic op) and post-processes with CUDA texture/surface
 * kernels; neither can be built or run in this image and the reference's tests
 * hold no golden vector for this path.  This file restates the reference's
 * *source semantics*:
 *
 *   RefLine constructor (offsets)     library/include/tplcpp/utils.hpp:184-214
 *   interpRefLine, shortAngleDist     utils.hpp:21-53, 148-173
 *   insertGeometry                    library/src/dyn_prog/env.cu:352-374
 *   OccupancyRenderer::render         library/src/dyn_prog/occupancy_renderer.cu:275-320
 *   occupancyToFrenet                 env.cu:11-63
 *   closeIntersections                env.cu:65-93
 *   updateDistMapLon                  env.cu:95-129
 *   update (driver order)             env.cu:381-437
 *
 * Rules frozen here beyond R1-R3 of dp_lon_oracle.c (types as the source promotes them, no
 * FMA contraction, tex filter with 8-bit weights):
 *  G1  OpenGL leaves the exact fragment coverage to the implementation.  Rule used: window
 *      coordinates w = (v*scale)*(W/2) + W/2 in float, snapped to 1/256 pixel with
 *      round-to-nearest-even; a pixel is covered iff its centre lies inside the triangle;
 *      a centre exactly on an edge is covered iff, with the triangle taken counter-clockwise
 *      (y up), the edge runs downwards or horizontally to the left ("top-left" rule);
 *      zero-area triangles are dropped; both windings are drawn; the colour is the flags
 *      word of the triangle's LAST vertex (flat shading, last-vertex convention).
 *  G2  tex2D point sampling with clamp addressing reads texel floor(coordinate).
 *  G3  (occ & (1 << t)) is 0 for t >= 32.
 */
#define _GNU_SOURCE            /* M_PI under -std=c11 */
#include "../include/tplhip.h"

#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define DPO_API __attribute__((visibility("default")))

#define PPM 4.0f                      /* env.cu:395 */

typedef struct { double x, y, heading, s, k, v_max, d_left, d_right, semantic; } RefPt;   /* utils.hpp:135-146 */

static int clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }

/* utils.hpp:21-31 */
static double normalize_angle(double x)
{
    x = fmod(x, 2 * M_PI);
    x = fmod(x + 2 * M_PI, 2 * M_PI);
    if (x > M_PI) x -= 2 * M_PI;
    return x;
}

/* utils.hpp:33-53 */
static double short_angle_dist(double x, double y)
{
    x = normalize_angle(x);
    y = normalize_angle(y);
    double a0 = y - x;
    double a1 = y - x + 2 * M_PI;
    double a2 = y - x - 2 * M_PI;
    if (fabs(a1) < fabs(a0)) a0 = a1;
    if (fabs(a2) < fabs(a0)) a0 = a2;
    return a0;
}

/* utils.hpp:148-173 */
static RefPt interp_ref_line(const RefPt* rl, int len, float step_size, float s)
{
    float a = s / step_size;
    int i0 = clampi((int)floorf(a), 0, len - 1);          /* max(0, min(len-1, (int)floor(a))) */
    int i1 = clampi((int)ceilf(a), 0, len - 1);
    a -= (float)i0;
    float ia = (float)(1.0 - (double)a);
    RefPt p;
    memset(&p, 0, sizeof p);
    p.x = rl[i0].x * ia + rl[i1].x * a;
    p.y = rl[i0].y * ia + rl[i1].y * a;
    p.heading = rl[i0].heading + short_angle_dist(rl[i0].heading, rl[i1].heading) * a;
    p.s = rl[i0].s * ia + rl[i1].s * a;
    p.k = rl[i0].k * ia + rl[i1].k * a;
    p.v_max = rl[i0].v_max * ia + rl[i1].v_max * a;
    p.d_left = rl[i0].d_left * ia + rl[i1].d_left * a;
    p.d_right = rl[i0].d_right * ia + rl[i1].d_right * a;
    p.semantic = (ia < 0.5) ? rl[i0].semantic : rl[i1].semantic;
    return p;
}

/* rule R2 of dp_lon_oracle.c for one axis; x_plus_half is the coordinate handed to tex3D */
static float tex1_linear(const float* texels, int n, float x_plus_half)
{
    float xb = x_plus_half - 0.5f;
    float fl = floorf(xb);
    float alpha = xb - fl;
    float w = rintf(alpha * 256.0f) / 256.0f;
    float lo = fminf(fmaxf(fl, -1.0f), (float)n);
    int i = (int)lo;
    int i0 = clampi(i, 0, n - 1), i1 = clampi(i + 1, 0, n - 1);
    return fmaf(texels[i1], w, texels[i0] * (1.0f - w));
}

/* RefLine(data, len, step), utils.hpp:184-214: subtract the bounding-box centre.
 * out_ref receives the shifted points, offsets[0..1] the centre. */
DPO_API void dpo_env_center_ref_line(const double* ref_line, int n, double* out_ref, double* offsets)
{
    const RefPt* in = (const RefPt*)ref_line;
    RefPt* out = (RefPt*)out_ref;
    double x_min = INFINITY, x_max = -INFINITY, y_min = INFINITY, y_max = -INFINITY;
    for (int i = 0; i < n; ++i) {
        if (in[i].x < x_min) x_min = in[i].x;
        if (in[i].x > x_max) x_max = in[i].x;
        if (in[i].y < y_min) y_min = in[i].y;
        if (in[i].y > y_max) y_max = in[i].y;
    }
    offsets[0] = x_min + (x_max - x_min) / 2.0;
    offsets[1] = y_min + (y_max - y_min) / 2.0;
    for (int i = 0; i < n; ++i) {
        out[i] = in[i];
        out[i].x -= offsets[0];
        out[i].y -= offsets[1];
    }
}

/* insertGeometry, env.cu:352-374: geom float64[n][3] -> (x, y) float relative to the offsets
 * and the flags word.  vx, vy, flags: [n]. */
DPO_API void dpo_env_insert_geometry(const TplDpEnvParams* ep, const double* geom, int n, int stationary,
                                     const double* offsets, float* vx, float* vy, uint32_t* flags)
{
    for (int i = 0; i < n; ++i) {
        const double* v = geom + 3 * (size_t)i;
        int t_idx = v[2] < ep->dt_start ? 0 : (int)(roundf((float)((v[2] - ep->dt_start) / ep->dt)) + 1.0);
        uint32_t f = (t_idx >= 0 && t_idx < 32) ? ((uint32_t)1 << t_idx) : 0u;   /* (uint32_t)pow(2, t_idx) */
        if (stationary) f |= ((uint32_t)1) << 31;
        vx[i] = (float)(v[0] - offsets[0]);
        vy[i] = (float)(v[1] - offsets[1]);
        flags[i] = f;
    }
}

/* occupancy_renderer.cu:277-279 */
DPO_API int dpo_env_cart_size(const TplDpEnvParams* ep)
{
    float size = ep->s_max + ep->l_max * 2.0f;
    return (int)ceil(size * PPM);
}

static void snap(float vx, float vy, float scale, float half_w, int64_t* sx, int64_t* sy)
{
    float cx = vx * scale, cy = vy * scale;                 /* vertex shader */
    float wx = cx * half_w + half_w, wy = cy * half_w + half_w;   /* viewport transform */
    const float lim = 4194304.0f;
    if (wx < -lim) wx = -lim;
    if (wx > lim) wx = lim;
    if (wy < -lim) wy = -lim;
    if (wy > lim) wy = lim;
    *sx = (int64_t)rintf(wx * 256.0f);
    *sy = (int64_t)rintf(wy * 256.0f);
}

static int64_t edge(int64_t ax, int64_t ay, int64_t bx, int64_t by, int64_t px, int64_t py)
{
    return (bx - ax) * (py - ay) - (by - ay) * (px - ax);
}

static int owns(int64_t ax, int64_t ay, int64_t bx, int64_t by)
{
    int64_t dx = bx - ax, dy = by - ay;
    return dy < 0 || (dy == 0 && dx < 0);
}

/* OccupancyRenderer::render, rule G1.  img: uint32[W][W], cleared here. */
DPO_API void dpo_env_render(const TplDpEnvParams* ep, const float* vx, const float* vy, const uint32_t* flags,
                            int n_vtx, uint32_t* img)
{
    const int W = dpo_env_cart_size(ep);
    const float size = ep->s_max + ep->l_max * 2.0f;
    const float scale = (float)(1.0 / (size * 0.5));
    const float half_w = (float)W * 0.5f;
    memset(img, 0, sizeof(uint32_t) * (size_t)W * W);
    for (int t = 0; t + 2 < n_vtx; t += 3) {
        int64_t x0, y0, x1, y1, x2, y2;
        snap(vx[t], vy[t], scale, half_w, &x0, &y0);
        snap(vx[t + 1], vy[t + 1], scale, half_w, &x1, &y1);
        snap(vx[t + 2], vy[t + 2], scale, half_w, &x2, &y2);
        const uint32_t colour = flags[t + 2];
        int64_t area = edge(x0, y0, x1, y1, x2, y2);
        if (area == 0) continue;
        if (area < 0) { int64_t tx = x1, ty = y1; x1 = x2; y1 = y2; x2 = tx; y2 = ty; }
        const int o01 = owns(x0, y0, x1, y1), o12 = owns(x1, y1, x2, y2), o20 = owns(x2, y2, x0, y0);
        /* only pixels whose centre can lie in the bounding box (one pixel of slack) */
        int64_t bx0 = x0 < x1 ? x0 : x1, bx1 = x0 > x1 ? x0 : x1, by0 = y0 < y1 ? y0 : y1, by1 = y0 > y1 ? y0 : y1;
        if (x2 < bx0) bx0 = x2;
        if (x2 > bx1) bx1 = x2;
        if (y2 < by0) by0 = y2;
        if (y2 > by1) by1 = y2;
        int64_t i_lo = bx0 / 256 - 2, i_hi = bx1 / 256 + 2, j_lo = by0 / 256 - 2, j_hi = by1 / 256 + 2;
        if (i_lo < 0) i_lo = 0;
        if (j_lo < 0) j_lo = 0;
        if (i_hi > W - 1) i_hi = W - 1;
        if (j_hi > W - 1) j_hi = W - 1;
        for (int64_t j = j_lo; j <= j_hi; ++j) {
            const int64_t py = (int64_t)j * 256 + 128;
            for (int64_t i = i_lo; i <= i_hi; ++i) {
                const int64_t px = (int64_t)i * 256 + 128;
                const int64_t w0 = edge(x0, y0, x1, y1, px, py);
                if (!(w0 > 0 || (w0 == 0 && o01))) continue;
                const int64_t w1 = edge(x1, y1, x2, y2, px, py);
                if (!(w1 > 0 || (w1 == 0 && o12))) continue;
                const int64_t w2 = edge(x2, y2, x0, y0, px, py);
                if (!(w2 > 0 || (w2 == 0 && o20))) continue;
                img[(size_t)j * W + i] |= colour;
            }
        }
    }
}

/* rule G2 */
static uint32_t tex2d_point(const uint32_t* img, int W, float x, float y)
{
    int ix = clampi((int)fminf(fmaxf(floorf(x), -1.0f), (float)W), 0, W - 1);
    int iy = clampi((int)fminf(fmaxf(floorf(y), -1.0f), (float)W), 0, W - 1);
    return img[(size_t)iy * W + ix];
}

/* occupancyToFrenet, env.cu:11-63.  occ: float[l][s][t] (the layout getOccMap hands out). */
DPO_API void dpo_env_occupancy_to_frenet(const TplDpEnvParams* ep, const double* ref_centered, int n_ref,
                                         float ref_step, const uint32_t* img, float* occ)
{
    const RefPt* rl = (const RefPt*)ref_centered;
    const int W = dpo_env_cart_size(ep);
    for (int idx_t = 0; idx_t < ep->t_steps; ++idx_t)
        for (int idx_s = 0; idx_s < ep->s_steps; ++idx_s)
            for (int idx_l = 0; idx_l < ep->l_steps; ++idx_l) {
                const float s = ep->s_min + idx_s * ep->_s_step_size;
                const int dil_steps = (int)ceilf(ep->dilation * PPM);
                const float l = ep->l_min + idx_l * ep->_l_step_size;
                const RefPt rp = interp_ref_line(rl, n_ref, ref_step, s);
                const float x = (float)(rp.x - l * sin(rp.heading));
                const float y = (float)(rp.y + l * cos(rp.heading));
                const float x_pixels = (float)(x * PPM + (float)W / 2.0);
                const float y_pixels = (float)(y * PPM + (float)W / 2.0);
                uint32_t o = 0;
                for (int i = -dil_steps; i < dil_steps; ++i)
                    for (int j = -dil_steps; j < dil_steps; ++j) {
                        if (sqrtf((float)(i * i + j * j)) > dil_steps) continue;
                        o |= tex2d_point(img, W, x_pixels + i, y_pixels + j);
                    }
                const uint32_t mask = 1;
                float val = (idx_t < 32 && (o & (mask << idx_t)) != 0) ? 1.0f : 0.0f;     /* rule G3 */
                if ((o & (mask << 31)) != 0) val = 3.0f;
                if (l > rp.d_left || l < -rp.d_right) val = 2.0f;
                else if (idx_l == 0 || idx_l == ep->l_steps - 1) val = 2.0f;
                else if (rp.v_max < 0.1) val = 2.0f;
                occ[((size_t)idx_l * ep->s_steps + idx_s) * ep->t_steps + idx_t] = val;
            }
}

/* closeIntersections, env.cu:65-93 */
DPO_API void dpo_env_close_intersections(const TplDpEnvParams* ep, const double* ref_centered, int n_ref,
                                         float ref_step, float* occ)
{
    const RefPt* rl = (const RefPt*)ref_centered;
    float* sem = (float*)malloc(sizeof(float) * (size_t)(n_ref > 0 ? n_ref : 1));
    for (int i = 0; i < n_ref; ++i) sem[i] = (float)rl[i].semantic;        /* common.cuh:59-65 */
    for (int idx_t = 0; idx_t < ep->t_steps; ++idx_t)
        for (int idx_s = 0; idx_s < ep->s_steps; ++idx_s) {
            const float s = ep->s_min + idx_s * ep->_s_step_size;
            const float semantic = tex1_linear(sem, n_ref, s / ref_step + 0.5f);   /* texLerp(s).w */
            if (semantic < 0.5f) continue;
            int close = 0;
            for (int idx_l = 1; idx_l < ep->l_steps - 1; ++idx_l)
                if (occ[((size_t)idx_l * ep->s_steps + idx_s) * ep->t_steps + idx_t] == 1.0f) close = 1;
            if (close)
                for (int idx_l = 1; idx_l < ep->l_steps - 1; ++idx_l)
                    occ[((size_t)idx_l * ep->s_steps + idx_s) * ep->t_steps + idx_t] = 1.0f;
        }
    free(sem);
}

/* updateDistMapLon, env.cu:95-129.  dist_lon: float[l][s][t][2] */
DPO_API void dpo_env_dist_map_lon(const TplDpEnvParams* ep, const float* occ, float* dist_lon)
{
    for (int idx_t = 0; idx_t < ep->t_steps; ++idx_t)
        for (int idx_l = 0; idx_l < ep->l_steps; ++idx_l) {
            float d = 10000.0f;
            for (int idx_s = ep->s_steps - 1; idx_s >= 0; idx_s -= 1) {
                const size_t c = ((size_t)idx_l * ep->s_steps + idx_s) * ep->t_steps + idx_t;
                if (occ[c] > 0) d = 0.0f; else d += ep->_s_step_size;
                dist_lon[2 * c] = d;
                dist_lon[2 * c + 1] = 0.0f;
            }
            d = 10000.0f;
            for (int idx_s = 0; idx_s < ep->s_steps; idx_s += 1) {
                const size_t c = ((size_t)idx_l * ep->s_steps + idx_s) * ep->t_steps + idx_t;
                if (occ[c] > 0) d = 0.0f; else d += ep->_s_step_size;
                dist_lon[2 * c + 1] = d;
            }
        }
}

/* update(), env.cu:381-437, on vertices already run through dpo_env_insert_geometry.
 * Returns -1 when the reference line is shorter than s_max (the reference throws). */
DPO_API int dpo_env_update(const TplDpEnvParams* ep, const double* ref_centered, int n_ref, float ref_step,
                           const float* vx, const float* vy, const uint32_t* flags, int n_vtx,
                           uint32_t* img, float* occ, float* dist_lon)
{
    if ((float)n_ref * ref_step < ep->s_max) return -1;
    dpo_env_render(ep, vx, vy, flags, n_vtx, img);
    dpo_env_occupancy_to_frenet(ep, ref_centered, n_ref, ref_step, img, occ);
    dpo_env_close_intersections(ep, ref_centered, n_ref, ref_step, occ);
    dpo_env_dist_map_lon(ep, occ, dist_lon);
    return 0;
}
