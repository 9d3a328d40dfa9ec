// dp_env_internal.h -- the environment handle shared by dp_lon.hip (planner, distance map along
// the path) and dp_env.hip (occupancy pipeline: triangles -> Cartesian bit image -> Frenet
// occupancy -> longitudinal distance map).
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

#include "tplhip_common.h"

namespace tplhip {

// RefLinePoint, library/include/tplcpp/utils.hpp:135-146 (nine doubles, the row layout of the
// float64[n][9] array that set_ref_line takes)
struct RefPoint { double x, y, heading, s, k, v_max, d_left, d_right, semantic; };

// Vertex of the occupancy renderer (occupancy_renderer.cuh): position relative to the reference
// line's bounding-box centre, time flags (bit t = occupied in slice t, bit 31 = stationary)
struct EnvVertex { float x, y; uint32_t flags; };

// What occupancyToFrenet (env.cu:11-63) needs of interpRefLine(s) for one station: computed once
// per station on the host in the reference's mixed float/double arithmetic (the reference
// recomputes it, sin and cos included, for every (t, s, l) cell)
struct StationFrame { double x, y, sin_h, cos_h, d_left, d_right, v_max, k, heading; };

// normalizeAngle / shortAngleDist, utils.hpp:21-53
inline double normalize_angle(double x)
{
    x = std::fmod(x, 2 * M_PI);
    x = std::fmod(x + 2 * M_PI, 2 * M_PI);
    if (x > M_PI) x -= 2 * M_PI;
    return x;
}

inline double short_angle_dist(double x, double y)
{
    x = normalize_angle(x);
    y = normalize_angle(y);
    double a0 = y - x;
    const double a1 = y - x + 2 * M_PI;
    const double a2 = y - x - 2 * M_PI;
    if (std::fabs(a1) < std::fabs(a0)) a0 = a1;
    if (std::fabs(a2) < std::fabs(a0)) a0 = a2;
    return a0;
}

// interpRefLine, utils.hpp:148-173, at one station (host)
inline StationFrame station_frame(const std::vector<RefPoint>& rl, float step_size, float s)
{
    const int len = (int)rl.size();
    float a = s / step_size;
    const int i0 = std::max(0, std::min(len - 1, (int)std::floor(a)));
    const int i1 = std::max(0, std::min(len - 1, (int)std::ceil(a)));
    a -= (float)i0;
    const float ia = (float)(1.0 - (double)a);
    StationFrame f;
    f.x = rl[i0].x * (double)ia + rl[i1].x * (double)a;
    f.y = rl[i0].y * (double)ia + rl[i1].y * (double)a;
    f.heading = rl[i0].heading + short_angle_dist(rl[i0].heading, rl[i1].heading) * (double)a;
    f.sin_h = std::sin(f.heading);
    f.cos_h = std::cos(f.heading);
    f.k = rl[i0].k * (double)ia + rl[i1].k * (double)a;
    f.v_max = rl[i0].v_max * (double)ia + rl[i1].v_max * (double)a;
    f.d_left = rl[i0].d_left * (double)ia + rl[i1].d_left * (double)a;
    f.d_right = rl[i0].d_right * (double)ia + rl[i1].d_right * (double)a;
    return f;
}

}  // namespace tplhip

struct tplhip_dp_env {
    int device = 0;
    TplDpEnvParams params;
    bool have_params = false;
    float* d_occ = nullptr;     // [l][s][t]
    float* d_dist = nullptr;    // [t][s]        distance map along the planner's path
    bool have_occ = false;

    // ---- occupancy pipeline (dp_env.hip) ----
    std::vector<tplhip::RefPoint> ref_line;      // offsets already removed (RefLine ctor, utils.hpp:184-214)
    float ref_step = 0.0f;
    double x_offset = 0.0, y_offset = 0.0;
    std::vector<tplhip::EnvVertex> vertices;     // pending geometry, cleared by update()
    uint32_t* d_cart = nullptr;                  // [size_px][size_px] OR-ed time flags
    int cart_px = 0;                             // size of the last rendered image
    size_t cart_cap = 0;
    tplhip::EnvVertex* d_vtx = nullptr;
    size_t vtx_cap = 0;
    tplhip::StationFrame* d_frames = nullptr;    // [s_steps]
    float* d_semantic = nullptr;                 // [ref steps], float as in the reference's texture copy
    size_t sem_cap = 0;
    float* d_dist_lon = nullptr;                 // [l][s][t][2]
    // the occupancy image is copied to page-locked host memory behind the rasteriser, on a side stream, while the
    // Frenet stages run: the environment plugin fetches it every cycle (dp_env.py:183), and a 2.8 MB hipMemcpy into
    // fresh pageable memory pins and unpins the pages each time -- sporadic 20-30 ms waits in the NEXT device call
    uint32_t* h_cart = nullptr;
    size_t h_cart_cap = 0;
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_raster = nullptr, ev_cart = nullptr;
    bool cart_staged = false;
    float* d_slices = nullptr;                   // scratch of the time-slice getters, [t][l][s][1 or 2]
    size_t slices_cap = 0;
    bool rendered = false;
};
