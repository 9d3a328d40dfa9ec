// dp_env_internal.h -- the environment handle shared by dp_lon.hip (planner, distance map along
// the path) and dp_env.hip (occupancy pipeline: triangles -> Cartesian bit image -> Frenet
// occupancy -> longitudinal distance map).
#pragma once

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
struct StationFrame { double x, y, sin_h, cos_h, d_left, d_right, v_max; };

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
    bool rendered = false;
};
