// tplhip_debug.h -- the ONE environment switch of the library.
//   TPLHIP_DEBUG="token[=int],token[=int],..."     tuning and test overrides, read when a handle is (re)initialised:
//     no_v3          large grids: the general two-pass sweep (k_backward) instead of k_backward3v
//     no_s3          small grids: k_backward instead of k_backward3s
//     no_fastdiv     IEEE division in the sweeps (skips the brute-force proof of the reciprocal forms)
//     div_form=N     plain-fp32 sweeps: division form N (3, 4, 5; dp_sweep.hip) where it is proven, IEEE division elsewhere
//     no_bufload     plain global loads instead of buffer loads in k_backward
//     no_fwd_overlap plan_resident: the forward rollout stays on the main stream
//     no_fwd_spec    forward rollout without speculation on every grid (the form large table sets take anyway)
//     no_front       distance map, stage prelude and terminal stage as three launches instead of k_plan_front
//     kchunk=N       actions per barrier round of k_backward (1 or 3)
//     big=0|1        treat the grid as small / as several residency rounds deep (k_backward3s / k_backward3v), whatever its size
//     tile_v=N, tile_a=N   traversal tile of the sweeps
//     skew=0|1       k_backward3v: rectangular traversal / strips skewed along v + a dt / 2, whatever the table size
//     ilqr_no_lds, ilqr_lds_kb=N   iLQR: where the solver mirrors live (tests force both homes)
//     ilqr_split=0|1|2, ilqr_split_lds_kb=N iLQR: single launch / split-launch sequence / fused launch whatever the batch size; LDS budget of the split and fused kernels
// Nothing else in the library reads the environment (tpl_amd/_capi.py reads TPLHIP_LIB to locate the .so).
#pragma once

#include <cstdlib>
#include <cstring>

namespace tplhip {

// true if `name` is among the tokens; *value receives the integer behind '=' (left alone if there is none)
inline bool debug_opt(const char* name, int* value = nullptr)
{
    const char* s = std::getenv("TPLHIP_DEBUG");
    if (!s) return false;
    const size_t n = std::strlen(name);
    while (*s) {
        const char* e = std::strchr(s, ',');
        const size_t len = e ? (size_t)(e - s) : std::strlen(s);
        if (len >= n && std::strncmp(s, name, n) == 0 && (len == n || s[n] == '=')) {
            if (value && len > n) *value = std::atoi(s + n + 1);
            return true;
        }
        s += len;
        if (*s == ',') ++s;
    }
    return false;
}

} // namespace tplhip
