// tplhip_common.h -- error plumbing shared by the translation units of libtplhip.so
#pragma once

#include <hip/hip_runtime.h>
#include "../../include/tplhip.h"

// Records a printf-style message in the calling thread's error slot and returns
// `code`.  The reference surfaces CUDA failures as std::runtime_error("<cuda
// msg>: <context>") (library/src/dyn_prog/common.cu:3-10); HIP_TRY builds the
// same "<hip msg>: <context>" text.
int tplhip_fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
// device of a handle created with device < 0: the one tplhip_set_device chose, else the last one
int tplhip_default_device(int n_devices);

#define HIP_TRY(expr, context)                                                        \
    do {                                                                              \
        hipError_t tplhip_err__ = (expr);                                             \
        if (tplhip_err__ != hipSuccess) {                                             \
            return tplhip_fail(TPLHIP_EDEVICE, "%s: %s", hipGetErrorString(tplhip_err__), context); \
        }                                                                             \
    } while (0)
