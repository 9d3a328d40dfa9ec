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

// Device -> pageable host memory through the calling thread's page-locked bounce buffer (one asynchronous copy, one
// wait on `stream`, one host memcpy) instead of hipMemcpy, which makes the runtime pin and unpin the caller's pages
// around every copy: measured on tpl's planner cycle, a 2.8 MB image fetched that way every cycle into fresh numpy
// memory left sporadic 20-30 ms waits in the NEXT device call (12-20 of 200 cycles).  Copies above 64 MB (whole value
// tables of the largest grids, debug reads) still take the plain path.
int tplhip_copy_to_host(void* dst, const void* src, size_t bytes, hipStream_t stream, const char* context);

#define HIP_TRY(expr, context)                                                        \
    do {                                                                              \
        hipError_t tplhip_err__ = (expr);                                             \
        if (tplhip_err__ != hipSuccess) {                                             \
            return tplhip_fail(TPLHIP_EDEVICE, "%s: %s", hipGetErrorString(tplhip_err__), context); \
        }                                                                             \
    } while (0)
