// common.hip -- error slot, device discovery, version string of libtplhip.so
#include <cstdarg>
#include <cstdio>

#include "tplhip_common.h"

namespace {
thread_local char g_error[1024] = "";
}

int tplhip_fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof g_error, fmt, ap);
    va_end(ap);
    return code;
}

extern "C" {

const char* tplhip_last_error(void) { return g_error; }

int tplhip_device_count(void)
{
    // tplcpp.cuda_get_device_count, library/src/python/bindings_dp_common.cu:38-45
    int n = -1;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n < 0 ? 0 : n;
}

const char* tplhip_version(void) { return "tplhip 0.1 (gfx950)"; }

} // extern "C"
