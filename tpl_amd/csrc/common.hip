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

static int g_default_device = -1;

int tplhip_default_device(int n_devices)
{
    // tpl/planning/__init__.py:2-7 picks the last device unless cuda_set_device chose one
    return (g_default_device >= 0 && g_default_device < n_devices) ? g_default_device : n_devices - 1;
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

int tplhip_set_device(int idx)
{
    // tplcpp.cuda_set_device, bindings_dp_common.cu:47-49: the device every handle created with device < 0 lives on
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || idx < 0 || idx >= n)
        return tplhip_fail(TPLHIP_EDEVICE, "no usable HIP device (requested %d)", idx);
    if (hipSetDevice(idx) != hipSuccess) return tplhip_fail(TPLHIP_EDEVICE, "hipSetDevice(%d) failed", idx);
    g_default_device = idx;
    return TPLHIP_OK;
}

const char* tplhip_version(void) { return "tplhip 0.1 (gfx950)"; }

} // extern "C"
