// common.hip -- error slot, device discovery, version string of libtplhip.so
#include <cstdarg>
#include <cstdio>

#include <cstring>

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

int tplhip_copy_to_host(void* dst, const void* src, size_t bytes, hipStream_t stream, const char* context)
{
    // one page-locked buffer per calling thread, given back when the thread ends (a short-lived worker thread that calls
    // a getter must not leave up to 64 MB of pinned memory behind for the life of the process)
    struct Bounce {
        void* p = nullptr;
        size_t cap = 0;
        ~Bounce() { if (p) (void)hipHostFree(p); }
    };
    static thread_local Bounce tl;
    void*& bounce = tl.p;
    size_t& cap = tl.cap;
    if (bytes == 0) return TPLHIP_OK;
    if (bytes > ((size_t)64 << 20)) {                       // whole value tables of the largest grids: no 64 MB+ of page-locked memory for a debug read
        HIP_TRY(hipStreamSynchronize(stream), context);
        HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost), context);
        return TPLHIP_OK;
    }
    if (bytes > cap) {
        if (bounce) (void)hipHostFree(bounce);
        bounce = nullptr; cap = 0;
        const size_t want = (bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
        if (hipHostMalloc(&bounce, want, hipHostMallocPortable) != hipSuccess) {
            (void)hipGetLastError();
            bounce = nullptr;                               // no page-locked memory to be had: the plain copy still works
            HIP_TRY(hipStreamSynchronize(stream), context);
            HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost), context);
            return TPLHIP_OK;
        }
        cap = want;
    }
    HIP_TRY(hipMemcpyAsync(bounce, src, bytes, hipMemcpyDeviceToHost, stream), context);
    HIP_TRY(hipStreamSynchronize(stream), context);
    std::memcpy(dst, bounce, bytes);
    return TPLHIP_OK;
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
