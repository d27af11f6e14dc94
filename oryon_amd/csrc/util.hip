// Library-level plumbing: version, last error, device check.
#include <stdarg.h>
#include "common.h"

namespace oryon {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace oryon

namespace oryon {
static thread_local hipEvent_t g_ev_start = nullptr, g_ev_stop = nullptr;
static thread_local const char *g_dominant = "";       // per host thread, like the events it describes
void profile_begin(hipStream_t st, const char *kernel_name)
{
    if (!g_ev_start) return;
    (void)hipEventRecord(g_ev_start, st);
    if (kernel_name) g_dominant = kernel_name;          // the kernel these events bracket
}
void profile_end(hipStream_t st)
{
    if (g_ev_stop) (void)hipEventRecord(g_ev_stop, st);
    g_ev_start = g_ev_stop = nullptr;
}
}  // namespace oryon

namespace oryon {
unsigned *x3_range_flag()
{
    static std::mutex mu;
    static unsigned *flags[64] = {nullptr};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64) dev = 0;
    std::lock_guard<std::mutex> lock(mu);
    if (!flags[dev]) {
        if (hipMalloc(reinterpret_cast<void **>(&flags[dev]), 256) != hipSuccess) return nullptr;
        (void)hipMemset(flags[dev], 0, 256);
    }
    return flags[dev];
}
}  // namespace oryon

extern "C" int oryon_x3_range_flag(int *value_out, int reset, void *stream)
{
    ORYON_CHECK_ARG(value_out);
    unsigned *f = oryon::x3_range_flag();
    if (!f) { oryon::set_error("oryon_x3_range_flag: no device memory for the flag"); return ORYON_ERR_HIP; }
    hipStream_t st = oryon::as_stream(stream);
    unsigned v = 0;
    ORYON_CHECK_HIP(hipMemcpyAsync(&v, f, sizeof(v), hipMemcpyDeviceToHost, st));
    if (reset) ORYON_CHECK_HIP(hipMemsetAsync(f, 0, sizeof(unsigned), st));
    ORYON_CHECK_HIP(hipStreamSynchronize(st));
    *value_out = (int)v;
    return ORYON_OK;
}

extern "C" const char *oryon_dominant_kernel(void) { return oryon::g_dominant; }

extern "C" int oryon_profile_events(void *start_event, void *stop_event)
{
    oryon::g_ev_start = static_cast<hipEvent_t>(start_event);
    oryon::g_ev_stop = static_cast<hipEvent_t>(stop_event);
    return ORYON_OK;
}

extern "C" const char *oryon_version(void) { return "oryon_hip 0.1 (gfx950)"; }
extern "C" const char *oryon_last_error(void) { return oryon::g_err; }

extern "C" int oryon_device_check(int device)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) {
        oryon::set_error("no HIP device %d (count %d)", device, n);
        return ORYON_ERR_NO_DEVICE;
    }
    hipDeviceProp_t p;
    ORYON_CHECK_HIP(hipGetDeviceProperties(&p, device));
    if (strncmp(p.gcnArchName, "gfx950", 6) != 0) {
        oryon::set_error("device %d is %s, this library is built for gfx950 only", device, p.gcnArchName);
        return ORYON_ERR_NO_DEVICE;
    }
    return ORYON_OK;
}
