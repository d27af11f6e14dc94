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
