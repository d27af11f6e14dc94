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
// The fp16x3 range flag is one word PER (device, stream) (round 6; it was one word per device, so a flag raised by another stream's
// forward - or cleared by it - reached the wrong reader): a device holds a table of 64 words, a stream owns the word it was first seen
// with (first come, first served; stream 64 and later share the last word - still correct, merely shared), every fp16x3 launch on a
// stream raises that stream's word, and oryon_x3_range_flag(stream) reads / resets it in stream order.  The table is allocated at the
// first call: oryon_amd.backbone.enable_fp16x3 makes that call (oryon_x3_range_flag with value_out = NULL) so that no launch path
// allocates - a launch that cannot get its word returns ORYON_ERR_HIP instead of handing a null pointer to atomicOr.
constexpr int X3_FLAG_WORDS = 64;
unsigned *x3_range_flag(hipStream_t st)
{
    static std::mutex mu;
    static unsigned *flags[64] = {nullptr};
    static hipStream_t owner[64][X3_FLAG_WORDS];
    static int n_owner[64] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64) dev = 0;
    std::lock_guard<std::mutex> lock(mu);
    if (!flags[dev]) {
        if (hipMalloc(reinterpret_cast<void **>(&flags[dev]), X3_FLAG_WORDS * sizeof(unsigned)) != hipSuccess) {
            flags[dev] = nullptr;
            (void)hipGetLastError();
            set_error("fp16x3 range flag: no device memory for the flag table");
            return nullptr;
        }
        (void)hipMemset(flags[dev], 0, X3_FLAG_WORDS * sizeof(unsigned));
    }
    int idx = -1;
    for (int i = 0; i < n_owner[dev]; ++i)
        if (owner[dev][i] == st) { idx = i; break; }
    if (idx < 0) {
        if (n_owner[dev] < X3_FLAG_WORDS - 1) {
            idx = n_owner[dev]++;
            owner[dev][idx] = st;
        } else {
            idx = X3_FLAG_WORDS - 1;                   // the shared overflow word
        }
    }
    return flags[dev] + idx;
}
}  // namespace oryon

extern "C" int oryon_x3_range_flag(int *value_out, int reset, void *stream)
{
    hipStream_t st = oryon::as_stream(stream);
    unsigned *f = oryon::x3_range_flag(st);
    if (!f) return ORYON_ERR_HIP;
    if (!value_out) {
        // no read-back: allocate the table (first call) and, with reset, queue the clear on the stream - never synchronises
        if (reset) ORYON_CHECK_HIP(hipMemsetAsync(f, 0, sizeof(unsigned), st));
        return ORYON_OK;
    }
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
