// Shared helpers for the gfx950 kernels of liboryon_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <set>
#include <utility>
#include "../../include/oryon_hip.h"

namespace oryon {

void set_error(const char *fmt, ...);
// measurement hook (oryon_profile_events): HIP events recorded around the dominant kernel launch of the next matcher call
// kernel_name (a string literal): what oryon_dominant_kernel() reports for the launch the events bracket
void profile_begin(hipStream_t st, const char *kernel_name = nullptr);
void profile_end(hipStream_t st);

#define ORYON_CHECK_ARG(cond)                                                           \
    do {                                                                                \
        if (!(cond)) {                                                                  \
            oryon::set_error("%s: invalid argument: %s", __func__, #cond);              \
            return ORYON_ERR_INVALID_ARG;                                               \
        }                                                                               \
    } while (0)

#define ORYON_CHECK_LAUNCH()                                                            \
    do {                                                                                \
        hipError_t e_ = hipGetLastError();                                              \
        if (e_ != hipSuccess) {                                                         \
            oryon::set_error("%s: launch failed: %s", __func__, hipGetErrorString(e_)); \
            return ORYON_ERR_HIP;                                                       \
        }                                                                               \
    } while (0)

#define ORYON_CHECK_HIP(expr)                                                           \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess) {                                                         \
            oryon::set_error("%s: %s failed: %s", __func__, #expr, hipGetErrorString(e_)); \
            return ORYON_ERR_HIP;                                                       \
        }                                                                               \
    } while (0)

// Development switches.  The SHIPPED library (make all) reads no environment variable: dev_env_* return their defaults.  The dev build
// (make dev -> liboryon_hip_dev.so, -DORYON_DEV; what the A/B scripts under tools/ load) reads them.
#ifdef ORYON_DEV
static inline int dev_env_int(const char *name, int dflt) { const char *s = getenv(name); return s ? atoi(s) : dflt; }
static inline bool dev_env_set(const char *name) { return getenv(name) != nullptr; }
#else
static inline int dev_env_int(const char *, int dflt) { return dflt; }
static inline bool dev_env_set(const char *) { return false; }
#endif

// Range flag of the fp16x3 kernels (round 5).  An fp32 operand of magnitude >= 65520 splits into hi = +-inf (lo = -+inf or NaN) and every
// product it enters comes out inf or NaN, so a kernel that looks at the magnitude bits of its raw ACCUMULATORS (pre-activation) sees an
// overflow of any of its inputs - and one that merely produced a value outside float16's range flags it before the next kernel splits
// it.  The flag is one word per (device, stream), set with atomicOr, reset at the start and read at the end of a forward on that stream
// by oryon_x3_range_flag.
constexpr unsigned X3_RANGE_LIMIT_BITS = 0x476a6000u;            // 60000.0f: below float16's 65504 with room for the rounding of hi
unsigned *x3_range_flag(hipStream_t st);                         // the flag word of (current device, stream) (util.hip); nullptr = no memory, error set
__device__ __forceinline__ unsigned x3_mag(float v) { return __float_as_uint(v) & 0x7fffffffu; }     // NaN / inf compare above any finite value
__device__ __forceinline__ void x3_raise(unsigned *flag, unsigned mag_max)
{
    if (mag_max >= X3_RANGE_LIMIT_BITS) atomicOr(flag, 1u);
}

static inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }
static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// 64-bit mix (splitmix64 finaliser): the counter-based RNG of the batched sampling kernels.
__host__ __device__ static inline uint64_t mix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__host__ __device__ static inline uint32_t rng_u32(uint64_t seed, uint64_t key, uint32_t stream, uint32_t i)
{
    return (uint32_t)(mix64(mix64(seed ^ (key * 0xD1B54A32D192ED03ull)) + ((uint64_t)stream << 32 | i)) >> 32);
}

// Raise a kernel's dynamic-LDS limit once per (kernel, device): thread-safe, and right when one process drives several GPUs
// (the attribute belongs to the device's copy of the function).
inline void allow_dynamic_lds(const void *kernel, int bytes)
{
    static std::mutex mu;
    static std::set<std::pair<const void *, int>> done;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    if (done.insert({kernel, dev}).second) (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

// Correctly rounded fp32 square root.  NOT __fsqrt_rn: this toolchain's HIP headers define it as __ocml_native_sqrt_f32 (the
// approximate one) unless OCML_BASIC_ROUNDED_OPERATIONS is set, and whether the compiler then emits the IEEE fix-up after
// v_sqrt_f32 depends on the surrounding code.  sqrtf is __ocml_sqrt_f32: correctly rounded with the (default, and explicit in the
// Makefile) -fhip-fp32-correctly-rounded-divide-sqrt.
__device__ __forceinline__ float sqrt_rn(float x) { return sqrtf(x); }

}  // namespace oryon
