// Fused element-wise epilogues for the PyTorch-ROCm backbone (the towers' GEMMs stay on hipBLASLt by design; what is fused
// here is the activation torch would run as three bandwidth-bound kernels).
//
// QuickGELU of the CLIP residual blocks (third-party `clip` model.py: x * sigmoid(1.702 * x); models/vlm.py:19 loads it):
// one read and one write of the [tokens, 4*width] activation instead of three of each.  bf16 in / bf16 out, arithmetic in fp32
// with a single final rounding (torch's bf16 chain rounds three times); the fp32 path of the backbone keeps torch's own ops so
// that the reference-pinned fp32 numerics are untouched.
#include <hip/hip_bf16.h>
#include "common.h"

namespace oryon {

__device__ __forceinline__ float bf16_bits_to_float(unsigned short b) { return __uint_as_float((unsigned)b << 16); }
__device__ __forceinline__ unsigned short float_to_bf16_bits(float f)
{
    // round to nearest even (NaN stays NaN: the mantissa msb is forced)
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float quick_gelu(float x) { return x / (1.0f + __expf(-1.702f * x)); }

__global__ __launch_bounds__(256) void quick_gelu_bf16_kernel(const uint4 *__restrict__ x, uint4 *__restrict__ y, int64_t n8,
                                                               const unsigned short *__restrict__ xt, unsigned short *__restrict__ yt,
                                                               int tail)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
        const uint4 v = x[i];
        uint4 o;
        const unsigned in[4] = {v.x, v.y, v.z, v.w};
        unsigned out[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float lo = quick_gelu(bf16_bits_to_float((unsigned short)(in[j] & 0xffffu)));
            const float hi = quick_gelu(bf16_bits_to_float((unsigned short)(in[j] >> 16)));
            out[j] = (unsigned)float_to_bf16_bits(lo) | ((unsigned)float_to_bf16_bits(hi) << 16);
        }
        o.x = out[0]; o.y = out[1]; o.z = out[2]; o.w = out[3];
        y[i] = o;
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < tail) yt[threadIdx.x] = float_to_bf16_bits(quick_gelu(bf16_bits_to_float(xt[threadIdx.x])));
}

}  // namespace oryon

using namespace oryon;

extern "C" int oryon_quick_gelu_bf16(const void *x, void *y, int64_t n, void *stream)
{
    ORYON_CHECK_ARG(x && y && n >= 0);
    ORYON_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0);
    if (n == 0) return ORYON_OK;
    const int64_t n8 = n / 8;
    const int tail = (int)(n - n8 * 8);
    int64_t blocks = (n8 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    const unsigned short *xt = static_cast<const unsigned short *>(x) + n8 * 8;
    unsigned short *yt = static_cast<unsigned short *>(y) + n8 * 8;
    hipLaunchKernelGGL(quick_gelu_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), static_cast<const uint4 *>(x),
                       static_cast<uint4 *>(y), n8, xt, yt, tail);
    ORYON_CHECK_LAUNCH();
    return ORYON_OK;
}
