// Premise check for a register-light K0 (development aid): a streaming kernel with a small register footprint and a given LDS claim, run beside
// the int8 screening kernel on another stream.  Does it get its bandwidth, and what does it cost the screen?  tools/probe_coresidency.py drives it.
#include <hip/hip_runtime.h>
extern "C" __global__ __launch_bounds__(64) void stream_probe(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n_vec, int write_every)
{
    extern __shared__ char lds_claim[];
    const size_t stride = (size_t)gridDim.x * 64;
    float4 acc = {0.f, 0.f, 0.f, 0.f};
    size_t k = 0;
    for (size_t i = (size_t)blockIdx.x * 64 + threadIdx.x; i + 3 * stride < n_vec; i += 4 * stride, ++k) {
        const float4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        acc.x += a.x + b.x + c.x + d.x; acc.y += a.y + b.y + c.y + d.y; acc.z += a.z + b.z + c.z + d.z; acc.w += a.w + b.w + c.w + d.w;
        if ((int)(k % write_every) == 0) dst[i / 4] = acc;                      // 4 reads : 1/write_every writes
    }
    if (acc.x == 12345.f) lds_claim[threadIdx.x] = 1;                          // keep the LDS claim alive
}
extern "C" int launch_stream_probe(const void *src, void *dst, size_t bytes, int groups, int lds_bytes, int write_every, void *stream)
{
    hipFuncSetAttribute(reinterpret_cast<const void *>(stream_probe), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipLaunchKernelGGL(stream_probe, dim3(groups), dim3(64), lds_bytes, static_cast<hipStream_t>(stream), static_cast<const float4 *>(src),
                       static_cast<float4 *>(dst), bytes / 16, write_every);
    return (int)hipGetLastError();
}
