// Probe: sustained MFMA rate per input type on this GPU (all 256 CUs, 2 waves per SIMD, 4 independent accumulators per wave,
// random-ish operands so that the power draw is realistic).  hipcc -O3 --offload-arch=gfx950 tools/probe_mfma_rates.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ inline unsigned mix(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int KIND>   // 0: f16 32x32x16, 1: i8 32x32x32, 2: fp8 32x32x16
__global__ __launch_bounds__(256, 2) void rate(float *out, int n_iter)
{
    const unsigned seed = mix(threadIdx.x * 977u + blockIdx.x * 131071u + 7u);
    float r = 0.f;
    if (KIND == 0) {
        half8 a[4], b;
        for (int j = 0; j < 4; ++j) for (int i = 0; i < 8; ++i) a[j][i] = (_Float16)(((int)(mix(seed + 8 * j + i) & 1023) - 512) * (1.0f / 8192.0f));
        for (int i = 0; i < 8; ++i) b[i] = (_Float16)(((int)(mix(seed + 99 + i) & 1023) - 512) * (1.0f / 8192.0f));
        f32x16 acc[4];
        for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
        for (int it = 0; it < n_iter; ++it)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j], b, acc[j], 0, 0, 0);
        for (int j = 0; j < 4; ++j) r += acc[j][0];
    } else if (KIND == 1) {
        i32x4 a[4], b;
        for (int j = 0; j < 4; ++j) for (int i = 0; i < 4; ++i) a[j][i] = (int)mix(seed + 4 * j + i);
        for (int i = 0; i < 4; ++i) b[i] = (int)mix(seed + 77 + i);
        i32x16 acc[4];
        for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0;
        for (int it = 0; it < n_iter; ++it)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[j], b, acc[j], 0, 0, 0);
        for (int j = 0; j < 4; ++j) r += (float)acc[j][0];
    } else {
        long a[4], b;
        for (int j = 0; j < 4; ++j) a[j] = ((long)(mix(seed + j) & 0x3f3f3f3fu) << 32) | (mix(seed + 31 + j) & 0x3f3f3f3fu);
        b = ((long)(mix(seed + 55) & 0x3f3f3f3fu) << 32) | (mix(seed + 56) & 0x3f3f3f3fu);
        f32x16 acc[4];
        for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
        for (int it = 0; it < n_iter; ++it)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a[j], b, acc[j], 0, 0, 0);
        for (int j = 0; j < 4; ++j) r += acc[j][0];
    }
    if (r == 123.456f) out[0] = r;
}

template <int KIND>
static void run(const char *name, double flops_per_mfma)
{
    float *d; (void)hipMalloc(&d, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int n_iter = 1 << 15, blocks = 256 * 2;      // 2 workgroups of 4 waves per CU: 2 waves per SIMD
    hipLaunchKernelGGL(rate<KIND>, dim3(blocks), dim3(256), 0, 0, d, n_iter);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(rate<KIND>, dim3(blocks), dim3(256), 0, 0, d, n_iter);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double total = 3.0 * blocks * 4.0 * n_iter * 4.0 * flops_per_mfma;
    printf("%-14s %8.1f ms  %8.1f T(FL)OP/s sustained\n", name, ms, total / (ms * 1e-3) / 1e12);
    (void)hipFree(d);
}

int main()
{
    run<0>("f16 32x32x16", 2.0 * 32 * 32 * 16);
    run<1>("i8  32x32x32", 2.0 * 32 * 32 * 32);
    run<2>("fp8 32x32x16", 2.0 * 32 * 32 * 16);
    run<0>("f16 32x32x16", 2.0 * 32 * 32 * 16);
    return 0;
}
