// Probe: does v_mfma_f32_32x32x16_f16 on gfx950 flush subnormal half INPUTS?  (K1s' error bound assumes it does not.)
// hipcc --offload-arch=gfx950 tools/probe_mfma_f16_denorm.hip -o gpurun_out/probe_denorm && gpurun_out/probe_denorm
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void probe(float *out, float tiny, float big)
{
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = 0; b[i] = 0; }
    a[0] = (_Float16)tiny;      // subnormal in half when tiny < 2^-14
    b[0] = (_Float16)big;
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.0f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = c[0]; out[1] = (float)a[0]; }
}
int main()
{
    float *d, h[2];
    hipMalloc(&d, 8);
    const float tiny = 1.0f / (1 << 20), big = 1024.0f;
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, tiny, big);
    hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    // lane 0 holds k = 0..7 of row 0 / col 0: two lanes (0 and 32) contribute k 0..7 and 8..15 -> expect 2 * tiny * big? no: lane 32 holds
    // k = 8..15 with the same register contents, so D[0][0] = 2 * tiny * big when subnormals are honoured.
    printf("half(tiny)=%g  D00=%g  expected(no flush)=%g  -> %s\n", h[1], h[0], 2 * tiny * big, h[0] > 0 ? "SUBNORMAL INPUTS HONOURED" : "FLUSHED");
    return 0;
}
