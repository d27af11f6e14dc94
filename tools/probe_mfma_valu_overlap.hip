// Probe: do VALU instructions of one wave overlap with MFMAs of ANOTHER wave on the same SIMD?  (gfx950)
// 512-thread workgroups, 1 per CU: waves 0-3 run MFMA chains, waves 4-7 run VALU fma chains (one of each per SIMD).
// hipcc -O3 --offload-arch=gfx950 tools/probe_mfma_valu_overlap.hip -o /tmp/probe_ov && /tmp/probe_ov
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

template <int KIND>   // 0: f32 32x32x2, 1: f16 32x32x16
__global__ __launch_bounds__(512) void probe(float *out, int n_mfma, int n_valu, int mode)
{
    const int wave = threadIdx.x >> 6;
    float r = 0.f;
    if (wave < 4) {
        if (mode & 1) {
            f32x16 acc[4];
            for (int a = 0; a < 4; ++a) for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
            float x = threadIdx.x * 1e-3f, y = 1.0f;
            half8 hx, hy;
            for (int i = 0; i < 8; ++i) { hx[i] = (_Float16)x; hy[i] = (_Float16)1; }
            for (int it = 0; it < n_mfma; it += 4) {
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    if (KIND == 0) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
                    else acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hx, hy, acc[a], 0, 0, 0);
                }
            }
            for (int a = 0; a < 4; ++a) r += acc[a][0];
        }
    } else if (mode & 2) {
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-4f + i;
        for (int it = 0; it < n_valu; it += 8) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], 1.0001f, 0.5f);
        }
        for (int i = 0; i < 8; ++i) r += v[i];
    }
    if (r == 123.456f) out[0] = r;
}

// same-wave variant: every wave issues 4 independent MFMAs and NV independent VALU fmas per iteration
template <int KIND, int NV>
__global__ __launch_bounds__(256) void probe_same_wave(float *out, int n_iter)
{
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
    float x = threadIdx.x * 1e-3f, y = 1.0f;
    half8 hx, hy;
    for (int i = 0; i < 8; ++i) { hx[i] = (_Float16)x; hy[i] = (_Float16)1; }
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 1e-4f + i;
    for (int it = 0; it < n_iter; ++it) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            if (KIND == 0) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
            else acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hx, hy, acc[a], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NV / 4; ++i) v[(a * (NV / 4) + i) & 15] = __builtin_fmaf(v[(a * (NV / 4) + i) & 15], 1.0001f, 0.5f);
        }
    }
    float r = 0.f;
    for (int a = 0; a < 4; ++a) r += acc[a][0];
    for (int i = 0; i < 16; ++i) r += v[i];
    if (r == 123.456f) out[0] = r;
}

template <int KIND, int NV>
static float run_same(int n_iter)
{
    float *d; (void)hipMalloc(&d, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((probe_same_wave<KIND, NV>), dim3(256), dim3(256), 0, 0, d, n_iter);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((probe_same_wave<KIND, NV>), dim3(256), dim3(256), 0, 0, d, n_iter);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipFree(d);
    return ms * 1e3f;
}

template <int KIND>
static float run(int n_mfma, int n_valu, int mode)
{
    float *d; hipMalloc(&d, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<KIND>, dim3(256), dim3(512), 0, 0, d, n_mfma, n_valu, mode);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<KIND>, dim3(256), dim3(512), 0, 0, d, n_mfma, n_valu, mode);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipFree(d);
    return ms * 1e3f;
}

int main()
{
    const int NM = 1 << 14;
    for (int kind = 0; kind < 2; ++kind) {
        const int cyc = kind == 0 ? 64 : 32;                 // pipe cycles per MFMA
        const int NV = NM * cyc / 4;                         // same nominal duration: 4 cycles per VALU op
        float tm = kind ? run<1>(NM, NV, 1) : run<0>(NM, NV, 1);
        float tv = kind ? run<1>(NM, NV, 2) : run<0>(NM, NV, 2);
        float tb = kind ? run<1>(NM, NV, 3) : run<0>(NM, NV, 3);
        printf("%s: mfma-only %.1f us, valu-only %.1f us, both (different waves, same SIMD) %.1f us -> %s\n",
               kind ? "f16 32x32x16" : "f32 32x32x2 ", tm, tv, tb, tb < 0.75f * (tm + tv) ? "OVERLAP" : "SERIALISED");
    }
    printf("same wave, f16 32x32x16 (4 MFMAs = 128 cycles per iteration): +0 VALU %.1f us, +16 VALU %.1f us, +32 VALU %.1f us, +64 VALU %.1f us\n",
           run_same<1, 0>(4096), run_same<1, 16>(4096), run_same<1, 32>(4096), run_same<1, 64>(4096));
    printf("same wave, f32 32x32x2  (4 MFMAs = 256 cycles per iteration): +0 VALU %.1f us, +16 VALU %.1f us, +32 VALU %.1f us, +64 VALU %.1f us\n",
           run_same<0, 0>(4096), run_same<0, 16>(4096), run_same<0, 32>(4096), run_same<0, 64>(4096));
    return 0;
}
