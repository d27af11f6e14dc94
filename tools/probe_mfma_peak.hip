// What the fp16 matrix pipe sustains on this part with nothing else going on, and with the GEMM's LDS read mix next to it:
//   hipcc --offload-arch=gfx950 -O3 tools/probe_mfma_peak.hip -o /tmp/probe_mfma && /tmp/probe_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16acc __attribute__((ext_vector_type(16)));

template <int LDS_READS, int VALU = 0>
__global__ __launch_bounds__(512, 2) void probe(float *out, int iters)
{
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 65536 / 4; i += 512) reinterpret_cast<float *>(lds)[i] = 0.001f * i;
    __syncthreads();
    f16acc acc[8];
    float dummy[4] = {1.0f, 2.0f, 3.0f, 4.0f};
    for (int b = 0; b < 8; ++b)
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.0f;
    h8 a[2], w[4];
    for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const h8 *>(lds + lane * 16 + i * 1024);
    for (int i = 0; i < 4; ++i) w[i] = *reinterpret_cast<const h8 *>(lds + 4096 + lane * 16 + i * 1024);
    const char *p = lds + lane * 16 + (threadIdx.x >> 6) * 4096;
    // the GEMM's fragment pattern: row (lane & 31) of 64-byte rows, 16-byte slot (lane >> 5) ^ ((row >> 2) & 3)
    const char *pg = lds + (lane & 31) * 64 + ((((lane >> 5) ^ (((lane & 31) >> 2) & 3))) << 4) + (threadIdx.x >> 6) * 2048;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 3; ++rep) {
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[b >> 2], w[b & 3], acc[b], 0, 0, 0);
                if (LDS_READS && ((b & 1) == 0 || LDS_READS == 2)) {
                    const int slot = LDS_READS == 2 ? (rep * 8 + b) % 12 : rep * 4 + (b >> 1);   // 12 (or 24) reads per 24 MFMAs
                    h8 v = *reinterpret_cast<const h8 *>((LDS_READS == 3 ? pg : p) + ((slot * 1024 + it * 64) & 32767));
                    if (slot < 2) a[slot] = v; else if (slot < 6) w[slot - 2] = v;
                    else asm volatile("" :: "v"(v));
                }
                if (VALU && (b & 1)) {                             // VALU ops between MFMAs, like the GEMM's in-loop fp32 -> hi/lo split
#pragma unroll
                    for (int v = 0; v < VALU; ++v) asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(dummy[v & 3]) : "v"(dummy[(v + 1) & 3]));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = 0.0f;
    for (int b = 0; b < 8; ++b)
        for (int r = 0; r < 16; ++r) s += acc[b][r];
    out[blockIdx.x * 512 + threadIdx.x] = s + dummy[0] + dummy[1] + dummy[2] + dummy[3];
}

template <int L, int V = 0>
static void run(const char *name, int wgs, int iters)
{
    float *out;
    (void)hipMalloc(&out, sizeof(float) * wgs * 512);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int pass = 0; pass < 2; ++pass) {
        (void)hipEventRecord(e0);
        for (int r = 0; r < 20; ++r) hipLaunchKernelGGL((probe<L, V>), dim3(wgs), dim3(512), 0, 0, out, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
    }
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = 20.0 * wgs * 8 * (double)iters * 24 * 32768.0;
    printf("%-28s %d WGs x 8 waves, %d iters: %.3f ms / launch, %.0f TF/s\n", name, wgs, iters, ms / 20, flops / (ms * 1e-3) / 1e12);
    (void)hipFree(out);
}

int main()
{
    run<0>("MFMA only", 256, 4000);
    run<1>("MFMA + 12 ds_read_b128 / 24", 256, 4000);
    run<0>("MFMA only, 512 WGs", 512, 2000);
    run<2>("MFMA + 24 ds_read_b128 / 24", 256, 4000);
    run<3>("MFMA + 12 reads, GEMM pattern", 256, 4000);
    run<1, 3>("+ 36 VALU / 24 MFMA", 256, 4000);
    run<1, 6>("+ 72 VALU / 24 MFMA", 256, 4000);
    return 0;
}
