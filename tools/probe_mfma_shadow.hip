// Probe: how many independent VALU instructions of the SAME wave hide behind a v_mfma_f32_32x32x16_f16 (8 passes) on gfx950,
// at one and at two waves per SIMD?  The loop body is 4 MFMAs (4 independent accumulators), each followed by K VALU instructions
// (8 independent chains); everything is volatile inline asm so that the order in the binary is the order written here.
// hipcc -O3 --offload-arch=gfx950 tools/probe_mfma_shadow.hip -o /tmp/probe_shadow && /tmp/probe_shadow
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

template <int K, int KIND, bool MFMA, bool DEP = false>   // DEP: the four MFMAs of the body chain through ONE accumulator; KIND 0: v_fma_f32, 1: v_exp_f32, 2: v_cvt_pk_f16_f32 + v_fma_mix (pairs)
__global__ __launch_bounds__(512) void probe(float *out, int iters)
{
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
    half8 hx, hy;
    for (int i = 0; i < 8; ++i) { hx[i] = (_Float16)(threadIdx.x * 1e-3f); hy[i] = (_Float16)1; }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-4f + i;
    const float c1 = 1.0001f, c2 = 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            if (MFMA) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[DEP ? 0 : a]) : "v"(hx), "v"(hy));
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int i = (a * K + k) & 7;
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c1), "v"(c2));
                else if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
                else asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c2));
            }
        }
    }
    float r = 0.f;
    for (int a = 0; a < 4; ++a) r += acc[a][0];
    for (int i = 0; i < 8; ++i) r += v[i];
    if (r == 123.456f) out[0] = r;
}

template <int K, int KIND, bool MFMA, bool DEP = false>
static float run(int threads, int iters)
{
    float *out;
    hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<K, KIND, MFMA, DEP><<<256, threads>>>(out, 64);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<K, KIND, MFMA, DEP><<<256, threads>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipFree(out);
    return ms * 1e6f / iters / 4;       // ns per (MFMA + K VALU) group
}

#define ROW(K, KIND) printf("  K=%d  mfma+valu %6.2f %6.2f   valu only %6.2f %6.2f\n", K, run<K, KIND, true>(256, N), run<K, KIND, true>(512, N), \
                            run<K, KIND, false>(256, N), run<K, KIND, false>(512, N))
int main()
{
    const int N = 200000;
    printf("ns per group of (1 MFMA 32x32x16 f16 + K VALU), columns: 1 wave per SIMD, 2 waves per SIMD (each wave runs the whole loop)\n");
    printf("v_fma_f32:\n");
    ROW(0, 0); ROW(1, 0); ROW(2, 0); ROW(4, 0); ROW(6, 0); ROW(8, 0); ROW(12, 0);
    printf("v_exp_f32:\n");
    ROW(1, 1); ROW(2, 1); ROW(4, 1);
    printf("dependent chain (one accumulator), v_fma_f32: K, 1 wave, 2 waves per SIMD\n");
    printf("  K=0 %6.2f %6.2f\n", run<0, 0, true, true>(256, N), run<0, 0, true, true>(512, N));
    printf("  K=4 %6.2f %6.2f\n", run<4, 0, true, true>(256, N), run<4, 0, true, true>(512, N));
    printf("  K=6 %6.2f %6.2f\n", run<6, 0, true, true>(256, N), run<6, 0, true, true>(512, N));
    printf("v_max_f32:\n");
    ROW(4, 2); ROW(8, 2);
    return 0;
}
