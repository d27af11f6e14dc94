// Probe: the sustained rate of v_mfma_f32_32x32x16_f16 on all 256 CUs (two waves per SIMD) with constant operands vs operands of
// random bits (four operand register sets in rotation): how much of the 2.5 PF/s the board's power limit leaves with real data.
// hipcc -O3 --offload-arch=gfx950 tools/probe_mfma_power.hip -o /tmp/probe_power && /tmp/probe_power
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: constant small operands, 1: random bits (finite halves, |x| < 2), 2: random magnitudes like unit-norm rows split hi / lo
__global__ __launch_bounds__(512) void probe(float *out, int iters)
{
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
    half8 A[4], B[4];
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int k = 0; k < 4; ++k) {
        u4 ua, ub;
        for (int i = 0; i < 4; ++i) {
            s = s * 1664525u + 1013904223u; unsigned x = s;
            s = s * 1664525u + 1013904223u; unsigned y = s;
            if (MODE == 0) { x = 0x3c003c00u; y = 0x3c003c00u; }
            else { x = (x & 0xbfffbfffu) & ~0x40004000u; y = (y & 0xbfffbfffu) & ~0x40004000u; }   // exponent < 16: |value| < 2
            ua[i] = x; ub[i] = y;
        }
        A[k] = __builtin_bit_cast(half8, ua); B[k] = __builtin_bit_cast(half8, ub);
    }
    for (int it = 0; it < iters; it += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int a = 0; a < 4; ++a)
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[a]) : "v"(A[a]), "v"(B[(a + u) & 3]));
    }
    float r = 0.f;
    for (int a = 0; a < 4; ++a) for (int i = 0; i < 16; ++i) r += acc[a][i];
    if (r == 123.456f) out[0] = r;
}

template <int MODE>
static void run(const char *name, int wgs)
{
    float *out; (void)hipMalloc(&out, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 400000;
    probe<MODE><<<wgs, 512>>>(out, 1000);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    probe<MODE><<<wgs, 512>>>(out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)wgs * 8 * iters * 4, flops = mfma * 32 * 32 * 16 * 2;
    printf("%-28s %3d workgroups: %.1f ms, %.2f ns per MFMA per SIMD, %.2f PF/s\n", name, wgs, ms, ms * 1e6 / (iters * 4.0 * 2), flops / ms / 1e12);
    (void)hipFree(out);
}
int main()
{
    run<0>("constant operands", 256);
    run<1>("random operand bits", 256);
    run<0>("constant operands", 64);
    run<1>("random operand bits", 64);
    return 0;
}
