// Probe (round 6): does v_cvt_pk_f16_f32 (packed, what __builtin_convertvector(float2 -> half2) compiles to on gfx950) give the bits of the
// scalar v_cvt_f16_f32 for every input - normals, the fp16 denormal range, ties?  And is float(hi) via v_fma_mix_f32 the scalar conversion's?
// hipcc -O3 --offload-arch=gfx950 tools/probe_cvt_pk_f16.hip -o /tmp/probe_cvt && /tmp/probe_cvt
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float xf32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 xf16x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float *x, int n, unsigned *diff_hi, unsigned *diff_lo, unsigned *diff_mix, float *first_bad)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float a = x[i];
    // scalar
    _Float16 hs = (_Float16)a;
    asm volatile("" : "+v"(hs));
    const float back_s = (float)hs;
    const _Float16 ls = (_Float16)(a - back_s);
    // packed
    const xf32x2 v = {a, a};
    xf16x2 hp = __builtin_convertvector(v, xf16x2);
    const unsigned hb = __builtin_bit_cast(unsigned, hp);
    float m0;
    asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(m0) : "v"(hb), "v"(a));
    const xf32x2 lv = {m0, m0};
    const xf16x2 lp = __builtin_convertvector(lv, xf16x2);
    const unsigned short hs_b = __builtin_bit_cast(unsigned short, hs), hp_b = (unsigned short)(hb & 0xffffu);
    const unsigned short ls_b = __builtin_bit_cast(unsigned short, ls), lp_b = (unsigned short)(__builtin_bit_cast(unsigned, lp) & 0xffffu);
    if (hs_b != hp_b) { if (atomicAdd(diff_hi, 1u) == 0) first_bad[0] = a; }
    if (ls_b != lp_b) { if (atomicAdd(diff_lo, 1u) == 0) first_bad[1] = a; }
    if (m0 != a - back_s) { if (atomicAdd(diff_mix, 1u) == 0) first_bad[2] = a; }
}
int main()
{
    const int n = 1 << 24;
    float *h = (float *)malloc(n * sizeof(float));
    uint64_t s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const int e = (int)(s % 40) - 30;                      // magnitudes 2^-30 .. 2^9: normals, the fp16 denormal range, underflow
        const float m = 1.0f + (float)((s >> 20) & 0xfffff) / 1048576.0f;
        h[i] = ldexpf(m, e) * ((s >> 60) & 1 ? -1.0f : 1.0f);
    }
    float *d; unsigned *c; float *fb;
    hipMalloc(&d, n * sizeof(float)); hipMalloc(&c, 12); hipMalloc(&fb, 12);
    hipMemcpy(d, h, n * sizeof(float), hipMemcpyHostToDevice); hipMemset(c, 0, 12); hipMemset(fb, 0, 12);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, n, c, c + 1, c + 2, fb);
    unsigned hc[3]; float hf[3];
    hipMemcpy(hc, c, 12, hipMemcpyDeviceToHost); hipMemcpy(hf, fb, 12, hipMemcpyDeviceToHost);
    printf("%d values: hi differs %u (first %.9g), lo differs %u (first %.9g), fma_mix residual differs %u (first %.9g)\n", n, hc[0], hf[0], hc[1], hf[1],
           hc[2], hf[2]);
    return 0;
}
