// Probe: semantics of v_cvt_scalef32_2xpk16_fp6_f32 (32 x f32 -> 32 packed fp6 e2m3) and v_cvt_scalef32_pk32_f32_fp6 (back):
// element order inside the 192-bit result, the role of the scale operand, rounding and saturation.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v32f __attribute__((ext_vector_type(32)));
typedef unsigned v6u __attribute__((ext_vector_type(6)));
__global__ void k(const v16f *a, v6u *o, v32f *d, float sc)
{
    v6u c = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a[0], a[1], sc);
    o[0] = c;
    d[0] = __builtin_amdgcn_cvt_scalef32_pk32_f32_fp6(c, sc);
}
static float fp6(int c)
{
    const int s = c >> 5, e = (c >> 3) & 3, m = c & 7;
    const float v = e == 0 ? m / 8.0f : (1.0f + m / 8.0f) * (float)(1 << (e - 1));
    return s ? -v : v;
}
int main()
{
    float h[32];
    const float vals[32] = {0.0f, 0.06f, 0.0625f, 0.07f, 0.125f, 0.19f, 0.9f, 0.97f, 1.0f, 1.06f, 1.0625f, 1.07f, 1.19f, 1.9f, 1.97f, 2.1f,
                            -2.125f, 2.2f, 3.9f, 4.1f, -4.25f, 5.3f, 7.3f, 7.5f, 7.8f, 9.0f, -100.0f, 0.03f, -0.03125f, 0.032f, 1e-6f, -0.5f};
    for (float sc : {1.0f, 0.25f, 4.0f}) {
        for (int i = 0; i < 32; ++i) h[i] = vals[i] * (sc == 1.0f ? 1.0f : sc);
        v16f *da; v6u *d_o; v32f *dd;
        hipMalloc(&da, 128); hipMalloc(&d_o, 24); hipMalloc(&dd, 128);
        hipMemcpy(da, h, 128, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, da, d_o, dd, sc);
        uint32_t c[6]; float back[32];
        hipMemcpy(c, d_o, 24, hipMemcpyDeviceToHost); hipMemcpy(back, dd, 128, hipMemcpyDeviceToHost);
        printf("scale operand %g (inputs = table * scale)\n", sc);
        for (int i = 0; i < 32; ++i) {
            int code = 0;
            for (int q = 0; q < 6; ++q) code |= ((c[(6 * i + q) >> 5] >> ((6 * i + q) & 31)) & 1) << q;
            printf("  [%2d] in %9.5f  code 0x%02x = %7.4f   hw decode %9.5f\n", i, h[i], code, fp6(code), back[i]);
        }
    }
    return 0;
}
