// Probe: operand layout of v_mfma_scale_f32_32x32x64_f8f6f4 for fp4 (e2m1) and fp6 (e2m3) inputs with per-lane E8M0 scales.
// Hypothesis tested against a CPU evaluation: lane l carries row (A) / column (B) l & 31 and the 32 consecutive k values
// k = 32 * (l >> 5) + t, t = 0..31, packed little-endian (fp4: nibble t of the 128-bit operand; fp6: bits [6t, 6t+6)); the scale
// VGPR's byte `opsel` is that lane's E8M0 exponent for its 32 values.   hipcc --offload-arch=gfx950 -O2 -o probe tools/probe_mfma_mx.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int FMT>
__global__ void k(const v8i *a, const v8i *b, const int *sa, const int *sb, v16f *c)
{
    v16f acc = {0};
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[threadIdx.x], b[threadIdx.x], acc, FMT, FMT, 0, sa[threadIdx.x], 0, sb[threadIdx.x]);
    c[threadIdx.x] = acc;
}

static float fp4(int c) { static const float v[8] = {0, .5f, 1, 1.5f, 2, 3, 4, 6}; return (c & 8 ? -1.f : 1.f) * v[c & 7]; }
static float fp6(int c)
{   // e2m3: sign, 2 exponent bits (bias 1), 3 mantissa bits; exponent 0 = subnormal
    const int s = c >> 5, e = (c >> 3) & 3, m = c & 7;
    const float v = e == 0 ? m / 8.0f : (1.0f + m / 8.0f) * (float)(1 << (e - 1));
    return s ? -v : v;
}

int main()
{
    for (int fmt : {4, 2}) {
        const int bits = fmt == 4 ? 4 : 6;
        static float A[32][64], B[64][32], SA[32][2], SB[32][2];
        v8i ha[64], hb[64];
        int hsa[64], hsb[64];
        srand(7 + fmt);
        for (int l = 0; l < 64; ++l) {
            uint32_t wa[8] = {0}, wb[8] = {0};
            const int ea = 125 + rand() % 5, eb = 124 + rand() % 5;
            hsa[l] = ea | 0x11223300;              // other bytes: garbage that opsel 0 must ignore
            hsb[l] = eb | 0x55000000;
            for (int t = 0; t < 32; ++t) {
                const int ca = rand() & ((1 << bits) - 1), cb = rand() & ((1 << bits) - 1);
                const int bit = t * bits;
                for (int q = 0; q < bits; ++q) {
                    if (ca >> q & 1) wa[(bit + q) >> 5] |= 1u << ((bit + q) & 31);
                    if (cb >> q & 1) wb[(bit + q) >> 5] |= 1u << ((bit + q) & 31);
                }
                A[l & 31][32 * (l >> 5) + t] = (fmt == 4 ? fp4(ca) : fp6(ca)) * ldexpf(1.0f, ea - 127);
                B[32 * (l >> 5) + t][l & 31] = (fmt == 4 ? fp4(cb) : fp6(cb)) * ldexpf(1.0f, eb - 127);
            }
            for (int d = 0; d < 8; ++d) { ha[l][d] = (int)wa[d]; hb[l][d] = (int)wb[d]; }
        }
        v8i *da, *db; int *dsa, *dsb; v16f *dc;
        hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dsa, sizeof(hsa)); hipMalloc(&dsb, sizeof(hsb)); hipMalloc(&dc, 64 * sizeof(v16f));
        hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
        hipMemcpy(dsa, hsa, sizeof(hsa), hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, sizeof(hsb), hipMemcpyHostToDevice);
        if (fmt == 4) hipLaunchKernelGGL(k<4>, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dc);
        else hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dc);
        v16f hc[64];
        hipMemcpy(hc, dc, sizeof(hc), hipMemcpyDeviceToHost);
        double worst = 0;
        int bad = 0;
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 16; ++r) {
                const int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                double ref = 0;
                for (int kk = 0; kk < 64; ++kk) ref += (double)A[row][kk] * B[kk][col];
                const double e = fabs(ref - hc[l][r]);
                worst = fmax(worst, e);
                bad += e > 1e-3 * (1 + fabs(ref));
            }
        printf("fmt %d (%s): max |D - ref| = %.3g, mismatching entries %d / 1024\n", fmt, fmt == 4 ? "fp4 e2m1" : "fp6 e2m3", worst, bad);
    }
    return 0;
}
