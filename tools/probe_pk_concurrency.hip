// Stand-alone probe for the packed-fp32 finding (DESIGN.md "Concurrency and the packed-fp32 finding").
// A "victim" kernel evaluates a rigid transform + squared distance per point (the inner loop of the pose refinement) twice: once through
// whatever the compiler makes of the plain expression (v_pk_mul/fma/add_f32 on gfx950 by default) and once through single v_fma_f32 /
// v_mul_f32 / v_add_f32 instructions (inline asm).  It runs on stream A, alone and then beside an "aggressor" on stream B that does
// nothing but MFMAs of one type; every victim output is compared bit for bit with its own serial result.
// Measured on the MI355X boxes of this project (ROCm 7.2.0, 2000 victim launches per line; wrong lanes always in the last quarter 48-63 of a
// wave, the single-op result never wrong):
//     beside nothing / v_mfma_f32_32x32x2_f32 / LDS reads + conversions without MFMA ............... 0
//     beside two INDEPENDENT interleaved chains of v_mfma_f32_32x32x16_f16 or v_mfma_i32_32x32x32_i8 ... 0
//     beside three DEPENDENT v_mfma_f32_32x32x16_f16 per step (acc -> acc -> acc), registers only ...... 21
//     beside LDS reads + hi/lo split + three dependent f16 MFMAs (the fp16x3 kernels' shape) ........... 305-324
// i.e. the trigger on the other side is a chain of accumulate-dependent double-rate MFMAs; nothing in the victim is needed beyond
// v_pk_{mul,fma,add}_f32.  A second victim with the other multi-element instruction classes the library still contains
// (v_cvt_pk_f16_f32 splits, v_dot4 int8 chains, fp64 FMA chains) runs in the same iterations and never differs (0 of 2000 on every line).  The library is therefore built without packed fp32 ops (oryon_amd/csrc/Makefile NOPK).
//   hipcc -O3 --offload-arch=gfx950 tools/probe_pk_concurrency.hip -o /tmp/probe_pk && /tmp/probe_pk
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float s_mul(float a, float b) { float r; asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float s_add(float a, float b) { float r; asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float s_sub(float a, float b) { float r; asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float s_fma(float a, float b, float c) { float r; asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// out[0][i] = sum over the points of thread i of |T x - b|^2 (compiler's code), out[1][i] the same through single-lane-pair-free ops
__global__ __launch_bounds__(256) void victim(const float *__restrict__ T_in, const float *__restrict__ src, const float *__restrict__ tgt,
                                              int n, int rounds, float *__restrict__ out)
{
    __shared__ float sT[12];
    const int b = blockIdx.x, t = threadIdx.x;
    if (t < 12) sT[t] = T_in[b * 12 + t];
    __syncthreads();
    float T[12];
    for (int i = 0; i < 12; ++i) T[i] = sT[i];
    const float *sp = src + (size_t)b * n * 3, *tp = tgt + (size_t)b * n * 3;
    float acc_pk = 0.f, acc_s = 0.f;
    for (int r = 0; r < rounds; ++r)
        for (int j = t; j < n; j += 256) {
            const float x = sp[3 * j], y = sp[3 * j + 1], z = sp[3 * j + 2];
            const float bx = tp[3 * j], by = tp[3 * j + 1], bz = tp[3 * j + 2];
            {
                const float dx = (T[0] * x + T[1] * y + T[2] * z + T[3]) - bx;
                const float dy = (T[4] * x + T[5] * y + T[6] * z + T[7]) - by;
                const float dz = (T[8] * x + T[9] * y + T[10] * z + T[11]) - bz;
                acc_pk += dx * dx + dy * dy + dz * dz;
            }
            {
                const float dx = s_sub(s_add(s_fma(T[2], z, s_fma(T[1], y, s_mul(T[0], x))), T[3]), bx);
                const float dy = s_sub(s_add(s_fma(T[6], z, s_fma(T[5], y, s_mul(T[4], x))), T[7]), by);
                const float dz = s_sub(s_add(s_fma(T[10], z, s_fma(T[9], y, s_mul(T[8], x))), T[11]), bz);
                acc_s = s_add(acc_s, s_add(s_fma(dy, dy, s_mul(dx, dx)), s_mul(dz, dz)));
            }
        }
    out[(size_t)b * 256 + t] = acc_pk;
    out[(size_t)gridDim.x * 256 + (size_t)b * 256 + t] = acc_s;
}

// the other multi-element instruction classes the library's code objects still contain (v_cvt_pk_f16_f32, v_dot4c_i32_i8) and its fp64 chains:
// out[0] = sum of the hi/lo fp16 split residuals, out[1] = an int8 dot-product chain, out[2] = an fp64 FMA chain
__global__ __launch_bounds__(256) void victim_other(const float *__restrict__ src, int n, int rounds, float *__restrict__ out)
{
    const int b = blockIdx.x, t = threadIdx.x;
    const float *sp = src + (size_t)b * n * 3;
    float acc_cvt = 0.f;
    int acc_dot = 0;
    double acc_d = 0.0;
    for (int r = 0; r < rounds; ++r)
        for (int j = t; j < n; j += 256) {
            const float x = sp[3 * j], y = sp[3 * j + 1], z = sp[3 * j + 2];
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            typedef float f2 __attribute__((ext_vector_type(2)));
            const f2 v = {x, y};
            const h2 hi = __builtin_convertvector(v, h2);                   // v_cvt_pk_f16_f32
            const f2 lo = v - __builtin_convertvector(hi, f2);
            const h2 lo_h = __builtin_convertvector(lo, h2);
            acc_cvt += (float)hi[0] + (float)hi[1] + 1024.f * ((float)lo_h[0] + (float)lo_h[1]);
            const int pa = __builtin_bit_cast(int, x) ^ (j * 0x01010101), pb = __builtin_bit_cast(int, z) + r;
            acc_dot = __builtin_amdgcn_sdot4(pa, pb, acc_dot, false);       // v_dot4_i32_i8 / v_dot4c_i32_i8
            acc_d = __builtin_fma((double)x, (double)y, acc_d) + (double)z * 0.5;
        }
    const size_t N = (size_t)gridDim.x * 256;
    out[(size_t)b * 256 + t] = acc_cvt;
    out[N + (size_t)b * 256 + t] = (float)acc_dot;
    out[2 * N + (size_t)b * 256 + t] = (float)acc_d;
}

template <int KIND> __global__ __launch_bounds__(256) void aggressor(int iters, float *sink)
{
    const int t = threadIdx.x;
    if (KIND == 0) {          // v_mfma_f32_32x32x16_f16 (gfx950, 8 passes, 128-bit A/B operands)
        f32x16 c0 = {}, c1 = {};
        f16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (t + i)); b[i] = (_Float16)(0.002f * (t - i)); }
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c1, 0, 0, 0);
        }
        sink[blockIdx.x * 256 + t] = c0[0] + c1[3];
    } else if (KIND == 1) {   // v_mfma_i32_32x32x32_i8 (gfx950)
        i32x16 c0 = {}, c1 = {};
        i32x4 a = {t, t * 3, t * 5, t * 7}, b = {t * 11, t * 13, t * 17, t * 19};
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(b, a, c1, 0, 0, 0);
        }
        sink[blockIdx.x * 256 + t] = (float)(c0[0] + c1[3]);
    } else if (KIND == 5) {   // three DEPENDENT f16 MFMAs per step on constant registers (no LDS, no conversions)
        f32x16 c0 = {};
        f16x8 a, b, c;
        for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (t + i)); b[i] = (_Float16)(0.002f * (t - i)); c[i] = (_Float16)(0.003f * i); }
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(c, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, c, c0, 0, 0, 0);
        }
        sink[blockIdx.x * 256 + t] = c0[0] + c0[5];
    } else if (KIND == 6 || KIND == 7) {   // 6: LDS reads + conversions only (no MFMA);  7: LDS reads + plain conversion (no lo part) + 3 MFMAs
        __shared__ float tile[4096];
        for (int i = t; i < 4096; i += 256) tile[i] = 0.001f * i;
        __syncthreads();
        f32x16 c0 = {};
        float keep = 0.f;
        for (int i = 0; i < iters; ++i) {
            float v[8], w[8];
            for (int k = 0; k < 8; ++k) { v[k] = tile[(t * 8 + k + i * 64) & 4095]; w[k] = tile[(t * 8 + k + i * 32 + 7) & 4095]; }
            f16x8 ah, al, bh;
            for (int k = 0; k < 8; ++k) {
                ah[k] = (_Float16)v[k]; al[k] = (_Float16)(v[k] - (float)ah[k]);
                bh[k] = (_Float16)w[k];
            }
            if (KIND == 7) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, ah, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, ah, c0, 0, 0, 0);
            } else {
                for (int k = 0; k < 8; ++k) keep += (float)al[k] * (float)bh[k];
            }
        }
        sink[blockIdx.x * 256 + t] = c0[0] + c0[5] + keep;
    } else if (KIND == 3 || KIND == 4) {   // fp16x3-style loop: LDS reads -> hi/lo split (v_cvt_pk_f16_f32, packed fp32 subtract) -> 3 MFMAs
        __shared__ float tile[4096];
        for (int i = t; i < 4096; i += 256) tile[i] = 0.001f * i;
        __syncthreads();
        f32x16 c0 = {};
        for (int i = 0; i < iters; ++i) {
            float v[8], w[8];
            for (int k = 0; k < 8; ++k) { v[k] = tile[(t * 8 + k + i * 64) & 4095]; w[k] = tile[(t * 8 + k + i * 32 + 7) & 4095]; }
            f16x8 ah, al, bh, bl;
            for (int k = 0; k < 8; ++k) {
                ah[k] = (_Float16)v[k]; al[k] = (_Float16)(v[k] - (float)ah[k]);
                bh[k] = (_Float16)w[k]; bl[k] = (_Float16)(w[k] - (float)bh[k]);
            }
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c0, 0, 0, 0);
            if (KIND == 3) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, c0, 0, 0, 0);
            }
        }
        sink[blockIdx.x * 256 + t] = c0[0] + c0[5];
    } else {                  // v_mfma_f32_32x32x2_f32 (the fp32 MFMA the exact kernels use)
        f32x16 c0 = {}, c1 = {};
        float a = 0.001f * t, b = 0.002f * t;
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, c1, 0, 0, 0);
        }
        sink[blockIdx.x * 256 + t] = c0[0] + c1[3];
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main()
{
    const int B = 64, n = 500, rounds = 8, NV = 2 * B * 256;
    std::vector<float> hT(B * 12), hs((size_t)B * n * 3), ht((size_t)B * n * 3);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) * (1.0f / 16777216.0f); };
    for (auto &v : hs) v = rnd();
    for (size_t i = 0; i < ht.size(); ++i) ht[i] = hs[i] + 0.02f * (rnd() - 0.5f);
    for (int b = 0; b < B; ++b) {
        const float e = 0.01f * rnd();
        const float T[12] = {1.f - e * e, -e, 0.5f * e, 0.001f, e, 1.f - e * e, -e, -0.002f, -0.5f * e, e, 1.f, 0.0015f};
        memcpy(&hT[b * 12], T, sizeof(T));
    }
    float *dT, *ds, *dt, *dout, *dsink;
    CK(hipMalloc(&dT, hT.size() * 4)); CK(hipMalloc(&ds, hs.size() * 4)); CK(hipMalloc(&dt, ht.size() * 4));
    CK(hipMalloc(&dout, (size_t)NV * 4)); CK(hipMalloc(&dsink, (size_t)4096 * 256 * 4));
    CK(hipMemcpy(dT, hT.data(), hT.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(ds, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dt, ht.data(), ht.size() * 4, hipMemcpyHostToDevice));
    hipStream_t sa, sb;
    CK(hipStreamCreate(&sa)); CK(hipStreamCreate(&sb));
    std::vector<float> ref(NV), got(NV), ref2(3 * B * 256), got2(3 * B * 256);
    float *dout2;
    CK(hipMalloc(&dout2, ref2.size() * 4));
    hipLaunchKernelGGL(victim_other, dim3(B), dim3(256), 0, sa, ds, n, rounds, dout2);
    CK(hipMemcpyAsync(ref2.data(), dout2, ref2.size() * 4, hipMemcpyDeviceToHost, sa));
    hipLaunchKernelGGL(victim, dim3(B), dim3(256), 0, sa, dT, ds, dt, n, rounds, dout);
    CK(hipMemcpyAsync(ref.data(), dout, (size_t)NV * 4, hipMemcpyDeviceToHost, sa));
    CK(hipStreamSynchronize(sa));
    const char *names[9] = {"nothing", "v_mfma_f32_32x32x16_f16", "v_mfma_i32_32x32x32_i8", "v_mfma_f32_32x32x2_f32", "LDS + split + 3 f16 MFMAs", "LDS + split + 1 f16 MFMA",
                            "3 dependent f16 MFMAs only", "LDS + split, no MFMA", "LDS + cvt + 3 f16 MFMAs"};
    for (int kind = -1; kind < 8; ++kind) {
        int bad_pk = 0, bad_s = 0, bad_other = 0, launches = 0, lane_hist[4] = {0, 0, 0, 0};
        for (int it = 0; it < 200; ++it) {
            if (kind == 0) hipLaunchKernelGGL(aggressor<0>, dim3(2048), dim3(256), 0, sb, 4000, dsink);
            if (kind == 1) hipLaunchKernelGGL(aggressor<1>, dim3(2048), dim3(256), 0, sb, 4000, dsink);
            if (kind == 2) hipLaunchKernelGGL(aggressor<2>, dim3(2048), dim3(256), 0, sb, 2000, dsink);
            if (kind == 3) hipLaunchKernelGGL(aggressor<3>, dim3(2048), dim3(256), 0, sb, 1500, dsink);
            if (kind == 4) hipLaunchKernelGGL(aggressor<4>, dim3(2048), dim3(256), 0, sb, 1500, dsink);
            if (kind == 5) hipLaunchKernelGGL(aggressor<5>, dim3(2048), dim3(256), 0, sb, 3000, dsink);
            if (kind == 6) hipLaunchKernelGGL(aggressor<6>, dim3(2048), dim3(256), 0, sb, 1500, dsink);
            if (kind == 7) hipLaunchKernelGGL(aggressor<7>, dim3(2048), dim3(256), 0, sb, 1500, dsink);
            for (int k = 0; k < 10; ++k) {
                hipLaunchKernelGGL(victim, dim3(B), dim3(256), 0, sa, dT, ds, dt, n, rounds, dout);
                CK(hipMemcpyAsync(got.data(), dout, (size_t)NV * 4, hipMemcpyDeviceToHost, sa));
                CK(hipStreamSynchronize(sa));
                ++launches;
                bool b1 = false, b2 = false;
                for (int i = 0; i < NV / 2; ++i)
                    if (memcmp(&got[i], &ref[i], 4)) { b1 = true; ++lane_hist[(i & 63) >> 4]; }
                for (int i = NV / 2; i < NV; ++i)
                    if (memcmp(&got[i], &ref[i], 4)) b2 = true;
                bad_pk += b1; bad_s += b2;
                hipLaunchKernelGGL(victim_other, dim3(B), dim3(256), 0, sa, ds, n, rounds, dout2);
                CK(hipMemcpyAsync(got2.data(), dout2, got2.size() * 4, hipMemcpyDeviceToHost, sa));
                CK(hipStreamSynchronize(sa));
                bad_other += memcmp(got2.data(), ref2.data(), got2.size() * 4) != 0;
            }
            CK(hipStreamSynchronize(sb));
        }
        printf("victim beside %-28s: %4d of %d launches differ in the compiler's (packed) result, %d in the single-op result, %d in the cvt_pk / dot4 / fp64 victim; wrong lanes by quarter 0-15/16-31/32-47/48-63: %d/%d/%d/%d\n",
               names[kind + 1], bad_pk, launches, bad_s, bad_other, lane_hist[0], lane_hist[1], lane_hist[2], lane_hist[3]);
    }
    return 0;
}
