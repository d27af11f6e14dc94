// B4: error-compensated fp16x3 linear layer for the frozen fp32 towers of Oryon.forward (net.py:142-167, models/vlm.py:43-61):
//     C[M,N] = act(A[M,K] * W[N,K]^T + bias[N])      A, C fp32;  W given pre-split into two fp16 matrices W = Whi + Wlo
// Every fp32 operand is split x = hi + lo (hi = half(x), lo = half(x - hi): 22 significant bits) and the product is accumulated in
// fp32 as  Ahi*Whi + Ahi*Wlo + Alo*Whi  - three v_mfma_f32_32x32x16_f16 (16 k per 32 cycles) instead of eight fp32-input MFMAs
// (2 k per 64 cycles); the dropped Alo*Wlo term is ~2^-22 |a||w|, the size of fp32's own accumulation error.  The split of the
// activations happens on the way from HBM to LDS (the round-1 experiment split them with three torch passes per linear and lost
// the gain to that traffic); weights are split once (oryon_split_f16x3) and cached by the caller.  Optional fused QuickGELU
// (x * sigmoid(1.702 x), CLIP's activation) in the epilogue.
//
// Tile 128 x 256 x 32, 4 waves (2 x 2, 64 x 128 each = eight 32x32 accumulators: 12 LDS fragment reads and one activation split
// per 24 MFMAs), operands in padded LDS rows (40 halves: conflict-free
// ds_read_b128 for the 32x32x16 fragment layout), next tile's global loads in flight under the current tile's 48 MFMAs per wave.
// Workgroups are dealt to the 8 XCDs in 8 x 8 super-tiles so that an XCD's concurrent workgroups share their A and W panels in its L2.
// Magnitudes must stay below 65504 (fp16 range); CLIP / Swin activations and weights are O(10).
#include <hip/hip_fp16.h>
#include "common.h"

namespace oryon {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16acc __attribute__((ext_vector_type(16)));

constexpr int GX_BM = 128, GX_BN = 256, GX_BK = 32;     // 4 waves as 2 (M) x 2 (N): 64 x 128 per wave = eight 32x32 accumulators
constexpr int GX_LD = GX_BK + 8;                 // halves per LDS row (80 bytes)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// x = hi + lo with two packed conversions per pair (v_cvt_pk_f16_f32 on gfx950, round-to-nearest-even)
__device__ __forceinline__ void split4(const float4 v, uint2 &hi, uint2 &lo)
{
    const f32x2 a = {v.x, v.y}, b = {v.z, v.w};
    const f16x2 ha = __builtin_convertvector(a, f16x2), hb = __builtin_convertvector(b, f16x2);
    const f16x2 la = __builtin_convertvector(a - __builtin_convertvector(ha, f32x2), f16x2);
    const f16x2 lb = __builtin_convertvector(b - __builtin_convertvector(hb, f32x2), f16x2);
    hi.x = __builtin_bit_cast(unsigned, ha); hi.y = __builtin_bit_cast(unsigned, hb);
    lo.x = __builtin_bit_cast(unsigned, la); lo.y = __builtin_bit_cast(unsigned, lb);
}

template <int ACT>
__global__ __launch_bounds__(256, 2) void linear_f16x3_kernel(const float *__restrict__ A, int M, int K, const __half *__restrict__ Whi,
                                                               const __half *__restrict__ Wlo, const float *__restrict__ bias, int N,
                                                               float *__restrict__ C, int tiles_m, int tiles_n, int sup_n, int sup_rows, int sup_cols)
{
    __shared__ __attribute__((aligned(16))) __half sAh[GX_BM * GX_LD], sAl[GX_BM * GX_LD], sWh[GX_BN * GX_LD], sWl[GX_BN * GX_LD];
    // block -> (tile_m, tile_n): XCD x (= blockIdx % 8) owns every 8th 8x8 super-tile
    const int xcd = blockIdx.x & 7, pos = blockIdx.x >> 3;
    const int sup = (pos >> 6) * 8 + xcd, within = pos & 63;
    // a super-tile is sup_rows x sup_cols tiles (<= 64): sup_cols divides the N tiles evenly so that every XCD gets the same share
    const int wr = within / sup_cols, wc = within % sup_cols;
    const int tm = (sup / sup_n) * sup_rows + wr, tn = (sup % sup_n) * sup_cols + wc;
    if (wr >= sup_rows || tm >= tiles_m || tn >= tiles_n) return;
    const int m0 = tm * GX_BM, n0 = tn * GX_BN;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, kh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;

    // global -> register staging: A 128 x 32 floats = 1024 float4 (4 per thread); Whi / Wlo 128 x 32 halves = 512 uint4 each (2 + 2)
    float4 ra[4];
    uint4 rwh[4], rwl[4];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = t + 256 * i, row = f >> 3, c4 = f & 7;
            const int m = m0 + row;
            ra[i] = m < M ? *reinterpret_cast<const float4 *>(A + (size_t)m * K + k0 + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = t + 256 * i, row = f >> 2, c8 = f & 3;
            rwh[i] = *reinterpret_cast<const uint4 *>(Whi + (size_t)(n0 + row) * K + k0 + c8 * 8);
            rwl[i] = *reinterpret_cast<const uint4 *>(Wlo + (size_t)(n0 + row) * K + k0 + c8 * 8);
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = t + 256 * i, row = f >> 3, c4 = f & 7;
            uint2 hi, lo;
            split4(ra[i], hi, lo);
            *reinterpret_cast<uint2 *>(sAh + row * GX_LD + c4 * 4) = hi;
            *reinterpret_cast<uint2 *>(sAl + row * GX_LD + c4 * 4) = lo;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = t + 256 * i, row = f >> 2, c8 = f & 3;
            *reinterpret_cast<uint4 *>(sWh + row * GX_LD + c8 * 8) = rwh[i];
            *reinterpret_cast<uint4 *>(sWl + row * GX_LD + c8 * 8) = rwl[i];
        }
    };

    f16acc acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    const int nk = K / GX_BK;
    gload(0);
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();                       // everyone is done reading the previous tile
        lstore();
        __syncthreads();
        if (kt + 1 < nk) gload((kt + 1) * GX_BK);
#pragma unroll
        for (int ks = 0; ks < GX_BK / 16; ++ks) {
            h8 ah[2], al[2], wh[4], wl[4];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int off = (wm * 64 + a * 32 + l31) * GX_LD + ks * 16 + kh * 8;
                ah[a] = *reinterpret_cast<const h8 *>(sAh + off);
                al[a] = *reinterpret_cast<const h8 *>(sAl + off);
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int off = (wn * 128 + b * 32 + l31) * GX_LD + ks * 16 + kh * 8;
                wh[b] = *reinterpret_cast<const h8 *>(sWh + off);
                wl[b] = *reinterpret_cast<const h8 *>(sWl + off);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    // smallest terms first
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[a], wh[b], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], wl[b], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], wh[b], acc[a][b], 0, 0, 0);
                }
        }
    }
    // epilogue: lane owns column l31 of each 32x32 block and rows (r & 3) + 8 (r >> 2) + 4 kh
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int n = n0 + wn * 128 + b * 32 + l31;
        const float bv = bias ? bias[n] : 0.0f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                float v = acc[a][b][r] + bv;
                if (ACT == 1) v = v * (1.0f / (1.0f + __expf(-1.702f * v)));
                if (m < M) C[(size_t)m * N + n] = v;
            }
    }
}

__global__ void split_f16x3_kernel(const float *__restrict__ x, int64_t n, __half *__restrict__ hi, __half *__restrict__ lo)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        const __half h = __float2half_rn(v);
        hi[i] = h;
        lo[i] = __float2half_rn(v - __half2float(h));
    }
}

}  // namespace oryon

using namespace oryon;

extern "C" int oryon_split_f16x3(const float *x, int64_t n, void *hi, void *lo, void *stream)
{
    ORYON_CHECK_ARG(x && hi && lo && n >= 0);
    if (n == 0) return ORYON_OK;
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(split_f16x3_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), x, n, static_cast<__half *>(hi),
                       static_cast<__half *>(lo));
    ORYON_CHECK_LAUNCH();
    return ORYON_OK;
}

extern "C" int oryon_linear_f16x3(const float *A, int M, int K, const void *W_hi, const void *W_lo, const float *bias, int N, int act,
                                  float *C, void *stream)
{
    ORYON_CHECK_ARG(A && W_hi && W_lo && C && M >= 0 && K > 0 && N > 0);
    ORYON_CHECK_ARG(K % GX_BK == 0 && N % GX_BN == 0 && (act == 0 || act == 1));
    if (M == 0) return ORYON_OK;
    const int tiles_m = (M + GX_BM - 1) / GX_BM, tiles_n = N / GX_BN;
    const int sup_n = (tiles_n + 7) / 8;                                   // super-tile columns
    const int sup_cols = (tiles_n + sup_n - 1) / sup_n;                    // N tiles per super-tile (<= 8), evenly spread
    const int sup_rows = 64 / sup_cols;
    const int sup_m = (tiles_m + sup_rows - 1) / sup_rows;
    const int supers = ((sup_m * sup_n + 7) / 8) * 8;
    const dim3 grid(supers * 64);
    hipStream_t st = as_stream(stream);
    const __half *wh = static_cast<const __half *>(W_hi), *wl = static_cast<const __half *>(W_lo);
    if (act == 1)
        hipLaunchKernelGGL((linear_f16x3_kernel<1>), grid, dim3(256), 0, st, A, M, K, wh, wl, bias, N, C, tiles_m, tiles_n, sup_n, sup_rows, sup_cols);
    else
        hipLaunchKernelGGL((linear_f16x3_kernel<0>), grid, dim3(256), 0, st, A, M, K, wh, wl, bias, N, C, tiles_m, tiles_n, sup_n, sup_rows, sup_cols);
    ORYON_CHECK_LAUNCH();
    return ORYON_OK;
}
