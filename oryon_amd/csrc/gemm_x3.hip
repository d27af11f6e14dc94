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
#include <stdlib.h>
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

// Second-generation kernel: 256 x 256 x 32 tiles, 8 waves (4 x 2, 64 x 128 each), two LDS stages (128 KB, one workgroup per CU =
// two waves per SIMD) and ONE barrier per k-tile:
//   * the pre-split weight tiles travel HBM/L2 -> LDS by LDS-DMA (global_load_lds, 4 x 1 KB per wave and tile) - no staging registers;
//   * activation tile kt+1 is split and stored into the idle stage while tile kt is being multiplied; tile kt+2 is in flight;
//   * fragment reads run one k-step ahead of the MFMAs (two fragment sets);
//   * 64-byte LDS rows, 16-byte slots XOR-swizzled with (row >> 2) & 3: conflict-free ds_read_b128 for the 32x32x16 fragment pattern
//     and lane-linear DMA writes (the swizzle is applied to the DMA's source address).
constexpr int G2_BM = 256, G2_BN = 256, G2_BK = 32;
constexpr int G2_A_BYTES = G2_BM * G2_BK * 2, G2_W_BYTES = G2_BN * G2_BK * 2;            // one fp16 matrix tile each (16 KB)
constexpr int G2_STAGE = 2 * G2_A_BYTES + 2 * G2_W_BYTES;                                // Ahi | Alo | Whi | Wlo = 64 KB

template <int ACT>
__global__ __launch_bounds__(512, 2) void linear_f16x3_v2_kernel(const float *__restrict__ A, int M, int K, const __half *__restrict__ Whi,
                                                                  const __half *__restrict__ Wlo, const float *__restrict__ bias, int N,
                                                                  float *__restrict__ C, int tiles_m, int tiles_n, int sup_n, int sup_rows,
                                                                  int sup_cols)
{
    extern __shared__ __attribute__((aligned(1024))) char g2_lds[];
    const int xcd = blockIdx.x & 7, pos = blockIdx.x >> 3;
    const int sup = (pos >> 6) * 8 + xcd, within = pos & 63;
    const int wr = within / sup_cols, wc = within % sup_cols;
    const int tm = (sup / sup_n) * sup_rows + wr, tn = (sup % sup_n) * sup_cols + wc;
    if (wr >= sup_rows || tm >= tiles_m || tn >= tiles_n) return;
    const int m0 = tm * G2_BM, n0 = tn * G2_BN;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, kh = lane >> 5;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wave >> 1, wn = wave & 1;

    // activation staging: 256 x 32 floats = 2048 float4, 4 per thread
    float4 ra[4];
    auto gloadA = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = t + 512 * i, row = f >> 3, c4 = f & 7;
            const int m = m0 + row;
            ra[i] = m < M ? *reinterpret_cast<const float4 *>(A + (size_t)m * K + k0 + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto storeA = [&](int stage) {
        char *base = g2_lds + stage * G2_STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = t + 512 * i, row = f >> 3, c4 = f & 7;
            uint2 hi, lo;
            split4(ra[i], hi, lo);
            const unsigned off = (unsigned)(row * 64 + ((((c4 >> 1) ^ ((row >> 2) & 3)) << 4) | ((c4 & 1) << 3)));
            *reinterpret_cast<uint2 *>(base + off) = hi;
            *reinterpret_cast<uint2 *>(base + G2_A_BYTES + off) = lo;
        }
    };
    // weight tiles by LDS-DMA: 2 matrices x 16 pieces of 1 KB (16 rows x 64 bytes); wave w issues pieces 4w .. 4w+3 of the 32
    unsigned w_src[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int piece = (wave * 4 + j) & 15;
        const int row = piece * 16 + (lane >> 2), ps = lane & 3;
        const int ls = ps ^ ((row >> 2) & 3);
        w_src[j] = (unsigned)(((size_t)(n0 + row) * K + ls * 8) * 2);                 // byte offset inside the matrix; fits 32 bits for N*K < 2^31
    }
    auto dmaW = [&](int k0, int stage) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int pj = wave_u * 4 + j;                                            // 0..31: < 16 -> Whi, >= 16 -> Wlo
            const char *src = reinterpret_cast<const char *>(pj < 16 ? Whi : Wlo) + w_src[j] + (size_t)k0 * 2;
            char *dst = g2_lds + stage * G2_STAGE + 2 * G2_A_BYTES + (pj < 16 ? 0 : G2_W_BYTES) + (pj & 15) * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        }
    };
    // fragment addresses (bytes inside a matrix tile): row (l31 within the 32-row block), slot (ks * 2 + kh) ^ ((row >> 2) & 3)
    auto frag_off = [&](int row, int ks) -> unsigned { return (unsigned)(row * 64 + ((((ks << 1) | kh) ^ ((row >> 2) & 3)) << 4)); };

    f16acc acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    const int nk = K / G2_BK;
    // prologue: tile 0 complete in stage 0, tile 1's weights in flight into stage 1, tile 1's activations in registers
    gloadA(0);
    dmaW(0, 0);
    storeA(0);
    if (nk > 1) { gloadA(G2_BK); dmaW(G2_BK, 1); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const char *base = g2_lds + cur * G2_STAGE;
        if (kt + 1 < nk) storeA(cur ^ 1);                     // tile kt+1's activations (loaded during the previous iteration)
        if (kt + 2 < nk) gloadA((kt + 2) * G2_BK);            // in flight across this iteration's MFMAs and the barrier
        h8 ah[2][2], al[2][2], wh[2][4], wl[2][4];
        auto load_frags = [&](int ks, int set) {
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const unsigned off = frag_off(wm * 64 + a * 32 + l31, ks);
                ah[set][a] = *reinterpret_cast<const h8 *>(base + off);
                al[set][a] = *reinterpret_cast<const h8 *>(base + G2_A_BYTES + off);
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const unsigned off = frag_off(wn * 128 + b * 32 + l31, ks);
                wh[set][b] = *reinterpret_cast<const h8 *>(base + 2 * G2_A_BYTES + off);
                wl[set][b] = *reinterpret_cast<const h8 *>(base + 2 * G2_A_BYTES + G2_W_BYTES + off);
            }
        };
        load_frags(0, 0);
#pragma unroll
        for (int ks = 0; ks < G2_BK / 16; ++ks) {
            const int set = ks & 1;
            if (ks + 1 < G2_BK / 16) load_frags(ks + 1, set ^ 1);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[set][a], wh[set][b], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[set][a], wl[set][b], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[set][a], wh[set][b], acc[a][b], 0, 0, 0);
                }
        }
        // tile kt+1 must be complete (its weights' DMA was issued one iteration ago) and everyone done with stage `cur`
        if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // all but this iteration's 4 activation loads
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 2 < nk) dmaW((kt + 2) * G2_BK, cur);         // stage `cur` is free now
    }
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
    static const int variant = getenv("ORYON_GEMM_X3_VARIANT") ? atoi(getenv("ORYON_GEMM_X3_VARIANT")) : 2;
    if (variant == 2 && (size_t)N * (size_t)K < (1ull << 30)) {
        const int tiles_m = (M + G2_BM - 1) / G2_BM, tiles_n = N / G2_BN;
        const int sup_n = (tiles_n + 7) / 8;
        const int sup_cols = (tiles_n + sup_n - 1) / sup_n;
        const int sup_rows = 64 / sup_cols;
        const int sup_m = (tiles_m + sup_rows - 1) / sup_rows;
        const int supers = ((sup_m * sup_n + 7) / 8) * 8;
        hipStream_t st2 = as_stream(stream);
        const __half *wh2 = static_cast<const __half *>(W_hi), *wl2 = static_cast<const __half *>(W_lo);
        if (act == 1) {
            allow_dynamic_lds(reinterpret_cast<const void *>(linear_f16x3_v2_kernel<1>), 2 * G2_STAGE);
            hipLaunchKernelGGL((linear_f16x3_v2_kernel<1>), dim3(supers * 64), dim3(512), 2 * G2_STAGE, st2, A, M, K, wh2, wl2, bias, N, C, tiles_m,
                               tiles_n, sup_n, sup_rows, sup_cols);
        } else {
            allow_dynamic_lds(reinterpret_cast<const void *>(linear_f16x3_v2_kernel<0>), 2 * G2_STAGE);
            hipLaunchKernelGGL((linear_f16x3_v2_kernel<0>), dim3(supers * 64), dim3(512), 2 * G2_STAGE, st2, A, M, K, wh2, wl2, bias, N, C, tiles_m,
                               tiles_n, sup_n, sup_rows, sup_cols);
        }
        ORYON_CHECK_LAUNCH();
        return ORYON_OK;
    }
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
