// B4: error-compensated fp16x3 linear layer for the frozen fp32 towers of Oryon.forward (net.py:142-167, models/vlm.py:43-61):
//     C[M,N] = act(A[M,K] * W[N,K]^T + bias[N])      A, C fp32;  W given pre-split into two fp16 matrices W = Whi + Wlo
// Every fp32 operand is split x = hi + lo (hi = half(x), lo = half(x - hi): 22 significant bits) and the product is accumulated in
// fp32 as  Ahi*Whi + Ahi*Wlo + Alo*Whi  - three v_mfma_f32_32x32x16_f16 (16 k per 32 cycles) instead of eight fp32-input MFMAs
// (2 k per 64 cycles); the dropped Alo*Wlo term is ~2^-22 |a||w|, the size of fp32's own accumulation error.  The split of the
// activations happens on the way from HBM to LDS (the round-1 experiment split them with three torch passes per linear and lost
// the gain to that traffic); weights are split once (oryon_split_f16x3) and cached by the caller.  Optional fused QuickGELU
// (x * sigmoid(1.702 x), CLIP's activation) in the epilogue.
//
// Two kernels: the persistent 256 x 256 x 32 stream kernel below (the one that runs; K >= 64), and a small-tile kernel for K = 32 and
// as a cross-check (ORYON_GEMM_X3_VARIANT=1): tile 128 x 256 x 32, 4 waves (2 x 2, 64 x 128 each = eight 32x32 accumulators),
// operands in padded LDS rows (40 halves: conflict-free ds_read_b128 for the 32x32x16 fragment layout), next tile's global loads in
// flight under the current tile's 48 MFMAs per wave.  Both accumulate every output in the same order (k ascending, per k-step
// lo*hi, hi*lo, hi*hi), so their results are bit-identical.
// Workgroups are dealt to the 8 XCDs in 8 x 8 super-tiles so that an XCD's concurrent workgroups share their A and W panels in its L2.
// Range: magnitudes must stay below 65504 (fp16 range) or the split overflows to inf without a diagnostic; the random-init towers of
// the tests have O(10) activations, released checkpoints may not - oryon_amd.backbone.enable_fp16x3(True, guard=True) validates every
// call on the host first.  Precision: an operand below 2^-3 has its low half in float16's subnormal range (absolute split error
// <= 2^-25 instead of the relative 2^-22).
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

// x = hi + lo with two packed conversions per pair (v_cvt_pk_f16_f32 on gfx950, round-to-nearest-even) and - round 6 - the residuals
// x - float(hi) as one v_fma_mix_f32 each (hi's half read as the f16 source of an fp32 fma: the bits of the subtraction it replaces,
// tools/probe_cvt_pk_f16.hip): four instructions per pair instead of six.  VALU and MFMA do not overlap on this part.
__device__ __forceinline__ void split2(float x, float y, unsigned &hi, unsigned &lo)
{
    const f32x2 a = {x, y};
    const unsigned hb = __builtin_bit_cast(unsigned, __builtin_convertvector(a, f16x2));
    float l0, l1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(hb), "v"(x));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(hb), "v"(y));
    const f32x2 lv = {l0, l1};
    hi = hb;
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(lv, f16x2));
}
__device__ __forceinline__ void split4(const float4 v, uint2 &hi, uint2 &lo)
{
    split2(v.x, v.y, hi.x, lo.x);
    split2(v.z, v.w, hi.y, lo.y);
}

template <int ACT>
__global__ __launch_bounds__(256, 2) void linear_f16x3_kernel(const float *__restrict__ A, int M, int K, const __half *__restrict__ Whi,
                                                               const __half *__restrict__ Wlo, const float *__restrict__ bias, int N,
                                                               float *__restrict__ C, int tiles_m, int tiles_n, int sup_n, int sup_rows, int sup_cols,
                                                               unsigned *__restrict__ range_flag)
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
    unsigned mag = 0u;                                   // range flag (common.h): largest magnitude among the raw accumulators of live rows
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
                if (m < M) mag = max(mag, x3_mag(v));             // pre-activation: what a later split would see is bounded by it
                if (ACT == 1) v = v * (1.0f / (1.0f + __expf(-1.702f * v)));
                if (ACT == 2) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
                if (m < M) C[(size_t)m * N + n] = v;
            }
    }
    x3_raise(range_flag, mag);
}

constexpr int G2_BM = 256, G2_BN = 256, G2_BK = 32;
constexpr int G2_A_BYTES = G2_BM * G2_BK * 2, G2_W_BYTES = G2_BN * G2_BK * 2;            // one fp16 matrix tile each (16 KB)
constexpr int G2_STAGE = 2 * G2_A_BYTES + 2 * G2_W_BYTES;                                // Ahi | Alo | Whi | Wlo = 64 KB

// Main kernel: 256 x 256 x 32 tiles, 8 waves (4 x 2, 64 x 128 each = eight 32x32 accumulators), two 64 KB LDS stages (one workgroup
// per CU, two waves per SIMD), persistent workgroups.
//   * the pre-split weight tiles travel HBM/L2 -> LDS by LDS-DMA (global_load_lds, 4 x 1 KB per wave and k-tile) - no staging registers;
//     activation k-tile g+1 is split (fp32 -> fp16 hi + lo, v_cvt_pk_f16_f32) and stored into the idle stage while k-tile g is being
//     multiplied; k-tile g+2 is in flight;
//   * 64-byte LDS rows, 16-byte slots XOR-swizzled with (row >> 2) & 3: conflict-free ds_read_b128 for the 32x32x16 fragment pattern
//     and lane-linear DMA writes (the swizzle is applied to the DMA's source address);
//   * the k-loop is phased so that no fragment read is ever waited for right after it was issued (left to itself the compiler sinks
//     every ds_read to just before its first use: seven read -> s_waitcnt lgkmcnt(0) -> MFMA sequences per k-step):
//       phase 1  MFMAs of k-step 0 (fragments already in registers)  ||  reads of k-step 1, split + store of activation k-tile g+1,
//                global loads of activation k-tile g+2
//       sync     lgkmcnt(0), vmcnt (weights of g+1 landed), ONE barrier per k-tile - in the middle of the tile's MFMAs
//       phase 2  MFMAs of k-step 1  ||  LDS-DMA of weight k-tile g+2 into the stage just released, reads of k-tile g+1's k-step 0
//     MFMAs are issued term-major (all lo*hi, then hi*lo, then hi*hi: 8 independent MFMAs between dependent ones; per accumulator
//     the order is smallest term first); the hi fragments are double-buffered, the lo fragments are re-read in place right after their
//     last use (224 -> 255 VGPRs with the accumulators; a full second fragment set spills); sched_barrier pins the order;
//   * a workgroup walks its output tiles as ONE stream of k-tiles: the last two iterations of a tile already request the next
//     tile's first two k-tiles, so there is no per-tile prologue, and the epilogue's stores drain under the next tile's loads;
//   * full tiles store without predicates: with a per-row branch the compiler waits vmcnt(0) before every store (128 round trips).
// Measured (M = 73856, CLIP ViT-L shapes): 1.0-1.13 PFLOP/s on the fp16 pipe (small-tile kernel: 0.55-0.6), socket power at the 1.4 kW cap with
// sclk throttled to ~1.87 GHz: the kernel is power-limited, not stall-limited (a bare MFMA loop sustains ~1.7 PFLOP/s at the same cap,
// tools/probe_mfma_peak.hip; LDS reads at twice this kernel's rate cost that loop nothing).
struct G3Frags { h8 ah[2], wh[4]; };           // the hi halves are double-buffered; the lo halves (al, wl) are re-read in place

// WEX ("weights exact"): every weight is exactly representable in fp16 (W_lo would be all zeros - what `clip.load` leaves in the
// reference's CLIPEncoder, models/vlm.py:19-22: an fp16 checkpoint widened to fp32): the a_hi * w_lo term, its fragments and its DMA
// are left out - 16 instead of 24 MFMAs per k-step, bit-identical results (the omitted products are exact zeros).
// ACC ("accumulate"): C += A W^T + bias - the result is ADDED to what C holds with fire-and-forget global_atomic_add_f32 (every element is
// touched by exactly one lane once: one fp32 addition, same rounding as a separate residual add, deterministic).  The towers' residual
// stream is updated in place by the out-projection / second MLP linear, and the LayerNorm pass that follows reads ONE tensor instead of two.
template <int ACT, bool WEX, bool ACC = false>
__global__ __launch_bounds__(512, 2) void linear_f16x3_stream_kernel(const float *__restrict__ A, int M, int K, const __half *__restrict__ Whi,
                                                                  const __half *__restrict__ Wlo, const float *__restrict__ bias, int N,
                                                                  float *__restrict__ C, int tiles_m, int tiles_n, int sup_n, int sup_rows,
                                                                  int sup_cols, int n_slots,
                                                                       unsigned *__restrict__ range_flag)
{
    extern __shared__ __attribute__((aligned(1024))) char g2_lds[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, kh = lane >> 5;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wave >> 1, wn = wave & 1;
    // Persistent workgroup: slots blockIdx.x, + gridDim.x, ... of the XCD-aware slot -> tile map (slot % 8 = XCD, gridDim.x % 8 == 0,
    // so a workgroup's tiles all share its XCD's L2 with the neighbouring tiles of their super-tile, as under one-tile-per-block dispatch)
    auto decode = [&](int slot, int &m0, int &n0) -> bool {
        const int xcd = slot & 7, pos = slot >> 3;
        const int sup = (pos >> 6) * 8 + xcd, within = pos & 63;
        const int wr = within / sup_cols, wc = within % sup_cols;
        const int tm = (sup / sup_n) * sup_rows + wr, tn = (sup % sup_n) * sup_cols + wc;
        m0 = tm * G2_BM; n0 = tn * G2_BN;
        return wr < sup_rows && tm < tiles_m && tn < tiles_n;
    };
    auto next_valid = [&](int slot, int &m0, int &n0) -> int {          // first valid slot >= `slot` of this workgroup's sequence, or -1
        for (; slot < n_slots; slot += (int)gridDim.x)
            if (decode(slot, m0, n0)) return slot;
        return -1;
    };
    int m0, n0, nm0 = 0, nn0 = 0;
    int slot = next_valid((int)blockIdx.x, m0, n0);
    if (slot < 0) return;
    int nslot = next_valid(slot + (int)gridDim.x, nm0, nn0);

    // activation staging: 256 x 32 floats = 2048 float4, 4 per thread (rows t/8 + 64 i); rows past M read row M-1 (never stored).
    // All per-thread addressing is kept as ONE register + compile-time / wave-uniform offsets: the accumulators and the two fragment
    // sets leave ~30 registers for everything else.  The load side (a_tile, a_src, w_mat) runs two k-tiles ahead of the MFMAs and moves
    // on to the workgroup's next output tile two k-tiles before the current one ends.
    float4 ra[4];
    const int a_row0 = t >> 3, a_c4 = t & 7;
    const float *a_tile;
    unsigned a_src[4];
    const int w_piece0 = WEX ? wave_u * 2 : (wave_u * 4) & 15;
    const char *w_mat;
    auto set_load_tile = [&](int lm0, int ln0) {
        a_tile = A + (size_t)lm0 * K;
        const int a_rows = M - lm0;                                   // >= 1
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = a_row0 + 64 * i;
            a_src[i] = (unsigned)((row < a_rows ? row : a_rows - 1) * K + a_c4 * 4);      // float index inside the tile's rows: < 256 K
        }
        // N % 128 == 0: the last column tile may be half wide; the waves whose 64 weight rows fall past N re-read the tile's first rows
        // (their products land in accumulators that are never stored)
        const int w_row = ln0 + w_piece0 * 16 < N ? ln0 + w_piece0 * 16 : ln0;
        w_mat = reinterpret_cast<const char *>(WEX || wave_u < 4 ? Whi : Wlo) + ((size_t)w_row * K) * 2;
    };
    set_load_tile(m0, n0);
    const unsigned a_dst = (unsigned)(a_row0 * 64 + ((((a_c4 >> 1) ^ ((a_row0 >> 2) & 3)) << 4) | ((a_c4 & 1) << 3)));   // + 4096 i
    auto gloadA = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const float4 *>(a_tile + k0 + a_src[i]);
    };
    auto storeA1 = [&](int stage, int i) {
        char *base = g2_lds + stage * G2_STAGE + a_dst;
        uint2 hi, lo;
        split4(ra[i], hi, lo);
        *reinterpret_cast<uint2 *>(base + i * 4096) = hi;
        *reinterpret_cast<uint2 *>(base + G2_A_BYTES + i * 4096) = lo;
    };
    auto storeA = [&](int stage) {
#pragma unroll
        for (int i = 0; i < 4; ++i) storeA1(stage, i);
    };
    // weight tiles by LDS-DMA: wave w issues pieces 4 w .. 4 w + 3 of the 32 (16 Whi + 16 Wlo pieces of 16 rows x 64 bytes);
    // WEX: pieces 2 w, 2 w + 1 of the 16 Whi pieces
    const unsigned w_src = (unsigned)(((size_t)(lane >> 2) * K + (((lane & 3) ^ ((lane >> 4) & 3)) * 8)) * 2);       // row = lane / 4 (+ 16 j)
    auto dmaW = [&](int k0, int stage) {
#pragma unroll
        for (int j = 0; j < (WEX ? 2 : 4); ++j) {
            const char *src = w_mat + (size_t)j * 16 * K * 2 + (size_t)k0 * 2 + w_src;
            char *dst = g2_lds + stage * G2_STAGE + 2 * G2_A_BYTES + (WEX || wave_u < 4 ? 0 : G2_W_BYTES) + (w_piece0 + j) * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        }
    };
    // fragment offsets: row l31 of a 32-row block, slot (2 ks + kh) ^ ((row >> 2) & 3); blocks are 2048 bytes apart (same swizzle)
    const unsigned f_swz = (unsigned)((kh ^ ((l31 >> 2) & 3)) << 4);
    const unsigned fa0 = (unsigned)((wm * 64 + l31) * 64) + f_swz, fw0 = (unsigned)((wn * 128 + l31) * 64) + f_swz;
    f16acc acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    h8 al[2], wl[4];
#define G3_MFMA(X, Y, a, b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(X, Y, acc[a][b], 0, 0, 0)
#define G3_RD(p) (*reinterpret_cast<const h8 *>(p))
#define G3_SB __builtin_amdgcn_sched_barrier(0)            // the compiler otherwise sinks every read to just before its first use
    // One k-step: 24 MFMAs on (cur.ah, al) x (cur.wh, wl), term-major; between them the NEXT k-step's fragments are requested from
    // `nbase` (k-step `nks`): the hi halves into `nxt`, al right after the last MFMA that reads it (end of term 1), wl likewise (end
    // of term 2).  `mid` / `tail` are extra work to issue under terms 2 / 3.
    auto k_step = [&](const G3Frags &cur, G3Frags &nxt, const char *nbase, int nks, auto &&head, auto &&mid, auto &&tail) {
        const unsigned x = nks ? 32u : 0u;
        const char *pa = nbase + (fa0 ^ x), *pw = nbase + 2 * G2_A_BYTES + (fw0 ^ x);
        head();
        G3_SB;
        G3_MFMA(al[0], cur.wh[0], 0, 0); nxt.ah[0] = G3_RD(pa);
        G3_SB;
        G3_MFMA(al[0], cur.wh[1], 0, 1); nxt.ah[1] = G3_RD(pa + 2048);
        G3_SB;
        G3_MFMA(al[0], cur.wh[2], 0, 2); nxt.wh[0] = G3_RD(pw);
        G3_SB;
        G3_MFMA(al[0], cur.wh[3], 0, 3); nxt.wh[1] = G3_RD(pw + 2048);
        G3_SB;
        G3_MFMA(al[1], cur.wh[0], 1, 0); nxt.wh[2] = G3_RD(pw + 4096);
        G3_SB;
        G3_MFMA(al[1], cur.wh[1], 1, 1); nxt.wh[3] = G3_RD(pw + 6144);
        G3_SB;
        G3_MFMA(al[1], cur.wh[2], 1, 2);
        G3_SB;
        G3_MFMA(al[1], cur.wh[3], 1, 3);
        G3_SB;
        al[0] = G3_RD(pa + G2_A_BYTES);
        G3_SB;
        al[1] = G3_RD(pa + G2_A_BYTES + 2048);
        G3_SB;
        if constexpr (WEX) {
            G3_MFMA(cur.ah[0], cur.wh[0], 0, 0);
            G3_SB;
            mid(0);
            G3_SB;
            G3_MFMA(cur.ah[0], cur.wh[1], 0, 1);
            G3_SB;
            G3_MFMA(cur.ah[0], cur.wh[2], 0, 2);
            G3_SB;
            mid(1);
            G3_SB;
            G3_MFMA(cur.ah[0], cur.wh[3], 0, 3);
            G3_SB;
            mid(2);
            G3_SB;
            G3_MFMA(cur.ah[1], cur.wh[0], 1, 0);
            G3_SB;
            G3_MFMA(cur.ah[1], cur.wh[1], 1, 1);
            G3_SB;
            mid(3);
            G3_SB;
            G3_MFMA(cur.ah[1], cur.wh[2], 1, 2);
            G3_SB;
            tail();                                  // after the last mid(): it reloads the registers the mids store from
            G3_SB;
            G3_MFMA(cur.ah[1], cur.wh[3], 1, 3);
            G3_SB;
            return;
        }
        G3_MFMA(cur.ah[0], wl[0], 0, 0);
        G3_SB;
        mid(0);
        G3_SB;
        G3_MFMA(cur.ah[0], wl[1], 0, 1);
        G3_SB;
        G3_MFMA(cur.ah[0], wl[2], 0, 2);
        G3_SB;
        mid(1);
        G3_SB;
        G3_MFMA(cur.ah[0], wl[3], 0, 3);
        G3_SB;
        G3_MFMA(cur.ah[1], wl[0], 1, 0);
        G3_SB;
        mid(2);
        G3_SB;
        G3_MFMA(cur.ah[1], wl[1], 1, 1);
        G3_SB;
        G3_MFMA(cur.ah[1], wl[2], 1, 2);
        G3_SB;
        mid(3);
        G3_SB;
        G3_MFMA(cur.ah[1], wl[3], 1, 3);
        G3_SB;
        wl[0] = G3_RD(pw + G2_W_BYTES);
        G3_SB;
        wl[1] = G3_RD(pw + G2_W_BYTES + 2048);
        G3_SB;
        wl[2] = G3_RD(pw + G2_W_BYTES + 4096);
        G3_SB;
        wl[3] = G3_RD(pw + G2_W_BYTES + 6144);
        G3_SB;
        G3_MFMA(cur.ah[0], cur.wh[0], 0, 0);
        G3_SB;
        G3_MFMA(cur.ah[0], cur.wh[1], 0, 1);
        G3_SB;
        tail();
        G3_SB;
        G3_MFMA(cur.ah[0], cur.wh[2], 0, 2);
        G3_SB;
        G3_MFMA(cur.ah[0], cur.wh[3], 0, 3);
        G3_SB;
        G3_MFMA(cur.ah[1], cur.wh[0], 1, 0);
        G3_SB;
        G3_MFMA(cur.ah[1], cur.wh[1], 1, 1);
        G3_SB;
        G3_MFMA(cur.ah[1], cur.wh[2], 1, 2);
        G3_SB;
        G3_MFMA(cur.ah[1], cur.wh[3], 1, 3);
        G3_SB;
    };
    auto nothing = [] {};
    auto nothing1 = [](int) {};

    const int nk = K / G2_BK;                                         // >= 2 (checked by the launcher)
    // prologue of the workgroup's FIRST tile only: k-tile 0 complete in stage 0, its k-step 0 fragments in registers; k-tile 1: weights
    // in flight, activations in registers.  Every later tile's first k-tiles are requested by the previous tile's last iterations.
    gloadA(0);
    dmaW(0, 0);
    storeA(0);
    gloadA(G2_BK);
    dmaW(G2_BK, 1);
    if constexpr (WEX) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // k-tile 0's DMAs; k-tile 1's 4 loads + 2 DMAs may stay in flight
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");            // k-tile 0's 4 DMAs; k-tile 1's 4 loads + 4 DMAs may stay in flight
    __syncthreads();
    G3Frags f0, f1;
    {
        const char *pa = g2_lds + fa0, *pw = g2_lds + 2 * G2_A_BYTES + fw0;
#pragma unroll
        for (int a = 0; a < 2; ++a) { f0.ah[a] = G3_RD(pa + a * 2048); al[a] = G3_RD(pa + G2_A_BYTES + a * 2048); }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            f0.wh[b] = G3_RD(pw + b * 2048);
            if constexpr (!WEX) wl[b] = G3_RD(pw + G2_W_BYTES + b * 2048);
        }
    }

    int g = 0;                                                        // running k-tile count of this workgroup: stage = g & 1
    for (;;) {
        for (int kt = 0; kt < nk; ++kt, ++g) {
            const int cur = g & 1;
            const char *base = g2_lds + cur * G2_STAGE, *other = g2_lds + (cur ^ 1) * G2_STAGE;
            // what k-tile g+2 is: this tile's kt+2, the next tile's 0 / 1, or (last tile) a harmless re-request of the last k-tile
            int kload = (kt + 2) * G2_BK;
            if (kt + 2 >= nk) {
                if (nslot >= 0) {
                    if (kt + 2 == nk) set_load_tile(nm0, nn0);
                    kload = (kt + 2 - nk) * G2_BK;
                } else {
                    kload = (nk - 1) * G2_BK;
                }
            }
            // ---- phase 1: k-step 0's MFMAs; k-step 1's fragments; activation k-tile g+1 split + stored; k-tile g+2 requested
            k_step(f0, f1, base, 1, nothing, [&](int i) { storeA1(cur ^ 1, i); }, [&] { gloadA(kload); });
            // ---- sync: fragments and activation stores complete, weights of k-tile g+1 landed, stage `cur` no longer read by anyone
            asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
            __syncthreads();
            // ---- phase 2: k-step 1's MFMAs; weight k-tile g+2 by DMA into the stage just released; k-tile g+1's k-step 0 fragments
            k_step(f1, f0, other, 0, [&] { dmaW(kload, cur); }, nothing1, nothing);
        }
        // ---- epilogue of this tile; the next tile's first two k-tiles are already in flight / in LDS and keep arriving under the stores
        // Full tiles (all but the last row of tiles) store without a branch: with per-row predicates the compiler puts an
        // s_waitcnt vmcnt(0) in front of every store and the 128 stores of a wave complete one round trip at a time.
        float bv[4];
        const bool cols_in = n0 + wn * 128 < N;                      // wave-uniform (N % 128 == 0)
#pragma unroll
        for (int b = 0; b < 4; ++b) bv[b] = bias && cols_in ? bias[n0 + wn * 128 + b * 32 + l31] : 0.0f;
        const int mrow = m0 + wm * 64 + 4 * kh;
        float *ctile = C + (size_t)mrow * N + n0 + wn * 128 + l31;
        auto finish = [&](float v, int b) {
            v += bv[b];
            if (ACT == 1) v = v * (1.0f / (1.0f + __expf(-1.702f * v)));
            if (ACT == 2) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
            return v;
        };
        auto put = [&](float *p_, float v) {
            if constexpr (ACC) unsafeAtomicAdd(p_, v);
            else *p_ = v;
        };
        if (!cols_in) {
        } else if (m0 + G2_BM <= M) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r)
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        put(ctile + (size_t)(a * 32 + (r & 3) + 8 * (r >> 2)) * N + b * 32, finish(acc[a][b][r], b));
        } else {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dm = a * 32 + (r & 3) + 8 * (r >> 2);
                    if (mrow + dm < M) {
#pragma unroll
                        for (int b = 0; b < 4; ++b) put(ctile + (size_t)dm * N + b * 32, finish(acc[a][b][r], b));
                    }
                }
        }
        // range flag (common.h): accumulator + bias of the tile's live rows and columns, on the accumulators' way back to zero
        {
            unsigned mag = 0u;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const bool row_in = mrow + a * 32 + (r & 3) + 8 * (r >> 2) < M;
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        if (cols_in && row_in) mag = max(mag, x3_mag(acc[a][b][r] + bv[b]));      // pre-activation value
                        acc[a][b][r] = 0.0f;
                    }
                }
            x3_raise(range_flag, mag);
        }
        if (nslot < 0) break;
        m0 = nm0; n0 = nn0;
        nslot = next_valid(nslot + (int)gridDim.x, nm0, nn0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the tail's surplus DMAs must not outlive the workgroup's LDS
#undef G3_MFMA
#undef G3_RD
#undef G3_SB
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

static int linear_f16x3_impl(const float *A, int M, int K, const void *W_hi, const void *W_lo, const float *bias, int N, int act, float *C,
                             bool accumulate, void *stream);

extern "C" int oryon_linear_f16x3(const float *A, int M, int K, const void *W_hi, const void *W_lo, const float *bias, int N, int act,
                                  float *C, void *stream)
{
    return linear_f16x3_impl(A, M, K, W_hi, W_lo, bias, N, act, C, false, stream);
}

extern "C" int oryon_linear_f16x3_acc(const float *A, int M, int K, const void *W_hi, const void *W_lo, const float *bias, int N, float *C,
                                      void *stream)
{
    ORYON_CHECK_ARG(K >= 2 * G2_BK && N % 128 == 0 && (size_t)N * (size_t)K < (1ull << 30));      // the stream kernel's shapes only
    return linear_f16x3_impl(A, M, K, W_hi, W_lo, bias, N, 0, C, true, stream);
}

static int linear_f16x3_impl(const float *A, int M, int K, const void *W_hi, const void *W_lo, const float *bias, int N, int act, float *C,
                             bool accumulate, void *stream)
{
    ORYON_CHECK_ARG(A && W_hi && C && M >= 0 && K > 0 && N > 0);
    // W_lo == NULL: the weights are exactly representable in fp16 (stream-kernel shapes only)
    ORYON_CHECK_ARG(W_lo || (K >= 2 * G2_BK && (size_t)N * (size_t)K < (1ull << 30)));
    ORYON_CHECK_ARG(K % GX_BK == 0 && act >= 0 && act <= 2);
    ORYON_CHECK_ARG(N % GX_BN == 0 || (N % 128 == 0 && K >= 2 * G2_BK && (size_t)N * (size_t)K < (1ull << 30)));   // half-wide last column tile: stream kernel only
    if (M == 0) return ORYON_OK;
    static const int variant = dev_env_int("ORYON_GEMM_X3_VARIANT", 2);      // dev: 1 = small-tile kernel
    if ((variant != 1 || N % GX_BN != 0 || !W_lo || accumulate) && K >= 2 * G2_BK && (size_t)N * (size_t)K < (1ull << 30)) {
        const int tiles_m = (M + G2_BM - 1) / G2_BM, tiles_n = (N + G2_BN - 1) / G2_BN;
        const int sup_n = (tiles_n + 7) / 8;
        const int sup_cols = (tiles_n + sup_n - 1) / sup_n;
        const int sup_rows = 64 / sup_cols;
        const int sup_m = (tiles_m + sup_rows - 1) / sup_rows;
        const int n_slots = ((sup_m * sup_n + 7) / 8) * 8 * 64;
        int dev = 0, cus = 0;
        ORYON_CHECK_HIP(hipGetDevice(&dev));
        ORYON_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        int grid = cus - cus % 8;                                       // one persistent workgroup per CU (128 KB of LDS each)
        if (grid < 8) grid = 8;
        if (grid > n_slots) grid = n_slots;
        hipStream_t st2 = as_stream(stream);
        unsigned *rflag2 = x3_range_flag(st2);
        if (!rflag2) return ORYON_ERR_HIP;
        const __half *wh2 = static_cast<const __half *>(W_hi), *wl2 = static_cast<const __half *>(W_lo);
#define ORYON_LAUNCH_STREAM(ACT, WEX, ACC)                                                                                          \
    do {                                                                                                                            \
        allow_dynamic_lds(reinterpret_cast<const void *>(linear_f16x3_stream_kernel<ACT, WEX, ACC>), 2 * G2_STAGE);                 \
        hipLaunchKernelGGL((linear_f16x3_stream_kernel<ACT, WEX, ACC>), dim3(grid), dim3(512), 2 * G2_STAGE, st2, A, M, K, wh2, wl2,   \
                           bias, N, C, tiles_m, tiles_n, sup_n, sup_rows, sup_cols, n_slots, rflag2);                       \
    } while (0)
        if (accumulate) {
            if (!wl2) ORYON_LAUNCH_STREAM(0, true, true);
            else ORYON_LAUNCH_STREAM(0, false, true);
        } else if (!wl2) {
            if (act == 2) ORYON_LAUNCH_STREAM(2, true, false);
            else if (act == 1) ORYON_LAUNCH_STREAM(1, true, false);
            else ORYON_LAUNCH_STREAM(0, true, false);
        } else if (act == 2) ORYON_LAUNCH_STREAM(2, false, false);
        else if (act == 1) ORYON_LAUNCH_STREAM(1, false, false);
        else ORYON_LAUNCH_STREAM(0, false, false);
#undef ORYON_LAUNCH_STREAM
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
    unsigned *rflag = x3_range_flag(st);
    if (!rflag) return ORYON_ERR_HIP;
    const __half *wh = static_cast<const __half *>(W_hi), *wl = static_cast<const __half *>(W_lo);
    if (act == 2)
        hipLaunchKernelGGL((linear_f16x3_kernel<2>), grid, dim3(256), 0, st, A, M, K, wh, wl, bias, N, C, tiles_m, tiles_n, sup_n, sup_rows, sup_cols, rflag);
    else if (act == 1)
        hipLaunchKernelGGL((linear_f16x3_kernel<1>), grid, dim3(256), 0, st, A, M, K, wh, wl, bias, N, C, tiles_m, tiles_n, sup_n, sup_rows, sup_cols, rflag);
    else
        hipLaunchKernelGGL((linear_f16x3_kernel<0>), grid, dim3(256), 0, st, A, M, K, wh, wl, bias, N, C, tiles_m, tiles_n, sup_n, sup_rows, sup_cols, rflag);
    ORYON_CHECK_LAUNCH();
    return ORYON_OK;
}
