// K3 + K4: PointDSC feature encoder (NonLocalNet) on the fp32 matrix cores.
//
// Replaces models/pointdsc/PointDSC.py:150-156 (+ NonLocalBlock.forward :27-45, NonLocalNet.forward :65-77):
//   SC_ij   = clamp(1 - (|s_i-s_j| - |t_i-t_j|)^2 / sigma_d^2, 0)
//   layer   : feat = relu(BN(conv(feat)));  q,k,v = conv(feat);
//             w = softmax_j(SC_ij * q_i.k_j / sqrt(C));  msg = sum_j w_ij v_j;  feat += mlp(msg)
// The reference materialises SC [n,n] and the [n,n] attention matrix; here
//   * every 1x1 conv (+ folded eval-mode BatchNorm, ReLU, residual) is one LDS-tiled MFMA GEMM
//     (pdsc_linear_kernel), and
//   * the attention is a flash-style kernel that never stores an n x n object: SC is recomputed per
//     (query,key) from the six coordinates (keys' coordinates sit in LDS next to the K/V tiles) and the
//     softmax is online.  It works on TRANSPOSED tiles, S^T = K Q^T and O^T = V^T P^T: in the 32x32 MFMA
//     C/D layout a lane then owns ONE query column, so max / sum / rescale are lane-local, and the P
//     registers feed the second MFMA directly as its B operand (no LDS round trip, no shuffles).
#include <stdlib.h>
#include "common.h"
#include "pdsc.h"

namespace oryon {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// row index inside a 32x32 MFMA C/D block held by (register r, lane half hi)
__device__ __forceinline__ int crow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// ------------------------------------------------------------------------------------------------
// corr_pos = cat(src, tgt) - mean over the n rows (utils/pointdsc/init.py:18-19); rows >= n are zeroed.
// One workgroup per pair.
__global__ __launch_bounds__(256) void pdsc_center_kernel(const float *__restrict__ src, const float *__restrict__ tgt,
                                                           const int32_t *__restrict__ n_rows, int n_cap,
                                                           float *__restrict__ corr_pos /*[B,n_cap,8]*/)
{
    __shared__ double s_part[4][6];
    __shared__ float s_mean[6];
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int n = n_rows[b];
    const float *s = src + (size_t)b * n_cap * 3, *g = tgt + (size_t)b * n_cap * 3;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (int i = t; i < n; i += 256) {
#pragma unroll
        for (int d = 0; d < 3; ++d) { acc[d] += s[3 * i + d]; acc[3 + d] += g[3 * i + d]; }
    }
#pragma unroll
    for (int d = 0; d < 6; ++d) {
        double v = acc[d];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if (lane == 0) s_part[wave][d] = v;
    }
    __syncthreads();
    if (t < 6) s_mean[t] = (float)((s_part[0][t] + s_part[1][t] + s_part[2][t] + s_part[3][t]) / (double)(n > 0 ? n : 1));
    __syncthreads();
    float *o = corr_pos + (size_t)b * n_cap * 8;
    for (int i = t; i < n_cap; i += 256) {
        float v[8];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            v[d] = i < n ? s[3 * i + d] - s_mean[d] : 0.0f;
            v[3 + d] = i < n ? g[3 * i + d] - s_mean[3 + d] : 0.0f;
        }
        v[6] = v[7] = 0.0f;
#pragma unroll
        for (int d = 0; d < 8; ++d) o[(size_t)i * 8 + d] = v[d];
    }
}

// ------------------------------------------------------------------------------------------------
// Y[b, rows, :N] = act(X[b, rows, :K] W^T + bias) (+ R[b, rows, :N]);   W is [N, K] row-major.
// Tile: 64 rows x 128 columns per workgroup, k-tiles of 32 staged in LDS (LD = 33: conflict-free
// ds_read_b32 of a 32-row column); 4 waves as 2 (row halves) x 2 (column halves of 64 = two 32x32 blocks).
// Measured and NOT kept: prefetching k-tile t+1 under tile t's MFMAs (+-0), 16-byte staging loads (-20 %: LDS store conflicts).
constexpr int LIN_ROWS = 64, LIN_COLS = 128, LIN_BK = 32, LIN_LD = LIN_BK + 1;

template <bool RELU, bool RESID>
__global__ __launch_bounds__(256) void pdsc_linear_kernel(const float *__restrict__ X, int ldx, size_t x_batch,
                                                           const float *__restrict__ W, const float *__restrict__ bias,
                                                           const float *__restrict__ R, int ldr, size_t r_batch,
                                                           float *__restrict__ Y, int ldy, size_t y_batch, int K, int N,
                                                           const int32_t *__restrict__ n_rows)
{
    __shared__ float Xs[LIN_ROWS * LIN_LD];
    __shared__ float Ws[LIN_COLS * LIN_LD];
    const int b = blockIdx.z;
    const int m0 = blockIdx.x * LIN_ROWS, n0 = blockIdx.y * LIN_COLS;
    if (n_rows && m0 >= n_rows[b]) return;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, hi = lane >> 5;
    const float *x = X + (size_t)b * x_batch + (size_t)m0 * ldx;
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;

    for (int k0 = 0; k0 < K; k0 += LIN_BK) {
        // all 24 global loads of this k-tile are in flight before the first LDS store
        constexpr int NX = (LIN_ROWS * LIN_BK) / 256, NW = (LIN_COLS * LIN_BK) / 256;
        float xv[NX], wv[NW];
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int e = t + 256 * i, row = e >> 5, kk = e & 31;
            xv[i] = (k0 + kk < K) ? x[(size_t)row * ldx + k0 + kk] : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int e = t + 256 * i, col = e >> 5, kk = e & 31;
            wv[i] = (n0 + col < N && k0 + kk < K) ? W[(size_t)(n0 + col) * K + k0 + kk] : 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int e = t + 256 * i;
            Xs[(e >> 5) * LIN_LD + (e & 31)] = xv[i];
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int e = t + 256 * i;
            Ws[(e >> 5) * LIN_LD + (e & 31)] = wv[i];
        }
        __syncthreads();
        const float *xa = Xs + (wm * 32 + l31) * LIN_LD + hi;
        const float *wb = Ws + (wn * 64 + l31) * LIN_LD + hi;
#pragma unroll
        for (int ks = 0; ks < LIN_BK / 2; ++ks) {
            const float a = xa[2 * ks];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wb[2 * ks], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wb[32 * LIN_LD + 2 * ks], acc[1], 0, 0, 0);
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = n0 + wn * 64 + j * 32 + l31;
        if (c >= N) continue;
        const float bv = bias ? bias[c] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * 32 + crow(r, hi);
            float v = acc[j][r] + bv;
            if (RELU) v = v > 0.0f ? v : 0.0f;
            if (RESID) v += R[(size_t)b * r_batch + (size_t)row * ldr + c];
            Y[(size_t)b * y_batch + (size_t)row * ldy + c] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Spatial-consistency matrix (PointDSC.py:150-153), computed ONCE per registration and reused by all layers (as the reference
// does): SC_qk = clamp(1 - (|s_q - s_k| - |t_q - t_k|)^2 / sigma_d^2, 0).  Stored in the register layout of the attention
// kernel so that a lane fetches its 32 values of a (32-query, 64-key) tile with 8 coalesced 16-byte loads:
//   sc[b][q_block32][key_tile64][v4 = kb*4 + r/4][lane][r%4],  key = kb*32 + crow(r, lane/32), query = lane%32.
// Keys >= n hold -1 (masked).
constexpr int ATT_Q = 128, ATT_KT = 64;
__global__ __launch_bounds__(256) void pdsc_sc_kernel(const float *__restrict__ src, const float *__restrict__ tgt,
                                                       const int32_t *__restrict__ n_rows, int n_cap, float inv_sigma2,
                                                       float *__restrict__ sc)
{
    const int b = blockIdx.z, kt = blockIdx.y;
    const int n = n_rows[b];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, hi = lane >> 5;
    const int qb = blockIdx.x * 4 + wave;
    if (qb * 32 >= n || kt * ATT_KT >= n) return;
    const float *sp = src + (size_t)b * n_cap * 3, *tp = tgt + (size_t)b * n_cap * 3;
    const int q = qb * 32 + l31;
    float sq[3], tq[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) { sq[d] = sp[(size_t)q * 3 + d]; tq[d] = tp[(size_t)q * 3 + d]; }
    float4 *out = reinterpret_cast<float4 *>(sc) + ((((size_t)b * (n_cap / 32) + qb) * (n_cap / ATT_KT) + kt) * 8) * 64 + lane;
#pragma unroll
    for (int v4 = 0; v4 < 8; ++v4) {
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int r = (v4 & 3) * 4 + e, kb = v4 >> 2;
            const int k = kt * ATT_KT + kb * 32 + crow(r, hi);
            float v = -1.0f;
            if (k < n) {
                const float dx = sq[0] - sp[(size_t)k * 3], dy = sq[1] - sp[(size_t)k * 3 + 1], dz = sq[2] - sp[(size_t)k * 3 + 2];
                const float ex = tq[0] - tp[(size_t)k * 3], ey = tq[1] - tp[(size_t)k * 3 + 1], ez = tq[2] - tp[(size_t)k * 3 + 2];
                const float ds = sqrt_rn(dx * dx + dy * dy + dz * dz);
                const float dt = sqrt_rn(ex * ex + ey * ey + ez * ez);
                const float df = ds - dt;
                v = 1.0f - df * df * inv_sigma2;
                v = v > 0.0f ? v : 0.0f;
            }
            o[e] = v;
        }
        out[(size_t)v4 * 64] = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// Flash-style SC-modulated attention.  One workgroup = 128 queries (4 waves x 32), key tiles of 64.
// QKV: [B, n_cap, 3C] (q | k | v), sc: pdsc_sc_kernel's tiles;  msg: [B, n_cap, C].
// The K/V rows and the SC values of tile t+1 are fetched into registers while tile t is computed (the phases of a wave run
// back to back - measured: MFMA 43 %, softmax VALU 25 %, staging 16 % of the un-prefetched kernel, nothing overlapping).
// Key split (flash-decoding), used for small batches: blockIdx.y = split s of KS takes a share of the key tiles and writes
// un-normalised partials (O_s, m_s, l_s) that `pdsc_attention_merge_kernel` combines (exact softmax algebra, fp32).
template <int C>
__global__ __launch_bounds__(256) void pdsc_attention_kernel(const float *__restrict__ QKV, const float *__restrict__ sc,
                                                              const int32_t *__restrict__ n_rows, int n_cap,
                                                              float inv_sqrt_c, float *__restrict__ msg,
                                                              int KS, float *__restrict__ part_o, float *__restrict__ part_ml)
{
    constexpr int LD = C + 4;                 // 16-byte aligned rows: staging lands with ds_write_b128
    constexpr int CB = C / 32;
    constexpr int F4_PER_ROW = C / 4, PER_THREAD = ATT_KT * F4_PER_ROW / 256;
    __shared__ __attribute__((aligned(16))) float Ks[ATT_KT * LD];
    __shared__ __attribute__((aligned(16))) float Vs[ATT_KT * LD];
    const int b = blockIdx.z, split = blockIdx.y;
    const int n = n_rows[b];
    const int q0 = blockIdx.x * ATT_Q;
    if (q0 >= n) return;
    const int n_tiles = (n + ATT_KT - 1) / ATT_KT, per = (n_tiles + KS - 1) / KS;
    const int j_begin = split * per * ATT_KT;
    const int j_end = ((split + 1) * per * ATT_KT < n) ? (split + 1) * per * ATT_KT : n;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, hi = lane >> 5;
    const int qrow = q0 + wave * 32 + l31;  // this lane's query (< n_cap always)
    const float *base = QKV + (size_t)b * n_cap * 3 * C;
    const float4 *sc_q = reinterpret_cast<const float4 *>(sc) + (((size_t)b * (n_cap / 32) + (q0 / 32 + wave)) * (n_cap / ATT_KT)) * 8 * 64 + lane;
    const bool q_live = q0 + wave * 32 < n;         // query blocks past n have no SC tiles (their rows are never consumed)

    float4 kv[PER_THREAD], vv[PER_THREAD], scv[8];
    auto fetch = [&](int j0) {
#pragma unroll
        for (int i = 0; i < PER_THREAD; ++i) {
            const int e = t + 256 * i, row = e / F4_PER_ROW, c4 = e % F4_PER_ROW;
            const bool in = j0 + row < n_cap;
            const float *rp = base + (size_t)(in ? j0 + row : 0) * 3 * C;
            kv[i] = in ? *reinterpret_cast<const float4 *>(rp + C + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
            vv[i] = in ? *reinterpret_cast<const float4 *>(rp + 2 * C + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const float4 *sp = sc_q + (size_t)(j0 / ATT_KT) * 8 * 64;
#pragma unroll
        for (int v4 = 0; v4 < 8; ++v4) scv[v4] = q_live ? sp[(size_t)v4 * 64] : make_float4(-1.f, -1.f, -1.f, -1.f);
    };
    auto land = [&]() {
#pragma unroll
        for (int i = 0; i < PER_THREAD; ++i) {
            const int e = t + 256 * i, row = e / F4_PER_ROW, c4 = e % F4_PER_ROW;
            *reinterpret_cast<float4 *>(Ks + row * LD + 4 * c4) = kv[i];
            *reinterpret_cast<float4 *>(Vs + row * LD + 4 * c4) = vv[i];
        }
    };

    // Q^T as B operand: lane (query l31, half hi) holds Q[query][2ks + hi]
    float qreg[C / 2];
    {
        const float2 *qv = reinterpret_cast<const float2 *>(base + (size_t)qrow * 3 * C);
#pragma unroll
        for (int ks = 0; ks < C / 2; ++ks) {
            const float2 v = qv[ks];
            qreg[ks] = hi ? v.y : v.x;
        }
    }
    f32x16 acc_o[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[cb][r] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;

    if (j_begin < j_end) fetch(j_begin);
    for (int j0 = j_begin; j0 < j_end; j0 += ATT_KT) {
        __syncthreads();                       // everyone is done reading the previous tile
        land();
        float4 sct[8];
#pragma unroll
        for (int v4 = 0; v4 < 8; ++v4) sct[v4] = scv[v4];
        __syncthreads();
        if (j0 + ATT_KT < j_end) fetch(j0 + ATT_KT);      // in flight during this tile's MFMAs and softmax

        // S^T = K Q^T : rows = keys (two blocks of 32, alternating so that back-to-back MFMAs are independent),
        // columns = this wave's 32 queries
        f32x16 s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = 0.0f;
        {
            const float *ka0 = Ks + l31 * LD + hi, *ka1 = Ks + (32 + l31) * LD + hi;
#pragma unroll
            for (int ks = 0; ks < C / 2; ++ks) {
                s[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ka0[2 * ks], qreg[ks], s[0], 0, 0, 0);
                s[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ka1[2 * ks], qreg[ks], s[1], 0, 0, 0);
            }
        }
        // logits = SC * (q.k) / sqrt(C);  running max
        float m_tile = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float4 q4 = sct[kb * 4 + (r >> 2)];
                const float scq = (r & 3) == 0 ? q4.x : (r & 3) == 1 ? q4.y : (r & 3) == 2 ? q4.z : q4.w;
                float v = scq * (s[kb][r] * inv_sqrt_c);
                v = (scq >= 0.0f) ? v : -INFINITY;
                s[kb][r] = v;
                m_tile = fmaxf(m_tile, v);
            }
        m_tile = fmaxf(m_tile, __shfl_xor(m_tile, 32));
        const float m_new = fmaxf(m_run, m_tile);        // finite for live queries: key j0 < n is always live
        const float alpha = __expf(m_run - m_new);       // exp(-inf) = 0 on the first tile
        float l_tile = 0.0f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = expf(s[kb][r] - m_new);
                s[kb][r] = p;
                l_tile += p;
            }
        l_run = l_run * alpha + l_tile;
        m_run = m_new;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc_o[cb][r] *= alpha;
        // O^T += V^T P^T : A = V^T (lane (channel l31, half hi) reads V[key(r,hi)][channel]), B = own P registers
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float *va = Vs + (kb * 32 + crow(r, hi)) * LD + l31;
                const float p = s[kb][r];
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) acc_o[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[cb * 32], p, acc_o[cb], 0, 0, 0);
            }
    }
    const float l_all = l_run + __shfl_xor(l_run, 32);
    if (KS > 1) {
        // partials: O un-normalised (empty key ranges leave O = 0, m = -inf, l = 0 and drop out of the merge)
        const size_t prow = ((size_t)split * gridDim.z + b) * n_cap + qrow;
        float *po = part_o + prow * C;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 v;
                v.x = acc_o[cb][4 * g + 0]; v.y = acc_o[cb][4 * g + 1]; v.z = acc_o[cb][4 * g + 2]; v.w = acc_o[cb][4 * g + 3];
                *reinterpret_cast<float4 *>(po + cb * 32 + 8 * g + 4 * hi) = v;
            }
        if (hi == 0) { part_ml[prow * 2] = m_run; part_ml[prow * 2 + 1] = l_all; }
        return;
    }
    const float inv_l = 1.0f / l_all;
    float *mo = msg + ((size_t)b * n_cap + qrow) * C;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 v;
            v.x = acc_o[cb][4 * g + 0] * inv_l;
            v.y = acc_o[cb][4 * g + 1] * inv_l;
            v.z = acc_o[cb][4 * g + 2] * inv_l;
            v.w = acc_o[cb][4 * g + 3] * inv_l;
            *reinterpret_cast<float4 *>(mo + cb * 32 + 8 * g + 4 * hi) = v;
        }
}

// ------------------------------------------------------------------------------------------------
// The same attention on the fp16 matrix pipe with error-compensated operands ("fp16x3"): every fp32 operand x is split as
// x = hi + lo with hi = half(x), lo = half(x - hi) (22 significant bits together) and a product a.b is accumulated as
// a_hi.b_hi + a_hi.b_lo + a_lo.b_hi on v_mfma_f32_32x32x16_f16 (exact fp16 x fp16 products, fp32 accumulate).  The dropped
// a_lo.b_lo term and the split residuals are ~2^-22 |a.b| - the order of the fp32 accumulation error itself - while three
// fp16 MFMAs replace eight fp32 ones (16 k per 32 cycles vs 2 k per 64 cycles).  Range: operands must stay below 65504 in
// magnitude (PointDSC activations are O(1..100)).  Layout notes:
//   K tile   [key][channel] halves (hi and lo arrays): a lane's 8 consecutive channels are one ds_read_b128
//   V tile   arranged so that the MFMA k-slot (lane half h, element e) of block (kb, t) IS the key the softmax registers hold:
//            Vp[kb][t][h][channel][e] with key = kb*32 + 16t + 8(e>>2) + 4h + (e&3)  - again one ds_read_b128 per operand
//   P        split in registers right after the softmax (B operand of the second product)
typedef _Float16 xhalf8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split_half(float x, _Float16 &hi, _Float16 &lo)
{
    hi = (_Float16)x;
    lo = (_Float16)(x - (float)hi);
}

// Two values at a time: packed conversions (v_cvt_pk_f16_f32 on gfx950, round-to-nearest-even) - same results as split_half.
typedef float xf32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 xf16x2 __attribute__((ext_vector_type(2)));
// Round 6: the residuals x - float(hi) come from v_fma_mix_f32 (hi's half read as the f16 source of an fp32 fma: float(hi) * -1 + x, one
// rounding of an exactly representable difference - the bits of the subtraction it replaces), four instructions per pair instead of six.
__device__ __forceinline__ void split_pair(float a, float b, unsigned &hi, unsigned &lo)
{
    const xf32x2 v = {a, b};
    const xf16x2 h = __builtin_convertvector(v, xf16x2);
    const unsigned hb = __builtin_bit_cast(unsigned, h);
    float l0, l1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(hb), "v"(a));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(hb), "v"(b));
    const xf32x2 lv = {l0, l1};
    hi = hb;
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(lv, xf16x2));
}

template <int C>
__global__ __launch_bounds__(256) void pdsc_attention_x3_kernel(const float *__restrict__ QKV, const float *__restrict__ sc,
                                                                 const int32_t *__restrict__ n_rows, int n_cap,
                                                                 float inv_sqrt_c, float *__restrict__ msg,
                                                                 int KS, float *__restrict__ part_o, float *__restrict__ part_ml)
{
    constexpr int CB = C / 32;
    constexpr int NS = C / 16;                    // k16 steps of the first product
    constexpr int KLD = C + 8;                    // halves per K row (16-byte pad: rows 16 bytes apart in bank space)
    constexpr int KF4 = ATT_KT * (C / 4) / 256;   // float4 of K per thread and tile
    constexpr int VPT = ATT_KT * C / 256;         // V scalars per thread and tile
    constexpr int VOCT = VPT / 8;                 // key octets per thread
    static_assert(C % 32 == 0 && C <= 256 && VPT % 8 == 0 && 256 % C == 0, "tile geometry");
    __shared__ __attribute__((aligned(16))) _Float16 Kh[ATT_KT * KLD], Kl[ATT_KT * KLD];
    __shared__ __attribute__((aligned(16))) _Float16 Vh[ATT_KT * C], Vl[ATT_KT * C];
    const int b = blockIdx.z, split = blockIdx.y;
    const int n = n_rows[b];
    const int q0 = blockIdx.x * ATT_Q;
    if (q0 >= n) return;
    const int n_tiles = (n + ATT_KT - 1) / ATT_KT, per = (n_tiles + KS - 1) / KS;
    const int j_begin = split * per * ATT_KT;
    const int j_end = ((split + 1) * per * ATT_KT < n) ? (split + 1) * per * ATT_KT : n;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, hi = lane >> 5;
    const int qrow = q0 + wave * 32 + l31;
    const float *base = QKV + (size_t)b * n_cap * 3 * C;
    const float4 *sc_q = reinterpret_cast<const float4 *>(sc) + (((size_t)b * (n_cap / 32) + (q0 / 32 + wave)) * (n_cap / ATT_KT)) * 8 * 64 + lane;
    const bool q_live = q0 + wave * 32 < n;

    // staging registers of the NEXT tile: K as float4 (4 channels of a key), V as scalars (thread = one channel, 8 keys per octet)
    float4 kv[KF4], scv[8];
    float vv[VPT];
    constexpr int GROUPS = 256 / C;               // thread groups along the key octets (2 at C = 128)
    const int vch = t % C, vgrp = t / C;
    auto fetch = [&](int j0) {
#pragma unroll
        for (int i = 0; i < KF4; ++i) {
            const int e = t + 256 * i, row = e / (C / 4), c4 = e % (C / 4);
            kv[i] = *reinterpret_cast<const float4 *>(base + (size_t)(j0 + row) * 3 * C + C + 4 * c4);   // n_cap % ATT_KT == 0: the tile is allocated
        }
#pragma unroll
        for (int o = 0; o < VOCT; ++o) {
            const int oct = vgrp + GROUPS * o;     // octet index = (kb*2 + t2)*2 + h
            const int kb = oct >> 2, t2 = (oct >> 1) & 1, h = oct & 1;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int key = j0 + kb * 32 + 16 * t2 + 8 * (e >> 2) + 4 * h + (e & 3);
                vv[o * 8 + e] = base[(size_t)key * 3 * C + 2 * C + vch];
            }
        }
    };
    // the SC values of a tile are fetched right after the softmax of the previous one has consumed them (no second copy live)
    auto fetch_sc = [&](int j0) {
        const float4 *sp = sc_q + (size_t)(j0 / ATT_KT) * 8 * 64;
#pragma unroll
        for (int v4 = 0; v4 < 8; ++v4) scv[v4] = q_live ? sp[(size_t)v4 * 64] : make_float4(-1.f, -1.f, -1.f, -1.f);
    };
    auto land = [&]() {
#pragma unroll
        for (int i = 0; i < KF4; ++i) {
            const int e = t + 256 * i, row = e / (C / 4), c4 = e % (C / 4);
            uint2 ph, pl;
            split_pair(kv[i].x, kv[i].y, ph.x, pl.x);
            split_pair(kv[i].z, kv[i].w, ph.y, pl.y);
            *reinterpret_cast<uint2 *>(Kh + row * KLD + 4 * c4) = ph;
            *reinterpret_cast<uint2 *>(Kl + row * KLD + 4 * c4) = pl;
        }
#pragma unroll
        for (int o = 0; o < VOCT; ++o) {
            const int oct = vgrp + GROUPS * o;
            uint4 ph, pl;
            split_pair(vv[o * 8 + 0], vv[o * 8 + 1], ph.x, pl.x);
            split_pair(vv[o * 8 + 2], vv[o * 8 + 3], ph.y, pl.y);
            split_pair(vv[o * 8 + 4], vv[o * 8 + 5], ph.z, pl.z);
            split_pair(vv[o * 8 + 6], vv[o * 8 + 7], ph.w, pl.w);
            *reinterpret_cast<uint4 *>(Vh + ((size_t)oct * C + vch) * 8) = ph;
            *reinterpret_cast<uint4 *>(Vl + ((size_t)oct * C + vch) * 8) = pl;
        }
    };

    // Q^T as B operand: lane (query l31, half hi), k16 step s -> channels 16s + 8hi .. +7, split once
    xhalf8 qh[NS], ql[NS];
    {
        const float4 *qv = reinterpret_cast<const float4 *>(base + (size_t)qrow * 3 * C);
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) {
            const float4 a = qv[4 * s_ + 2 * hi], c = qv[4 * s_ + 2 * hi + 1];
            const float x[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                _Float16 h_, l_;
                split_half(x[e], h_, l_);
                qh[s_][e] = h_;
                ql[s_][e] = l_;
            }
        }
    }
    f32x16 acc_o[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[cb][r] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;

    if (j_begin < j_end) { fetch(j_begin); fetch_sc(j_begin); }
    for (int j0 = j_begin; j0 < j_end; j0 += ATT_KT) {
        __syncthreads();
        land();
        __syncthreads();
        if (j0 + ATT_KT < j_end) fetch(j0 + ATT_KT);

        // S^T = K Q^T, three fp16 products per k16 step and key block
        f32x16 s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = 0.0f;
        // One wave per SIMD: nobody hides an LDS read that is waited for right after its issue, and that is how the compiler orders
        // this loop when left alone (ds_read -> s_waitcnt lgkmcnt(0) -> 1-2 MFMAs, 32 times per tile).  The fragments of k16 step
        // s_+1 are therefore requested before the MFMAs of step s_, and sched_barrier keeps it that way.
        xhalf8 kf[2][2][2];                           // [buffer][key block][hi | lo]
        auto read_k = [&](int s_, int buf) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                kf[buf][kb][0] = *reinterpret_cast<const xhalf8 *>(Kh + (kb * 32 + l31) * KLD + 16 * s_ + 8 * hi);
                kf[buf][kb][1] = *reinterpret_cast<const xhalf8 *>(Kl + (kb * 32 + l31) * KLD + 16 * s_ + 8 * hi);
            }
        };
        read_k(0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) {
            const int buf = s_ & 1;
            if (s_ + 1 < NS) read_k(s_ + 1, buf ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            // the two key blocks alternate: back-to-back MFMAs are independent; per accumulator the order stays hi*hi, hi*lo, lo*hi
            s[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[buf][0][0], qh[s_], s[0], 0, 0, 0);
            s[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[buf][1][0], qh[s_], s[1], 0, 0, 0);
            s[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[buf][0][0], ql[s_], s[0], 0, 0, 0);
            s[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[buf][1][0], ql[s_], s[1], 0, 0, 0);
            s[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[buf][0][1], qh[s_], s[0], 0, 0, 0);
            s[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[buf][1][1], qh[s_], s[1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        float m_tile = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float4 q4 = scv[kb * 4 + (r >> 2)];
                const float scq = (r & 3) == 0 ? q4.x : (r & 3) == 1 ? q4.y : (r & 3) == 2 ? q4.z : q4.w;
                float v = scq * (s[kb][r] * inv_sqrt_c);
                v = (scq >= 0.0f) ? v : -INFINITY;
                s[kb][r] = v;
                m_tile = fmaxf(m_tile, v);
            }
        if (j0 + ATT_KT < j_end) fetch_sc(j0 + ATT_KT);
        m_tile = fmaxf(m_tile, __shfl_xor(m_tile, 32));
        const float m_new = fmaxf(m_run, m_tile);
        const float alpha = __expf(m_run - m_new);
        float l_tile = 0.0f;
        xhalf8 ph[2][2], pl[2][2];                  // [kb][t2]: keys crow(8*t2 + e, hi)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const float p0 = __expf(s[kb][r] - m_new);    // v_exp_f32 path: ~1e-6 relative, VALU issue is what bounds this kernel
                const float p1 = __expf(s[kb][r + 1] - m_new);
                l_tile += p0;
                l_tile += p1;
                unsigned uh, ul;
                split_pair(p0, p1, uh, ul);
                const xf16x2 h2 = __builtin_bit_cast(xf16x2, uh), l2 = __builtin_bit_cast(xf16x2, ul);
                ph[kb][r >> 3][r & 7] = h2[0]; ph[kb][r >> 3][(r & 7) + 1] = h2[1];
                pl[kb][r >> 3][r & 7] = l2[0]; pl[kb][r >> 3][(r & 7) + 1] = l2[1];
            }
        l_run = l_run * alpha + l_tile;
        m_run = m_new;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc_o[cb][r] *= alpha;
        // O^T += V^T P^T, the V fragments of key octet pair o+1 requested before the MFMAs of pair o (as above)
        xhalf8 vf[2][CB][2];                          // [buffer][channel block][hi | lo]
        auto read_v = [&](int o, int buf) {           // o = kb * 2 + t2
            const int oct = o * 2 + hi;
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                vf[buf][cb][0] = *reinterpret_cast<const xhalf8 *>(Vh + ((size_t)oct * C + cb * 32 + l31) * 8);
                vf[buf][cb][1] = *reinterpret_cast<const xhalf8 *>(Vl + ((size_t)oct * C + cb * 32 + l31) * 8);
            }
        };
        read_v(0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const int buf = o & 1, kb = o >> 1, t2 = o & 1;
            if (o + 1 < 4) read_v(o + 1, buf ^ 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) acc_o[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[buf][cb][0], ph[kb][t2], acc_o[cb], 0, 0, 0);
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) acc_o[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[buf][cb][0], pl[kb][t2], acc_o[cb], 0, 0, 0);
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) acc_o[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[buf][cb][1], ph[kb][t2], acc_o[cb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const float l_all = l_run + __shfl_xor(l_run, 32);
    if (KS > 1) {
        const size_t prow = ((size_t)split * gridDim.z + b) * n_cap + qrow;
        float *po = part_o + prow * C;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 v;
                v.x = acc_o[cb][4 * g + 0]; v.y = acc_o[cb][4 * g + 1]; v.z = acc_o[cb][4 * g + 2]; v.w = acc_o[cb][4 * g + 3];
                *reinterpret_cast<float4 *>(po + cb * 32 + 8 * g + 4 * hi) = v;
            }
        if (hi == 0) { part_ml[prow * 2] = m_run; part_ml[prow * 2 + 1] = l_all; }
        return;
    }
    const float inv_l = 1.0f / l_all;
    float *mo = msg + ((size_t)b * n_cap + qrow) * C;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 v;
            v.x = acc_o[cb][4 * g + 0] * inv_l;
            v.y = acc_o[cb][4 * g + 1] * inv_l;
            v.z = acc_o[cb][4 * g + 2] * inv_l;
            v.w = acc_o[cb][4 * g + 3] * inv_l;
            *reinterpret_cast<float4 *>(mo + cb * 32 + 8 * g + 4 * hi) = v;
        }
}

// The same kernel fed by pdsc_pcn_qkv_x3_kernel's K / V tile images (C = 128, no key split across workgroups): K / V of a tile arrive
// pre-split and in LDS layout, 66 pieces of 1 KB by LDS-DMA - no staging registers, no conversion work (the fp32-fed kernel splits the same
// K / V in each of a pair's four query-block workgroups) - and the next tile lands in a second buffer while this one is multiplied: one
// barrier per tile instead of two.
template <int C>
__global__ __launch_bounds__(256) void pdsc_attention_x3_img_kernel(const float *__restrict__ QKV, const char *__restrict__ kv_img,
                                                                     const float *__restrict__ sc, const int32_t *__restrict__ n_rows,
                                                                     int n_cap, float inv_sqrt_c, float *__restrict__ msg, int n_pairs)
{
    static_assert(C == 128, "tile image geometry");
    constexpr int CB = C / 32;
    constexpr int NS = C / 16;                    // k16 steps of the first product
    // (K rows of the tile image: pdsc_k_img_elem, pdsc.h)
    extern __shared__ __attribute__((aligned(1024))) char att_lds[];            // two tile images
    // XCD-aware block map: the query blocks of one pair read the same K / V tile images, so they go to ONE XCD (linear block id mod 8) and
    // share the images through its L2 (with the plain (query block, pair) grid a pair's four blocks landed on four XCDs and each fetched
    // the pair's 540 KB of images for itself).  The grid's z extent is B rounded up to a multiple of 8.
    const int lin = blockIdx.x + gridDim.x * blockIdx.z;
    const int b = (lin / 8 / (int)gridDim.x) * 8 + (lin & 7);
    const int qblk = (lin / 8) % (int)gridDim.x;
    if (b >= n_pairs) return;
    const int n = n_rows[b];
    const int q0 = qblk * ATT_Q;
    if (q0 >= n) return;
    const int j_begin = 0, j_end = n;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, hi = lane >> 5;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int qrow = q0 + wave * 32 + l31;
    const float *base = QKV + (size_t)b * n_cap * 3 * C;
    const char *img = kv_img + (size_t)b * (n_cap / ATT_KT) * PDSC_KV_TILE_BYTES;
    const float4 *sc_q = reinterpret_cast<const float4 *>(sc) + (((size_t)b * (n_cap / 32) + (q0 / 32 + wave)) * (n_cap / ATT_KT)) * 8 * 64 + lane;
    const bool q_live = q0 + wave * 32 < n;
    float4 scv[8];
    auto dma_tile = [&](int j0, int buf) {
        const char *src = img + (size_t)(j0 / ATT_KT) * PDSC_KV_TILE_BYTES;
#pragma unroll
        for (int j = 0; j < 17; ++j) {
            const int piece = wave_u * 17 + j;
            if (piece < PDSC_KV_TILE_BYTES / 1024)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + piece * 1024 + lane * 16),
                                                 (__attribute__((address_space(3))) void *)(att_lds + buf * PDSC_KV_TILE_BYTES + piece * 1024), 16, 0, 0);
        }
    };
    auto fetch_sc = [&](int j0) {
        const float4 *sp = sc_q + (size_t)(j0 / ATT_KT) * 8 * 64;
#pragma unroll
        for (int v4 = 0; v4 < 8; ++v4) scv[v4] = q_live ? sp[(size_t)v4 * 64] : make_float4(-1.f, -1.f, -1.f, -1.f);
    };

    // Q^T as B operand: lane (query l31, half hi), k16 step s -> channels 16s + 8hi .. +7, split once
    xhalf8 qh[NS], ql[NS];
    {
        const float4 *qv = reinterpret_cast<const float4 *>(base + (size_t)qrow * 3 * C);
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) {
            const float4 a = qv[4 * s_ + 2 * hi], c = qv[4 * s_ + 2 * hi + 1];
            const float x[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                _Float16 h_, l_;
                split_half(x[e], h_, l_);
                qh[s_][e] = h_;
                ql[s_][e] = l_;
            }
        }
    }
    f32x16 acc_o[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[cb][r] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;

    dma_tile(0, 0);
    fetch_sc(0);
    int buf = 0;
    for (int j0 = j_begin; j0 < j_end; j0 += ATT_KT, buf ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // this tile's pieces (and SC values) have landed
        __syncthreads();                                                 // ... everybody's; and the other buffer is no longer read
        if (j0 + ATT_KT < j_end) dma_tile(j0 + ATT_KT, buf ^ 1);
        const _Float16 *Kh = reinterpret_cast<const _Float16 *>(att_lds + buf * PDSC_KV_TILE_BYTES);
        const _Float16 *Kl = reinterpret_cast<const _Float16 *>(att_lds + buf * PDSC_KV_TILE_BYTES + PDSC_KV_KL);
        const _Float16 *Vh = reinterpret_cast<const _Float16 *>(att_lds + buf * PDSC_KV_TILE_BYTES + PDSC_KV_VH);
        const _Float16 *Vl = reinterpret_cast<const _Float16 *>(att_lds + buf * PDSC_KV_TILE_BYTES + PDSC_KV_VL);

        // S^T = K Q^T, three fp16 products per k16 step and key block
        f32x16 s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = 0.0f;
        // One wave per SIMD: nobody hides an LDS read that is waited for right after its issue, and that is how the compiler orders
        // this loop when left alone (ds_read -> s_waitcnt lgkmcnt(0) -> 1-2 MFMAs, 32 times per tile).  The fragments of k16 step
        // s_+1 are therefore requested before the MFMAs of step s_, and sched_barrier keeps it that way.
        xhalf8 kf[2][2][2];                           // [buffer][key block][hi | lo]
        auto read_k = [&](int s_, int buf) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                kf[buf][kb][0] = *reinterpret_cast<const xhalf8 *>(Kh + pdsc_k_img_elem(kb * 32 + l31, 2 * s_ + hi));
                kf[buf][kb][1] = *reinterpret_cast<const xhalf8 *>(Kl + pdsc_k_img_elem(kb * 32 + l31, 2 * s_ + hi));
            }
        };
        read_k(0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) {
            const int buf = s_ & 1;
            if (s_ + 1 < NS) read_k(s_ + 1, buf ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            // the two key blocks alternate: back-to-back MFMAs are independent; per accumulator the order stays hi*hi, hi*lo, lo*hi
            s[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[buf][0][0], qh[s_], s[0], 0, 0, 0);
            s[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[buf][1][0], qh[s_], s[1], 0, 0, 0);
            s[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[buf][0][0], ql[s_], s[0], 0, 0, 0);
            s[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[buf][1][0], ql[s_], s[1], 0, 0, 0);
            s[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[buf][0][1], qh[s_], s[0], 0, 0, 0);
            s[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[buf][1][1], qh[s_], s[1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        float m_tile = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float4 q4 = scv[kb * 4 + (r >> 2)];
                const float scq = (r & 3) == 0 ? q4.x : (r & 3) == 1 ? q4.y : (r & 3) == 2 ? q4.z : q4.w;
                float v = scq * (s[kb][r] * inv_sqrt_c);
                v = (scq >= 0.0f) ? v : -INFINITY;
                s[kb][r] = v;
                m_tile = fmaxf(m_tile, v);
            }
        if (j0 + ATT_KT < j_end) fetch_sc(j0 + ATT_KT);
        m_tile = fmaxf(m_tile, __shfl_xor(m_tile, 32));
        const float m_new = fmaxf(m_run, m_tile);
        const float alpha = __expf(m_run - m_new);
        float l_tile = 0.0f;
        xhalf8 ph[2][2], pl[2][2];                  // [kb][t2]: keys crow(8*t2 + e, hi)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const float p0 = __expf(s[kb][r] - m_new);    // v_exp_f32 path: ~1e-6 relative, VALU issue is what bounds this kernel
                const float p1 = __expf(s[kb][r + 1] - m_new);
                l_tile += p0;
                l_tile += p1;
                unsigned uh, ul;
                split_pair(p0, p1, uh, ul);
                const xf16x2 h2 = __builtin_bit_cast(xf16x2, uh), l2 = __builtin_bit_cast(xf16x2, ul);
                ph[kb][r >> 3][r & 7] = h2[0]; ph[kb][r >> 3][(r & 7) + 1] = h2[1];
                pl[kb][r >> 3][r & 7] = l2[0]; pl[kb][r >> 3][(r & 7) + 1] = l2[1];
            }
        l_run = l_run * alpha + l_tile;
        m_run = m_new;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc_o[cb][r] *= alpha;
        // O^T += V^T P^T.  ONE fragment buffer, channel-block-major: the three MFMAs of block cb are followed by the read of cb's fragments
        // for the next octet pair, which then has the nine MFMAs of the other blocks to arrive (a second buffer costs 32 registers this
        // kernel does not have: every value beyond 256 is parked in AGPRs and copied back, ~100 v_accvgpr moves per tile)
        xhalf8 vf[CB][2];
        auto read_v1 = [&](int o, int cb) {
            const int oct = o * 2 + hi;
            vf[cb][0] = *reinterpret_cast<const xhalf8 *>(Vh + ((size_t)oct * C + cb * 32 + l31) * 8);
            vf[cb][1] = *reinterpret_cast<const xhalf8 *>(Vl + ((size_t)oct * C + cb * 32 + l31) * 8);
        };
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) read_v1(0, cb);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const int kb = o >> 1, t2 = o & 1;
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                acc_o[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[cb][0], ph[kb][t2], acc_o[cb], 0, 0, 0);
                acc_o[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[cb][0], pl[kb][t2], acc_o[cb], 0, 0, 0);
                acc_o[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[cb][1], ph[kb][t2], acc_o[cb], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (o + 1 < 4) read_v1(o + 1, cb);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const float l_all = l_run + __shfl_xor(l_run, 32);
    const float inv_l = 1.0f / l_all;
    float *mo = msg + ((size_t)b * n_cap + qrow) * C;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 v;
            v.x = acc_o[cb][4 * g + 0] * inv_l;
            v.y = acc_o[cb][4 * g + 1] * inv_l;
            v.z = acc_o[cb][4 * g + 2] * inv_l;
            v.w = acc_o[cb][4 * g + 3] * inv_l;
            *reinterpret_cast<float4 *>(mo + cb * 32 + 8 * g + 4 * hi) = v;
        }
}

// The same attention with EIGHT waves per workgroup (round 3): the two 32-key blocks of every 64-key tile go to two different waves of a
// query block (wave = query block w | key half kh), each with its own online softmax over its keys; the two partial results (O, m, l) of
// a query block are merged once at the end through LDS - the exact softmax algebra of pdsc_attention_merge_kernel.  A wave then holds ONE
// score block, half of the P / K / V fragments and SC values: 248 registers instead of 312 + AGPR parking, so two waves share a SIMD and
// one's softmax VALU runs under the other's MFMAs (the 4-wave kernel is a strict sequence of phases with one wave per SIMD: MFMA 43 %,
// softmax VALU 25 %, nothing overlapping).  Same K / V tile images, same DMA, one barrier per tile.
template <int C>
__global__ __launch_bounds__(512) void pdsc_attention_x3_img8_kernel(const float *__restrict__ QKV, const char *__restrict__ kv_img,
                                                                      const float *__restrict__ sc, const int32_t *__restrict__ n_rows,
                                                                      int n_cap, float inv_sqrt_c, float *__restrict__ msg, int n_pairs)
{
    static_assert(C == 128, "tile image geometry");
    constexpr int CB = C / 32;
    constexpr int NS = C / 16;
    extern __shared__ __attribute__((aligned(1024))) char att_lds[];
    const int lin = blockIdx.x + gridDim.x * blockIdx.z;
    const int b = (lin / 8 / (int)gridDim.x) * 8 + (lin & 7);
    const int qblk = (lin / 8) % (int)gridDim.x;
    if (b >= n_pairs) return;
    const int n = n_rows[b];
    const int q0 = qblk * ATT_Q;
    if (q0 >= n) return;
    const int t = threadIdx.x, lane = t & 63, wave8 = t >> 6, l31 = lane & 31, hi = lane >> 5;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave8);
    const int wave = wave8 & 3, kb = wave8 >> 2;          // query block of the workgroup, key half of every tile
    const int qrow = q0 + wave * 32 + l31;
    const float *base = QKV + (size_t)b * n_cap * 3 * C;
    const char *img = kv_img + (size_t)b * (n_cap / ATT_KT) * PDSC_KV_TILE_BYTES;
    const float4 *sc_q = reinterpret_cast<const float4 *>(sc) + (((size_t)b * (n_cap / 32) + (q0 / 32 + wave)) * (n_cap / ATT_KT)) * 8 * 64 + lane;
    const bool q_live = q0 + wave * 32 < n;
    float4 scv[4];
    auto dma_tile = [&](int j0, int buf) {
        const char *src = img + (size_t)(j0 / ATT_KT) * PDSC_KV_TILE_BYTES;
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const int piece = wave_u * 9 + j;
            if (piece < PDSC_KV_TILE_BYTES / 1024)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + piece * 1024 + lane * 16),
                                                 (__attribute__((address_space(3))) void *)(att_lds + buf * PDSC_KV_TILE_BYTES + piece * 1024), 16, 0, 0);
        }
    };
    auto fetch_sc = [&](int j0) {
        const float4 *sp = sc_q + (size_t)(j0 / ATT_KT) * 8 * 64 + (size_t)kb * 4 * 64;
#pragma unroll
        for (int v4 = 0; v4 < 4; ++v4) scv[v4] = q_live ? sp[(size_t)v4 * 64] : make_float4(-1.f, -1.f, -1.f, -1.f);
    };
    xhalf8 qh[NS], ql[NS];
    {
        const float4 *qv = reinterpret_cast<const float4 *>(base + (size_t)qrow * 3 * C);
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) {
            const float4 a = qv[4 * s_ + 2 * hi], c = qv[4 * s_ + 2 * hi + 1];
            const float x[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                _Float16 h_, l_;
                split_half(x[e], h_, l_);
                qh[s_][e] = h_;
                ql[s_][e] = l_;
            }
        }
    }
    f32x16 acc_o[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[cb][r] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;

    dma_tile(0, 0);
    fetch_sc(0);
    int buf = 0;
    for (int j0 = 0; j0 < n; j0 += ATT_KT, buf ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (j0 + ATT_KT < n) dma_tile(j0 + ATT_KT, buf ^ 1);
        const _Float16 *Kh = reinterpret_cast<const _Float16 *>(att_lds + buf * PDSC_KV_TILE_BYTES);
        const _Float16 *Kl = reinterpret_cast<const _Float16 *>(att_lds + buf * PDSC_KV_TILE_BYTES + PDSC_KV_KL);
        const _Float16 *Vh = reinterpret_cast<const _Float16 *>(att_lds + buf * PDSC_KV_TILE_BYTES + PDSC_KV_VH);
        const _Float16 *Vl = reinterpret_cast<const _Float16 *>(att_lds + buf * PDSC_KV_TILE_BYTES + PDSC_KV_VL);
        // S^T (this wave's 32 keys x 32 queries)
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.0f;
        xhalf8 kf[2][2];                              // [buffer][hi | lo]
        auto read_k = [&](int s_, int bf) {
            kf[bf][0] = *reinterpret_cast<const xhalf8 *>(Kh + pdsc_k_img_elem(kb * 32 + l31, 2 * s_ + hi));
            kf[bf][1] = *reinterpret_cast<const xhalf8 *>(Kl + pdsc_k_img_elem(kb * 32 + l31, 2 * s_ + hi));
        };
        read_k(0, 0);
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) {
            const int bf = s_ & 1;
            if (s_ + 1 < NS) read_k(s_ + 1, bf ^ 1);
            s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[bf][0], qh[s_], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[bf][0], ql[s_], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[bf][1], qh[s_], s, 0, 0, 0);
        }
        float m_tile = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float4 q4 = scv[r >> 2];
            const float scq = (r & 3) == 0 ? q4.x : (r & 3) == 1 ? q4.y : (r & 3) == 2 ? q4.z : q4.w;
            float v = scq * (s[r] * inv_sqrt_c);
            v = (scq >= 0.0f) ? v : -INFINITY;
            s[r] = v;
            m_tile = fmaxf(m_tile, v);
        }
        if (j0 + ATT_KT < n) fetch_sc(j0 + ATT_KT);
        m_tile = fmaxf(m_tile, __shfl_xor(m_tile, 32));
        const float m_new = fmaxf(m_run, m_tile);
        // a block whose keys are all masked so far keeps m = -inf: exp(-inf - (-inf)) must not produce NaN
        const float alpha = m_new == -INFINITY ? 1.0f : __expf(m_run - m_new);
        float l_tile = 0.0f;
        xhalf8 ph[2], pl[2];
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const float p0 = m_new == -INFINITY ? 0.0f : __expf(s[r] - m_new);
            const float p1 = m_new == -INFINITY ? 0.0f : __expf(s[r + 1] - m_new);
            l_tile += p0;
            l_tile += p1;
            unsigned uh, ul;
            split_pair(p0, p1, uh, ul);
            const xf16x2 h2 = __builtin_bit_cast(xf16x2, uh), l2 = __builtin_bit_cast(xf16x2, ul);
            ph[r >> 3][r & 7] = h2[0]; ph[r >> 3][(r & 7) + 1] = h2[1];
            pl[r >> 3][r & 7] = l2[0]; pl[r >> 3][(r & 7) + 1] = l2[1];
        }
        l_run = l_run * alpha + l_tile;
        m_run = m_new;
        if (__ballot(alpha != 1.0f) != 0ull) {               // the running maximum moved for some query of the wave
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc_o[cb][r] *= alpha;
        }
        // O^T += V^T P^T over this wave's two key octet pairs
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            const int oct = (kb * 2 + t2) * 2 + hi;
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                const xhalf8 vh = *reinterpret_cast<const xhalf8 *>(Vh + ((size_t)oct * C + cb * 32 + l31) * 8);
                const xhalf8 vl = *reinterpret_cast<const xhalf8 *>(Vl + ((size_t)oct * C + cb * 32 + l31) * 8);
                acc_o[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph[t2], acc_o[cb], 0, 0, 0);
                acc_o[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl[t2], acc_o[cb], 0, 0, 0);
                acc_o[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph[t2], acc_o[cb], 0, 0, 0);
            }
        }
    }
    // merge the two key halves of every query block: the kb = 1 waves hand (O, m, l) over through LDS (the tile buffers are free now)
    float l_all = l_run + __shfl_xor(l_run, 32);
    __syncthreads();
    float *xo = reinterpret_cast<float *>(att_lds) + (size_t)wave * (64 * (CB * 16 + 2));
    if (kb == 1) {
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) xo[(cb * 16 + r) * 64 + lane] = acc_o[cb][r];
        xo[(CB * 16) * 64 + lane] = m_run;
        xo[(CB * 16 + 1) * 64 + lane] = l_all;
    }
    __syncthreads();
    if (kb == 1) return;
    const float m_b = xo[(CB * 16) * 64 + lane], l_b = xo[(CB * 16 + 1) * 64 + lane];
    const float m = fmaxf(m_run, m_b);
    const float wa = m_run == -INFINITY ? 0.0f : __expf(m_run - m), wb = m_b == -INFINITY ? 0.0f : __expf(m_b - m);
    const float den = wa * l_all + wb * l_b;
    const float inv_l = den > 0.0f ? 1.0f / den : 0.0f;      // query rows of a dead 32-row block (every key masked): zeros
    float *mo = msg + ((size_t)b * n_cap + qrow) * C;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 v;
            v.x = (wa * acc_o[cb][4 * g + 0] + wb * xo[(cb * 16 + 4 * g + 0) * 64 + lane]) * inv_l;
            v.y = (wa * acc_o[cb][4 * g + 1] + wb * xo[(cb * 16 + 4 * g + 1) * 64 + lane]) * inv_l;
            v.z = (wa * acc_o[cb][4 * g + 2] + wb * xo[(cb * 16 + 4 * g + 2) * 64 + lane]) * inv_l;
            v.w = (wa * acc_o[cb][4 * g + 3] + wb * xo[(cb * 16 + 4 * g + 3) * 64 + lane]) * inv_l;
            *reinterpret_cast<float4 *>(mo + cb * 32 + 8 * g + 4 * hi) = v;
        }
}

// fp16x3 version of pdsc_linear_kernel (same tile, same epilogue) for K % 32 == 0: X and W tiles are split into hi/lo halves
// while they are staged (8-byte loads, 4-byte LDS stores; rows 80 bytes apart: the ds_read_b128 of a 16-lane group is
// conflict-free), 12 fp16 MFMAs per k-tile and wave instead of 32 fp32 ones.
template <bool RELU, bool RESID>
__global__ __launch_bounds__(256) void pdsc_linear_x3_kernel(const float *__restrict__ X, int ldx, size_t x_batch,
                                                              const float *__restrict__ W, const float *__restrict__ bias,
                                                              const float *__restrict__ R, int ldr, size_t r_batch,
                                                              float *__restrict__ Y, int ldy, size_t y_batch, int K, int N,
                                                              const int32_t *__restrict__ n_rows)
{
    constexpr int XLD = LIN_BK + 8;              // halves per LDS row
    __shared__ __attribute__((aligned(16))) _Float16 Xh[LIN_ROWS * XLD], Xl[LIN_ROWS * XLD];
    __shared__ __attribute__((aligned(16))) _Float16 Wh[LIN_COLS * XLD], Wl[LIN_COLS * XLD];
    const int b = blockIdx.z;
    const int m0 = blockIdx.x * LIN_ROWS, n0 = blockIdx.y * LIN_COLS;
    if (n_rows && m0 >= n_rows[b]) return;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, hi = lane >> 5;
    const float *x = X + (size_t)b * x_batch + (size_t)m0 * ldx;
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    constexpr int NX = (LIN_ROWS * LIN_BK) / 512, NW = (LIN_COLS * LIN_BK) / 512;     // float2 per thread: 4 + 8
    for (int k0 = 0; k0 < K; k0 += LIN_BK) {
        float2 xv[NX], wv[NW];
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int e = t + 256 * i, row = e >> 4, kk = (e & 15) * 2;
            xv[i] = *reinterpret_cast<const float2 *>(x + (size_t)row * ldx + k0 + kk);
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int e = t + 256 * i, col = e >> 4, kk = (e & 15) * 2;
            wv[i] = (n0 + col < N) ? *reinterpret_cast<const float2 *>(W + (size_t)(n0 + col) * K + k0 + kk) : make_float2(0.f, 0.f);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int e = t + 256 * i, row = e >> 4, kk = (e & 15) * 2;
            union { _Float16 h[2]; unsigned u; } ph, pl;
            split_half(xv[i].x, ph.h[0], pl.h[0]);
            split_half(xv[i].y, ph.h[1], pl.h[1]);
            *reinterpret_cast<unsigned *>(Xh + row * XLD + kk) = ph.u;
            *reinterpret_cast<unsigned *>(Xl + row * XLD + kk) = pl.u;
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int e = t + 256 * i, col = e >> 4, kk = (e & 15) * 2;
            union { _Float16 h[2]; unsigned u; } ph, pl;
            split_half(wv[i].x, ph.h[0], pl.h[0]);
            split_half(wv[i].y, ph.h[1], pl.h[1]);
            *reinterpret_cast<unsigned *>(Wh + col * XLD + kk) = ph.u;
            *reinterpret_cast<unsigned *>(Wl + col * XLD + kk) = pl.u;
        }
        __syncthreads();
#pragma unroll
        for (int s_ = 0; s_ < LIN_BK / 16; ++s_) {
            const int ko = 16 * s_ + 8 * hi;
            const xhalf8 ah = *reinterpret_cast<const xhalf8 *>(Xh + (wm * 32 + l31) * XLD + ko);
            const xhalf8 al = *reinterpret_cast<const xhalf8 *>(Xl + (wm * 32 + l31) * XLD + ko);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const xhalf8 bh = *reinterpret_cast<const xhalf8 *>(Wh + (wn * 64 + j * 32 + l31) * XLD + ko);
                const xhalf8 bl = *reinterpret_cast<const xhalf8 *>(Wl + (wn * 64 + j * 32 + l31) * XLD + ko);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[j], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = n0 + wn * 64 + j * 32 + l31;
        if (c >= N) continue;
        const float bv = bias ? bias[c] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * 32 + crow(r, hi);
            float v = acc[j][r] + bv;
            if (RELU) v = v > 0.0f ? v : 0.0f;
            if (RESID) v += R[(size_t)b * r_batch + (size_t)row * ldr + c];
            Y[(size_t)b * y_batch + (size_t)row * ldy + c] = v;
        }
    }
}

// fc_message of one NonLocal layer as ONE kernel (PointDSC.py:27-45: conv C->C/2 + BN + ReLU, conv C/2->C/2 + BN + ReLU, conv C/2->C,
// + the residual; C = 128): feat = feat1 + W3 relu(W2 relu(W1 msg + b1) + b2) + b3.  Three dependent launches of tiny GEMMs (29 us per layer:
// launch latency and five HBM round trips of [rows, 64..128] activations) become one.
// The chain runs TRANSPOSED, Y^T = W X^T: the weights are the MFMA's A operand (rows = output channels), the activations its B operand
// (columns = points), so an accumulator register holds (point = lane % 32, channel = crow(r, lane / 32)) - and the 8 registers 8j .. 8j+7
// of a lane are exactly one B-operand fragment of the NEXT layer, provided that layer's weights list their input channels in that register
// order.  oryon_pointdsc_finalize stores the three matrices that way (pre-split into fp16 hi / lo, rows swizzled: PDSC_MLP_* image), so the
// intermediate activations never leave the registers: bias + ReLU + split, next MFMA.  A wave owns 32 points and all channels; a workgroup
// (4 waves, 128 points) copies the 80 KB weight image into LDS once (LDS-DMA) while its waves fetch and split their input rows.
// A bias float4 at a WAVE-UNIFORM offset for lane half `hi` (channels off + 4 hi .. + 3): both halves come through the scalar cache
// (uniform address -> s_load, lgkmcnt) and the lane picks one.  Round 6: as vector loads (address + 16 hi) they sat on the vector-memory
// counter between the epilogue's stores, and the compiler's s_waitcnt vmcnt(0) in front of each use made every store of the per-point
// chain wait for the acknowledgement of the one before it - 5 to 8 us per 16 KB of output (the phase clocks of ORYON_PDSC_CLOCKS).
__device__ __forceinline__ float4 bias4(const float *__restrict__ b, int off, int hi)
{
    const float4 lo = *reinterpret_cast<const float4 *>(b + off), up = *reinterpret_cast<const float4 *>(b + off + 4);
    return hi ? up : lo;
}

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void pdsc_mlp3_x3_kernel(const float *__restrict__ msg, const float *__restrict__ resid,
                                                            const char *__restrict__ img, const float *__restrict__ b1,
                                                            const float *__restrict__ b2, const float *__restrict__ b3,
                                                            const int32_t *__restrict__ n_rows, int n_cap, float *__restrict__ out)
{
    constexpr int C = 128;
    extern __shared__ __attribute__((aligned(1024))) char mlp_lds[];
    const int b = blockIdx.y, q0 = blockIdx.x * (32 * WAVES);
    if (q0 >= n_rows[b]) return;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, hi = lane >> 5;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // weight image -> LDS: 80 pieces of 1 KB, 80 / WAVES per wave, lane-linear
    static_assert((PDSC_MLP_IMG_BYTES / 1024) % WAVES == 0, "pieces per wave");
#pragma unroll
    for (int j = 0; j < PDSC_MLP_IMG_BYTES / 1024 / WAVES; ++j) {
        const int piece = wave_u * (PDSC_MLP_IMG_BYTES / 1024 / WAVES) + j;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(img + piece * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void *)(mlp_lds + piece * 1024), 16, 0, 0);
    }
    // this lane's point: channels 16 s + 8 hi .. + 7 of k-step s, split into B-operand fragments
    const size_t prow = ((size_t)b * n_cap + q0 + wave * 32 + l31) * C;
    xhalf8 xh[8], xl[8];
    {
        const float4 *xp = reinterpret_cast<const float4 *>(msg + prow);
        float4 raw[16];
#pragma unroll
        for (int s_ = 0; s_ < 8; ++s_) { raw[2 * s_] = xp[4 * s_ + 2 * hi]; raw[2 * s_ + 1] = xp[4 * s_ + 2 * hi + 1]; }
#pragma unroll
        for (int s_ = 0; s_ < 8; ++s_) {
            uint4 uh, ul;
            split_pair(raw[2 * s_].x, raw[2 * s_].y, uh.x, ul.x);
            split_pair(raw[2 * s_].z, raw[2 * s_].w, uh.y, ul.y);
            split_pair(raw[2 * s_ + 1].x, raw[2 * s_ + 1].y, uh.z, ul.z);
            split_pair(raw[2 * s_ + 1].z, raw[2 * s_ + 1].w, uh.w, ul.w);
            xh[s_] = __builtin_bit_cast(xhalf8, uh);
            xl[s_] = __builtin_bit_cast(xhalf8, ul);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // A fragments: 256-byte rows (W1) slot ^ (row & 15), 128-byte rows (W2, W3) slot ^ ((row >> 1) & 7)
    auto w1_frag = [&](int base, int rb, int s_) {
        const int o = rb * 32 + l31;
        return *reinterpret_cast<const xhalf8 *>(mlp_lds + base + o * 256 + (((2 * s_ + hi) ^ (o & 15)) << 4));
    };
    auto w23_frag = [&](int base, int rb, int s_) {
        const int o = rb * 32 + l31;
        return *reinterpret_cast<const xhalf8 *>(mlp_lds + base + o * 128 + (((2 * s_ + hi) ^ ((o >> 1) & 7)) << 4));
    };
    // bias + ReLU + split of two 32-channel accumulator blocks -> the 4 B fragments (k-steps) of the next layer
    auto next_operand = [&](const f32x16 (&acc)[2], const float *bias, xhalf8 (&oh)[4], xhalf8 (&ol)[4]) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                uint4 uh, ul;
                unsigned *ph = &uh.x, *pl = &ul.x;
#pragma unroll
                for (int g2 = 0; g2 < 2; ++g2) {                 // registers 8 j + 4 g2 .. + 3: channels rb*32 + 8 (2 j + g2) + 4 hi + 0..3
                    const float4 bv = *reinterpret_cast<const float4 *>(bias + rb * 32 + 8 * (2 * j + g2) + 4 * hi);
                    const int r0 = 8 * j + 4 * g2;
                    const float v0 = fmaxf(acc[rb][r0] + bv.x, 0.0f), v1 = fmaxf(acc[rb][r0 + 1] + bv.y, 0.0f);
                    const float v2 = fmaxf(acc[rb][r0 + 2] + bv.z, 0.0f), v3 = fmaxf(acc[rb][r0 + 3] + bv.w, 0.0f);
                    split_pair(v0, v1, ph[2 * g2], pl[2 * g2]);
                    split_pair(v2, v3, ph[2 * g2 + 1], pl[2 * g2 + 1]);
                }
                oh[rb * 2 + j] = __builtin_bit_cast(xhalf8, uh);
                ol[rb * 2 + j] = __builtin_bit_cast(xhalf8, ul);
            }
    };
    // Two 32-channel output blocks (rb0, rb0 + 1) over NS k-steps: the four weight fragments of step s+1 are requested before the six
    // MFMAs of step s (left alone the compiler reads each fragment right before its first use: one exposed LDS latency per fragment).
    // The two accumulators alternate, per accumulator the order is hi*hi, hi*lo, lo*hi.
    auto two_blocks = [&](auto &&frag, int base_h, int base_l, int rb0, int NS, const xhalf8 *bh, const xhalf8 *bl, f32x16 (&acc)[2]) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
        xhalf8 w[2][2][2];                            // [buffer][block][hi | lo]
#pragma unroll
        for (int i = 0; i < 2; ++i) { w[0][i][0] = frag(base_h, rb0 + i, 0); w[0][i][1] = frag(base_l, rb0 + i, 0); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s_ = 0; s_ < 8; ++s_) {
            if (s_ < NS) {
                const int cur = s_ & 1;
                if (s_ + 1 < NS) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) { w[cur ^ 1][i][0] = frag(base_h, rb0 + i, s_ + 1); w[cur ^ 1][i][1] = frag(base_l, rb0 + i, s_ + 1); }
                }
                __builtin_amdgcn_sched_barrier(0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[cur][0][0], bh[s_], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[cur][1][0], bh[s_], acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[cur][0][0], bl[s_], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[cur][1][0], bl[s_], acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[cur][0][1], bh[s_], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[cur][1][1], bh[s_], acc[1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    // layer 1: [64, 128] x [128, points]
    f32x16 a1[2];
    two_blocks(w1_frag, PDSC_MLP_W1H, PDSC_MLP_W1L, 0, 8, xh, xl, a1);
    xhalf8 h1h[4], h1l[4];
    next_operand(a1, b1, h1h, h1l);
    // layer 2: [64, 64] x [64, points]
    f32x16 a2[2];
    two_blocks(w23_frag, PDSC_MLP_W2H, PDSC_MLP_W2L, 0, 4, h1h, h1l, a2);
    xhalf8 h2h[4], h2l[4];
    next_operand(a2, b2, h2h, h2l);
    // layer 3: [128, 64] x [64, points], + bias + residual, stored as float4 groups (channels rb*32 + 8 g + 4 hi + 0..3 of the lane's point)
#pragma unroll
    for (int rp = 0; rp < 2; ++rp) {
        float4 rv[2][4];                              // the residual rows travel under the MFMAs
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) rv[i][g] = *reinterpret_cast<const float4 *>(resid + prow + (2 * rp + i) * 32 + 8 * g + 4 * hi);
        f32x16 a3[2];
        two_blocks(w23_frag, PDSC_MLP_W3H, PDSC_MLP_W3L, 2 * rp, 4, h2h, h2l, a3);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = (2 * rp + i) * 32 + 8 * g + 4 * hi;
                const float4 bv = *reinterpret_cast<const float4 *>(b3 + c);
                float4 o;
                o.x = a3[i][4 * g + 0] + bv.x + rv[i][g].x;
                o.y = a3[i][4 * g + 1] + bv.y + rv[i][g].y;
                o.z = a3[i][4 * g + 2] + bv.z + rv[i][g].z;
                o.w = a3[i][4 * g + 3] + bv.w + rv[i][g].w;
                *reinterpret_cast<float4 *>(out + prow + c) = o;
            }
    }
}

// PointCN + q|k|v of one layer as ONE kernel, same transposed register chain as pdsc_mlp3_x3_kernel:
//   feat1 = relu(Wp feat + bp)  (stored: it is the residual of the layer's fc_message)      qkv = Wq feat1 + bq
// The weights do not fit LDS together (4 x 64 KB: PointCN, q, k, v - hi and lo halves of a 128 x 128 matrix each), so they stream through
// two 64 KB areas: PointCN | q are requested up front, k replaces PointCN once every wave has finished layer 1, v replaces q.
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void pdsc_pcn_qkv_x3_kernel(const float *__restrict__ feat, const char *__restrict__ img,
                                                               const float *__restrict__ bp, const float *__restrict__ bq,
                                                               const int32_t *__restrict__ n_rows, int n_cap, float *__restrict__ feat1,
                                                               float *__restrict__ qkv, char *__restrict__ kv_img, int g4)
{
    // g4 != 0: feat1 and q leave in the G4 row-fragment layout (pdsc.h) that pdsc_att_chain_x3_kernel reads and writes (layer 0 of the
    // one-launch-per-layer path); 0: [row][C] / [row][3 C] rows for the separate attention / fc_message kernels
    constexpr int C = 128, HALF = PDSC_PQ_CHUNK_BYTES / 2;
    extern __shared__ __attribute__((aligned(1024))) char pq_lds[];
    const int b = blockIdx.y, q0 = blockIdx.x * (32 * WAVES);
    if (q0 >= n_rows[b]) return;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, hi = lane >> 5;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto dma_chunk = [&](int chunk, int area) {                       // 64 pieces of 1 KB, 64 / WAVES per wave
#pragma unroll
        for (int j = 0; j < 64 / WAVES; ++j) {
            const int piece = wave_u * (64 / WAVES) + j;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(img + (size_t)chunk * PDSC_PQ_CHUNK_BYTES + piece * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void *)(pq_lds + area * PDSC_PQ_CHUNK_BYTES + piece * 1024), 16, 0, 0);
        }
    };
    dma_chunk(0, 0);
    dma_chunk(1, 1);
    const size_t prow = (size_t)b * n_cap + q0 + wave * 32 + l31;
    xhalf8 xh[8], xl[8];
    {
        const float4 *xp = reinterpret_cast<const float4 *>(feat + prow * C);
        float4 raw[16];
#pragma unroll
        for (int s_ = 0; s_ < 8; ++s_) { raw[2 * s_] = xp[4 * s_ + 2 * hi]; raw[2 * s_ + 1] = xp[4 * s_ + 2 * hi + 1]; }
#pragma unroll
        for (int s_ = 0; s_ < 8; ++s_) {
            uint4 uh, ul;
            split_pair(raw[2 * s_].x, raw[2 * s_].y, uh.x, ul.x);
            split_pair(raw[2 * s_].z, raw[2 * s_].w, uh.y, ul.y);
            split_pair(raw[2 * s_ + 1].x, raw[2 * s_ + 1].y, uh.z, ul.z);
            split_pair(raw[2 * s_ + 1].z, raw[2 * s_ + 1].w, uh.w, ul.w);
            xh[s_] = __builtin_bit_cast(xhalf8, uh);
            xl[s_] = __builtin_bit_cast(xhalf8, ul);
        }
    }
    // v's bias is per lane (lane = channel): fetched here, with the inputs, so that the wait below covers it - a load waited for between the
    // parts' stores would be waited for with vmcnt(0), i.e. behind every earlier store's acknowledgement
    float bias_v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bias_v[i] = bq[2 * C + i * 32 + l31];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" ::"v"(bias_v[0]), "v"(bias_v[1]), "v"(bias_v[2]), "v"(bias_v[3]));
    __syncthreads();
    auto frag = [&](int base, int rb, int s_) {
        const int o = rb * 32 + l31;
        return *reinterpret_cast<const xhalf8 *>(pq_lds + base + o * 256 + (((2 * s_ + hi) ^ (o & 15)) << 4));
    };
    // two 32-channel output blocks over the 8 k-steps, fragments of step s+1 requested before the MFMAs of step s (see pdsc_mlp3_x3_kernel)
    // swap = false: weights are the A operand (accumulator: lane = point, registers = channels); swap = true: the activations are (lane =
    // channel, registers = points crow(r, hi)) - the same fragments either way, the 32x32x16 A and B register layouts are mirror images
    auto two_blocks = [&](int area, int rb0, const xhalf8 *bh, const xhalf8 *bl, f32x16 (&acc)[2], bool swap) {
        const int base_h = area * PDSC_PQ_CHUNK_BYTES, base_l = base_h + HALF;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
        xhalf8 w[2][2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) { w[0][i][0] = frag(base_h, rb0 + i, 0); w[0][i][1] = frag(base_l, rb0 + i, 0); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s_ = 0; s_ < 8; ++s_) {
            const int cur = s_ & 1;
            if (s_ + 1 < 8) {
#pragma unroll
                for (int i = 0; i < 2; ++i) { w[cur ^ 1][i][0] = frag(base_h, rb0 + i, s_ + 1); w[cur ^ 1][i][1] = frag(base_l, rb0 + i, s_ + 1); }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!swap) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[cur][0][0], bh[s_], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[cur][1][0], bh[s_], acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[cur][0][0], bl[s_], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[cur][1][0], bl[s_], acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[cur][0][1], bh[s_], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[cur][1][1], bh[s_], acc[1], 0, 0, 0);
            } else {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[s_], w[cur][0][0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[s_], w[cur][1][0], acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[s_], w[cur][0][0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[s_], w[cur][1][0], acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[s_], w[cur][0][1], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[s_], w[cur][1][1], acc[1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // layer 1 (PointCN, area 0): bias + ReLU, stored as feat1 and kept as the 8 B fragments of layer 2
    xhalf8 fh[8], fl[8];
#pragma unroll
    for (int rp = 0; rp < 2; ++rp) {
        f32x16 acc[2];
        two_blocks(0, 2 * rp, xh, xl, acc, false);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int rb = 2 * rp + i;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                uint4 uh, ul;
                unsigned *ph = &uh.x, *pl = &ul.x;
#pragma unroll
                for (int g2 = 0; g2 < 2; ++g2) {
                    const int c = rb * 32 + 8 * (2 * j + g2) + 4 * hi, r0 = 8 * j + 4 * g2;
                    const float4 bv = bias4(bp, rb * 32 + 8 * (2 * j + g2), hi);
                    float4 v;
                    v.x = fmaxf(acc[i][r0] + bv.x, 0.0f); v.y = fmaxf(acc[i][r0 + 1] + bv.y, 0.0f);
                    v.z = fmaxf(acc[i][r0 + 2] + bv.z, 0.0f); v.w = fmaxf(acc[i][r0 + 3] + bv.w, 0.0f);
                    if (g4) reinterpret_cast<float4 *>(feat1 + (size_t)b * n_cap * C)[pdsc_g4_index(n_cap, q0 + wave * 32 + l31, c >> 2)] = v;
                    else *reinterpret_cast<float4 *>(feat1 + prow * C + c) = v;
                    split_pair(v.x, v.y, ph[2 * g2], pl[2 * g2]);
                    split_pair(v.z, v.w, ph[2 * g2 + 1], pl[2 * g2 + 1]);
                }
                fh[rb * 2 + j] = __builtin_bit_cast(xhalf8, uh);
                fl[rb * 2 + j] = __builtin_bit_cast(xhalf8, ul);
            }
        }
    }
    // q from area 1 while k lands in area 0, k from area 0 while v lands in area 1, v from area 1
#pragma unroll
    for (int part = 0; part < 3; ++part) {
        const int area = (part + 1) & 1;
        // this part's chunk has landed: it was requested one part ago, BEFORE that part's stores (16 float4 for q, 32 eight-byte pieces for k
        // with the tile image), and the vector-memory counter retires in order - a counted wait covers the DMA without waiting for the
        // stores' acknowledgements; a raw barrier (a __syncthreads() would fence with vmcnt(0)).  Round 6, see pdsc_att_chain_x3_kernel.
        if (part == 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        if (part == 2) { if (kv_img) asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                            // every wave is done with the other area; part's chunk visible
        asm volatile("" ::: "memory");
        if (part < 2) dma_chunk(part + 2, area ^ 1);
        // q (and k, v without an image buffer): fp32 rows of the q|k|v array.  With `kv_img`: k and v leave as the attention kernel's LDS
        // tile image, already split into fp16 hi / lo - k rows [key][136 halves] from the transposed product (lane = key: 8-byte pieces),
        // v from the un-transposed one (lane = channel, registers 8 t2 .. 8 t2 + 7 = the 8 keys of octet (kb, t2, hi): 16-byte pieces).
        const int p_pair = q0 + wave * 32;                           // first point of this wave inside its pair
        char *tile = kv_img ? kv_img + ((size_t)b * (n_cap / 64) + (p_pair >> 6)) * PDSC_KV_TILE_BYTES : nullptr;
        const bool as_v = kv_img && part == 2;
#pragma unroll
        for (int rp = 0; rp < 2; ++rp) {
            f32x16 acc[2];
            two_blocks(area, 2 * rp, fh, fl, acc, as_v);
            if (as_v) {
                const int kb = (p_pair >> 5) & 1;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int ch = (2 * rp + i) * 32 + l31;
                    const float bv = bias_v[2 * rp + i];
#pragma unroll
                    for (int t2 = 0; t2 < 2; ++t2) {
                        uint4 uh, ul;
                        split_pair(acc[i][8 * t2 + 0] + bv, acc[i][8 * t2 + 1] + bv, uh.x, ul.x);
                        split_pair(acc[i][8 * t2 + 2] + bv, acc[i][8 * t2 + 3] + bv, uh.y, ul.y);
                        split_pair(acc[i][8 * t2 + 4] + bv, acc[i][8 * t2 + 5] + bv, uh.z, ul.z);
                        split_pair(acc[i][8 * t2 + 6] + bv, acc[i][8 * t2 + 7] + bv, uh.w, ul.w);
                        const int oct = (kb * 2 + t2) * 2 + hi;
                        *reinterpret_cast<uint4 *>(tile + PDSC_KV_VH + ((size_t)oct * C + ch) * 16) = uh;
                        *reinterpret_cast<uint4 *>(tile + PDSC_KV_VL + ((size_t)oct * C + ch) * 16) = ul;
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int cc = (2 * rp + i) * 32 + 8 * g + 4 * hi, c = part * C + cc;
                        const float4 bv = bias4(bq, part * C + (2 * rp + i) * 32 + 8 * g, hi);
                        float4 o;
                        o.x = acc[i][4 * g + 0] + bv.x; o.y = acc[i][4 * g + 1] + bv.y;
                        o.z = acc[i][4 * g + 2] + bv.z; o.w = acc[i][4 * g + 3] + bv.w;
                        if (kv_img && part == 1) {
                            uint2 uh, ul;
                            split_pair(o.x, o.y, uh.x, ul.x);
                            split_pair(o.z, o.w, uh.y, ul.y);
                            const size_t off = (size_t)pdsc_k_img_elem((p_pair & 63) + l31, cc >> 3) * 2 + (cc & 7) * 2;
                            *reinterpret_cast<uint2 *>(tile + off) = uh;
                            *reinterpret_cast<uint2 *>(tile + PDSC_KV_KL + off) = ul;
                        } else if (g4 && part == 0) {
                            reinterpret_cast<float4 *>(qkv + (size_t)b * n_cap * 3 * C)[pdsc_g4_index(n_cap, q0 + wave * 32 + l31, cc >> 2)] = o;
                        } else {
                            *reinterpret_cast<float4 *>(qkv + prow * 3 * C + c) = o;
                        }
                    }
            }
        }
    }
}

// fc_message of layer l AND PointCN + q|k|v of layer l + 1 as ONE kernel (round 4): both are per-point chains on the same transposed
// register layout, so the layer output  feat = feat1 + W3 relu(W2 relu(W1 msg + b1) + b2) + b3  never leaves the registers - bias +
// residual + split and it IS the B operand of the next layer's PointCN, whose weights therefore come with their K axis in
// accumulator-register order (chunk 4 of the next layer's image; chunk 0 keeps the natural order for layer 0, whose input comes from
// memory).  One launch, one fetch of the rows and one write of the features less per layer (11 of the encoder's 36 launches).
// LDS: the 80 KB fc_message image at 0, PointCN' (64 KB) beside it from the start; q, k, v then stream through the two areas as in
// pdsc_pcn_qkv_x3_kernel (area 1 = offset 0 once the fc_message image is dead, area 0 = offset 80 KB).
// resid / feat1 may be the same buffer (a lane reads its residual row before it writes the row of the next layer): no __restrict__.
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void pdsc_mlp3_pcn_qkv_x3_kernel(const float *__restrict__ msg, const float *resid,
                                                                    const char *__restrict__ mlp_img, const float *__restrict__ b1,
                                                                    const float *__restrict__ b2, const float *__restrict__ b3,
                                                                    const char *__restrict__ pq_img, const float *__restrict__ bp,
                                                                    const float *__restrict__ bq, const int32_t *__restrict__ n_rows, int n_cap,
                                                                    float *feat1, float *__restrict__ qkv, char *__restrict__ kv_img)
{
    constexpr int C = 128, HALF = PDSC_PQ_CHUNK_BYTES / 2;
    constexpr int AREA0 = PDSC_MLP_IMG_BYTES, AREA1 = 0;               // byte offsets of the two 64 KB weight areas
    extern __shared__ __attribute__((aligned(1024))) char fz_lds[];
    const int b = blockIdx.y, q0 = blockIdx.x * (32 * WAVES);
    if (q0 >= n_rows[b]) return;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, hi = lane >> 5;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    static_assert((PDSC_MLP_IMG_BYTES / 1024) % WAVES == 0, "pieces per wave");
#pragma unroll
    for (int j = 0; j < PDSC_MLP_IMG_BYTES / 1024 / WAVES; ++j) {
        const int piece = wave_u * (PDSC_MLP_IMG_BYTES / 1024 / WAVES) + j;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(mlp_img + piece * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void *)(fz_lds + piece * 1024), 16, 0, 0);
    }
    auto dma_chunk = [&](int chunk, int area_off) {                   // 64 pieces of 1 KB, 64 / WAVES per wave
#pragma unroll
        for (int j = 0; j < 64 / WAVES; ++j) {
            const int piece = wave_u * (64 / WAVES) + j;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(pq_img + (size_t)chunk * PDSC_PQ_CHUNK_BYTES + piece * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void *)(fz_lds + area_off + piece * 1024), 16, 0, 0);
        }
    };
    dma_chunk(4, AREA0);                                               // PointCN with the permuted K axis
    const size_t prow = (size_t)b * n_cap + q0 + wave * 32 + l31;
    xhalf8 xh[8], xl[8];
    {
        const float4 *xp = reinterpret_cast<const float4 *>(msg + prow * C);
        float4 raw[16];
#pragma unroll
        for (int s_ = 0; s_ < 8; ++s_) { raw[2 * s_] = xp[4 * s_ + 2 * hi]; raw[2 * s_ + 1] = xp[4 * s_ + 2 * hi + 1]; }
#pragma unroll
        for (int s_ = 0; s_ < 8; ++s_) {
            uint4 uh, ul;
            split_pair(raw[2 * s_].x, raw[2 * s_].y, uh.x, ul.x);
            split_pair(raw[2 * s_].z, raw[2 * s_].w, uh.y, ul.y);
            split_pair(raw[2 * s_ + 1].x, raw[2 * s_ + 1].y, uh.z, ul.z);
            split_pair(raw[2 * s_ + 1].z, raw[2 * s_ + 1].w, uh.w, ul.w);
            xh[s_] = __builtin_bit_cast(xhalf8, uh);
            xl[s_] = __builtin_bit_cast(xhalf8, ul);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    auto w1_frag = [&](int base, int rb, int s_) {
        const int o = rb * 32 + l31;
        return *reinterpret_cast<const xhalf8 *>(fz_lds + base + o * 256 + (((2 * s_ + hi) ^ (o & 15)) << 4));
    };
    auto w23_frag = [&](int base, int rb, int s_) {
        const int o = rb * 32 + l31;
        return *reinterpret_cast<const xhalf8 *>(fz_lds + base + o * 128 + (((2 * s_ + hi) ^ ((o >> 1) & 7)) << 4));
    };
    auto next_operand = [&](const f32x16 (&acc)[2], const float *bias, xhalf8 (&oh)[4], xhalf8 (&ol)[4]) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                uint4 uh, ul;
                unsigned *ph = &uh.x, *pl = &ul.x;
#pragma unroll
                for (int g2 = 0; g2 < 2; ++g2) {
                    const float4 bv = *reinterpret_cast<const float4 *>(bias + rb * 32 + 8 * (2 * j + g2) + 4 * hi);
                    const int r0 = 8 * j + 4 * g2;
                    const float v0 = fmaxf(acc[rb][r0] + bv.x, 0.0f), v1 = fmaxf(acc[rb][r0 + 1] + bv.y, 0.0f);
                    const float v2 = fmaxf(acc[rb][r0 + 2] + bv.z, 0.0f), v3 = fmaxf(acc[rb][r0 + 3] + bv.w, 0.0f);
                    split_pair(v0, v1, ph[2 * g2], pl[2 * g2]);
                    split_pair(v2, v3, ph[2 * g2 + 1], pl[2 * g2 + 1]);
                }
                oh[rb * 2 + j] = __builtin_bit_cast(xhalf8, uh);
                ol[rb * 2 + j] = __builtin_bit_cast(xhalf8, ul);
            }
    };
    // two 32-row output blocks over NS k-steps (see pdsc_mlp3_x3_kernel); swap: the activations are the A operand (lane = channel)
    auto two_blocks = [&](auto &&frag, int base_h, int base_l, int rb0, int NS, const xhalf8 *bh, const xhalf8 *bl, f32x16 (&acc)[2], bool swap) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
        xhalf8 w[2][2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) { w[0][i][0] = frag(base_h, rb0 + i, 0); w[0][i][1] = frag(base_l, rb0 + i, 0); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s_ = 0; s_ < 8; ++s_) {
            if (s_ < NS) {
                const int cur = s_ & 1;
                if (s_ + 1 < NS) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) { w[cur ^ 1][i][0] = frag(base_h, rb0 + i, s_ + 1); w[cur ^ 1][i][1] = frag(base_l, rb0 + i, s_ + 1); }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (!swap) {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[cur][0][0], bh[s_], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[cur][1][0], bh[s_], acc[1], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[cur][0][0], bl[s_], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[cur][1][0], bl[s_], acc[1], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[cur][0][1], bh[s_], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[cur][1][1], bh[s_], acc[1], 0, 0, 0);
                } else {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[s_], w[cur][0][0], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[s_], w[cur][1][0], acc[1], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[s_], w[cur][0][0], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[s_], w[cur][1][0], acc[1], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[s_], w[cur][0][1], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[s_], w[cur][1][1], acc[1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    // ---- fc_message of layer l
    f32x16 a1[2];
    two_blocks(w1_frag, PDSC_MLP_W1H, PDSC_MLP_W1L, 0, 8, xh, xl, a1, false);
    xhalf8 h1h[4], h1l[4];
    next_operand(a1, b1, h1h, h1l);
    f32x16 a2[2];
    two_blocks(w23_frag, PDSC_MLP_W2H, PDSC_MLP_W2L, 0, 4, h1h, h1l, a2, false);
    xhalf8 h2h[4], h2l[4];
    next_operand(a2, b2, h2h, h2l);
    // layer 3 + bias + residual = the layer's output features, kept as the 8 B fragments (permuted K order) of the next layer's PointCN
#pragma unroll
    for (int rp = 0; rp < 2; ++rp) {
        float4 rv[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) rv[i][g] = *reinterpret_cast<const float4 *>(resid + prow * C + (2 * rp + i) * 32 + 8 * g + 4 * hi);
        f32x16 a3[2];
        two_blocks(w23_frag, PDSC_MLP_W3H, PDSC_MLP_W3L, 2 * rp, 4, h2h, h2l, a3, false);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int rb = 2 * rp + i;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                uint4 uh, ul;
                unsigned *ph = &uh.x, *pl = &ul.x;
#pragma unroll
                for (int g2 = 0; g2 < 2; ++g2) {
                    const int g = 2 * j + g2;
                    const float4 bv = *reinterpret_cast<const float4 *>(b3 + rb * 32 + 8 * g + 4 * hi);
                    const float v0 = a3[i][4 * g + 0] + bv.x + rv[i][g].x, v1 = a3[i][4 * g + 1] + bv.y + rv[i][g].y;
                    const float v2 = a3[i][4 * g + 2] + bv.z + rv[i][g].z, v3 = a3[i][4 * g + 3] + bv.w + rv[i][g].w;
                    split_pair(v0, v1, ph[2 * g2], pl[2 * g2]);
                    split_pair(v2, v3, ph[2 * g2 + 1], pl[2 * g2 + 1]);
                }
                xh[rb * 2 + j] = __builtin_bit_cast(xhalf8, uh);       // (the message fragments are dead: their registers take the features)
                xl[rb * 2 + j] = __builtin_bit_cast(xhalf8, ul);
            }
        }
    }
    // ---- PointCN + q|k|v of layer l + 1.  The fc_message image is dead once every wave is here: q lands on top of it.
    __syncthreads();
    dma_chunk(1, AREA1);
    auto frag = [&](int base, int rb, int s_) {
        const int o = rb * 32 + l31;
        return *reinterpret_cast<const xhalf8 *>(fz_lds + base + o * 256 + (((2 * s_ + hi) ^ (o & 15)) << 4));
    };
    xhalf8 fh[8], fl[8];
#pragma unroll
    for (int rp = 0; rp < 2; ++rp) {
        f32x16 acc[2];
        two_blocks(frag, AREA0, AREA0 + HALF, 2 * rp, 8, xh, xl, acc, false);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int rb = 2 * rp + i;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                uint4 uh, ul;
                unsigned *ph = &uh.x, *pl = &ul.x;
#pragma unroll
                for (int g2 = 0; g2 < 2; ++g2) {
                    const int c = rb * 32 + 8 * (2 * j + g2) + 4 * hi, r0 = 8 * j + 4 * g2;
                    const float4 bv = *reinterpret_cast<const float4 *>(bp + c);
                    float4 v;
                    v.x = fmaxf(acc[i][r0] + bv.x, 0.0f); v.y = fmaxf(acc[i][r0 + 1] + bv.y, 0.0f);
                    v.z = fmaxf(acc[i][r0 + 2] + bv.z, 0.0f); v.w = fmaxf(acc[i][r0 + 3] + bv.w, 0.0f);
                    *reinterpret_cast<float4 *>(feat1 + prow * C + c) = v;
                    split_pair(v.x, v.y, ph[2 * g2], pl[2 * g2]);
                    split_pair(v.z, v.w, ph[2 * g2 + 1], pl[2 * g2 + 1]);
                }
                fh[rb * 2 + j] = __builtin_bit_cast(xhalf8, uh);
                fl[rb * 2 + j] = __builtin_bit_cast(xhalf8, ul);
            }
        }
    }
    // q from area 1 while k lands in area 0, k from area 0 while v lands in area 1, v from area 1
#pragma unroll
    for (int part = 0; part < 3; ++part) {
        const int area_off = ((part + 1) & 1) ? AREA1 : AREA0, other_off = ((part + 1) & 1) ? AREA0 : AREA1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this part's chunk has landed (the feat1 stores drain with it)
        __syncthreads();                                                // every wave is done with the other area; this part's chunk is visible
        if (part < 2) dma_chunk(part + 2, other_off);
        const int p_pair = q0 + wave * 32;
        char *tile = kv_img ? kv_img + ((size_t)b * (n_cap / 64) + (p_pair >> 6)) * PDSC_KV_TILE_BYTES : nullptr;
        const bool as_v = kv_img && part == 2;
#pragma unroll
        for (int rp = 0; rp < 2; ++rp) {
            f32x16 acc[2];
            two_blocks(frag, area_off, area_off + HALF, 2 * rp, 8, fh, fl, acc, as_v);
            if (as_v) {
                const int kb = (p_pair >> 5) & 1;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int ch = (2 * rp + i) * 32 + l31;
                    const float bv = bq[2 * C + ch];
#pragma unroll
                    for (int t2 = 0; t2 < 2; ++t2) {
                        uint4 uh, ul;
                        split_pair(acc[i][8 * t2 + 0] + bv, acc[i][8 * t2 + 1] + bv, uh.x, ul.x);
                        split_pair(acc[i][8 * t2 + 2] + bv, acc[i][8 * t2 + 3] + bv, uh.y, ul.y);
                        split_pair(acc[i][8 * t2 + 4] + bv, acc[i][8 * t2 + 5] + bv, uh.z, ul.z);
                        split_pair(acc[i][8 * t2 + 6] + bv, acc[i][8 * t2 + 7] + bv, uh.w, ul.w);
                        const int oct = (kb * 2 + t2) * 2 + hi;
                        *reinterpret_cast<uint4 *>(tile + PDSC_KV_VH + ((size_t)oct * C + ch) * 16) = uh;
                        *reinterpret_cast<uint4 *>(tile + PDSC_KV_VL + ((size_t)oct * C + ch) * 16) = ul;
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int cc = (2 * rp + i) * 32 + 8 * g + 4 * hi, c = part * C + cc;
                        const float4 bv = *reinterpret_cast<const float4 *>(bq + c);
                        float4 o;
                        o.x = acc[i][4 * g + 0] + bv.x; o.y = acc[i][4 * g + 1] + bv.y;
                        o.z = acc[i][4 * g + 2] + bv.z; o.w = acc[i][4 * g + 3] + bv.w;
                        if (kv_img && part == 1) {
                            uint2 uh, ul;
                            split_pair(o.x, o.y, uh.x, ul.x);
                            split_pair(o.z, o.w, uh.y, ul.y);
                            const size_t off = (size_t)pdsc_k_img_elem((p_pair & 63) + l31, cc >> 3) * 2 + (cc & 7) * 2;
                            *reinterpret_cast<uint2 *>(tile + off) = uh;
                            *reinterpret_cast<uint2 *>(tile + PDSC_KV_KL + off) = ul;
                        } else {
                            *reinterpret_cast<float4 *>(qkv + prow * 3 * C + c) = o;
                        }
                    }
            }
        }
    }
}

// Attention + fc_message of layer l + PointCN + q|k|v of layer l + 1 as ONE kernel (round 4): the merged attention output of a query block
// already sits in the accumulator layout the per-point chain consumes (lane = point, registers = channels), so the message never goes to
// memory either: W1 comes with its K axis in accumulator-register order (second fc_message image).  The four key-half-0 waves run
// fc_message and PointCN for the workgroup's 128 points; q | k | v are shared by both waves of a query block (round 6); the key-half-1 waves
// request the q | k | v weight chunks (LDS-DMA).  HAS_NEXT = false (last layer): the chain stops after fc_message and writes the features.
// One launch per encoder layer instead of three.
// Round 6 (DESIGN.md "PointDSC encoder: what round 6 found"; phase clocks: DBG / ORYON_PDSC_CLOCKS in the development build):
//   * nothing on the vector-memory counter between the chain's stores: biases in LDS (bias4l), v's per-lane bias and the residual rows
//     fetched before the first store, chunk DMA by waves that wait with counted vmcnt, raw s_barrier between the parts - vmcnt is ONE
//     in-order counter, and a vector load between stores makes every store wait for the acknowledgement of the one before it;
//   * every store whole 128-byte lines: K tile image [channel octet][key][8 halves], q and the PointCN output in the G4 row-fragment layout
//     (pdsc.h) - the store path costs ~4 cycles per line a store instruction touches;
//   * q / resid / feat1 in G4, so this kernel's only partners on those arrays are itself and pdsc_pcn_qkv_x3_kernel<8>(g4 = 1) for layer 0.
template <int C, bool HAS_NEXT, bool DBG = false>
__global__ __launch_bounds__(512) void pdsc_att_chain_x3_kernel(const float *__restrict__ QKV, const char *__restrict__ kv_img_in,
                                                                 const float *__restrict__ sc, const int32_t *__restrict__ n_rows, int n_cap,
                                                                 float inv_sqrt_c, int n_pairs, const float *resid, const char *__restrict__ mlp_img,
                                                                 const float *__restrict__ b1, const float *__restrict__ b2,
                                                                 const float *__restrict__ b3, const char *__restrict__ pq_img,
                                                                 const float *__restrict__ bp, const float *__restrict__ bq, float *feat1,
                                                                 float *qkv, char *kv_img, float *__restrict__ feat_out, long long *dbg_clk = nullptr)
{
    constexpr int WAVES = 8, HALF = PDSC_PQ_CHUNK_BYTES / 2;
    // development aid (ORYON_PDSC_CLOCKS, dev build only): phase timestamps (100 MHz wall clock) of waves 0 (key half 0) and 4 (key half 1) of workgroup 0
#define CLK(i) do { if constexpr (DBG) { if (blockIdx.x == 0 && blockIdx.z == 0 && (threadIdx.x & 255) == 0) dbg_clk[(threadIdx.x >> 8) * 16 + (i)] = wall_clock64(); } } while (0)
    CLK(0);
    constexpr int AREA0 = PDSC_MLP_IMG_BYTES, AREA1 = 0;               // byte offsets of the two 64 KB weight areas (see pdsc_mlp3_pcn_qkv_x3_kernel)
    const char *kv_img_rd = kv_img_in;
#define kv_img kv_img_rd
    static_assert(C == 128, "tile image geometry");
    constexpr int CB = C / 32;
    constexpr int NS = C / 16;
    extern __shared__ __attribute__((aligned(1024))) char att_lds[];
    // the chain's biases live in LDS (3 KB behind the merge area, filled under the attention tiles): b1 | b2 | b3 | bp | bq as FOUR planes of
    // 192 floats (plane e = element e of every channel quad), so that a lane's quad is four broadcasting ds_read_b32 - neither the
    // vector-memory counter its stores sit on nor a cold scalar cache in front of every epilogue (round 6, see bias4)
    float *lds_bias = reinterpret_cast<float *>(att_lds + PDSC_AC_BIAS_OFF);
    constexpr int SEG_B1 = 0, SEG_B2 = 64, SEG_B3 = 128, SEG_BP = 256, SEG_BQ = 384;
    auto bias4l = [&](int seg_off, int hi_) {                       // seg_off: wave-uniform multiple of 8; lane half hi_ -> + 4 channels
        const int q = (seg_off >> 2) + hi_;
        return make_float4(lds_bias[q], lds_bias[192 + q], lds_bias[384 + q], lds_bias[576 + q]);
    };
    const int lin = blockIdx.x + gridDim.x * blockIdx.z;
    const int b = (lin / 8 / (int)gridDim.x) * 8 + (lin & 7);
    const int qblk = (lin / 8) % (int)gridDim.x;
    if (b >= n_pairs) return;
    const int n = n_rows[b];
    const int q0 = qblk * ATT_Q;
    if (q0 >= n) return;
    const int t = threadIdx.x, lane = t & 63, wave8 = t >> 6, l31 = lane & 31, hi = lane >> 5;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave8);
    const int wave = wave8 & 3, kb = wave8 >> 2;          // query block of the workgroup, key half of every tile
    for (int i = t; i < (HAS_NEXT ? 768 : 256); i += 512) {
        const float v = i < SEG_B2 ? b1[i] : i < SEG_B3 ? b2[i - SEG_B2] : i < SEG_BP ? b3[i - SEG_B3] : i < SEG_BQ ? bp[i - SEG_BP] : bq[i - SEG_BQ];
        lds_bias[(i & 3) * 192 + (i >> 2)] = v;                // (visible behind the first barrier of the tile loop)
    }
    const int qrow = q0 + wave * 32 + l31;
    const float *base = QKV + (size_t)b * n_cap * 3 * C;
    const char *img = kv_img + (size_t)b * (n_cap / ATT_KT) * PDSC_KV_TILE_BYTES;
    const float4 *sc_q = reinterpret_cast<const float4 *>(sc) + (((size_t)b * (n_cap / 32) + (q0 / 32 + wave)) * (n_cap / ATT_KT)) * 8 * 64 + lane;
    const bool q_live = q0 + wave * 32 < n;
    float4 scv[4];
    auto dma_tile = [&](int j0, int buf) {
        const char *src = img + (size_t)(j0 / ATT_KT) * PDSC_KV_TILE_BYTES;
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const int piece = wave_u * 9 + j;
            if (piece < PDSC_KV_TILE_BYTES / 1024)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + piece * 1024 + lane * 16),
                                                 (__attribute__((address_space(3))) void *)(att_lds + buf * PDSC_KV_TILE_BYTES + piece * 1024), 16, 0, 0);
        }
    };
    auto fetch_sc = [&](int j0) {
        const float4 *sp = sc_q + (size_t)(j0 / ATT_KT) * 8 * 64 + (size_t)kb * 4 * 64;
#pragma unroll
        for (int v4 = 0; v4 < 4; ++v4) scv[v4] = q_live ? sp[(size_t)v4 * 64] : make_float4(-1.f, -1.f, -1.f, -1.f);
    };
    // round 6: the first tile's DMA and SC rows are requested BEFORE the query rows are split (they used to be issued behind the ~300
    // instructions of the split, their latency exposed in front of the first tile)
    dma_tile(0, 0);
    fetch_sc(0);
    xhalf8 qh[NS], ql[NS];
    {
        // q in the G4 row-fragment layout (pdsc.h): the 8 channels of k-step s_ for lane half hi are the quads 4 s_ + 2 hi, + 1
        const float4 *qv = reinterpret_cast<const float4 *>(base);
        // the query rows carry the scores' constant factor: 1 / sqrt(C) and, since the exponentials below are exp2, log2(e) - one fp32
        // multiplication per query value here instead of two per score in every tile (softmax(SC * s / sqrt(C)) is unchanged:
        // exp(x) = exp2(x log2 e); 64 registrations 0.901 -> 0.891 ms on one box)
        const float qs = inv_sqrt_c * 1.44269504088896340736f;
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) {
            const float4 a = qv[pdsc_g4_index(n_cap, qrow, 4 * s_ + 2 * hi)], c = qv[pdsc_g4_index(n_cap, qrow, 4 * s_ + 2 * hi + 1)];
            uint4 uh, ul;
            split_pair(a.x * qs, a.y * qs, uh.x, ul.x);
            split_pair(a.z * qs, a.w * qs, uh.y, ul.y);
            split_pair(c.x * qs, c.y * qs, uh.z, ul.z);
            split_pair(c.z * qs, c.w * qs, uh.w, ul.w);
            qh[s_] = __builtin_bit_cast(xhalf8, uh);
            ql[s_] = __builtin_bit_cast(xhalf8, ul);
        }
    }
    f32x16 acc_o[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[cb][r] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;

    CLK(1);
    // every query row of this wave exists: its tiles whose 64 keys all exist need no masking (SC >= 0 there, scores finite)
    const bool q_full = q0 + wave * 32 + 32 <= n;
    int buf = 0;
    for (int j0 = 0; j0 < n; j0 += ATT_KT, buf ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (j0 + ATT_KT < n) dma_tile(j0 + ATT_KT, buf ^ 1);
        const _Float16 *Kh = reinterpret_cast<const _Float16 *>(att_lds + buf * PDSC_KV_TILE_BYTES);
        const _Float16 *Kl = reinterpret_cast<const _Float16 *>(att_lds + buf * PDSC_KV_TILE_BYTES + PDSC_KV_KL);
        const _Float16 *Vh = reinterpret_cast<const _Float16 *>(att_lds + buf * PDSC_KV_TILE_BYTES + PDSC_KV_VH);
        const _Float16 *Vl = reinterpret_cast<const _Float16 *>(att_lds + buf * PDSC_KV_TILE_BYTES + PDSC_KV_VL);
        // S^T (this wave's 32 keys x 32 queries)
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.0f;
        xhalf8 kf[2][2];                              // [buffer][hi | lo]
        auto read_k = [&](int s_, int bf) {
            kf[bf][0] = *reinterpret_cast<const xhalf8 *>(Kh + pdsc_k_img_elem(kb * 32 + l31, 2 * s_ + hi));
            kf[bf][1] = *reinterpret_cast<const xhalf8 *>(Kl + pdsc_k_img_elem(kb * 32 + l31, 2 * s_ + hi));
        };
        read_k(0, 0);
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) {
            const int bf = s_ & 1;
            if (s_ + 1 < NS) read_k(s_ + 1, bf ^ 1);
            s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[bf][0], qh[s_], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[bf][0], ql[s_], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[bf][1], qh[s_], s, 0, 0, 0);
        }
        float m_tile = -INFINITY;
        xhalf8 ph[2], pl[2];
        if (q_full && j0 + ATT_KT <= n) {
            // ---- full tile (round 6): every (query, key) of it exists - no mask, no -inf guards (m_new is finite; exp(-inf - m) = 0 on the
            // first tile).  Same arithmetic per element as the masked path below.
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float4 q4 = scv[r >> 2];
                const float scq = (r & 3) == 0 ? q4.x : (r & 3) == 1 ? q4.y : (r & 3) == 2 ? q4.z : q4.w;
                const float v = scq * s[r];
                s[r] = v;
                m_tile = fmaxf(m_tile, v);
            }
            if (j0 + ATT_KT < n) fetch_sc(j0 + ATT_KT);
            m_tile = fmaxf(m_tile, __shfl_xor(m_tile, 32));
            const float m_new = fmaxf(m_run, m_tile);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            float l_tile = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const float p0 = __builtin_amdgcn_exp2f(s[r] - m_new);
                const float p1 = __builtin_amdgcn_exp2f(s[r + 1] - m_new);
                l_tile += p0;
                l_tile += p1;
                unsigned uh, ul;
                split_pair(p0, p1, uh, ul);
                const xf16x2 h2 = __builtin_bit_cast(xf16x2, uh), l2 = __builtin_bit_cast(xf16x2, ul);
                ph[r >> 3][r & 7] = h2[0]; ph[r >> 3][(r & 7) + 1] = h2[1];
                pl[r >> 3][r & 7] = l2[0]; pl[r >> 3][(r & 7) + 1] = l2[1];
            }
            l_run = l_run * alpha + l_tile;
            m_run = m_new;
            if (__ballot(alpha != 1.0f) != 0ull) {
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc_o[cb][r] *= alpha;
            }
        } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float4 q4 = scv[r >> 2];
            const float scq = (r & 3) == 0 ? q4.x : (r & 3) == 1 ? q4.y : (r & 3) == 2 ? q4.z : q4.w;
            float v = scq * s[r];
            v = (scq >= 0.0f) ? v : -INFINITY;
            s[r] = v;
            m_tile = fmaxf(m_tile, v);
        }
        if (j0 + ATT_KT < n) fetch_sc(j0 + ATT_KT);
        m_tile = fmaxf(m_tile, __shfl_xor(m_tile, 32));
        const float m_new = fmaxf(m_run, m_tile);
        // a block whose keys are all masked so far keeps m = -inf: exp(-inf - (-inf)) must not produce NaN
        const float alpha = m_new == -INFINITY ? 1.0f : __builtin_amdgcn_exp2f(m_run - m_new);
        float l_tile = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const float p0 = m_new == -INFINITY ? 0.0f : __builtin_amdgcn_exp2f(s[r] - m_new);
            const float p1 = m_new == -INFINITY ? 0.0f : __builtin_amdgcn_exp2f(s[r + 1] - m_new);
            l_tile += p0;
            l_tile += p1;
            unsigned uh, ul;
            split_pair(p0, p1, uh, ul);
            const xf16x2 h2 = __builtin_bit_cast(xf16x2, uh), l2 = __builtin_bit_cast(xf16x2, ul);
            ph[r >> 3][r & 7] = h2[0]; ph[r >> 3][(r & 7) + 1] = h2[1];
            pl[r >> 3][r & 7] = l2[0]; pl[r >> 3][(r & 7) + 1] = l2[1];
        }
        l_run = l_run * alpha + l_tile;
        m_run = m_new;
        if (__ballot(alpha != 1.0f) != 0ull) {               // the running maximum moved for some query of the wave
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc_o[cb][r] *= alpha;
        }
        }
        // O^T += V^T P^T over this wave's two key octet pairs
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            const int oct = (kb * 2 + t2) * 2 + hi;
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                const xhalf8 vh = *reinterpret_cast<const xhalf8 *>(Vh + ((size_t)oct * C + cb * 32 + l31) * 8);
                const xhalf8 vl = *reinterpret_cast<const xhalf8 *>(Vl + ((size_t)oct * C + cb * 32 + l31) * 8);
                acc_o[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph[t2], acc_o[cb], 0, 0, 0);
                acc_o[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl[t2], acc_o[cb], 0, 0, 0);
                acc_o[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph[t2], acc_o[cb], 0, 0, 0);
            }
        }
    }
#undef kv_img
    // ---- the tiles are done: the fc_message image starts landing at [0, 80 KB) while the two key halves merge through [80 KB, 146 KB)
    CLK(2);
    float l_all = l_run + __shfl_xor(l_run, 32);
    __syncthreads();
    static_assert((PDSC_MLP_IMG_BYTES / 1024) % WAVES == 0, "pieces per wave");
#pragma unroll
    for (int j = 0; j < PDSC_MLP_IMG_BYTES / 1024 / WAVES; ++j) {
        const int piece = wave_u * (PDSC_MLP_IMG_BYTES / 1024 / WAVES) + j;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(mlp_img + piece * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void *)(att_lds + piece * 1024), 16, 0, 0);
    }
    float *xo = reinterpret_cast<float *>(att_lds + PDSC_MLP_IMG_BYTES) + (size_t)wave * (64 * (CB * 16 + 2));
    if (kb == 1) {
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) xo[(cb * 16 + r) * 64 + lane] = acc_o[cb][r];
        xo[(CB * 16) * 64 + lane] = m_run;
        xo[(CB * 16 + 1) * 64 + lane] = l_all;
    }
    __syncthreads();
    const bool live = kb == 0;
    const size_t prow = (size_t)b * n_cap + qrow;
    xhalf8 xh[8], xl[8];                                              // the message as the chain's B fragments (k-step 2 cb + j: registers 8 j .. 8 j + 7 of block cb)
    if (live) {
        const float m_b = xo[(CB * 16) * 64 + lane], l_b = xo[(CB * 16 + 1) * 64 + lane];
        const float m = fmaxf(m_run, m_b);
        const float wa = m_run == -INFINITY ? 0.0f : __builtin_amdgcn_exp2f(m_run - m), wb = m_b == -INFINITY ? 0.0f : __builtin_amdgcn_exp2f(m_b - m);
        const float den = wa * l_all + wb * l_b;
        const float inv_l = den > 0.0f ? 1.0f / den : 0.0f;      // query rows of a dead 32-row block (every key masked): zeros
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                uint4 uh, ul;
                unsigned *ph = &uh.x, *pl = &ul.x;
#pragma unroll
                for (int g2 = 0; g2 < 2; ++g2) {
                    const int r0 = 8 * j + 4 * g2;
                    const float v0 = (wa * acc_o[cb][r0 + 0] + wb * xo[(cb * 16 + r0 + 0) * 64 + lane]) * inv_l;
                    const float v1 = (wa * acc_o[cb][r0 + 1] + wb * xo[(cb * 16 + r0 + 1) * 64 + lane]) * inv_l;
                    const float v2 = (wa * acc_o[cb][r0 + 2] + wb * xo[(cb * 16 + r0 + 2) * 64 + lane]) * inv_l;
                    const float v3 = (wa * acc_o[cb][r0 + 3] + wb * xo[(cb * 16 + r0 + 3) * 64 + lane]) * inv_l;
                    split_pair(v0, v1, ph[2 * g2], pl[2 * g2]);
                    split_pair(v2, v3, ph[2 * g2 + 1], pl[2 * g2 + 1]);
                }
                xh[cb * 2 + j] = __builtin_bit_cast(xhalf8, uh);
                xl[cb * 2 + j] = __builtin_bit_cast(xhalf8, ul);
            }
    }
    CLK(3);
    __syncthreads();                                                    // the merge area is free: PointCN' lands there
    auto dma_chunk = [&](int chunk, int area_off) {                   // 64 pieces of 1 KB, 64 / WAVES per wave
#pragma unroll
        for (int j = 0; j < 64 / WAVES; ++j) {
            const int piece = wave_u * (64 / WAVES) + j;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(pq_img + (size_t)chunk * PDSC_PQ_CHUNK_BYTES + piece * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void *)(att_lds + area_off + piece * 1024), 16, 0, 0);
        }
    };
    // The q | k | v chunks are requested by the four key-half-1 waves alone (16 pieces each): they have no stores, so their
    // s_waitcnt vmcnt(0) waits for the LDS-DMA only, and the four live waves - whose feat1 / q / k / v stores sit on the same in-order
    // counter - never wait on it between the parts: their stores drain behind the next part's MFMAs (round 6; with every wave requesting
    // an eighth and waiting vmcnt(0), each part paid the acknowledgement of its 16 KB of stores)
    auto dma_chunk_idle = [&](int chunk, int area_off) {
        if (kb == 1) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int piece = (wave_u - 4) * 16 + j;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(pq_img + (size_t)chunk * PDSC_PQ_CHUNK_BYTES + piece * 1024 + lane * 16),
                                                 (__attribute__((address_space(3))) void *)(att_lds + area_off + piece * 1024), 16, 0, 0);
            }
        }
    };
    if constexpr (HAS_NEXT) dma_chunk(4, AREA0);                       // PointCN with the permuted K axis
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    auto w1_frag = [&](int base, int rb, int s_) {
        const int o = rb * 32 + l31;
        return *reinterpret_cast<const xhalf8 *>(att_lds + base + o * 256 + (((2 * s_ + hi) ^ (o & 15)) << 4));
    };
    auto w23_frag = [&](int base, int rb, int s_) {
        const int o = rb * 32 + l31;
        return *reinterpret_cast<const xhalf8 *>(att_lds + base + o * 128 + (((2 * s_ + hi) ^ ((o >> 1) & 7)) << 4));
    };
    auto next_operand = [&](const f32x16 (&acc)[2], int bias_seg, xhalf8 (&oh)[4], xhalf8 (&ol)[4]) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                uint4 uh, ul;
                unsigned *ph = &uh.x, *pl = &ul.x;
#pragma unroll
                for (int g2 = 0; g2 < 2; ++g2) {
                    const float4 bv = bias4l(bias_seg + rb * 32 + 8 * (2 * j + g2), hi);
                    const int r0 = 8 * j + 4 * g2;
                    const float v0 = fmaxf(acc[rb][r0] + bv.x, 0.0f), v1 = fmaxf(acc[rb][r0 + 1] + bv.y, 0.0f);
                    const float v2 = fmaxf(acc[rb][r0 + 2] + bv.z, 0.0f), v3 = fmaxf(acc[rb][r0 + 3] + bv.w, 0.0f);
                    split_pair(v0, v1, ph[2 * g2], pl[2 * g2]);
                    split_pair(v2, v3, ph[2 * g2 + 1], pl[2 * g2 + 1]);
                }
                oh[rb * 2 + j] = __builtin_bit_cast(xhalf8, uh);
                ol[rb * 2 + j] = __builtin_bit_cast(xhalf8, ul);
            }
    };
    // two 32-row output blocks over NS k-steps (see pdsc_mlp3_x3_kernel); swap: the activations are the A operand (lane = channel)
    auto two_blocks = [&](auto &&frag, int base_h, int base_l, int rb0, int NS, const xhalf8 *bh, const xhalf8 *bl, f32x16 (&acc)[2], bool swap) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
        xhalf8 w[2][2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) { w[0][i][0] = frag(base_h, rb0 + i, 0); w[0][i][1] = frag(base_l, rb0 + i, 0); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s_ = 0; s_ < 8; ++s_) {
            if (s_ < NS) {
                const int cur = s_ & 1;
                if (s_ + 1 < NS) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) { w[cur ^ 1][i][0] = frag(base_h, rb0 + i, s_ + 1); w[cur ^ 1][i][1] = frag(base_l, rb0 + i, s_ + 1); }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (!swap) {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[cur][0][0], bh[s_], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[cur][1][0], bh[s_], acc[1], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[cur][0][0], bl[s_], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[cur][1][0], bl[s_], acc[1], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[cur][0][1], bh[s_], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[cur][1][1], bh[s_], acc[1], 0, 0, 0);
                } else {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[s_], w[cur][0][0], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[s_], w[cur][1][0], acc[1], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[s_], w[cur][0][0], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[s_], w[cur][1][0], acc[1], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[s_], w[cur][0][1], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[s_], w[cur][1][1], acc[1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    CLK(4);
    // v's bias is per lane (lane = channel in the swapped product): fetched here, long before the first store of the chain - a vector load
    // issued between stores waits for every earlier store's acknowledgement (s_waitcnt vmcnt is one in-order counter)
    float bias_v[2] = {0.0f, 0.0f};                                  // (round 6: a wave computes the row blocks 2 kb, 2 kb + 1 of q | k | v)
    if constexpr (HAS_NEXT) {
#pragma unroll
        for (int i = 0; i < 2; ++i) bias_v[i] = bq[2 * C + (2 * kb + i) * 32 + l31];
    }
    if (live) {
    // the residual rows (this layer's PointCN output, 32 floats per lane) are requested before W1: behind the W3 blocks, where they used to
    // be issued, a wave alone on its SIMD waited out their latency twice
    float4 rv[2][2][4];
#pragma unroll
    for (int rp = 0; rp < 2; ++rp)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                rv[rp][i][g] = reinterpret_cast<const float4 *>(resid + (size_t)b * n_cap * C)[pdsc_g4_index(n_cap, qrow, (2 * rp + i) * 8 + 2 * g + hi)];
    // ---- fc_message of layer l
    f32x16 a1[2];
    two_blocks(w1_frag, PDSC_MLP_W1H, PDSC_MLP_W1L, 0, 8, xh, xl, a1, false);
    xhalf8 h1h[4], h1l[4];
    next_operand(a1, SEG_B1, h1h, h1l);
    CLK(5);
    f32x16 a2[2];
    two_blocks(w23_frag, PDSC_MLP_W2H, PDSC_MLP_W2L, 0, 4, h1h, h1l, a2, false);
    xhalf8 h2h[4], h2l[4];
    next_operand(a2, SEG_B2, h2h, h2l);
    CLK(6);
    // layer 3 + bias + residual = the layer's output features, kept as the 8 B fragments (permuted K order) of the next layer's PointCN
#pragma unroll
    for (int rp = 0; rp < 2; ++rp) {
        f32x16 a3[2];
        two_blocks(w23_frag, PDSC_MLP_W3H, PDSC_MLP_W3L, 2 * rp, 4, h2h, h2l, a3, false);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int rb = 2 * rp + i;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                uint4 uh, ul;
                unsigned *ph = &uh.x, *pl = &ul.x;
#pragma unroll
                for (int g2 = 0; g2 < 2; ++g2) {
                    const int g = 2 * j + g2;
                    const float4 bv = bias4l(SEG_B3 + rb * 32 + 8 * g, hi);
                    const float v0 = a3[i][4 * g + 0] + bv.x + rv[rp][i][g].x, v1 = a3[i][4 * g + 1] + bv.y + rv[rp][i][g].y;
                    const float v2 = a3[i][4 * g + 2] + bv.z + rv[rp][i][g].z, v3 = a3[i][4 * g + 3] + bv.w + rv[rp][i][g].w;
                    split_pair(v0, v1, ph[2 * g2], pl[2 * g2]);
                    split_pair(v2, v3, ph[2 * g2 + 1], pl[2 * g2 + 1]);
                }
                xh[rb * 2 + j] = __builtin_bit_cast(xhalf8, uh);       // (the message fragments are dead: their registers take the features)
                xl[rb * 2 + j] = __builtin_bit_cast(xhalf8, ul);
            }
            if constexpr (!HAS_NEXT) {                                 // last layer: the features leave as fp32 rows
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = rb * 32 + 8 * g + 4 * hi;
                    const float4 bv = bias4l(SEG_B3 + rb * 32 + 8 * g, hi);
                    float4 o;
                    o.x = a3[i][4 * g + 0] + bv.x + rv[rp][i][g].x; o.y = a3[i][4 * g + 1] + bv.y + rv[rp][i][g].y;
                    o.z = a3[i][4 * g + 2] + bv.z + rv[rp][i][g].z; o.w = a3[i][4 * g + 3] + bv.w + rv[rp][i][g].w;
                    if (live) *reinterpret_cast<float4 *>(feat_out + prow * C + c) = o;
                }
            }
        }
    }
    }
    CLK(7);
    if constexpr (!HAS_NEXT) return;
    // v's bias registers are "used" here, before the chain's first store: the compiler waits for their loads at the first use, and behind
    // this kernel's hand-written counted waits it can only do so with vmcnt(0) - in the v part that meant every wave waiting for the
    // acknowledgement of its k stores
    asm volatile("" ::"v"(bias_v[0]), "v"(bias_v[1]));
    // ---- PointCN + q|k|v of layer l + 1.  The fc_message image is dead once every wave is here: q lands on top of it.
    __syncthreads();
    dma_chunk_idle(1, AREA1);
    auto frag = [&](int base, int rb, int s_) {
        const int o = rb * 32 + l31;
        return *reinterpret_cast<const xhalf8 *>(att_lds + base + o * 256 + (((2 * s_ + hi) ^ (o & 15)) << 4));
    };
    xhalf8 fh[8], fl[8];
    if (live) {
#pragma unroll
    for (int rp = 0; rp < 2; ++rp) {
        f32x16 acc[2];
        two_blocks(frag, AREA0, AREA0 + HALF, 2 * rp, 8, xh, xl, acc, false);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int rb = 2 * rp + i;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                uint4 uh, ul;
                unsigned *ph = &uh.x, *pl = &ul.x;
#pragma unroll
                for (int g2 = 0; g2 < 2; ++g2) {
                    const int c = rb * 32 + 8 * (2 * j + g2) + 4 * hi, r0 = 8 * j + 4 * g2;
                    const float4 bv = bias4l(SEG_BP + rb * 32 + 8 * (2 * j + g2), hi);
                    float4 v;
                    v.x = fmaxf(acc[i][r0] + bv.x, 0.0f); v.y = fmaxf(acc[i][r0 + 1] + bv.y, 0.0f);
                    v.z = fmaxf(acc[i][r0 + 2] + bv.z, 0.0f); v.w = fmaxf(acc[i][r0 + 3] + bv.w, 0.0f);
                    if (live) reinterpret_cast<float4 *>(feat1 + (size_t)b * n_cap * C)[pdsc_g4_index(n_cap, qrow, c >> 2)] = v;
                    split_pair(v.x, v.y, ph[2 * g2], pl[2 * g2]);
                    split_pair(v.z, v.w, ph[2 * g2 + 1], pl[2 * g2 + 1]);
                }
                fh[rb * 2 + j] = __builtin_bit_cast(xhalf8, uh);
                fl[rb * 2 + j] = __builtin_bit_cast(xhalf8, ul);
            }
        }
    }
    }
    // q from area 1 while k lands in area 0, k from area 0 while v lands in area 1, v from area 1.
    // Round 6: BOTH waves of a query block compute q | k | v - wave (qb, kb) the output row blocks 2 kb, 2 kb + 1 of each part - so the
    // 288 of the chain's 504 MFMAs and their stores run at two waves per SIMD.  The PointCN output (the B operand: 16 fragments per lane)
    // goes from the wave that made it to its partner through the query block's 16 KB slice of the dead PointCN weight area.
    CLK(8);
    CLK(13);
    if (kb == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the q chunk (requested by these waves) has landed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                       // ... and every wave is done with the PointCN weights (area 0)
    asm volatile("" ::: "memory");
    {
        xhalf8 *ex = reinterpret_cast<xhalf8 *>(att_lds + AREA0 + wave * 16384);
        if (live) {
#pragma unroll
            for (int f = 0; f < 8; ++f) { ex[f * 64 + lane] = fh[f]; ex[(8 + f) * 64 + lane] = fl[f]; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (!live) {
#pragma unroll
            for (int f = 0; f < 8; ++f) { fh[f] = ex[f * 64 + lane]; fl[f] = ex[(8 + f) * 64 + lane]; }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the slice is read: this wave's own DMA pieces may overwrite it
        }
    }
    CLK(9);
#pragma unroll
    for (int part = 0; part < 3; ++part) {
        const int area_off = ((part + 1) & 1) ? AREA1 : AREA0, other_off = ((part + 1) & 1) ? AREA0 : AREA1;
        if (part > 0) {
            CLK(13 + part);
            // this part's chunk has landed: the key-half-1 waves requested it BEFORE their previous part's stores (>= 8 of them), and the
            // vector-memory counter retires in order - vmcnt(8) covers the DMA without waiting for the stores' acknowledgements; the other
            // waves never wait on it.  Raw barrier (a __syncthreads() would fence with vmcnt(0)).
            // (behind the q part: its 8 float4 stores; behind the k part: its 16 eight-byte stores - one instruction each, none mergeable)
            if (kb == 1) { if (part == 1 || !kv_img) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                               // every wave is done with the other area; this part's chunk is visible
            asm volatile("" ::: "memory");
            CLK(9 + part);
        }
        if (part < 2) dma_chunk_idle(part + 2, other_off);
        const int p_pair = q0 + wave * 32;                           // (wave = query block 0..3 here)
        char *tile = kv_img ? kv_img + ((size_t)b * (n_cap / 64) + (p_pair >> 6)) * PDSC_KV_TILE_BYTES : nullptr;
        const bool as_v = kv_img && part == 2;
        {
        const int rp = kb;
        {
            f32x16 acc[2];
            two_blocks(frag, area_off, area_off + HALF, 2 * rp, 8, fh, fl, acc, as_v);
            if (as_v) {
                const int kb = (p_pair >> 5) & 1;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int ch = (2 * rp + i) * 32 + l31;
                    const float bv = bias_v[i];
#pragma unroll
                    for (int t2 = 0; t2 < 2; ++t2) {
                        uint4 uh, ul;
                        split_pair(acc[i][8 * t2 + 0] + bv, acc[i][8 * t2 + 1] + bv, uh.x, ul.x);
                        split_pair(acc[i][8 * t2 + 2] + bv, acc[i][8 * t2 + 3] + bv, uh.y, ul.y);
                        split_pair(acc[i][8 * t2 + 4] + bv, acc[i][8 * t2 + 5] + bv, uh.z, ul.z);
                        split_pair(acc[i][8 * t2 + 6] + bv, acc[i][8 * t2 + 7] + bv, uh.w, ul.w);
                        const int oct = (kb * 2 + t2) * 2 + hi;
                        *reinterpret_cast<uint4 *>(tile + PDSC_KV_VH + ((size_t)oct * C + ch) * 16) = uh;
                        *reinterpret_cast<uint4 *>(tile + PDSC_KV_VL + ((size_t)oct * C + ch) * 16) = ul;
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int cc = (2 * rp + i) * 32 + 8 * g + 4 * hi, c = part * C + cc;
                        const float4 bv = bias4l(SEG_BQ + part * C + (2 * rp + i) * 32 + 8 * g, hi);
                        float4 o;
                        o.x = acc[i][4 * g + 0] + bv.x; o.y = acc[i][4 * g + 1] + bv.y;
                        o.z = acc[i][4 * g + 2] + bv.z; o.w = acc[i][4 * g + 3] + bv.w;
                        if (kv_img && part == 1) {
                            uint2 uh, ul;
                            split_pair(o.x, o.y, uh.x, ul.x);
                            split_pair(o.z, o.w, uh.y, ul.y);
                            const size_t off = (size_t)pdsc_k_img_elem((p_pair & 63) + l31, cc >> 3) * 2 + (cc & 7) * 2;
                            *reinterpret_cast<uint2 *>(tile + off) = uh;
                            *reinterpret_cast<uint2 *>(tile + PDSC_KV_KL + off) = ul;
                        } else if (part == 0) {
                            reinterpret_cast<float4 *>(qkv + (size_t)b * n_cap * 3 * C)[pdsc_g4_index(n_cap, qrow, cc >> 2)] = o;
                        } else {
                            *reinterpret_cast<float4 *>(qkv + prow * 3 * C + c) = o;       // (no image buffer: never on this path)
                        }
                    }
            }
        }
        }
    }
    CLK(12);
#undef CLK
}

// Combine the key-split partials: msg = sum_s e^{m_s - m} O_s / sum_s e^{m_s - m} l_s,  m = max_s m_s.
__global__ __launch_bounds__(256) void pdsc_attention_merge_kernel(const float *__restrict__ part_o, const float *__restrict__ part_ml,
                                                                    const int32_t *__restrict__ n_rows, int n_cap, int C, int KS,
                                                                    int B, float *__restrict__ msg)
{
    const int b = blockIdx.y;
    const int n_live = (n_rows[b] + ATT_Q - 1) / ATT_Q * ATT_Q;          // rows the attention kernel produced
    const int c4n = C / 4;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n_live * c4n; e += gridDim.x * blockDim.x) {
        const int row = e / c4n, c4 = e % c4n;
        float m = -INFINITY;
        for (int s = 0; s < KS; ++s) m = fmaxf(m, part_ml[(((size_t)s * B + b) * n_cap + row) * 2]);
        float L = 0.0f;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int s = 0; s < KS; ++s) {
            const size_t prow = ((size_t)s * B + b) * n_cap + row;
            const float ms = part_ml[prow * 2];
            if (ms == -INFINITY) continue;
            const float w = __expf(ms - m);
            L += w * part_ml[prow * 2 + 1];
            const float4 v = *reinterpret_cast<const float4 *>(part_o + prow * C + 4 * c4);
            o.x += w * v.x; o.y += w * v.y; o.z += w * v.z; o.w += w * v.w;
        }
        const float inv = 1.0f / L;
        o.x *= inv; o.y *= inv; o.z *= inv; o.w *= inv;
        *reinterpret_cast<float4 *>(msg + ((size_t)b * n_cap + row) * C + 4 * c4) = o;
    }
}

// F.normalize(feat, p=2, dim=-1) (PointDSC.py:156): one wave per row.
__global__ __launch_bounds__(256) void pdsc_normalise_kernel(const float *__restrict__ feat, int C, int n_cap,
                                                              const int32_t *__restrict__ n_rows, float *__restrict__ out)
{
    const int b = blockIdx.y;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= n_cap) return;
    const float *f = feat + ((size_t)b * n_cap + row) * C;
    float *o = out + ((size_t)b * n_cap + row) * C;
    float s = 0.0f;
    for (int c = lane; c < C; c += 64) s += f[c] * f[c];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    float d = sqrt_rn(s);
    d = d < 1e-12f ? 1e-12f : d;
    const bool live = row < n_rows[b];
    for (int c = lane; c < C; c += 64) o[c] = live ? f[c] / d : 0.0f;
}

// Confidence head + feature normalisation as ONE launch (round 5; C = 128): conf = W3 relu(W2 relu(W1 feat + b1) + b2) + b3
// (PointDSC.py:107-113, 128 -> 32 -> 32 -> 1) and feat_n = F.normalize(feat) (PointDSC.py:156) were three pdsc_linear_x3_kernel launches
// and one pdsc_normalise_kernel launch, 30 us + three launch gaps for 17 MB of reads.  One workgroup keeps a 64-row tile's two hidden
// activations in LDS.  Every output element is accumulated by exactly the MFMA sequence pdsc_linear_x3_kernel runs for it (same splits,
// same k order: k-tiles of 32, two 16-steps each, hi*hi then hi*lo then lo*hi) and the normalisation is pdsc_normalise_kernel's loop
// verbatim, so the results are bit-identical to the four launches (ORYON_PDSC_FUSED_HEAD = 0 in the development build runs those).
constexpr int HEAD_H = 32;       // hidden width of the confidence head
__global__ __launch_bounds__(256) void pdsc_head_x3_kernel(const float *__restrict__ feat, const float *__restrict__ W1, const float *__restrict__ b1,
                                                            const float *__restrict__ W2, const float *__restrict__ b2,
                                                            const float *__restrict__ W3, const float *__restrict__ b3,
                                                            const int32_t *__restrict__ n_rows, int n_cap, float *__restrict__ conf,
                                                            float *__restrict__ feat_n)
{
    constexpr int C = 128, XLD = LIN_BK + 8;
    __shared__ __attribute__((aligned(16))) _Float16 Xh[LIN_ROWS * XLD], Xl[LIN_ROWS * XLD];
    __shared__ __attribute__((aligned(16))) _Float16 Wh[HEAD_H * XLD], Wl[HEAD_H * XLD];
    __shared__ float Hs[LIN_ROWS * HEAD_H];
    const int b = blockIdx.y, m0 = blockIdx.x * LIN_ROWS;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, hi = lane >> 5;
    const int nr = n_rows[b];
    const float *x = feat + ((size_t)b * n_cap + m0) * C;
    if (m0 < nr) {                                                   // the linear kernels skip tiles without a live row
        // one layer of the head: X rows from global (ldx floats apart) or from Hs, W [n_out, K] row-major, result (+ bias, ReLU) into Hs
        // or, for the last layer, column 0 into conf.  Waves 0 / 1 own the tile's two 32-row blocks; all four stage.
        auto layer = [&](const float *xg, int ldx, bool from_lds, const float *W, const float *bias, int K, int n_out, bool relu, bool last) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            for (int k0 = 0; k0 < K; k0 += LIN_BK) {
                float2 xv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int e = t + 256 * i, row = e >> 4, kk = (e & 15) * 2;
                    xv[i] = from_lds ? *reinterpret_cast<const float2 *>(Hs + row * HEAD_H + kk)
                                     : *reinterpret_cast<const float2 *>(xg + (size_t)row * ldx + k0 + kk);
                }
                float2 wv[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int e = t + 256 * i, col = e >> 4, kk = (e & 15) * 2;
                    wv[i] = col < n_out ? *reinterpret_cast<const float2 *>(W + (size_t)col * K + k0 + kk) : make_float2(0.f, 0.f);
                }
                __syncthreads();
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int e = t + 256 * i, row = e >> 4, kk = (e & 15) * 2;
                    union { _Float16 h[2]; unsigned u; } ph, pl;
                    split_half(xv[i].x, ph.h[0], pl.h[0]);
                    split_half(xv[i].y, ph.h[1], pl.h[1]);
                    *reinterpret_cast<unsigned *>(Xh + row * XLD + kk) = ph.u;
                    *reinterpret_cast<unsigned *>(Xl + row * XLD + kk) = pl.u;
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int e = t + 256 * i, col = e >> 4, kk = (e & 15) * 2;
                    union { _Float16 h[2]; unsigned u; } ph, pl;
                    split_half(wv[i].x, ph.h[0], pl.h[0]);
                    split_half(wv[i].y, ph.h[1], pl.h[1]);
                    *reinterpret_cast<unsigned *>(Wh + col * XLD + kk) = ph.u;
                    *reinterpret_cast<unsigned *>(Wl + col * XLD + kk) = pl.u;
                }
                __syncthreads();
                if (wave < 2) {
#pragma unroll
                    for (int s_ = 0; s_ < LIN_BK / 16; ++s_) {
                        const int ko = 16 * s_ + 8 * hi;
                        const xhalf8 ah = *reinterpret_cast<const xhalf8 *>(Xh + (wave * 32 + l31) * XLD + ko);
                        const xhalf8 al = *reinterpret_cast<const xhalf8 *>(Xl + (wave * 32 + l31) * XLD + ko);
                        const xhalf8 bh = *reinterpret_cast<const xhalf8 *>(Wh + l31 * XLD + ko);
                        const xhalf8 bl = *reinterpret_cast<const xhalf8 *>(Wl + l31 * XLD + ko);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
                    }
                }
            }
            __syncthreads();                                          // Hs (the previous layer's output) has been read by everyone
            if (wave < 2 && l31 < n_out) {
                const float bv = bias ? bias[l31] : 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wave * 32 + crow(r, hi);
                    float v = acc[r] + bv;
                    if (relu) v = v > 0.0f ? v : 0.0f;
                    if (last) conf[(size_t)b * n_cap + m0 + row] = v;
                    else Hs[row * HEAD_H + l31] = v;
                }
            }
            __syncthreads();
        };
        layer(x, C, false, W1, b1, C, HEAD_H, true, false);
        layer(nullptr, 0, true, W2, b2, HEAD_H, HEAD_H, true, false);
        layer(nullptr, 0, true, W3, b3, HEAD_H, 1, false, true);
    }
    // F.normalize of the tile's rows: pdsc_normalise_kernel, one wave per row
    for (int rr = wave; rr < LIN_ROWS; rr += 4) {
        const int row = m0 + rr;
        const float *f = feat + ((size_t)b * n_cap + row) * C;
        float *o = feat_n + ((size_t)b * n_cap + row) * C;
        float s = 0.0f;
        for (int c = lane; c < C; c += 64) s += f[c] * f[c];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        float d = sqrt_rn(s);
        d = d < 1e-12f ? 1e-12f : d;
        const bool live = row < nr;
        for (int c = lane; c < C; c += 64) o[c] = live ? f[c] / d : 0.0f;
    }
}

// ------------------------------------------------------------------------------------------------ host side
static int launch_linear(bool relu, bool resid, const float *X, int ldx, size_t xb, const float *W, const float *bias,
                         const float *R, int ldr, size_t rb, float *Y, int ldy, size_t yb, int K, int N, int B, int n_cap,
                         const int32_t *n_rows, hipStream_t st)
{
    dim3 grid(n_cap / LIN_ROWS, ceil_div(N, LIN_COLS), B), block(256);
    static const bool x3 = !dev_env_set("ORYON_PDSC_FP32_MFMA");
    if (x3 && K % LIN_BK == 0 && ldx % 2 == 0) {
        if (relu && !resid) hipLaunchKernelGGL((pdsc_linear_x3_kernel<true, false>), grid, block, 0, st, X, ldx, xb, W, bias, R, ldr, rb, Y, ldy, yb, K, N, n_rows);
        else if (!relu && resid) hipLaunchKernelGGL((pdsc_linear_x3_kernel<false, true>), grid, block, 0, st, X, ldx, xb, W, bias, R, ldr, rb, Y, ldy, yb, K, N, n_rows);
        else if (!relu && !resid) hipLaunchKernelGGL((pdsc_linear_x3_kernel<false, false>), grid, block, 0, st, X, ldx, xb, W, bias, R, ldr, rb, Y, ldy, yb, K, N, n_rows);
        else hipLaunchKernelGGL((pdsc_linear_x3_kernel<true, true>), grid, block, 0, st, X, ldx, xb, W, bias, R, ldr, rb, Y, ldy, yb, K, N, n_rows);
        return hipGetLastError() == hipSuccess ? ORYON_OK : ORYON_ERR_HIP;
    }
    if (relu && !resid)
        hipLaunchKernelGGL((pdsc_linear_kernel<true, false>), grid, block, 0, st, X, ldx, xb, W, bias, R, ldr, rb, Y, ldy, yb, K, N, n_rows);
    else if (!relu && resid)
        hipLaunchKernelGGL((pdsc_linear_kernel<false, true>), grid, block, 0, st, X, ldx, xb, W, bias, R, ldr, rb, Y, ldy, yb, K, N, n_rows);
    else if (!relu && !resid)
        hipLaunchKernelGGL((pdsc_linear_kernel<false, false>), grid, block, 0, st, X, ldx, xb, W, bias, R, ldr, rb, Y, ldy, yb, K, N, n_rows);
    else
        hipLaunchKernelGGL((pdsc_linear_kernel<true, true>), grid, block, 0, st, X, ldx, xb, W, bias, R, ldr, rb, Y, ldy, yb, K, N, n_rows);
    return hipGetLastError() == hipSuccess ? ORYON_OK : ORYON_ERR_HIP;
}

void pdsc_launch_normalise(const float *feat, int C, int n_cap, int B, const int32_t *n_rows, float *out, hipStream_t st)
{
    hipLaunchKernelGGL(pdsc_normalise_kernel, dim3(n_cap / 4, B), dim3(256), 0, st, feat, C, n_cap, n_rows, out);
}

// Runs the whole encoder: features [B,n_cap,C] (un-normalised) into ws.feat, confidence into ws.conf.
int pdsc_run_encoder(const PdscModel &M, const PdscWorkspace &ws, const float *src, const float *tgt, const int32_t *n_rows,
                     int B, int n_cap, hipStream_t st)
{
    const int C = M.cfg.num_channels, H = C / 2;
    const size_t fb = (size_t)n_cap * C, qb = (size_t)n_cap * 3 * C, hb = (size_t)n_cap * H;
    hipLaunchKernelGGL(pdsc_center_kernel, dim3(B), dim3(256), 0, st, src, tgt, n_rows, n_cap, ws.corr_pos);
    // layer0: conv 6 -> C (PointDSC.py:72)
    int rc = launch_linear(false, false, ws.corr_pos, 8, (size_t)n_cap * 8, M.w0, M.b0, nullptr, 0, 0, ws.feat, C, fb, M.cfg.in_dim, C,
                           B, n_cap, n_rows, st);
    if (rc) return rc;
    const float inv_sigma2 = 1.0f / (M.sigma_d * M.sigma_d);
    const float inv_sqrt_c = 1.0f / sqrtf((float)C);
    // spatial-consistency tiles, shared by all layers
    hipLaunchKernelGGL(pdsc_sc_kernel, dim3(n_cap / 128, n_cap / ATT_KT, B), dim3(256), 0, st, src, tgt, n_rows, n_cap, inv_sigma2, ws.sc);
    for (int l = 0; l < M.cfg.num_layers; ++l) {
        const PdscLayer &L = M.layers[l];
        static const bool x3 = !dev_env_set("ORYON_PDSC_FP32_MFMA");     // fp16x3 unless the pure-fp32 kernels are asked for
        static const bool fused_pq = dev_env_int("ORYON_PDSC_FUSED_PQ", 1) != 0;       // dev: 0 = two launches
        static const bool att_img = dev_env_int("ORYON_PDSC_ATT_IMG", 1) != 0;           // dev: 0 = fp32 K / V
        bool use_img = false;
        // round 4: fc_message of layer l - 1 and PointCN + q|k|v of layer l came as ONE launch at the end of the previous iteration
        static const bool fuse_chain = dev_env_int("ORYON_PDSC_FUSED_CHAIN", 1) != 0;     // dev: 0 = separate launches
        static const bool fuse_att = dev_env_int("ORYON_PDSC_FUSED_ATT", 1) != 0;           // dev: 0 = attention as its own launch
        const bool chain_ok = C == 128 && x3 && fused_pq && fuse_chain && n_cap % 256 == 0 && ws.att_splits == 1 && ws.kv_img != nullptr && att_img &&
                              (dev_env_int("ORYON_PDSC_FUSED_MLP", 1) != 0) &&
                              (dev_env_int("ORYON_PDSC_WAVES", 8) == 8);
        // K / V images alternate between two buffers when the attention is part of the one-launch-per-layer kernel (its workgroups write
        // layer l + 1's tiles while others still read layer l's)
        // (every layer or none: the kernel hands q and the PointCN output from layer to layer in its own G4 layout)
        bool all_imgs = true;
        for (const PdscLayer &Lx : M.layers) all_imgs = all_imgs && Lx.mlp_img_p && Lx.mlp_img && Lx.pq_img;
        const bool att_chain = chain_ok && fuse_att && ws.kv_img2 != nullptr && all_imgs &&
                               (dev_env_int("ORYON_PDSC_ATT8", 1) != 0);
        char *kv_cur = (att_chain && (l & 1)) ? ws.kv_img2 : ws.kv_img, *kv_nxt = (att_chain && !(l & 1)) ? ws.kv_img2 : ws.kv_img;
        if (l > 0 && chain_ok && L.pq_img && M.layers[l - 1].mlp_img) {
            use_img = true;                                            // (launched at the end of the previous iteration)
        } else
        if (C == 128 && x3 && fused_pq && L.pq_img) {
            // PointCN (conv + BN + ReLU, BN folded) and the q | k | v projections in one launch
            // 8-wave workgroups (256 points): the same 128 KB of weights feed twice the points and the launch occupies half the CUs with
            // two waves per SIMD - latency-bound kernels lose nothing, and K0 / the other registration stream find free CUs beside them
            static const int pw = dev_env_int("ORYON_PDSC_WAVES", 8);
            const bool w8 = pw == 8 && n_cap % 256 == 0;
            if (w8) allow_dynamic_lds(reinterpret_cast<const void *>(pdsc_pcn_qkv_x3_kernel<8>), 2 * PDSC_PQ_CHUNK_BYTES);
            else allow_dynamic_lds(reinterpret_cast<const void *>(pdsc_pcn_qkv_x3_kernel<4>), 2 * PDSC_PQ_CHUNK_BYTES);
            use_img = ws.att_splits == 1 && ws.kv_img != nullptr && att_img;
            if (w8)
                hipLaunchKernelGGL(pdsc_pcn_qkv_x3_kernel<8>, dim3(n_cap / 256, B), dim3(512), 2 * PDSC_PQ_CHUNK_BYTES, st, ws.feat, L.pq_img, L.b_pcn,
                                   L.b_qkv, n_rows, n_cap, ws.feat1, ws.qkv, use_img ? ws.kv_img : nullptr, (att_chain && use_img) ? 1 : 0);
            else
                hipLaunchKernelGGL(pdsc_pcn_qkv_x3_kernel<4>, dim3(n_cap / 128, B), dim3(256), 2 * PDSC_PQ_CHUNK_BYTES, st, ws.feat, L.pq_img, L.b_pcn,
                                   L.b_qkv, n_rows, n_cap, ws.feat1, ws.qkv, use_img ? ws.kv_img : nullptr, 0);
            if (hipGetLastError() != hipSuccess) return ORYON_ERR_HIP;
        } else {
            // PointCN: conv + BN + ReLU (BN folded)
            rc = launch_linear(true, false, ws.feat, C, fb, L.w_pcn, L.b_pcn, nullptr, 0, 0, ws.feat1, C, fb, C, C, B, n_cap, n_rows, st);
            if (rc) return rc;
            // q | k | v projections as one GEMM with N = 3C
            rc = launch_linear(false, false, ws.feat1, C, fb, L.w_qkv, L.b_qkv, nullptr, 0, 0, ws.qkv, 3 * C, qb, C, 3 * C, B, n_cap, n_rows, st);
            if (rc) return rc;
        }
        if (att_chain && use_img) {
            // attention + fc_message (+ PointCN + q|k|v of the next layer) in one launch
            constexpr int AC_LDS = PDSC_AC_BIAS_OFF + PDSC_AC_BIAS_BYTES;        // tile buffers / weight areas / merge area, then the biases
            static_assert(PDSC_AC_BIAS_OFF >= 2 * PDSC_KV_TILE_BYTES && AC_LDS <= 160 * 1024, "att_chain LDS budget");
            const dim3 grid(n_cap / ATT_Q, 1, (B + 7) / 8 * 8);
            if (l + 1 < M.cfg.num_layers && M.layers[l + 1].pq_img) {
                const PdscLayer &N = M.layers[l + 1];
#ifdef ORYON_DEV
                static const int clocks = dev_env_int("ORYON_PDSC_CLOCKS", 0);      // > 0: print the phase clocks of layer `clocks` of every call
                if (clocks > 0 && l == clocks) {
                    static long long *dclk = nullptr;
                    if (!dclk) { (void)hipMalloc(reinterpret_cast<void **>(&dclk), 32 * sizeof(long long)); }
                    (void)hipMemsetAsync(dclk, 0, 32 * sizeof(long long), st);
                    allow_dynamic_lds(reinterpret_cast<const void *>(pdsc_att_chain_x3_kernel<128, true, true>), AC_LDS);
                    hipLaunchKernelGGL((pdsc_att_chain_x3_kernel<128, true, true>), grid, dim3(512), AC_LDS, st, ws.qkv, kv_cur, ws.sc, n_rows, n_cap, inv_sqrt_c, B,
                                       ws.feat1, L.mlp_img_p, L.b_m1, L.b_m2, L.b_m3, N.pq_img, N.b_pcn, N.b_qkv, ws.feat1, ws.qkv, kv_nxt, ws.feat, dclk);
                    long long hc[32];
                    (void)hipMemcpyAsync(hc, dclk, sizeof(hc), hipMemcpyDeviceToHost, st);
                    (void)hipStreamSynchronize(st);
                    fprintf(stderr, "att_chain clocks (us since entry) wave0:");
                    for (int i = 1; i <= 15; ++i) fprintf(stderr, " %.2f", hc[i] ? (hc[i] - hc[0]) * 0.01 : -1.0);
                    fprintf(stderr, "  wave4:");
                    for (int i = 1; i <= 12; ++i) fprintf(stderr, " %.2f", hc[16 + i] ? (hc[16 + i] - hc[16]) * 0.01 : -1.0);
                    fprintf(stderr, "\n");
                    continue;
                }
#endif
                allow_dynamic_lds(reinterpret_cast<const void *>(pdsc_att_chain_x3_kernel<128, true>), AC_LDS);
                hipLaunchKernelGGL((pdsc_att_chain_x3_kernel<128, true>), grid, dim3(512), AC_LDS, st, ws.qkv, kv_cur, ws.sc, n_rows, n_cap, inv_sqrt_c, B,
                                   ws.feat1, L.mlp_img_p, L.b_m1, L.b_m2, L.b_m3, N.pq_img, N.b_pcn, N.b_qkv, ws.feat1, ws.qkv, kv_nxt, ws.feat);
            } else {
                allow_dynamic_lds(reinterpret_cast<const void *>(pdsc_att_chain_x3_kernel<128, false>), AC_LDS);
                hipLaunchKernelGGL((pdsc_att_chain_x3_kernel<128, false>), grid, dim3(512), AC_LDS, st, ws.qkv, kv_cur, ws.sc, n_rows, n_cap, inv_sqrt_c, B,
                                   ws.feat1, L.mlp_img_p, L.b_m1, L.b_m2, L.b_m3, nullptr, nullptr, nullptr, ws.feat1, ws.qkv, kv_nxt, ws.feat);
            }
            if (hipGetLastError() != hipSuccess) return ORYON_ERR_HIP;
            continue;
        }
        const int KS = ws.att_splits;
        dim3 ag(n_cap / ATT_Q, KS, B);
        static const bool att8 = dev_env_int("ORYON_PDSC_ATT8", 1) != 0;       // 0: the 4-wave attention kernel
        if (C == 128 && x3 && use_img && att8) {
            allow_dynamic_lds(reinterpret_cast<const void *>(pdsc_attention_x3_img8_kernel<128>), 2 * PDSC_KV_TILE_BYTES);
            hipLaunchKernelGGL((pdsc_attention_x3_img8_kernel<128>), dim3(n_cap / ATT_Q, 1, (B + 7) / 8 * 8), dim3(512), 2 * PDSC_KV_TILE_BYTES, st,
                               ws.qkv, ws.kv_img, ws.sc, n_rows, n_cap, inv_sqrt_c, ws.msg, B);
        } else if (C == 128 && x3 && use_img) {
            allow_dynamic_lds(reinterpret_cast<const void *>(pdsc_attention_x3_img_kernel<128>), 2 * PDSC_KV_TILE_BYTES);
            hipLaunchKernelGGL((pdsc_attention_x3_img_kernel<128>), dim3(n_cap / ATT_Q, 1, (B + 7) / 8 * 8), dim3(256), 2 * PDSC_KV_TILE_BYTES, st,
                               ws.qkv, ws.kv_img, ws.sc, n_rows, n_cap, inv_sqrt_c, ws.msg, B);
        } else if (C == 128 && x3)
            hipLaunchKernelGGL((pdsc_attention_x3_kernel<128>), ag, dim3(256), 0, st, ws.qkv, ws.sc, n_rows, n_cap, inv_sqrt_c, ws.msg, KS, ws.att_o, ws.att_ml);
        else if (C == 128)
            hipLaunchKernelGGL((pdsc_attention_kernel<128>), ag, dim3(256), 0, st, ws.qkv, ws.sc, n_rows, n_cap, inv_sqrt_c, ws.msg, KS, ws.att_o, ws.att_ml);
        else if (C == 64)
            hipLaunchKernelGGL((pdsc_attention_kernel<64>), ag, dim3(256), 0, st, ws.qkv, ws.sc, n_rows, n_cap, inv_sqrt_c, ws.msg, KS, ws.att_o, ws.att_ml);
        else
            hipLaunchKernelGGL((pdsc_attention_kernel<32>), ag, dim3(256), 0, st, ws.qkv, ws.sc, n_rows, n_cap, inv_sqrt_c, ws.msg, KS, ws.att_o, ws.att_ml);
        if (KS > 1)
            hipLaunchKernelGGL(pdsc_attention_merge_kernel, dim3((n_cap * (C / 4) + 255) / 256, B), dim3(256), 0, st, ws.att_o, ws.att_ml,
                               n_rows, n_cap, C, KS, B, ws.msg);
        if (hipGetLastError() != hipSuccess) return ORYON_ERR_HIP;
        // fc_message: C -> C/2 -> C/2 -> C, residual onto the PointCN output
        static const bool fused_mlp = dev_env_int("ORYON_PDSC_FUSED_MLP", 1) != 0;   // dev: 0 = three launches
        if (chain_ok && L.mlp_img && l + 1 < M.cfg.num_layers && M.layers[l + 1].pq_img) {
            const PdscLayer &N = M.layers[l + 1];
            constexpr int FZ_LDS = PDSC_MLP_IMG_BYTES + PDSC_PQ_CHUNK_BYTES;
            allow_dynamic_lds(reinterpret_cast<const void *>(pdsc_mlp3_pcn_qkv_x3_kernel<8>), FZ_LDS);
            hipLaunchKernelGGL(pdsc_mlp3_pcn_qkv_x3_kernel<8>, dim3(n_cap / 256, B), dim3(512), FZ_LDS, st, ws.msg, ws.feat1, L.mlp_img, L.b_m1, L.b_m2,
                               L.b_m3, N.pq_img, N.b_pcn, N.b_qkv, n_rows, n_cap, ws.feat1, ws.qkv, ws.kv_img);
            if (hipGetLastError() != hipSuccess) return ORYON_ERR_HIP;
            continue;
        }
        if (C == 128 && x3 && fused_mlp && L.mlp_img) {
            static const int mw = dev_env_int("ORYON_PDSC_WAVES", 8);
            if (mw == 8 && n_cap % 256 == 0) {
                allow_dynamic_lds(reinterpret_cast<const void *>(pdsc_mlp3_x3_kernel<8>), PDSC_MLP_IMG_BYTES);
                hipLaunchKernelGGL(pdsc_mlp3_x3_kernel<8>, dim3(n_cap / 256, B), dim3(512), PDSC_MLP_IMG_BYTES, st, ws.msg, ws.feat1, L.mlp_img, L.b_m1,
                                   L.b_m2, L.b_m3, n_rows, n_cap, ws.feat);
            } else {
                allow_dynamic_lds(reinterpret_cast<const void *>(pdsc_mlp3_x3_kernel<4>), PDSC_MLP_IMG_BYTES);
                hipLaunchKernelGGL(pdsc_mlp3_x3_kernel<4>, dim3(n_cap / 128, B), dim3(256), PDSC_MLP_IMG_BYTES, st, ws.msg, ws.feat1, L.mlp_img, L.b_m1,
                                   L.b_m2, L.b_m3, n_rows, n_cap, ws.feat);
            }
            if (hipGetLastError() != hipSuccess) return ORYON_ERR_HIP;
            continue;
        }
        rc = launch_linear(true, false, ws.msg, C, fb, L.w_m1, L.b_m1, nullptr, 0, 0, ws.h1, H, hb, C, H, B, n_cap, n_rows, st);
        if (rc) return rc;
        rc = launch_linear(true, false, ws.h1, H, hb, L.w_m2, L.b_m2, nullptr, 0, 0, ws.h2, H, hb, H, H, B, n_cap, n_rows, st);
        if (rc) return rc;
        rc = launch_linear(false, true, ws.h2, H, hb, L.w_m3, L.b_m3, ws.feat1, C, fb, ws.feat, C, fb, H, C, B, n_cap, n_rows, st);
        if (rc) return rc;
    }
    static const bool fused_head = dev_env_int("ORYON_PDSC_FUSED_HEAD", 1) != 0;       // dev: 0 = three linears + the normalisation
    if (fused_head && C == 128 && n_cap % LIN_ROWS == 0) {
        hipLaunchKernelGGL(pdsc_head_x3_kernel, dim3(n_cap / LIN_ROWS, B), dim3(256), 0, st, ws.feat, M.w_c1, M.b_c1, M.w_c2, M.b_c2, M.w_c3, M.b_c3,
                           n_rows, n_cap, ws.conf, ws.feat_n);
        return hipGetLastError() == hipSuccess ? ORYON_OK : ORYON_ERR_HIP;
    }
    // confidence head C -> 32 -> 32 -> 1 (PointDSC.py:107-113)
    rc = launch_linear(true, false, ws.feat, C, fb, M.w_c1, M.b_c1, nullptr, 0, 0, ws.h1, 32, (size_t)n_cap * 32, C, 32, B, n_cap, n_rows, st);
    if (rc) return rc;
    rc = launch_linear(true, false, ws.h1, 32, (size_t)n_cap * 32, M.w_c2, M.b_c2, nullptr, 0, 0, ws.h2, 32, (size_t)n_cap * 32, 32, 32, B, n_cap, n_rows, st);
    if (rc) return rc;
    rc = launch_linear(false, false, ws.h2, 32, (size_t)n_cap * 32, M.w_c3, M.b_c3, nullptr, 0, 0, ws.conf, 1, (size_t)n_cap, 32, 1, B, n_cap, n_rows, st);
    if (rc) return rc;
    hipLaunchKernelGGL(pdsc_normalise_kernel, dim3(n_cap / 4, B), dim3(256), 0, st, ws.feat, C, n_cap, n_rows, ws.feat_n);
    return hipGetLastError() == hipSuccess ? ORYON_OK : ORYON_ERR_HIP;
}

}  // namespace oryon
