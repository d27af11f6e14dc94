// B5: multi-head self-attention of the frozen CLIP image tower (models/vlm.py:46-56 -> clip's ResidualAttentionBlock: nn.MultiheadAttention
// on [L = 577, N, 1024], 16 heads of 64) in fp32-grade arithmetic on the fp16 matrix pipe.  torch evaluates it with an fp32 flash kernel at
// ~70 TFLOP/s (2.5 ms per layer for 128 images); here both products run error-compensated (x = hi + lo halves, a.b accumulated as
// a_hi.b_hi + a_hi.b_lo + a_lo.b_hi on v_mfma_f32_32x32x16_f16), the scheme of pdsc_attention_x3_kernel without the spatial-consistency
// weights:
//     S^T = K Q^T / 8   (rows = keys, columns = queries: a lane owns ONE query column, so the softmax statistics are lane-local)
//     O^T = V^T P^T     (the P registers of the softmax ARE the B operand of the second product; the V tile is laid out in LDS so that
//                        MFMA k-slot (lane half h, element e) of block (kb, t) is the key those registers hold)
// flash-style over 64-key tiles with running (max, sum) per query; fp32 accumulation, ~1e-6 relative.
//   qkv [N, L, 3*Dm] fp32 = the in_proj output (q | k | v, head h at columns h*64 .. h*64+63 of each third), out [N, L, Dm] fp32.
// One workgroup = 128 queries of one (image, head); 4 waves x 32 queries.
#include <hip/hip_fp16.h>
#include "common.h"

namespace oryon {

typedef _Float16 ahalf8 __attribute__((ext_vector_type(8)));
typedef float af32x16 __attribute__((ext_vector_type(16)));

constexpr int MHA_D = 64, MHA_KT = 64, MHA_Q = 128;
constexpr int MHA_KLD = MHA_D + 8;

__device__ __forceinline__ void mha_split(float x, _Float16 &hi, _Float16 &lo)
{
    hi = (_Float16)x;
    lo = (_Float16)(x - (float)hi);
}

__global__ __launch_bounds__(256) void mha_x3_kernel(const float *__restrict__ qkv, int L, int Dm, float scale, float *__restrict__ out, int n_qblk,
                                                     int heads, int n_units)
{
    constexpr int C = MHA_D, CB = C / 32, NS = C / 16;
    constexpr int KF4 = MHA_KT * (C / 4) / 256;          // 4 float4 of K per thread and tile
    constexpr int VPT = MHA_KT * C / 256, VOCT = VPT / 8, GROUPS = 256 / C;
    __shared__ __attribute__((aligned(16))) _Float16 Kh[MHA_KT * MHA_KLD], Kl[MHA_KT * MHA_KLD];
    __shared__ __attribute__((aligned(16))) _Float16 Vh[MHA_KT * C], Vl[MHA_KT * C];
    // XCD-aware block map (1-D grid): the query blocks of one (image, head) read the same K / V rows, so they get linear ids that are equal
    // mod 8 - one XCD, one L2 - instead of landing on n_qblk different XCDs that each fetch the 295 KB of K / V for themselves
    const int lin = blockIdx.x;
    const int unit = (lin / 8 / n_qblk) * 8 + (lin & 7);
    if (unit >= n_units) return;
    const int img = unit / heads, head = unit % heads, q0 = ((lin / 8) % n_qblk) * MHA_Q;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, hi = lane >> 5;
    const size_t rs = (size_t)3 * Dm;                    // row stride of qkv
    const float *base = qkv + (size_t)img * L * rs + head * C;
    const int qrow = q0 + wave * 32 + l31;
    const int qsafe = qrow < L ? qrow : L - 1;

    float4 kv[KF4];
    float vv[VPT];
    const int vch = t % C, vgrp = t / C;
    auto fetch = [&](int j0) {
#pragma unroll
        for (int i = 0; i < KF4; ++i) {
            const int e = t + 256 * i, row = e / (C / 4), c4 = e % (C / 4);
            kv[i] = j0 + row < L ? *reinterpret_cast<const float4 *>(base + (size_t)(j0 + row) * rs + Dm + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int o = 0; o < VOCT; ++o) {
            const int oct = vgrp + GROUPS * o;           // octet index = (kb*2 + t2)*2 + h
            const int kb = oct >> 2, t2 = (oct >> 1) & 1, h = oct & 1;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int key = j0 + kb * 32 + 16 * t2 + 8 * (e >> 2) + 4 * h + (e & 3);
                vv[o * 8 + e] = key < L ? base[(size_t)key * rs + 2 * Dm + vch] : 0.0f;
            }
        }
    };
    auto land = [&]() {
#pragma unroll
        for (int i = 0; i < KF4; ++i) {
            const int e = t + 256 * i, row = e / (C / 4), c4 = e % (C / 4);
            union { _Float16 h[4]; uint2 u; } ph, pl;
            mha_split(kv[i].x, ph.h[0], pl.h[0]); mha_split(kv[i].y, ph.h[1], pl.h[1]);
            mha_split(kv[i].z, ph.h[2], pl.h[2]); mha_split(kv[i].w, ph.h[3], pl.h[3]);
            *reinterpret_cast<uint2 *>(Kh + row * MHA_KLD + 4 * c4) = ph.u;
            *reinterpret_cast<uint2 *>(Kl + row * MHA_KLD + 4 * c4) = pl.u;
        }
#pragma unroll
        for (int o = 0; o < VOCT; ++o) {
            const int oct = vgrp + GROUPS * o;
            union { _Float16 h[8]; uint4 u; } ph, pl;
#pragma unroll
            for (int e = 0; e < 8; ++e) mha_split(vv[o * 8 + e], ph.h[e], pl.h[e]);
            *reinterpret_cast<uint4 *>(Vh + ((size_t)oct * C + vch) * 8) = ph.u;
            *reinterpret_cast<uint4 *>(Vl + ((size_t)oct * C + vch) * 8) = pl.u;
        }
    };

    // Q^T as B operand (scaled once): lane (query l31, half hi), k16 step s -> channels 16s + 8hi .. +7
    ahalf8 qh[NS], ql[NS];
    {
        const float4 *qv = reinterpret_cast<const float4 *>(base + (size_t)qsafe * rs);
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) {
            const float4 a = qv[4 * s_ + 2 * hi], c = qv[4 * s_ + 2 * hi + 1];
            const float x[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                _Float16 h_, l_;
                mha_split(x[e] * scale, h_, l_);         // scale = 2^-3 for head dim 64: exact
                qh[s_][e] = h_;
                ql[s_][e] = l_;
            }
        }
    }
    af32x16 acc_o[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[cb][r] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;

    fetch(0);
    for (int j0 = 0; j0 < L; j0 += MHA_KT) {
        __syncthreads();
        land();
        __syncthreads();
        if (j0 + MHA_KT < L) fetch(j0 + MHA_KT);
        af32x16 s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = 0.0f;
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const ahalf8 ah = *reinterpret_cast<const ahalf8 *>(Kh + (kb * 32 + l31) * MHA_KLD + 16 * s_ + 8 * hi);
                const ahalf8 al = *reinterpret_cast<const ahalf8 *>(Kl + (kb * 32 + l31) * MHA_KLD + 16 * s_ + 8 * hi);
                s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, qh[s_], s[kb], 0, 0, 0);
                s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, ql[s_], s[kb], 0, 0, 0);
                s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, qh[s_], s[kb], 0, 0, 0);
            }
        }
        float m_tile = -INFINITY;
        const bool ragged = j0 + MHA_KT > L;             // only the last tile holds keys >= L
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = s[kb][r];
                if (ragged && j0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi >= L) v = -INFINITY;
                s[kb][r] = v;
                m_tile = fmaxf(m_tile, v);
            }
        m_tile = fmaxf(m_tile, __shfl_xor(m_tile, 32));
        const float m_new = fmaxf(m_run, m_tile);
        const float alpha = __expf(m_run - m_new);
        float l_tile = 0.0f;
        ahalf8 ph[2][2], pl[2][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __expf(s[kb][r] - m_new);
                l_tile += p;
                _Float16 h_, l_;
                mha_split(p, h_, l_);
                ph[kb][r >> 3][r & 7] = h_;
                pl[kb][r >> 3][r & 7] = l_;
            }
        l_run = l_run * alpha + l_tile;
        m_run = m_new;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc_o[cb][r] *= alpha;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                const int oct = (kb * 2 + t2) * 2 + hi;
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) {
                    const ahalf8 vh = *reinterpret_cast<const ahalf8 *>(Vh + ((size_t)oct * C + cb * 32 + l31) * 8);
                    const ahalf8 vl = *reinterpret_cast<const ahalf8 *>(Vl + ((size_t)oct * C + cb * 32 + l31) * 8);
                    acc_o[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph[kb][t2], acc_o[cb], 0, 0, 0);
                    acc_o[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl[kb][t2], acc_o[cb], 0, 0, 0);
                    acc_o[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph[kb][t2], acc_o[cb], 0, 0, 0);
                }
            }
    }
    const float l_all = l_run + __shfl_xor(l_run, 32);
    if (qrow >= L) return;
    const float inv_l = 1.0f / l_all;
    // O^T block cb: lane owns query column l31 and channels cb*32 + (r & 3) + 8 (r >> 2) + 4 hi: four float4 per block
    float *mo = out + ((size_t)img * L + qrow) * Dm + head * C;
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

}  // namespace oryon

using namespace oryon;

extern "C" int oryon_mha_f16x3(const float *qkv, int N, int L, int heads, int head_dim, float *out, void *stream)
{
    ORYON_CHECK_ARG(qkv && out && N >= 0 && L > 0 && heads > 0 && head_dim == MHA_D);
    ORYON_CHECK_ARG((((uintptr_t)qkv | (uintptr_t)out) & 15) == 0);
    if (N == 0) return ORYON_OK;
    const int n_qblk = (L + MHA_Q - 1) / MHA_Q, n_units = heads * N;
    hipLaunchKernelGGL(mha_x3_kernel, dim3(n_qblk * ((n_units + 7) / 8 * 8)), dim3(256), 0, as_stream(stream), qkv, L, heads * head_dim, 0.125f, out,
                       n_qblk, heads, n_units);
    ORYON_CHECK_LAUNCH();
    return ORYON_OK;
}
