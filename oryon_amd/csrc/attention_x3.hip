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
#include <type_traits>
#include "common.h"

namespace oryon {

typedef _Float16 ahalf8 __attribute__((ext_vector_type(8)));
typedef float af32x16 __attribute__((ext_vector_type(16)));

constexpr int MHA_D = 64, MHA_KT = 64, MHA_Q = 128;
constexpr int MHA_KLD = MHA_D + 8;

// Two values at a time (round 6): packed conversion for hi, x - float(hi) as ONE v_fma_mix_f32 (hi's half read as the f16 source of an fp32
// fma), packed conversion for lo - four instructions per pair where mha_split takes four per ELEMENT plus the packing; the same bits
// (tools/probe_cvt_pk_f16.hip).  The kernel's VALU and MFMA instructions do not overlap on this part, so the split is wave time.
typedef float mha_f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 mha_f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void mha_split2(float a, float b, unsigned &hi, unsigned &lo)
{
    const mha_f32x2 v = {a, b};
    const unsigned hb = __builtin_bit_cast(unsigned, __builtin_convertvector(v, mha_f16x2));
    float l0, l1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(hb), "v"(a));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(hb), "v"(b));
    const mha_f32x2 lv = {l0, l1};
    hi = hb;
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(lv, mha_f16x2));
}
__device__ __forceinline__ void mha_split(float x, _Float16 &hi, _Float16 &lo)
{
    hi = (_Float16)x;
    lo = (_Float16)(x - (float)hi);
}

// Round 5 rework of the loop around the same arithmetic (results bit-identical to the round-2 kernel): the ISA of that kernel spent ~790
// VALU-class instructions per 64-key tile and wave against 48 MFMAs (1536 matrix-pipe cycles), a quarter of them 64-bit address arithmetic
// and per-load `key < L` branches of the 16 scalar V loads, another quarter v_accvgpr moves of a spilled prefetch set.  Now: wave-uniform
// 64-bit bases in SGPRs + one 32-bit lane offset (saddr loads); predicates only in the ONE ragged tile (rows clamped instead of zeroed -
// their scores are masked to -inf, p = 0 exactly, so any finite K / V row gives the same result); the second 32-key half of the last
// tile (L = 577: one valid key of 64) and the waves of the last query block that hold no query at all are skipped; the accumulators are
// rescaled only when some lane's running maximum moved.
template <bool RAGGED>
__device__ __forceinline__ void mha_fetch(const float *__restrict__ kbase, const float *__restrict__ vbase, unsigned rs, int j0, int L,
                                          unsigned krow0, unsigned kc4, unsigned vkey0, unsigned vch, float4 (&kv)[4], float (&vv)[16])
{
    if constexpr (!RAGGED) {
        const float *kp = kbase + (size_t)j0 * rs, *vp = vbase + (size_t)j0 * rs;           // wave-uniform
        const unsigned koff = krow0 * rs + 4u * kc4;
#pragma unroll
        for (int i = 0; i < 4; ++i) kv[i] = *reinterpret_cast<const float4 *>(kp + (koff + (unsigned)(16 * i) * rs));
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int e = 0; e < 8; ++e) vv[o * 8 + e] = vp[(unsigned)(32 * o + 8 * (e >> 2) + (e & 3)) * rs + vch];
    } else {
        const unsigned last = (unsigned)(L - 1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned row = (unsigned)j0 + krow0 + 16u * i;
            row = row < last ? row : last;
            kv[i] = *reinterpret_cast<const float4 *>(kbase + ((size_t)row * rs + 4u * kc4));
        }
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                unsigned key = (unsigned)j0 + vkey0 + (unsigned)(32 * o + 8 * (e >> 2) + (e & 3));
                key = key < last ? key : last;
                vv[o * 8 + e] = vbase[((size_t)key - vkey0) * rs + vch];                    // vbase already holds this wave's key phase
            }
    }
}

__global__ __launch_bounds__(256, 2) void mha_x3_kernel(const float *__restrict__ qkv, int L, int Dm, float scale, float *__restrict__ out,
                                                        int n_qblk, int heads, int n_units)
{
    constexpr int C = MHA_D, CB = C / 32, NS = C / 16;
    static_assert(C == 64 && MHA_KT == 64, "fetch / land index maps are written for 64 x 64 tiles and 256 threads");
    __shared__ __attribute__((aligned(16))) _Float16 Kh[MHA_KT * MHA_KLD], Kl[MHA_KT * MHA_KLD];
    __shared__ __attribute__((aligned(16))) _Float16 Vh[MHA_KT * C], Vl[MHA_KT * C];
    // XCD-aware block map (1-D grid): the query blocks of one (image, head) read the same K / V rows, so they get linear ids that are equal
    // mod 8 - one XCD, one L2 - instead of landing on n_qblk different XCDs that each fetch the 295 KB of K / V for themselves
    const int lin = blockIdx.x;
    const int unit = (lin / 8 / n_qblk) * 8 + (lin & 7);
    if (unit >= n_units) return;
    const int img = unit / heads, head = unit % heads, q0 = ((lin / 8) % n_qblk) * MHA_Q;
    const int t = threadIdx.x, lane = t & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const unsigned rs = 3u * (unsigned)Dm;               // row stride of qkv (floats)
    const float *base = qkv + (size_t)img * L * rs + head * C;
    const int qrow = q0 + wave * 32 + l31;
    const int qsafe = qrow < L ? qrow : L - 1;
    const bool wave_live = q0 + wave * 32 < L;           // wave-uniform: a wave without a single query only helps landing the tiles

    // K tile: float4 e = t + 256 i -> row t / 16 + 16 i, columns 4 (t % 16) ..; V tile: octet (wave + 4 o) of channel `lane` = keys
    // 32 o + 16 ((wave >> 1) & 1) + 4 (wave & 1) + 8 (e >> 2) + (e & 3) - the MFMA k-slot order of the P registers
    float4 kv[4];
    float vv[16];
    const unsigned krow0 = (unsigned)t >> 4, kc4 = (unsigned)t & 15u, vch = (unsigned)lane;
    const unsigned vkey0 = (unsigned)(16 * ((wave >> 1) & 1) + 4 * (wave & 1));
    const float *kbase = base + Dm, *vbase = base + 2 * Dm + (size_t)vkey0 * rs;
    auto land = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (int)krow0 + 16 * i;
            uint2 ph, pl;
            mha_split2(kv[i].x, kv[i].y, ph.x, pl.x);
            mha_split2(kv[i].z, kv[i].w, ph.y, pl.y);
            *reinterpret_cast<uint2 *>(Kh + row * MHA_KLD + 4 * (int)kc4) = ph;
            *reinterpret_cast<uint2 *>(Kl + row * MHA_KLD + 4 * (int)kc4) = pl;
        }
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const int oct = wave + 4 * o;
            uint4 ph, pl;
            mha_split2(vv[o * 8 + 0], vv[o * 8 + 1], ph.x, pl.x);
            mha_split2(vv[o * 8 + 2], vv[o * 8 + 3], ph.y, pl.y);
            mha_split2(vv[o * 8 + 4], vv[o * 8 + 5], ph.z, pl.z);
            mha_split2(vv[o * 8 + 6], vv[o * 8 + 7], ph.w, pl.w);
            *reinterpret_cast<uint4 *>(Vh + ((size_t)oct * C + lane) * 8) = ph;
            *reinterpret_cast<uint4 *>(Vl + ((size_t)oct * C + lane) * 8) = pl;
        }
    };

    // Q^T as B operand (scaled once): lane (query l31, half hi), k16 step s -> channels 16s + 8hi .. +7
    ahalf8 qh[NS], ql[NS];
    {
        const float4 *qv = reinterpret_cast<const float4 *>(base + (size_t)qsafe * rs);
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) {
            const float4 a = qv[4 * s_ + 2 * hi], c = qv[4 * s_ + 2 * hi + 1];
            uint4 uh, ul;                                 // scale = 2^-3 for head dim 64: exact
            mha_split2(a.x * scale, a.y * scale, uh.x, ul.x);
            mha_split2(a.z * scale, a.w * scale, uh.y, ul.y);
            mha_split2(c.x * scale, c.y * scale, uh.z, ul.z);
            mha_split2(c.z * scale, c.w * scale, uh.w, ul.w);
            qh[s_] = __builtin_bit_cast(ahalf8, uh);
            ql[s_] = __builtin_bit_cast(ahalf8, ul);
        }
    }
    af32x16 acc_o[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[cb][r] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;

    // one 64-key tile (NKB = 2) or its first 32 keys only (NKB = 1: the rest of the last tile lies beyond L)
    auto tile = [&](auto nkb_tag, auto ragged_tag, int j0) {
        constexpr int NKB = decltype(nkb_tag)::value;
        constexpr bool RAGGED = decltype(ragged_tag)::value;
        af32x16 s[NKB];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = 0.0f;
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) {
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                const ahalf8 ah = *reinterpret_cast<const ahalf8 *>(Kh + (kb * 32 + l31) * MHA_KLD + 16 * s_ + 8 * hi);
                const ahalf8 al = *reinterpret_cast<const ahalf8 *>(Kl + (kb * 32 + l31) * MHA_KLD + 16 * s_ + 8 * hi);
                s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, qh[s_], s[kb], 0, 0, 0);
                s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, ql[s_], s[kb], 0, 0, 0);
                s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, qh[s_], s[kb], 0, 0, 0);
            }
        }
        float m_tile = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = s[kb][r];
                if constexpr (RAGGED) {
                    if (j0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi >= L) v = -INFINITY;
                    s[kb][r] = v;
                }
                m_tile = fmaxf(m_tile, v);
            }
        m_tile = fmaxf(m_tile, __shfl_xor(m_tile, 32));
        const float m_new = fmaxf(m_run, m_tile);
        if (__builtin_amdgcn_ballot_w64(m_new != m_run) != 0ull) {      // some lane's maximum moved: rescale (alpha is exactly 1 elsewhere)
            const float alpha = __expf(m_run - m_new);
            l_run *= alpha;
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc_o[cb][r] *= alpha;
            m_run = m_new;
        }
        float l_tile = 0.0f;
        ahalf8 ph[NKB][2], pl[NKB][2];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const float p0 = __expf(s[kb][r] - m_new), p1 = __expf(s[kb][r + 1] - m_new);
                l_tile += p0;
                l_tile += p1;
                unsigned uh, ul;
                mha_split2(p0, p1, uh, ul);
                const mha_f16x2 h2 = __builtin_bit_cast(mha_f16x2, uh), l2 = __builtin_bit_cast(mha_f16x2, ul);
                ph[kb][r >> 3][r & 7] = h2[0]; ph[kb][r >> 3][(r & 7) + 1] = h2[1];
                pl[kb][r >> 3][r & 7] = l2[0]; pl[kb][r >> 3][(r & 7) + 1] = l2[1];
            }
        l_run += l_tile;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
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
    };

    auto fetch = [&](int j) {
        if (j + MHA_KT <= L) mha_fetch<false>(kbase, vbase, rs, j, L, krow0, kc4, vkey0, vch, kv, vv);
        else mha_fetch<true>(kbase, vbase, rs, j, L, krow0, kc4, vkey0, vch, kv, vv);
    };
    // (tried: two tile buffers with the next tile's split placed after the QK^T MFMAs and one barrier per tile - 0.88 vs 0.87 ms, the
    // compiler keeps the split's VALU behind the MFMA block whatever sched_group_barrier asks for)
    fetch(0);
    for (int j0 = 0; j0 < L; j0 += MHA_KT) {
        __syncthreads();
        land();
        __syncthreads();
        const int jn = j0 + MHA_KT;
        if (jn < L) fetch(jn);
        if (!wave_live) continue;
        if (jn <= L) tile(std::integral_constant<int, 2>{}, std::false_type{}, j0);
        else if (j0 + 32 < L) tile(std::integral_constant<int, 2>{}, std::true_type{}, j0);
        else tile(std::integral_constant<int, 1>{}, std::true_type{}, j0);
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
