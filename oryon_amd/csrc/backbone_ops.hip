// Fused element-wise epilogues for the PyTorch-ROCm backbone (the towers' GEMMs stay on hipBLASLt by design; what is fused
// here is the activation torch would run as three bandwidth-bound kernels).
//
// QuickGELU of the CLIP residual blocks (third-party `clip` model.py: x * sigmoid(1.702 * x); models/vlm.py:19 loads it):
// one read and one write of the [tokens, 4*width] activation instead of three of each.  bf16 in / bf16 out, arithmetic in fp32
// with a single final rounding (torch's bf16 chain rounds three times); the fp32 path of the backbone keeps torch's own ops so
// that the reference-pinned fp32 numerics are untouched.
#include <hip/hip_bf16.h>
#include "common.h"

namespace oryon {

__device__ __forceinline__ float bf16_bits_to_float(unsigned short b) { return __uint_as_float((unsigned)b << 16); }
__device__ __forceinline__ unsigned short float_to_bf16_bits(float f)
{
    // round to nearest even (NaN stays NaN: the mantissa msb is forced)
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float quick_gelu(float x) { return x / (1.0f + __expf(-1.702f * x)); }

__global__ __launch_bounds__(256) void quick_gelu_bf16_kernel(const uint4 *__restrict__ x, uint4 *__restrict__ y, int64_t n8,
                                                               const unsigned short *__restrict__ xt, unsigned short *__restrict__ yt,
                                                               int tail)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
        const uint4 v = x[i];
        uint4 o;
        const unsigned in[4] = {v.x, v.y, v.z, v.w};
        unsigned out[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float lo = quick_gelu(bf16_bits_to_float((unsigned short)(in[j] & 0xffffu)));
            const float hi = quick_gelu(bf16_bits_to_float((unsigned short)(in[j] >> 16)));
            out[j] = (unsigned)float_to_bf16_bits(lo) | ((unsigned)float_to_bf16_bits(hi) << 16);
        }
        o.x = out[0]; o.y = out[1]; o.z = out[2]; o.w = out[3];
        y[i] = o;
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < tail) yt[threadIdx.x] = float_to_bf16_bits(quick_gelu(bf16_bits_to_float(xt[threadIdx.x])));
}

// Residual add + LayerNorm of the CLIP residual stream (clip model.py ResidualAttentionBlock: x = x + attn(ln_1(x)); x = x + mlp(ln_2(x))):
// torch runs the bf16 add (2 reads + 1 write) and nn.LayerNorm (1 read + 1 write, 1.3 TB/s measured for 1024-wide rows) as two
// launches; here one wave owns one row, keeps it in registers, and emits both the new residual stream and its normalised copy.
// Semantics equal torch's chain: s = bf16(x + delta) is rounded first, the statistics are fp32 over the rounded s (two-pass
// variance), h = bf16((s - mean) * rstd * gamma + beta).
// LPR lanes share one row (16 / 32 for the 128- and 256-wide Swin stages: 4 / 2 rows per wave, so that every lane has work).
constexpr int LN_MAX_CHUNKS = 8;                 // 8 chunks x 64 lanes x 8 values = rows up to 4096 wide
template <int LPR>
__global__ __launch_bounds__(256) void add_layernorm_bf16_kernel(const unsigned short *__restrict__ x,
                                                                  const unsigned short *__restrict__ delta,
                                                                  const unsigned short *__restrict__ gamma,
                                                                  const unsigned short *__restrict__ beta, int64_t rows, int D, float eps,
                                                                  unsigned short *__restrict__ x_out, unsigned short *__restrict__ h_out)
{
    constexpr int CH = LPR == 64 ? LN_MAX_CHUNKS : 1;
    constexpr int RPB = 256 / LPR;
    const int lane = threadIdx.x & (LPR - 1);
    const int64_t row = (int64_t)blockIdx.x * RPB + (threadIdx.x / LPR);
    if (row >= rows) return;                     // whole row groups leave together: the shuffles below stay inside a group
    const int chunks = (D + LPR * 8 - 1) / (LPR * 8);
    float v[CH][8];
    float sum = 0.0f;
    const unsigned short *xr = x + row * D;
    const unsigned short *dr = delta ? delta + row * D : nullptr;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int col = c * LPR * 8 + lane * 8;
        if (c < chunks && col < D) {
            const uint4 a = *reinterpret_cast<const uint4 *>(xr + col);
            const unsigned aw[4] = {a.x, a.y, a.z, a.w};
            unsigned sw[4];
            if (dr) {
                const uint4 b = *reinterpret_cast<const uint4 *>(dr + col);
                const unsigned bw[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float lo = bf16_bits_to_float((unsigned short)(aw[j] & 0xffffu)) + bf16_bits_to_float((unsigned short)(bw[j] & 0xffffu));
                    const float hi = bf16_bits_to_float((unsigned short)(aw[j] >> 16)) + bf16_bits_to_float((unsigned short)(bw[j] >> 16));
                    sw[j] = (unsigned)float_to_bf16_bits(lo) | ((unsigned)float_to_bf16_bits(hi) << 16);
                }
                if (x_out) *reinterpret_cast<uint4 *>(x_out + row * D + col) = make_uint4(sw[0], sw[1], sw[2], sw[3]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) sw[j] = aw[j];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[c][2 * j] = bf16_bits_to_float((unsigned short)(sw[j] & 0xffffu));
                v[c][2 * j + 1] = bf16_bits_to_float((unsigned short)(sw[j] >> 16));
                sum += v[c][2 * j] + v[c][2 * j + 1];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[c][j] = 0.0f;
        }
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum / (float)D;
    float sq = 0.0f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int col = c * LPR * 8 + lane * 8;
        if (c < chunks && col < D) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d = v[c][j] - mean;
                sq = fmaf(d, d, sq);
            }
        }
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    const float rstd = rsqrtf(sq / (float)D + eps);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int col = c * LPR * 8 + lane * 8;
        if (c < chunks && col < D) {
            const uint4 g = *reinterpret_cast<const uint4 *>(gamma + col);
            const uint4 b = *reinterpret_cast<const uint4 *>(beta + col);
            const unsigned gw[4] = {g.x, g.y, g.z, g.w}, bw[4] = {b.x, b.y, b.z, b.w};
            unsigned ow[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float lo = fmaf((v[c][2 * j] - mean) * rstd, bf16_bits_to_float((unsigned short)(gw[j] & 0xffffu)),
                                      bf16_bits_to_float((unsigned short)(bw[j] & 0xffffu)));
                const float hi = fmaf((v[c][2 * j + 1] - mean) * rstd, bf16_bits_to_float((unsigned short)(gw[j] >> 16)),
                                      bf16_bits_to_float((unsigned short)(bw[j] >> 16)));
                ow[j] = (unsigned)float_to_bf16_bits(lo) | ((unsigned)float_to_bf16_bits(hi) << 16);
            }
            *reinterpret_cast<uint4 *>(h_out + row * D + col) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
        }
    }
}

// fp32 twin of the kernel above for the fp32 / fp16x3 inference path (the towers' residual stream stays fp32 there): s = x + delta,
// two-pass fp32 statistics over s, h = (s - mean) * rstd * gamma + beta.  4 floats per lane per chunk (16-byte accesses).
constexpr int LN_F32_CHUNKS = 8;                 // 8 chunks x 64 lanes x 4 values = rows up to 2048 wide
template <int LPR>
__global__ __launch_bounds__(256) void add_layernorm_f32_kernel(const float *__restrict__ x, const float *__restrict__ delta,
                                                                 const float *__restrict__ gamma, const float *__restrict__ beta,
                                                                 int64_t rows, int D, float eps, float *__restrict__ x_out,
                                                                 float *__restrict__ h_out, unsigned *range_flag)
{
    constexpr int CH = LPR == 64 ? LN_F32_CHUNKS : 1;
    constexpr int RPB = 256 / LPR;
    const int lane = threadIdx.x & (LPR - 1);
    const int64_t row = (int64_t)blockIdx.x * RPB + (threadIdx.x / LPR);
    if (row >= rows) return;
    const int chunks = (D + LPR * 4 - 1) / (LPR * 4);
    float4 v[CH];
    float sum = 0.0f;
    unsigned x3m = 0u;
    const float *xr = x + row * D;
    const float *dr = delta ? delta + row * D : nullptr;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int col = c * LPR * 4 + lane * 4;
        v[c] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (c < chunks && col < D) {
            float4 a = *reinterpret_cast<const float4 *>(xr + col);
            if (dr) {
                const float4 b = *reinterpret_cast<const float4 *>(dr + col);
                a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
                *reinterpret_cast<float4 *>(x_out + row * D + col) = a;
            }
            v[c] = a;
            sum += (a.x + a.y) + (a.z + a.w);
            // the residual stream of the fp16x3 towers is range-checked HERE: whoever updated it (the in-place linear's atomics, the
            // delta of this very call), this pass reads every element of it before the next fp16x3 kernel splits anything derived from it
            x3m = max(max(x3m, x3_mag(a.x)), max(max(x3_mag(a.y), x3_mag(a.z)), x3_mag(a.w)));
        }
    }
    x3_raise(range_flag, x3m);
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum / (float)D;
    float sq = 0.0f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int col = c * LPR * 4 + lane * 4;
        if (c < chunks && col < D) {
            const float d0 = v[c].x - mean, d1 = v[c].y - mean, d2 = v[c].z - mean, d3 = v[c].w - mean;
            sq = fmaf(d0, d0, sq); sq = fmaf(d1, d1, sq); sq = fmaf(d2, d2, sq); sq = fmaf(d3, d3, sq);
        }
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    const float rstd = rsqrtf(sq / (float)D + eps);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int col = c * LPR * 4 + lane * 4;
        if (c < chunks && col < D) {
            const float4 g = *reinterpret_cast<const float4 *>(gamma + col);
            const float4 b = *reinterpret_cast<const float4 *>(beta + col);
            float4 o;
            o.x = fmaf((v[c].x - mean) * rstd, g.x, b.x);
            o.y = fmaf((v[c].y - mean) * rstd, g.y, b.y);
            o.z = fmaf((v[c].z - mean) * rstd, g.z, b.z);
            o.w = fmaf((v[c].w - mean) * rstd, g.w, b.w);
            *reinterpret_cast<float4 *>(h_out + row * D + col) = o;
        }
    }
}

// Shifted-window attention of the Swin guidance backbone (torchvision swin_transformer.shifted_window_attention, reached from
// net.py:60-75 of the reference through swin_b's feature extractor), bf16 inference.  torch runs it as pad + roll + window
// partition copy, q scaling, two batched 49x49 matmuls, bias add, mask add, softmax, transpose copy, window merge copy and the
// reverse roll: a dozen bandwidth-bound passes over [tokens, C] and [windows, heads, 49, 49] tensors.  The per-token q|k|v Linear
// commutes with all of that, so here it runs on the un-windowed tokens and ONE kernel does the rest: a workgroup owns a window,
// a wave owns a head, a lane owns a query token; roll, padding (pad tokens carry q|k|v = the Linear's bias, exactly what the zero
// padding produces), relative-position bias and the shift mask are index arithmetic.  fp32 arithmetic, one rounding at the end.
constexpr int SWIN_WS = 7, SWIN_N = SWIN_WS * SWIN_WS, SWIN_HD = 32, SWIN_LD = SWIN_HD + 4;
constexpr int SWIN_WAVE_FLOATS = 2 * SWIN_N * SWIN_LD + 64;

__device__ __forceinline__ void load_head_slice(const unsigned short *p, float (&f)[SWIN_HD])
{
#pragma unroll
    for (int c = 0; c < SWIN_HD / 8; ++c) {
        const uint4 u = *reinterpret_cast<const uint4 *>(p + c * 8);
        const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f[c * 8 + 2 * j] = bf16_bits_to_float((unsigned short)(w[j] & 0xffffu));
            f[c * 8 + 2 * j + 1] = bf16_bits_to_float((unsigned short)(w[j] >> 16));
        }
    }
}

__device__ __forceinline__ void load_head_slice(const float *p, float (&f)[SWIN_HD])
{
#pragma unroll
    for (int c = 0; c < SWIN_HD / 4; ++c) {
        const float4 u = *reinterpret_cast<const float4 *>(p + c * 4);
        f[c * 4] = u.x; f[c * 4 + 1] = u.y; f[c * 4 + 2] = u.z; f[c * 4 + 3] = u.w;
    }
}

// IO = unsigned short (bf16 bits) or float: the arithmetic is fp32 either way.  The float instantiation serves the fp32 evaluation
// of the guidance tower (`full` stage set: torch's dozen passes cost 54 ms per 128 images).
template <typename IO>
__global__ __launch_bounds__(512) void swin_window_attention_kernel(const IO *__restrict__ qkv, const IO *__restrict__ pad_qkv,
                                                                     const float *__restrict__ bias_t, int H, int W, int C, int shift,
                                                                     IO *__restrict__ out)
{
    constexpr bool F32 = sizeof(IO) == 4;
    extern __shared__ float swin_lds[];
    const int head = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *Ks = swin_lds + head * SWIN_WAVE_FLOATS;
    float *Vs = Ks + SWIN_N * SWIN_LD;
    int *labs = reinterpret_cast<int *>(Vs + SWIN_N * SWIN_LD);
    const int Hp = (H + SWIN_WS - 1) / SWIN_WS * SWIN_WS, Wp = (W + SWIN_WS - 1) / SWIN_WS * SWIN_WS;
    const int nwx = Wp / SWIN_WS;
    const int wy = blockIdx.x / nwx, wx = blockIdx.x % nwx, b = blockIdx.y;
    const bool tok = lane < SWIN_N;
    float q[SWIN_HD];
    int label = 0;
    bool real = false;
    size_t out_off = 0;
    if (tok) {
        const int py = wy * SWIN_WS + lane / SWIN_WS, px = wx * SWIN_WS + lane % SWIN_WS;     // rolled, padded frame
        const int sy = (py + shift) % Hp, sx = (px + shift) % Wp;                           // torch.roll(x, -shift): out[p] = in[p + shift]
        real = sy < H && sx < W;
        if (shift > 0) {
            const int by = py < Hp - SWIN_WS ? 0 : (py < Hp - shift ? 1 : 2);
            const int bx = px < Wp - SWIN_WS ? 0 : (px < Wp - shift ? 1 : 2);
            label = by * 3 + bx;
        }
        const size_t t = ((size_t)b * H + sy) * W + sx;
        out_off = t * C + head * SWIN_HD;
        const IO *src = real ? qkv + t * 3 * C + head * SWIN_HD : pad_qkv + head * SWIN_HD;
        float kv[SWIN_HD];
        load_head_slice(src, q);
        const float scale = 0.17677669529663687f;                                            // 32^-0.5
#pragma unroll
        for (int d = 0; d < SWIN_HD; ++d) q[d] *= scale;
        load_head_slice(src + C, kv);
#pragma unroll
        for (int d = 0; d < SWIN_HD; d += 4) *reinterpret_cast<float4 *>(Ks + lane * SWIN_LD + d) = make_float4(kv[d], kv[d + 1], kv[d + 2], kv[d + 3]);
        load_head_slice(src + 2 * C, kv);
#pragma unroll
        for (int d = 0; d < SWIN_HD; d += 4) *reinterpret_cast<float4 *>(Vs + lane * SWIN_LD + d) = make_float4(kv[d], kv[d + 1], kv[d + 2], kv[d + 3]);
        labs[lane] = label;
    }
    __syncthreads();
    if (!tok) return;
    // online softmax over the 49 keys (the fully unrolled two-pass form keeps 49 scores per lane and the compiler then hoists
    // every LDS read above the arithmetic: 1500 spilled VGPRs)
    const float *bt = bias_t + (size_t)head * SWIN_N * SWIN_N + lane;
    float m = -INFINITY, sum = 0.0f;
    float acc[SWIN_HD];
#pragma unroll
    for (int d = 0; d < SWIN_HD; ++d) acc[d] = 0.0f;
#pragma unroll 1
    for (int j = 0; j < SWIN_N; ++j) {
        float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
        for (int d = 0; d < SWIN_HD; d += 8) {
            const float4 k4 = *reinterpret_cast<const float4 *>(Ks + j * SWIN_LD + d);
            const float4 k5 = *reinterpret_cast<const float4 *>(Ks + j * SWIN_LD + d + 4);
            a0 = fmaf(q[d], k4.x, a0); a0 = fmaf(q[d + 1], k4.y, a0); a0 = fmaf(q[d + 2], k4.z, a0); a0 = fmaf(q[d + 3], k4.w, a0);
            a1 = fmaf(q[d + 4], k5.x, a1); a1 = fmaf(q[d + 5], k5.y, a1); a1 = fmaf(q[d + 6], k5.z, a1); a1 = fmaf(q[d + 7], k5.w, a1);
        }
        float a = a0 + a1 + bt[j * SWIN_N];
        if (labs[j] != label) a -= 100.0f;
        const float m_new = fmaxf(m, a);
        const float corr = F32 ? expf(m - m_new) : __expf(m - m_new);           // exp(-inf) = 0 on the first key
        const float p = F32 ? expf(a - m_new) : __expf(a - m_new);
        m = m_new;
        sum = fmaf(sum, corr, p);
#pragma unroll
        for (int d = 0; d < SWIN_HD; d += 4) {
            const float4 v4 = *reinterpret_cast<const float4 *>(Vs + j * SWIN_LD + d);
            acc[d] = fmaf(acc[d], corr, p * v4.x); acc[d + 1] = fmaf(acc[d + 1], corr, p * v4.y);
            acc[d + 2] = fmaf(acc[d + 2], corr, p * v4.z); acc[d + 3] = fmaf(acc[d + 3], corr, p * v4.w);
        }
    }
    if (!real) return;                                                                      // pad tokens are cropped away
    const float inv = 1.0f / sum;
    if constexpr (F32) {
#pragma unroll
        for (int c = 0; c < SWIN_HD / 4; ++c)
            *reinterpret_cast<float4 *>(out + out_off + c * 4) = make_float4(acc[c * 4] * inv, acc[c * 4 + 1] * inv, acc[c * 4 + 2] * inv, acc[c * 4 + 3] * inv);
    } else
#pragma unroll
    for (int c = 0; c < SWIN_HD / 8; ++c) {
        unsigned w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            w[j] = (unsigned)float_to_bf16_bits(acc[c * 8 + 2 * j] * inv) | ((unsigned)float_to_bf16_bits(acc[c * 8 + 2 * j + 1] * inv) << 16);
        *reinterpret_cast<uint4 *>(reinterpret_cast<unsigned short *>(out) + out_off + c * 8) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

}  // namespace oryon

using namespace oryon;

extern "C" int oryon_quick_gelu_bf16(const void *x, void *y, int64_t n, void *stream)
{
    ORYON_CHECK_ARG(x && y && n >= 0);
    ORYON_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0);
    if (n == 0) return ORYON_OK;
    const int64_t n8 = n / 8;
    const int tail = (int)(n - n8 * 8);
    int64_t blocks = (n8 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    const unsigned short *xt = static_cast<const unsigned short *>(x) + n8 * 8;
    unsigned short *yt = static_cast<unsigned short *>(y) + n8 * 8;
    hipLaunchKernelGGL(quick_gelu_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), static_cast<const uint4 *>(x),
                       static_cast<uint4 *>(y), n8, xt, yt, tail);
    ORYON_CHECK_LAUNCH();
    return ORYON_OK;
}

extern "C" int oryon_add_layernorm_bf16(const void *x, const void *delta, const void *gamma, const void *beta, int64_t rows, int D,
                                        float eps, void *x_out, void *h_out, void *stream)
{
    ORYON_CHECK_ARG(x && gamma && beta && h_out && rows >= 0 && D > 0 && D % 8 == 0 && D <= 512 * LN_MAX_CHUNKS && eps > 0.0f);
    ORYON_CHECK_ARG(!delta || x_out);
    ORYON_CHECK_ARG((((uintptr_t)x | (uintptr_t)delta | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)x_out | (uintptr_t)h_out) & 15) == 0);
    if (rows == 0) return ORYON_OK;
#define ORYON_LAUNCH_LN(LPR)                                                                                                       \
    hipLaunchKernelGGL((add_layernorm_bf16_kernel<LPR>), dim3((unsigned)((rows + 256 / LPR - 1) / (256 / LPR))), dim3(256), 0,      \
                       as_stream(stream), static_cast<const unsigned short *>(x), static_cast<const unsigned short *>(delta),        \
                       static_cast<const unsigned short *>(gamma), static_cast<const unsigned short *>(beta), rows, D, eps,          \
                       static_cast<unsigned short *>(x_out), static_cast<unsigned short *>(h_out))
    if (D <= 128) ORYON_LAUNCH_LN(16);
    else if (D <= 256) ORYON_LAUNCH_LN(32);
    else ORYON_LAUNCH_LN(64);
#undef ORYON_LAUNCH_LN
    ORYON_CHECK_LAUNCH();
    return ORYON_OK;
}

extern "C" int oryon_add_layernorm_f32(const float *x, const float *delta, const float *gamma, const float *beta, int64_t rows, int D,
                                       float eps, float *x_out, float *h_out, void *stream)
{
    ORYON_CHECK_ARG(x && gamma && beta && h_out && rows >= 0 && D > 0 && D % 4 == 0 && D <= 256 * LN_F32_CHUNKS && eps > 0.0f);
    ORYON_CHECK_ARG(!delta || x_out);
    ORYON_CHECK_ARG((((uintptr_t)x | (uintptr_t)delta | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)x_out | (uintptr_t)h_out) & 15) == 0);
    if (rows == 0) return ORYON_OK;
    unsigned *rflag = x3_range_flag(as_stream(stream));
    if (!rflag) return ORYON_ERR_HIP;
#define ORYON_LAUNCH_LN32(LPR)                                                                                                     \
    hipLaunchKernelGGL((add_layernorm_f32_kernel<LPR>), dim3((unsigned)((rows + 256 / LPR - 1) / (256 / LPR))), dim3(256), 0,       \
                       as_stream(stream), x, delta, gamma, beta, rows, D, eps, x_out, h_out, rflag)
    if (D <= 64) ORYON_LAUNCH_LN32(16);
    else if (D <= 128) ORYON_LAUNCH_LN32(32);
    else ORYON_LAUNCH_LN32(64);
#undef ORYON_LAUNCH_LN32
    ORYON_CHECK_LAUNCH();
    return ORYON_OK;
}

extern "C" int oryon_swin_window_attention_bf16(const void *qkv, const void *pad_qkv, const float *bias_t, int B, int H, int W, int C,
                                                int heads, int shift, void *out, void *stream)
{
    ORYON_CHECK_ARG(qkv && pad_qkv && bias_t && out && B >= 0 && H > 0 && W > 0 && heads >= 1 && heads <= 8 && C == heads * SWIN_HD);
    ORYON_CHECK_ARG(shift >= 0 && shift < SWIN_WS);
    ORYON_CHECK_ARG((((uintptr_t)qkv | (uintptr_t)pad_qkv | (uintptr_t)out) & 15) == 0);
    if (B == 0) return ORYON_OK;
    const int nwy = (H + SWIN_WS - 1) / SWIN_WS, nwx = (W + SWIN_WS - 1) / SWIN_WS;
    const size_t lds = (size_t)heads * SWIN_WAVE_FLOATS * sizeof(float);
    allow_dynamic_lds(reinterpret_cast<const void *>(swin_window_attention_kernel<unsigned short>), 8 * SWIN_WAVE_FLOATS * (int)sizeof(float));
    hipLaunchKernelGGL(swin_window_attention_kernel<unsigned short>, dim3(nwy * nwx, B), dim3(64 * heads), lds, as_stream(stream),
                       static_cast<const unsigned short *>(qkv), static_cast<const unsigned short *>(pad_qkv), bias_t, H, W, C, shift,
                       static_cast<unsigned short *>(out));
    ORYON_CHECK_LAUNCH();
    return ORYON_OK;
}


// ------------------------------------------------------------------------------------------------------------------------------------
// a4  window attention of ImageTextFusion's guided Swin blocks (models/fusion.py:75-103 WindowAttention.forward inside
//     SwinTransformerBlock.forward :173-213): torch.roll(-shift) + window_partition + softmax(q k^T * hd^-0.5 + shift mask) v +
//     window_reverse + torch.roll(+shift) as ONE kernel on un-windowed tokens - the q | k and v projections are per-token linears, so they
//     run on the natural [B, H, W, .] order and nothing is permuted or copied around the attention (torch: 2 rolls, 2 window copies, 3
//     head transposes, a scale, 2 batched GEMMs, a mask add and a softmax per block).
// Both products run on the fp16 matrix pipe with error-compensated operands (a first version on the fp32 VALU - one query per thread, K / V
// rows read from LDS as broadcasts - took 0.23 ms per call against 0.09 for this one).
// One workgroup per (window, image, head), five waves; wave w owns queries 32 w .. 32 w + 31 (144 = 4.5 blocks, the rest is padding).
// Everything is computed TRANSPOSED so that nothing has to change layout between the two products:
//   S^T = K Q^T : A = K rows (M = keys, straight from global memory), B = Q rows (N = queries, scaled by hd^-0.5 first, as the module does)
//                 -> a lane holds, for ITS query, the scores of 80 keys (16 per key block), its partner lane ^ 32 the other 80: the softmax
//                 statistics are in-lane loops plus one cross-lane exchange;
//   O^T = V^T P^T: B = P^T = the exponentials exactly where the accumulators left them (k-step ks = registers 8 (ks & 1) .. + 7 of key
//                 block ks / 2), A = V^T from an LDS tile whose key axis is stored in that register order.
// Operands are split hi + lo (fp16) and multiplied as lo.hi + hi.lo + hi.hi with fp32 accumulation, like every fp16x3 kernel here.
typedef _Float16 fwa_h8 __attribute__((ext_vector_type(8)));
typedef float fwa_acc __attribute__((ext_vector_type(16)));

static __device__ __forceinline__ void fwa_split8(const float (&x)[8], fwa_h8 &hi, fwa_h8 &lo)
{
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const _Float16 h = (_Float16)x[e];
        hi[e] = h;
        lo[e] = (_Float16)(x[e] - (float)h);
    }
}

// slot of key j on the key axis of the V^T tile: k-step (j / 32) * 2 + (j % 32) / 16, lane half ((j % 16) % 8) / 4, element 4 ((j % 16) / 8) + j % 4
// - the inverse of "register r of key block mb holds key 32 mb + 8 (r / 4) + 4 kg + r % 4"
static __device__ __forceinline__ int fwa_slot(int j)
{
    const int jj = j & 31, u = jj & 15;
    return ((j >> 5) * 2 + (jj >> 4)) * 16 + ((u & 7) >> 2) * 8 + 4 * (u >> 3) + (u & 3);
}

__global__ __launch_bounds__(320) void fusion_window_attention_x3_kernel(const float *__restrict__ qk, const float *__restrict__ v, int H, int W,
                                                                         int C, int shift, float scale, float *__restrict__ out)
{
    constexpr int WS = 12, N = 144, NP = 160, HD = 32;
    constexpr int VLD = 168;                                  // halves per V^T row (336 bytes: conflict-free 16-byte fragment reads)
    __shared__ __attribute__((aligned(16))) _Float16 Vh[HD * VLD];
    __shared__ __attribute__((aligned(16))) _Float16 Vl[HD * VLD];
    __shared__ int toks[NP];
    __shared__ int labs[NP];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, li = lane & 31, kg = lane >> 5;
    const int head = blockIdx.z, b = blockIdx.y;
    const int nwx = W / WS, wy = blockIdx.x / nwx, wx = blockIdx.x % nwx;
    if (t < NP) {
        int tk = -1, lab = -1;
        if (t < N) {
            const int py = wy * WS + t / WS, px = wx * WS + t % WS;
            const int sy = (py + shift) % H, sx = (px + shift) % W;
            tk = (b * H + sy) * W + sx;
            lab = 0;
            if (shift > 0) {
                const int by = py < H - WS ? 0 : (py < H - shift ? 1 : 2);
                const int bx = px < W - WS ? 0 : (px < W - shift ? 1 : 2);
                lab = by * 3 + bx;
            }
        }
        toks[t] = tk;
        labs[t] = lab;
    }
    // the padded key slots of V^T are zero (their probabilities are zero too, but 0 x garbage is not)
    for (int i = t; i < HD * 16; i += 320) {
        const int d = i >> 4, sig = fwa_slot(N + (i & 15));
        Vh[d * VLD + sig] = (_Float16)0.0f;
        Vl[d * VLD + sig] = (_Float16)0.0f;
    }
    __syncthreads();
    // V^T tile: key j -> slot sigma(j) (the order in which the accumulators of S^T hold the keys)
    for (int i = t; i < N * 8; i += 320) {
        const int j = i >> 3, dq = i & 7;
        const float4 x = *reinterpret_cast<const float4 *>(v + (size_t)toks[j] * C + head * HD + dq * 4);
        const int sig = fwa_slot(j);
        const float xv[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const _Float16 h = (_Float16)xv[c];
            Vh[(dq * 4 + c) * VLD + sig] = h;
            Vl[(dq * 4 + c) * VLD + sig] = (_Float16)(xv[c] - (float)h);
        }
    }
    // Q fragments of the wave's query block (B operand: column = query, 8 dims per lane and k-step)
    const int qn = wave * 32 + li;
    const int qtok = toks[qn];
    fwa_h8 qh[2], ql[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = 0.0f;
        if (qtok >= 0) {
            const float4 a0 = *reinterpret_cast<const float4 *>(qk + (size_t)qtok * 2 * C + head * HD + 16 * s + 8 * kg);
            const float4 a1 = *reinterpret_cast<const float4 *>(qk + (size_t)qtok * 2 * C + head * HD + 16 * s + 8 * kg + 4);
            x[0] = a0.x * scale; x[1] = a0.y * scale; x[2] = a0.z * scale; x[3] = a0.w * scale;
            x[4] = a1.x * scale; x[5] = a1.y * scale; x[6] = a1.z * scale; x[7] = a1.w * scale;
        }
        fwa_split8(x, qh[s], ql[s]);
    }
    // S^T = K Q^T, key block by key block
    fwa_acc sc[5];
#pragma unroll
    for (int mb = 0; mb < 5; ++mb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[mb][r] = 0.0f;
        const int ktok = toks[mb * 32 + li];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = 0.0f;
            if (ktok >= 0) {
                const float4 a0 = *reinterpret_cast<const float4 *>(qk + (size_t)ktok * 2 * C + C + head * HD + 16 * s + 8 * kg);
                const float4 a1 = *reinterpret_cast<const float4 *>(qk + (size_t)ktok * 2 * C + C + head * HD + 16 * s + 8 * kg + 4);
                x[0] = a0.x; x[1] = a0.y; x[2] = a0.z; x[3] = a0.w; x[4] = a1.x; x[5] = a1.y; x[6] = a1.z; x[7] = a1.w;
            }
            fwa_h8 kh, kl;
            fwa_split8(x, kh, kl);
            sc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[s], sc[mb], 0, 0, 0);
            sc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[s], sc[mb], 0, 0, 0);
            sc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[s], sc[mb], 0, 0, 0);
        }
    }
    // mask + softmax statistics of the lane's query: register r of key block mb = key 32 mb + 8 (r / 4) + 4 kg + r % 4
    const int qlab = labs[qn];
    float m = -3.0e38f;
#pragma unroll
    for (int mb = 0; mb < 5; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = 32 * mb + 8 * (r >> 2) + 4 * kg + (r & 3);
            const int klab = labs[j];
            float s = sc[mb][r];
            if (klab < 0) s = -3.0e38f;                                    // padding
            else if (klab != qlab) s += -100.0f;                           // the additive shift mask (models/fusion.py:166-167)
            sc[mb][r] = s;
            m = fmaxf(m, s);
        }
    m = fmaxf(m, __shfl_xor(m, 32));
    float sum = 0.0f;
#pragma unroll
    for (int mb = 0; mb < 5; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = sc[mb][r] > -1.0e38f ? __expf(sc[mb][r] - m) : 0.0f;
            sc[mb][r] = p;
            sum += p;
        }
    sum += __shfl_xor(sum, 32);
    __syncthreads();                                                        // the V^T tile is complete
    // O^T = V^T P^T: ten k-steps of 16 keys
    fwa_acc o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < 10; ++ks) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = sc[ks >> 1][8 * (ks & 1) + e];
        fwa_h8 ph, pl;
        fwa_split8(x, ph, pl);
        const fwa_h8 vh = *reinterpret_cast<const fwa_h8 *>(Vh + li * VLD + ks * 16 + kg * 8);
        const fwa_h8 vl = *reinterpret_cast<const fwa_h8 *>(Vl + li * VLD + ks * 16 + kg * 8);
        o = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph, o, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl, o, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph, o, 0, 0, 0);
    }
    if (qn < N) {
        const float inv = 1.0f / sum;
        float *dst = out + (size_t)qtok * C + head * HD;
#pragma unroll
        for (int g = 0; g < 4; ++g)                                         // registers 4 g .. 4 g + 3 = dims 8 g + 4 kg + 0..3
            *reinterpret_cast<float4 *>(dst + 8 * g + 4 * kg) = make_float4(o[4 * g] * inv, o[4 * g + 1] * inv, o[4 * g + 2] * inv, o[4 * g + 3] * inv);
    }
}

extern "C" int oryon_fusion_window_attention_f32(const float *qk, const float *v, int B, int H, int W, int C, int heads, int window, int shift,
                                                 float *out, void *stream)
{
    ORYON_CHECK_ARG(qk && v && out && B >= 0 && B < 65536 && H > 0 && W > 0 && heads >= 1 && heads <= 64 && C == heads * 32);
    ORYON_CHECK_ARG(window == 12 && H % window == 0 && W % window == 0 && shift >= 0 && shift < window);
    ORYON_CHECK_ARG((((uintptr_t)qk | (uintptr_t)v | (uintptr_t)out) & 15) == 0);
    if (B == 0) return ORYON_OK;
    hipLaunchKernelGGL(fusion_window_attention_x3_kernel, dim3((H / window) * (W / window), B, heads), dim3(320), 0, as_stream(stream), qk, v, H, W,
                       C, shift, 1.0f / sqrtf(32.0f), out);
    ORYON_CHECK_LAUNCH();
    return ORYON_OK;
}

// The Swin guidance tower's shifted-window attention (torchvision shifted_window_attention, window 7, head dim 32) on the fp16 matrix pipe for
// the fp32 evaluation - the same transposed scheme as fusion_window_attention_x3_kernel: one workgroup per (window, image, head), two waves of
// 32 queries (49 tokens = 1.5 blocks), S^T = K Q^T + relative-position bias + shift mask, in-lane softmax, O^T = V^T P^T.  Padding tokens (the
// map is padded to a multiple of 7 before the roll) carry q | k | v = the Linear's bias (`pad_qkv`) and take part as keys; their outputs are
// cropped.  The VALU kernel above (one query per lane, fp32 fmaf) took 1.2-2.4 ms per call on the tower's 96 x 96 / 48 x 48 maps.
__global__ __launch_bounds__(128) void swin_window_attention_x3_kernel(const float *__restrict__ qkv, const float *__restrict__ pad_qkv,
                                                                      const float *__restrict__ bias_t, int H, int W, int C, int shift,
                                                                      float *__restrict__ out)
{
    constexpr int WS = SWIN_WS, N = SWIN_N, NP = 64, HD = SWIN_HD;
    constexpr int VLD = 72;                                   // halves per V^T row (144 bytes)
    __shared__ __attribute__((aligned(16))) _Float16 Vh[HD * VLD];
    __shared__ __attribute__((aligned(16))) _Float16 Vl[HD * VLD];
    __shared__ int toks[NP];                                  // >= 0: token, -2: padding token, -1: no token (slots 49..63)
    __shared__ int labs[NP];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, li = lane & 31, kg = lane >> 5;
    const int head = blockIdx.z, b = blockIdx.y;
    const int Hp = (H + WS - 1) / WS * WS, Wp = (W + WS - 1) / WS * WS;
    const int nwx = Wp / WS, wy = blockIdx.x / nwx, wx = blockIdx.x % nwx;
    if (t < NP) {
        int tk = -1, lab = -1;
        if (t < N) {
            const int py = wy * WS + t / WS, px = wx * WS + t % WS;                          // rolled, padded frame
            const int sy = (py + shift) % Hp, sx = (px + shift) % Wp;
            tk = (sy < H && sx < W) ? (b * H + sy) * W + sx : -2;
            lab = 0;
            if (shift > 0) {
                const int by = py < Hp - WS ? 0 : (py < Hp - shift ? 1 : 2);
                const int bx = px < Wp - WS ? 0 : (px < Wp - shift ? 1 : 2);
                lab = by * 3 + bx;
            }
        }
        toks[t] = tk;
        labs[t] = lab;
    }
    for (int i = t; i < HD * (NP - N); i += 128) {                                          // zero the unused key slots of V^T
        const int d = i / (NP - N), sig = fwa_slot(N + i % (NP - N));
        Vh[d * VLD + sig] = (_Float16)0.0f;
        Vl[d * VLD + sig] = (_Float16)0.0f;
    }
    __syncthreads();
    auto row = [&](int tk) { return tk >= 0 ? qkv + (size_t)tk * 3 * C : pad_qkv; };
    for (int i = t; i < N * 8; i += 128) {
        const int j = i >> 3, dq = i & 7;
        const float4 x = *reinterpret_cast<const float4 *>(row(toks[j]) + 2 * C + head * HD + dq * 4);
        const int sig = fwa_slot(j);
        const float xv[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const _Float16 h = (_Float16)xv[c];
            Vh[(dq * 4 + c) * VLD + sig] = h;
            Vl[(dq * 4 + c) * VLD + sig] = (_Float16)(xv[c] - (float)h);
        }
    }
    const int qn = wave * 32 + li;
    const int qtok = toks[qn];
    const float scale = 0.17677669529663687f;                                                // 32^-0.5
    fwa_h8 qh[2], ql[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = 0.0f;
        if (qtok != -1) {
            const float *src = row(qtok) + head * HD + 16 * s + 8 * kg;
            const float4 a0 = *reinterpret_cast<const float4 *>(src), a1 = *reinterpret_cast<const float4 *>(src + 4);
            x[0] = a0.x * scale; x[1] = a0.y * scale; x[2] = a0.z * scale; x[3] = a0.w * scale;
            x[4] = a1.x * scale; x[5] = a1.y * scale; x[6] = a1.z * scale; x[7] = a1.w * scale;
        }
        fwa_split8(x, qh[s], ql[s]);
    }
    fwa_acc sc[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[mb][r] = 0.0f;
        const int ktok = toks[mb * 32 + li];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = 0.0f;
            if (ktok != -1) {
                const float *src = row(ktok) + C + head * HD + 16 * s + 8 * kg;
                const float4 a0 = *reinterpret_cast<const float4 *>(src), a1 = *reinterpret_cast<const float4 *>(src + 4);
                x[0] = a0.x; x[1] = a0.y; x[2] = a0.z; x[3] = a0.w; x[4] = a1.x; x[5] = a1.y; x[6] = a1.z; x[7] = a1.w;
            }
            fwa_h8 kh, kl;
            fwa_split8(x, kh, kl);
            sc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[s], sc[mb], 0, 0, 0);
            sc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[s], sc[mb], 0, 0, 0);
            sc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[s], sc[mb], 0, 0, 0);
        }
    }
    // relative-position bias (bias_t[head][key][query]: 32 consecutive queries per load), shift mask, softmax of the lane's query
    const int qlab = labs[qn];
    const float *bt = bias_t + (size_t)head * N * N + (qn < N ? qn : 0);
    float m = -3.0e38f;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = 32 * mb + 8 * (r >> 2) + 4 * kg + (r & 3);
            const int klab = labs[j];
            float s = -3.0e38f;
            if (klab >= 0) {
                s = sc[mb][r] + bt[j * N];
                if (klab != qlab) s -= 100.0f;
            }
            sc[mb][r] = s;
            m = fmaxf(m, s);
        }
    m = fmaxf(m, __shfl_xor(m, 32));
    float sum = 0.0f;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = sc[mb][r] > -1.0e38f ? expf(sc[mb][r] - m) : 0.0f;
            sc[mb][r] = p;
            sum += p;
        }
    sum += __shfl_xor(sum, 32);
    __syncthreads();
    fwa_acc o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = sc[ks >> 1][8 * (ks & 1) + e];
        fwa_h8 ph, pl;
        fwa_split8(x, ph, pl);
        const fwa_h8 vh = *reinterpret_cast<const fwa_h8 *>(Vh + li * VLD + ks * 16 + kg * 8);
        const fwa_h8 vl = *reinterpret_cast<const fwa_h8 *>(Vl + li * VLD + ks * 16 + kg * 8);
        o = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph, o, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl, o, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph, o, 0, 0, 0);
    }
    if (qtok >= 0) {                                                                         // padding tokens are cropped away
        const float inv = 1.0f / sum;
        float *dst = out + (size_t)qtok * C + head * HD;
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4 *>(dst + 8 * g + 4 * kg) = make_float4(o[4 * g] * inv, o[4 * g + 1] * inv, o[4 * g + 2] * inv, o[4 * g + 3] * inv);
    }
}

extern "C" int oryon_swin_window_attention_f32(const float *qkv, const float *pad_qkv, const float *bias_t, int B, int H, int W, int C,
                                               int heads, int shift, float *out, void *stream)
{
    ORYON_CHECK_ARG(qkv && pad_qkv && bias_t && out && B >= 0 && B < 65536 && H > 0 && W > 0 && heads >= 1 && heads <= 8 && C == heads * SWIN_HD);
    ORYON_CHECK_ARG(shift >= 0 && shift < SWIN_WS);
    ORYON_CHECK_ARG((((uintptr_t)qkv | (uintptr_t)pad_qkv | (uintptr_t)out) & 15) == 0);
    if (B == 0) return ORYON_OK;
    const int nwy = (H + SWIN_WS - 1) / SWIN_WS, nwx = (W + SWIN_WS - 1) / SWIN_WS;
    hipLaunchKernelGGL(swin_window_attention_x3_kernel, dim3(nwy * nwx, B, heads), dim3(128), 0, as_stream(stream), qkv, pad_qkv, bias_t, H, W, C,
                       shift, out);
    ORYON_CHECK_LAUNCH();
    return ORYON_OK;
}

// ------------------------------------------------------------------------------------------------------------------------------------
// a4  the class-aggregation layer of ImageTextFusion's aggregator (models/fusion.py:300-332 ClassTransformerLayer around :240-266
//     LinearAttention) for the reference's single prompt axis (T = 1) on 24 x 24 maps: AvgPool 6x6 -> LayerNorm -> q | k from
//     [tokens | text guidance], v -> elu + 1 linear attention over the T axis -> residual -> LayerNorm -> MLP (128 -> 512 -> 128, ReLU) ->
//     residual -> bilinear upsampling (align_corners) back to 24 x 24 -> residual on the map.  torch runs ~40 launch-bound passes over
//     [B * 16, 1, 128] tensors (0.42 ms per layer at 128 images); here one workgroup per image does all of it in fp32 on the VALU (the
//     layer is 0.2 MFLOP per token: nothing for the matrix pipe), weights streamed from L2, activations in LDS.  NHWC in, NHWC out.
struct FusClassW {
    const float *ln1_w, *ln1_b, *wq, *bq, *wk, *bk, *wv, *bv, *ln2_w, *ln2_b, *w1, *b1, *w2, *b2;
};

__global__ __launch_bounds__(256) void fusion_class_layer_kernel(const float *__restrict__ x, const float *__restrict__ tg, const FusClassW w,
                                                                 float *__restrict__ out)
{
    constexpr int S = 24, C = 128, NT = 16, PC = 6, HID = 512;
    extern __shared__ __attribute__((aligned(16))) float cl[];
    float *ys = cl;                       // [16][128]  pooled tokens, later the layer's token output
    float *xs = ys + NT * C;              // [16][256]  LayerNorm output | text guidance
    float *qs = xs + NT * 2 * C;          // [16][128]
    float *ks = qs + NT * C;              // [16][128]
    float *vs = ks + NT * C;              // [16][128]
    float *hs = vs + NT * C;              // [16][512]
    float *fs = hs + NT * HID;            // [16][4]
    const int t = threadIdx.x, j = t & 127, half = t >> 7, b = blockIdx.x;
    const int wave = t >> 6, lane = t & 63;
    const float *xb = x + (size_t)b * S * S * C;
    // P0: 6 x 6 average pooling (nn.AvgPool2d: sum / 36)
#pragma unroll 1
    for (int c8 = 0; c8 < 8; ++c8) {
        const int cell = half * 8 + c8, cy = cell >> 2, cx = cell & 3;
        float s = 0.0f;
#pragma unroll
        for (int py = 0; py < PC; ++py)
#pragma unroll
            for (int px = 0; px < PC; ++px) s += xb[((cy * PC + py) * S + cx * PC + px) * C + j];
        ys[cell * C + j] = s / 36.0f;
    }
    __syncthreads();
    auto layer_norm = [&](const float *src, const float *lw, const float *lb) {     // 4 tokens per wave, 2 channels per lane -> xs[tok][0..127]
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int tok = wave * 4 + i;
            const float a0 = src[tok * C + lane], a1 = src[tok * C + lane + 64];
            float s = a0 + a1;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
            const float mean = s * (1.0f / C);
            const float d0 = a0 - mean, d1 = a1 - mean;
            float q = d0 * d0 + d1 * d1;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off);
            const float rstd = 1.0f / sqrt_rn(q * (1.0f / C) + 1e-5f);
            xs[tok * 2 * C + lane] = d0 * rstd * lw[lane] + lb[lane];
            xs[tok * 2 * C + lane + 64] = d1 * rstd * lw[lane + 64] + lb[lane + 64];
        }
    };
    layer_norm(ys, w.ln1_w, w.ln1_b);
    if (t < C) {
        const float g = tg[(size_t)b * C + t];
#pragma unroll
        for (int tok = 0; tok < NT; ++tok) xs[tok * 2 * C + C + t] = g;
    }
    __syncthreads();
    // P2: q, k (256 inputs), v (128 inputs) for output channel j and the thread's 8 tokens
    {
        float aq[8], ak[8], av[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { aq[i] = 0.0f; ak[i] = 0.0f; av[i] = 0.0f; }
        const float4 *rq = reinterpret_cast<const float4 *>(w.wq + (size_t)j * 2 * C), *rk = reinterpret_cast<const float4 *>(w.wk + (size_t)j * 2 * C);
        const float4 *rv = reinterpret_cast<const float4 *>(w.wv + (size_t)j * C);
#pragma unroll 2
        for (int k4 = 0; k4 < 2 * C / 4; ++k4) {
            const float4 a = rq[k4], c = rk[k4];
            const float4 d = k4 < C / 4 ? rv[k4] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float4 xv = *reinterpret_cast<const float4 *>(xs + (half * 8 + i) * 2 * C + k4 * 4);
                aq[i] = fmaf(a.w, xv.w, fmaf(a.z, xv.z, fmaf(a.y, xv.y, fmaf(a.x, xv.x, aq[i]))));
                ak[i] = fmaf(c.w, xv.w, fmaf(c.z, xv.z, fmaf(c.y, xv.y, fmaf(c.x, xv.x, ak[i]))));
                av[i] = fmaf(d.w, xv.w, fmaf(d.z, xv.z, fmaf(d.y, xv.y, fmaf(d.x, xv.x, av[i]))));
            }
        }
        const float bq = w.bq[j], bk = w.bk[j], bv = w.bv[j];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int tok = half * 8 + i;
            const float q = aq[i] + bq, k = ak[i] + bk;
            qs[tok * C + j] = q > 0.0f ? q + 1.0f : expf(q);                  // elu(x) + 1
            ks[tok * C + j] = k > 0.0f ? k + 1.0f : expf(k);
            vs[tok * C + j] = av[i] + bv;                                      // / T with T = 1
        }
    }
    __syncthreads();
    // P3: T = 1: out = q (k v^T) / (q . k + eps) = v * (q . k) / (q . k + eps) per head (4 heads of 32)
    if (t < NT * 4) {
        const int tok = t >> 2, h = t & 3;
        float s = 0.0f;
#pragma unroll
        for (int d = 0; d < 32; ++d) s = fmaf(qs[tok * C + h * 32 + d], ks[tok * C + h * 32 + d], s);
        fs[t] = s * (1.0f / (s + 1e-6f));
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int tok = half * 8 + i;
        ys[tok * C + j] += vs[tok * C + j] * fs[tok * 4 + (j >> 5)];
    }
    __syncthreads();
    layer_norm(ys, w.ln2_w, w.ln2_b);
    __syncthreads();
    // P5: hidden = relu(W1 n + b1): outputs j, j + 128, j + 256, j + 384
    {
        float ah[4][8];
#pragma unroll
        for (int o = 0; o < 4; ++o)
#pragma unroll
            for (int i = 0; i < 8; ++i) ah[o][i] = 0.0f;
#pragma unroll 1
        for (int k4 = 0; k4 < C / 4; ++k4) {
            float4 wv4[4];
#pragma unroll
            for (int o = 0; o < 4; ++o) wv4[o] = reinterpret_cast<const float4 *>(w.w1 + (size_t)(j + o * C) * C)[k4];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float4 xv = *reinterpret_cast<const float4 *>(xs + (half * 8 + i) * 2 * C + k4 * 4);
#pragma unroll
                for (int o = 0; o < 4; ++o) ah[o][i] = fmaf(wv4[o].w, xv.w, fmaf(wv4[o].z, xv.z, fmaf(wv4[o].y, xv.y, fmaf(wv4[o].x, xv.x, ah[o][i]))));
            }
        }
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const float bb = w.b1[j + o * C];
#pragma unroll
            for (int i = 0; i < 8; ++i) hs[(half * 8 + i) * HID + j + o * C] = fmaxf(ah[o][i] + bb, 0.0f);
        }
    }
    __syncthreads();
    {
        float ao[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) ao[i] = 0.0f;
        const float4 *r2 = reinterpret_cast<const float4 *>(w.w2 + (size_t)j * HID);
#pragma unroll 2
        for (int k4 = 0; k4 < HID / 4; ++k4) {
            const float4 a = r2[k4];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float4 hv = *reinterpret_cast<const float4 *>(hs + (half * 8 + i) * HID + k4 * 4);
                ao[i] = fmaf(a.w, hv.w, fmaf(a.z, hv.z, fmaf(a.y, hv.y, fmaf(a.x, hv.x, ao[i]))));
            }
        }
        const float bb = w.b2[j];
#pragma unroll
        for (int i = 0; i < 8; ++i) ys[(half * 8 + i) * C + j] += ao[i] + bb;
    }
    __syncthreads();
    // P7: bilinear upsampling 4 x 4 -> 24 x 24 with align_corners = True (torch: src = dst * (in - 1) / (out - 1)) + the residual on the map
    float *ob = out + (size_t)b * S * S * C;
    const float sc = 3.0f / 23.0f;
#pragma unroll 1
    for (int p = half * (S * S / 2); p < (half + 1) * (S * S / 2); ++p) {
        const int py = p / S, px = p % S;
        const float fy = sc * (float)py, fx = sc * (float)px;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < 3 ? 1 : 0), x1 = x0 + (x0 < 3 ? 1 : 0);
        const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.0f - ly, hx = 1.0f - lx;
        const float v = hy * (hx * ys[(y0 * 4 + x0) * C + j] + lx * ys[(y0 * 4 + x1) * C + j]) +
                        ly * (hx * ys[(y1 * 4 + x0) * C + j] + lx * ys[(y1 * 4 + x1) * C + j]);
        ob[(size_t)p * C + j] = xb[(size_t)p * C + j] + v;
    }
}

extern "C" int oryon_fusion_class_layer_f32(const float *x, const float *text_guidance, const oryon_fusion_class_weights_t *wts, int B, float *out,
                                            void *stream)
{
    ORYON_CHECK_ARG(x && text_guidance && wts && out && B >= 0);
    ORYON_CHECK_ARG(wts->ln1_w && wts->ln1_b && wts->wq && wts->bq && wts->wk && wts->bk && wts->wv && wts->bv && wts->ln2_w && wts->ln2_b && wts->w1 &&
                    wts->b1 && wts->w2 && wts->b2);
    ORYON_CHECK_ARG((((uintptr_t)x | (uintptr_t)out | (uintptr_t)wts->wq | (uintptr_t)wts->wk | (uintptr_t)wts->wv | (uintptr_t)wts->w1 | (uintptr_t)wts->w2) & 15) == 0);
    if (B == 0) return ORYON_OK;
    FusClassW w{wts->ln1_w, wts->ln1_b, wts->wq, wts->bq, wts->wk, wts->bk, wts->wv, wts->bv, wts->ln2_w, wts->ln2_b, wts->w1, wts->b1, wts->w2, wts->b2};
    constexpr int dyn = (16 * 128 * 4 + 16 * 256 + 16 * 512 + 64) * 4;
    allow_dynamic_lds(reinterpret_cast<const void *>(&fusion_class_layer_kernel), dyn);
    hipLaunchKernelGGL(fusion_class_layer_kernel, dim3(B), dim3(256), dyn, as_stream(stream), x, text_guidance, w, out);
    ORYON_CHECK_LAUNCH();
    return ORYON_OK;
}
