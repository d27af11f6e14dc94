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
                                                                 float *__restrict__ h_out)
{
    constexpr int CH = LPR == 64 ? LN_F32_CHUNKS : 1;
    constexpr int RPB = 256 / LPR;
    const int lane = threadIdx.x & (LPR - 1);
    const int64_t row = (int64_t)blockIdx.x * RPB + (threadIdx.x / LPR);
    if (row >= rows) return;
    const int chunks = (D + LPR * 4 - 1) / (LPR * 4);
    float4 v[CH];
    float sum = 0.0f;
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
        }
    }
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
#define ORYON_LAUNCH_LN32(LPR)                                                                                                     \
    hipLaunchKernelGGL((add_layernorm_f32_kernel<LPR>), dim3((unsigned)((rows + 256 / LPR - 1) / (256 / LPR))), dim3(256), 0,       \
                       as_stream(stream), x, delta, gamma, beta, rows, D, eps, x_out, h_out)
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

extern "C" int oryon_swin_window_attention_f32(const float *qkv, const float *pad_qkv, const float *bias_t, int B, int H, int W, int C,
                                               int heads, int shift, float *out, void *stream)
{
    ORYON_CHECK_ARG(qkv && pad_qkv && bias_t && out && B >= 0 && H > 0 && W > 0 && heads >= 1 && heads <= 8 && C == heads * SWIN_HD);
    ORYON_CHECK_ARG(shift >= 0 && shift < SWIN_WS);
    ORYON_CHECK_ARG((((uintptr_t)qkv | (uintptr_t)pad_qkv | (uintptr_t)out) & 15) == 0);
    if (B == 0) return ORYON_OK;
    const int nwy = (H + SWIN_WS - 1) / SWIN_WS, nwx = (W + SWIN_WS - 1) / SWIN_WS;
    const size_t lds = (size_t)heads * SWIN_WAVE_FLOATS * sizeof(float);
    allow_dynamic_lds(reinterpret_cast<const void *>(swin_window_attention_kernel<float>), 8 * SWIN_WAVE_FLOATS * (int)sizeof(float));
    hipLaunchKernelGGL(swin_window_attention_kernel<float>, dim3(nwy * nwx, B), dim3(64 * heads), lds, as_stream(stream), qkv, pad_qkv, bias_t,
                       H, W, C, shift, out);
    ORYON_CHECK_LAUNCH();
    return ORYON_OK;
}

// ------------------------------------------------------------------------------------------------------------------------------------
// a4  window attention of ImageTextFusion's guided Swin blocks (models/fusion.py:75-103 WindowAttention.forward inside
//     SwinTransformerBlock.forward :173-213): torch.roll(-shift) + window_partition + softmax(q k^T * hd^-0.5 + shift mask) v +
//     window_reverse + torch.roll(+shift) as ONE kernel on un-windowed tokens - the q | k and v projections are per-token linears, so they
//     run on the natural [B, H, W, .] order and nothing is permuted or copied around the attention (torch: 2 rolls, 2 window copies, 3
//     head transposes, a scale, 2 batched GEMMs, a mask add and a softmax per block).  fp32 VALU arithmetic: one workgroup per (window,
//     image, head), one query per thread, K and V rows of the head in LDS (read as broadcasts), scores in registers.
template <int WS>
__global__ __launch_bounds__(192) void fusion_window_attention_kernel(const float *__restrict__ qk, const float *__restrict__ v, int H, int W,
                                                                      int C, int shift, float scale, float *__restrict__ out)
{
    constexpr int N = WS * WS, HD = 32;
    static_assert(N <= 192, "one query per thread");
    __shared__ __attribute__((aligned(16))) float Ks[N * HD];
    __shared__ __attribute__((aligned(16))) float Vs[N * HD];
    __shared__ int labs[N];
    const int t = threadIdx.x, head = blockIdx.z, b = blockIdx.y;
    const int nwx = W / WS, wy = blockIdx.x / nwx, wx = blockIdx.x % nwx;
    const bool active = t < N;
    float q[HD];
    int label = 0;
    size_t tok = 0;
    if (active) {
        const int py = wy * WS + t / WS, px = wx * WS + t % WS;             // position in the rolled frame
        const int sy = (py + shift) % H, sx = (px + shift) % W;             // torch.roll(x, -shift): rolled[p] = x[p + shift]
        if (shift > 0) {                                                    // the regions of SwinTransformerBlock's img_mask (:155-163)
            const int by = py < H - WS ? 0 : (py < H - shift ? 1 : 2);
            const int bx = px < W - WS ? 0 : (px < W - shift ? 1 : 2);
            label = by * 3 + bx;
        }
        labs[t] = label;
        tok = ((size_t)b * H + sy) * W + sx;
        const float4 *qs = reinterpret_cast<const float4 *>(qk + tok * 2 * C + head * HD);
        const float4 *ks = reinterpret_cast<const float4 *>(qk + tok * 2 * C + C + head * HD);
        const float4 *vs = reinterpret_cast<const float4 *>(v + tok * C + head * HD);
#pragma unroll
        for (int e = 0; e < HD / 4; ++e) {
            const float4 a = qs[e];
            q[4 * e + 0] = a.x * scale; q[4 * e + 1] = a.y * scale; q[4 * e + 2] = a.z * scale; q[4 * e + 3] = a.w * scale;
            reinterpret_cast<float4 *>(Ks + t * HD)[e] = ks[e];
            reinterpret_cast<float4 *>(Vs + t * HD)[e] = vs[e];
        }
    }
    __syncthreads();
    if (!active) return;
    float s[N];
    float m = -3.0e38f;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
        for (int e = 0; e < HD / 4; ++e) {
            const float4 kk = reinterpret_cast<const float4 *>(Ks + j * HD)[e];
            a0 = fmaf(q[4 * e + 0], kk.x, a0);
            a1 = fmaf(q[4 * e + 1], kk.y, a1);
            a0 = fmaf(q[4 * e + 2], kk.z, a0);
            a1 = fmaf(q[4 * e + 3], kk.w, a1);
        }
        float sc = a0 + a1;
        if (shift > 0 && labs[j] != label) sc += -100.0f;                   // the additive mask of :166-167
        s[j] = sc;
        m = fmaxf(m, sc);
    }
    float o[HD];
#pragma unroll
    for (int e = 0; e < HD; ++e) o[e] = 0.0f;
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const float p = __expf(s[j] - m);
        sum += p;
#pragma unroll
        for (int e = 0; e < HD / 4; ++e) {
            const float4 vv = reinterpret_cast<const float4 *>(Vs + j * HD)[e];
            o[4 * e + 0] = fmaf(p, vv.x, o[4 * e + 0]);
            o[4 * e + 1] = fmaf(p, vv.y, o[4 * e + 1]);
            o[4 * e + 2] = fmaf(p, vv.z, o[4 * e + 2]);
            o[4 * e + 3] = fmaf(p, vv.w, o[4 * e + 3]);
        }
    }
    const float inv = 1.0f / sum;
    float4 *dst = reinterpret_cast<float4 *>(out + tok * C + head * HD);
#pragma unroll
    for (int e = 0; e < HD / 4; ++e) dst[e] = make_float4(o[4 * e] * inv, o[4 * e + 1] * inv, o[4 * e + 2] * inv, o[4 * e + 3] * inv);
}

extern "C" int oryon_fusion_window_attention_f32(const float *qk, const float *v, int B, int H, int W, int C, int heads, int window, int shift,
                                                 float *out, void *stream)
{
    ORYON_CHECK_ARG(qk && v && out && B >= 0 && H > 0 && W > 0 && heads >= 1 && C == heads * 32);
    ORYON_CHECK_ARG(window == 12 && H % window == 0 && W % window == 0 && shift >= 0 && shift < window);
    ORYON_CHECK_ARG((((uintptr_t)qk | (uintptr_t)v | (uintptr_t)out) & 15) == 0);
    if (B == 0) return ORYON_OK;
    hipLaunchKernelGGL(fusion_window_attention_kernel<12>, dim3((H / window) * (W / window), B, heads), dim3(192), 0, as_stream(stream), qk, v, H, W,
                       C, shift, 1.0f / sqrtf(32.0f), out);
    ORYON_CHECK_LAUNCH();
    return ORYON_OK;
}
