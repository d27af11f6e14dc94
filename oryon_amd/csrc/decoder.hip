// a5  StandardDecoder.forward (models/decoder.py:82-108) as hand-written gfx950 kernels: the three Up blocks (ConvTranspose2d 2x2 s2 ->
// cat guidance -> (conv3x3 - GroupNorm - ReLU) x 2, models/decoder.py:9-42), the two guidance projections (conv3x3 + ReLU, :66-72) and
// the 3x3 head (:80), fp32 tensors in and out.
//
// Arithmetic: every convolution is an implicit GEMM on the fp16 matrix pipe with error-compensated operands, like the towers' linears
// (gemm_x3.hip): x = x_hi + x_lo, w = w_hi + w_lo (two float16 each), product = x_hi*w_hi + x_hi*w_lo + x_lo*w_hi accumulated in fp32 by
// three v_mfma_f32_32x32x16_f16 - ~2^-22 relative per product, fp32-grade results (the G5 golden of the imported reference holds to
// 1e-4, tests/test_gpu_decoder.py) at about three times the rate of the fp32-input MFMAs the library convolutions use.
//
// Data layout in HBM: activations between the layers are fp32 NHWC (the 32 channels of a K slab of one pixel are one 128-byte run, a
// halo row of a tile one contiguous piece); the module's inputs (x, the Swin guidance maps) are read in place as NCHW, the descriptor map
// goes out as NCHW fp32 (what Oryon.forward returns and the matcher's K0 reads).  Nothing is materialised that the reference's graph
// does not need: no im2col, the concatenations are channel ranges of one buffer that the up-convolution and the guidance projection
// write side by side, GroupNorm is two numbers per (image, channel) applied (with the ReLU) by the NEXT layer's tile loader, its
// statistics are per-tile partial sums written by the producing convolution's epilogue and reduced in a fixed order (no atomics: the
// descriptors are bit-reproducible run to run).
//
// conv3x3 kernel (dec_conv3x3_kernel): one workgroup = one 16 x 16 pixel tile of one image, all output channels (32 or 64).  Per 32-channel
// K slab the 18 x 18 halo tile is split into hi / lo halves and parked in LDS ([pixel][32 halves], 80-byte pixel stride: the 16-byte A
// reads of 8 consecutive pixels land on 8 different bank quads); a wave owns 4 rows x 16 columns = two 32-pixel M blocks and walks the
// 9 taps x 2 k-steps, its B fragments (pre-packed hi / lo weight images, fragment order, L2-resident) come straight from global memory.
// ~52 KB of LDS per workgroup: three workgroups per CU, the slab loads of one overlap the MFMAs of the others.
#include "common.h"
#include <hip/hip_fp16.h>

namespace oryon {

typedef _Float16 dh8 __attribute__((ext_vector_type(8)));
typedef _Float16 dh4 __attribute__((ext_vector_type(4)));
typedef float dacc16 __attribute__((ext_vector_type(16)));

constexpr int DEC_HALO = 18;                     // 16 + 2
constexpr int DEC_PIX = DEC_HALO * DEC_HALO;     // 324 halo pixels per tile
constexpr int DEC_PSTRIDE = 80;                  // bytes per pixel and plane: 32 halves + 16 bytes of padding
constexpr int DEC_PLANE = DEC_PIX * DEC_PSTRIDE; // 25920 bytes per plane (hi, lo)
constexpr int DEC_RING = 2;                      // steps of B fragments in flight per wave (divides 18)

struct DecConv {
    const float *in;          // NHWC [n, H, W, in_cstride] (channels in_coff ..) or NCHW [n, cin, H, W]
    const float *affine;      // [n, cin, 2] (a, b): the loader applies relu(a x + b) - the previous layer's GroupNorm + ReLU - or NULL
    const dh8 *wimg;          // packed weights, see dec_pack_conv3x3_kernel
    const float *bias;        // [cout] or NULL
    float *out;               // NHWC [n, H, W, out_cstride], channels out_coff .. out_coff + cout - 1
    float *stats;             // [n, tiles, cout / 16, 2] partial (sum, sum of squares) of the raw outputs, or NULL
    int H, W, cin, in_cstride, in_coff, out_cstride, out_coff, cout, relu;
    unsigned *range_flag;     // per-device flag word (common.h: x3_range_flag), or NULL
};

static __device__ __forceinline__ void split_h(float v, _Float16 &hi, _Float16 &lo)
{
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
}

// Weight image of a 3x3 convolution (torch layout w[cout, cin, 3, 3]): fragment ((((slab * 9 + tap) * 2 + kstep) * NB + nb) * 2 + part),
// 64 lanes x 8 halves each: lane l holds output channel nb * 32 + (l & 31), input channels slab * 32 + kstep * 16 + (l >> 5) * 8 + 0..7 -
// the B operand of v_mfma_f32_32x32x16_f16.  part 0 = hi halves, 1 = lo halves.  Channels beyond cout are zero columns.
__global__ void dec_pack_conv3x3_kernel(const float *__restrict__ w, int cout, int cin, int NB, _Float16 *__restrict__ img, int64_t total, int T = 9)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int e = idx & 7, lane = (idx >> 3) & 63, part = (idx >> 9) & 1;
    int64_t rest = idx >> 10;
    const int nb = rest % NB; rest /= NB;
    const int ks = rest & 1; rest >>= 1;
    const int tap = rest % T;
    const int s = rest / T;
    const int n = nb * 32 + (lane & 31), c = s * 32 + ks * 16 + (lane >> 5) * 8 + e;
    float v = 0.0f;
    if (n < cout && c < cin) v = w[((size_t)n * cin + c) * T + tap];
    _Float16 hi, lo;
    split_h(v, hi, lo);
    img[idx] = part ? lo : hi;
}

// Weight image of ConvTranspose2d(k = 2, s = 2) (torch layout w[cin, cout, 2, 2]): fragment ((nb * KS + kstep) * 2 + part); N block nb
// covers output position dydx = nb / nbp (dy = dydx >> 1, dx = dydx & 1), channels (nb % nbp) * 32 + (l & 31); K = input channels.
__global__ void dec_pack_upconv_kernel(const float *__restrict__ w, int cin, int cout, int nbp, _Float16 *__restrict__ img, int64_t total)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int e = idx & 7, lane = (idx >> 3) & 63, part = (idx >> 9) & 1;
    int64_t rest = idx >> 10;
    const int KS = cin / 16;
    const int ks = rest % KS;
    const int nb = rest / KS;
    const int dydx = nb / nbp, co = (nb % nbp) * 32 + (lane & 31), c = ks * 16 + (lane >> 5) * 8 + e;
    float v = 0.0f;
    if (co < cout) v = w[((size_t)c * cout + co) * 4 + dydx];
    _Float16 hi, lo;
    split_h(v, hi, lo);
    img[idx] = part ? lo : hi;
}

// WREG (single-slab convolutions, cin = 32, NB = 1 - decoder3's two and decoder2's second: half of the module's flops): the 36 weight
// fragments live in 144 registers of every wave and the workgroups are persistent (grid = 2 per CU, items strided).  Fetched per tile
// like the multi-slab variant does, the weights were 3.5x the bytes of the tile itself (4 waves x 36 KB from L2 per 41 KB tile).
template <int NB, bool IN_NCHW, bool IN_GN, bool WREG = false>
__global__ __launch_bounds__(256, (WREG ? 2 : (NB == 1 ? 3 : 2))) void dec_conv3x3_kernel(const DecConv a, const int tiles, const int total)
{
    static_assert(!WREG || (NB == 1 && !IN_NCHW), "register-resident weights: one slab, one N block");
    __shared__ __attribute__((aligned(16))) char lds[2 * DEC_PLANE];
    __shared__ float red[4][NB * 2][2];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int tiles_x = a.W / 16;
    const int slabs = WREG ? 1 : a.cin / 32;
    const int li = lane & 31, kg = lane >> 5;
    // byte offset of the lane's A piece for M block mb at tap (0, 0), k-step 0
    int a_off[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) a_off[mb] = ((4 * wave + 2 * mb + (li >> 4)) * DEC_HALO + (li & 15)) * DEC_PSTRIDE + kg * 16;

    // B fragments: a register ring RING steps deep (step = (tap, k-step), 18 per slab, consecutive in the image).  Fetched straight from
    // global memory (L2-resident) they need ~0.5 us; one step ahead - what the compiler schedules by itself - left every step waiting
    // for its weights (decoder3's convolutions: 0.49 ms, MFMA pipe 20 % busy).
    constexpr int RING = DEC_RING;
    const dh8 *wf0 = a.wimg + lane;
    dh8 wb[WREG ? 18 : 1][2];
    if constexpr (WREG) {
#pragma unroll
        for (int j = 0; j < 18; ++j) {
            wb[j][0] = wf0[(j * 2 + 0) * 64];
            wb[j][1] = wf0[(j * 2 + 1) * 64];
        }
    }

    for (int item = blockIdx.x; item < total; item += gridDim.x) {
    const int img = item / tiles, tile = item % tiles;
    const int y0 = (tile / tiles_x) * 16, x0 = (tile % tiles_x) * 16;
    dacc16 acc[2][NB];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.0f;
    dh8 rb[WREG ? 1 : RING][NB][2];
    if constexpr (!WREG) {
#pragma unroll
        for (int j = 0; j < RING; ++j)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                rb[j][nb][0] = wf0[((j * NB + nb) * 2 + 0) * 64];
                rb[j][nb][1] = wf0[((j * NB + nb) * 2 + 1) * 64];
            }
    }

    for (int s = 0; s < slabs; ++s) {
        __syncthreads();                                         // the previous slab's readers are done
        if constexpr (!IN_NCHW) {
            // 8 threads per pixel (4 channels each), 32 pixels per pass
            const int p8 = t >> 3, cq = t & 7;
            float4 v[11];
            float4 fa = make_float4(1.f, 1.f, 1.f, 1.f), fb = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (IN_GN) {
                const float *af = a.affine + ((size_t)img * a.cin + s * 32 + cq * 4) * 2;
                const float4 q0 = *reinterpret_cast<const float4 *>(af), q1 = *reinterpret_cast<const float4 *>(af + 4);
                fa = make_float4(q0.x, q0.z, q1.x, q1.z);
                fb = make_float4(q0.y, q0.w, q1.y, q1.w);
            }
            bool ok[11];
#pragma unroll
            for (int pp = 0; pp < 11; ++pp) {
                const int pix = pp * 32 + p8;
                const int hy = pix / DEC_HALO, hx = pix % DEC_HALO;
                const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
                ok[pp] = pix < DEC_PIX && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                v[pp] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok[pp]) v[pp] = *reinterpret_cast<const float4 *>(a.in + (((size_t)img * a.H + gy) * a.W + gx) * a.in_cstride + a.in_coff + s * 32 + cq * 4);
            }
#pragma unroll
            for (int pp = 0; pp < 11; ++pp) {
                const int pix = pp * 32 + p8;
                if (pix >= DEC_PIX) continue;
                float4 x = v[pp];
                if constexpr (IN_GN) {
                    if (ok[pp]) {                                // the zero padding is applied to the NORMALISED tensor
                        x.x = fmaxf(fmaf(x.x, fa.x, fb.x), 0.0f);
                        x.y = fmaxf(fmaf(x.y, fa.y, fb.y), 0.0f);
                        x.z = fmaxf(fmaf(x.z, fa.z, fb.z), 0.0f);
                        x.w = fmaxf(fmaf(x.w, fa.w, fb.w), 0.0f);
                    }
                }
                dh4 hi, lo;
                _Float16 h, l;
                split_h(x.x, h, l); hi[0] = h; lo[0] = l;
                split_h(x.y, h, l); hi[1] = h; lo[1] = l;
                split_h(x.z, h, l); hi[2] = h; lo[2] = l;
                split_h(x.w, h, l); hi[3] = h; lo[3] = l;
                *reinterpret_cast<dh4 *>(lds + pix * DEC_PSTRIDE + cq * 8) = hi;
                *reinterpret_cast<dh4 *>(lds + DEC_PLANE + pix * DEC_PSTRIDE + cq * 8) = lo;
            }
        } else {
            // NCHW input: element e = (channel, halo pixel), the pixel index fastest: runs of 18 floats of a plane row
            static_assert(!IN_NCHW || !IN_GN, "the module's NCHW inputs are raw tensors");
            const float *base = a.in + ((size_t)img * a.cin + s * 32) * a.H * a.W;
#pragma unroll 1
            for (int p0 = 0; p0 < 41; p0 += 8) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int e = (p0 + j) * 256 + t;
                    const int c = e / DEC_PIX, pix = e % DEC_PIX;
                    const int gy = y0 - 1 + pix / DEC_HALO, gx = x0 - 1 + pix % DEC_HALO;
                    v[j] = 0.0f;
                    if (e < 32 * DEC_PIX && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) v[j] = base[((size_t)c * a.H + gy) * a.W + gx];
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int e = (p0 + j) * 256 + t;
                    if (e >= 32 * DEC_PIX) continue;
                    const int c = e / DEC_PIX, pix = e % DEC_PIX;
                    _Float16 h, l;
                    split_h(v[j], h, l);
                    *reinterpret_cast<_Float16 *>(lds + pix * DEC_PSTRIDE + c * 2) = h;
                    *reinterpret_cast<_Float16 *>(lds + DEC_PLANE + pix * DEC_PSTRIDE + c * 2) = l;
                }
            }
        }
        __syncthreads();
        const dh8 *wf = wf0 + (size_t)(s * 18 + RING) * (NB * 2 * 64);          // the fragments RING steps ahead of this slab's step 0
#pragma unroll
        for (int step = 0; step < 18; ++step) {
            const int tap = step >> 1, ks = step & 1, slot = step % RING;
            const int toff = ((tap / 3) * DEC_HALO + (tap % 3)) * DEC_PSTRIDE;
            // the three products of an accumulator are issued 2 NB MFMAs apart (every accumulator's chain is dependent: back to back
            // they wait for each other's passes)
            dh8 ah[2], al[2];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                ah[mb] = *reinterpret_cast<const dh8 *>(lds + a_off[mb] + toff + ks * 32);
                al[mb] = *reinterpret_cast<const dh8 *>(lds + DEC_PLANE + a_off[mb] + toff + ks * 32);
            }
#pragma unroll
            for (int part = 0; part < 3; ++part)
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        const dh8 bh = WREG ? wb[WREG ? step : 0][0] : rb[WREG ? 0 : slot][nb][0];
                        const dh8 bl = WREG ? wb[WREG ? step : 0][1] : rb[WREG ? 0 : slot][nb][1];
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(part == 0 ? al[mb] : ah[mb], part == 1 ? bl : bh, acc[mb][nb], 0, 0, 0);
                    }
            if constexpr (WREG) continue;
            // refill the slot with step + RING (the image is padded by RING steps: the last slab reads zeros nobody uses)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                rb[WREG ? 0 : slot][nb][0] = wf[((step * NB + nb) * 2 + 0) * 64];
                rb[WREG ? 0 : slot][nb][1] = wf[((step * NB + nb) * 2 + 1) * 64];
            }
        }
    }

    // epilogue: accumulator element r of a lane = output channel nb * 32 + (lane & 31), pixel m = 8 (r / 4) + 4 (lane >> 5) + r % 4 of the
    // M block (row m >> 4, column m & 15)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int co = nb * 32 + li;
        const float bv = (a.bias && co < a.cout) ? a.bias[co] : 0.0f;
        float s1 = 0.0f;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = 8 * (r >> 2) + 4 * kg + (r & 3);
                const int gy = y0 + 4 * wave + 2 * mb + (m >> 4), gx = x0 + (m & 15);
                float v = acc[mb][nb][r];
                s1 += v;
                v += bv;
                if (a.relu) v = fmaxf(v, 0.0f);
                // (WREG: cout == 32 == the block: no predicate - with one the compiler waits for every store's acknowledgement (s_waitcnt
                //  vmcnt(0) in front of the next predicated store), 32 memory round trips per tile and wave: round 6)
                if (WREG || co < a.cout) a.out[(((size_t)img * a.H + gy) * a.W + gx) * a.out_cstride + a.out_coff + co] = v;
            }
        if (a.stats) {
            // GroupNorm statistics without cancellation: (sum, sum of squared deviations from the WAVE's own group mean) per wave, merged
            // below and in dec_gn_affine_kernel with the parallel-variance formula - E[x^2] - mean^2 in fp32 loses the variance as soon as
            // |mean| >> std (trained activations with a DC offset).  group = 16 channels = the 16 lanes li & 16 .. of both k-group halves
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) s1 += __shfl_xor(s1, off);
            s1 += __shfl_xor(s1, 32);
            const float mu = s1 * (1.0f / 1024.0f);                // 16 channels x 64 pixels of this wave's two M blocks
            float s2 = 0.0f;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float dv = acc[mb][nb][r] - mu;
                    s2 = fmaf(dv, dv, s2);
                }
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) s2 += __shfl_xor(s2, off);
            s2 += __shfl_xor(s2, 32);
            if ((lane & 47) == 0) {                              // lanes 0 and 16
                red[wave][nb * 2 + (lane >> 4)][0] = s1;
                red[wave][nb * 2 + (lane >> 4)][1] = s2;
            }
        }
        if constexpr (!WREG) if (a.range_flag && co < a.cout) {
            // range flag (common.h): the pre-activation values of this block once more, after the stores.  Not in the single-slab
            // persistent instantiations (WREG: 36 weight fragments resident, 248-252 registers - the check spilled 8-15 of them): their
            // inputs are the up-convolution's outputs (checked there) and GroupNorm-normalised maps, i.e. in range by construction,
            // and their raw outputs are only ever consumed through the next GroupNorm (fp32 statistics).
            unsigned mg = 0u;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) mg = max(mg, x3_mag(acc[mb][nb][r] + bv));
            x3_raise(a.range_flag, mg);
        }
    }
    if (a.stats) {
        __syncthreads();
        if (t < NB * 2) {
            // the tile's (sum, M2) from its four waves' (sum, M2): M2 = sum M2_w + 1024 sum (mean_w - mean_tile)^2
            const int g = t;
            const float sum = (red[0][g][0] + red[1][g][0]) + (red[2][g][0] + red[3][g][0]);
            const float mt = sum * (1.0f / 4096.0f);
            float m2 = (red[0][g][1] + red[1][g][1]) + (red[2][g][1] + red[3][g][1]);
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const float dm = red[w][g][0] * (1.0f / 1024.0f) - mt;
                m2 = fmaf(1024.0f * dm, dm, m2);
            }
            float2 o;
            o.x = sum;
            o.y = m2;
            *reinterpret_cast<float2 *>(a.stats + (((size_t)img * tiles + tile) * (NB * 2) + g) * 2) = o;
        }
    }
    }   // items
}

// GroupNorm(cout / 16 groups, eps) folded into y = a x + b per (image, channel): partial sums of the tiles added in tile order in double
__global__ __launch_bounds__(64) void dec_gn_affine_kernel(const float *__restrict__ stats, int n_img, int tiles, int NG, const float *__restrict__ gamma,
                                                           const float *__restrict__ beta, double eps, double inv_count, float *__restrict__ affine)
{
    // one wave per (image, group): lane l adds tiles l, l + 64, .. in order, then a fixed butterfly - the same sum every run
    const int idx = blockIdx.x, lane = threadIdx.x;
    const int img = idx / NG, g = idx % NG;
    // tile partials are (sum, M2 about the tile's own mean) of 4096 values each; merged in double with the parallel-variance formula
    double s1 = 0.0;
    for (int tl = lane; tl < tiles; tl += 64) s1 += (double)stats[(((size_t)img * tiles + tl) * NG + g) * 2];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s1 += __shfl_xor(s1, off);
    const double mean = s1 * inv_count;
    double m2 = 0.0;
    for (int tl = lane; tl < tiles; tl += 64) {
        const float2 q = *reinterpret_cast<const float2 *>(stats + (((size_t)img * tiles + tl) * NG + g) * 2);
        const double dm = (double)q.x * (1.0 / 4096.0) - mean;
        m2 += (double)q.y + 4096.0 * dm * dm;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m2 += __shfl_xor(m2, off);
    double var = m2 * inv_count;
    var = var < 0.0 ? 0.0 : var;
    const double rstd = 1.0 / sqrt(var + eps);
    if (lane < 16) {
        const int ch = g * 16 + lane;
        const double sc = (double)gamma[ch] * rstd;
        affine[((size_t)img * NG * 16 + ch) * 2 + 0] = (float)sc;
        affine[((size_t)img * NG * 16 + ch) * 2 + 1] = (float)((double)beta[ch] - mean * sc);
    }
}

// ConvTranspose2d(cin, cout, kernel 2, stride 2) + bias as a GEMM per input pixel: a wave owns 32 consecutive input pixels of one image
// (its A fragments - all of K - live in registers, read straight from global memory), and walks the 4 * nbp N blocks (output position,
// 32 channels); each accumulator row is one 128-byte run of the NHWC output.
template <int KS, bool IN_NCHW, bool IN_GN>
__global__ __launch_bounds__(256) void dec_upconv_kernel(const float *__restrict__ in, const float *__restrict__ affine, int Hin, int Win,
                                                         int in_cstride, const dh8 *__restrict__ wimg, const float *__restrict__ bias,
                                                         float *__restrict__ out, int out_cstride, int cout, int nbp, unsigned *__restrict__ range_flag)
{
    constexpr int CIN = KS * 16;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, kg = lane >> 5;
    const int img = blockIdx.y;
    const int p0 = (blockIdx.x * 4 + wave) * 32;
    if (p0 >= Hin * Win) return;
    const int p = p0 + li;
    dh8 ah[KS], al[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        float x[8];
        const int c0 = ks * 16 + kg * 8;
        if constexpr (IN_NCHW) {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = in[((size_t)img * CIN + c0 + e) * Hin * Win + p];
        } else {
            const float *src = in + ((size_t)img * Hin * Win + p) * in_cstride + c0;
            const float4 q0 = *reinterpret_cast<const float4 *>(src), q1 = *reinterpret_cast<const float4 *>(src + 4);
            x[0] = q0.x; x[1] = q0.y; x[2] = q0.z; x[3] = q0.w; x[4] = q1.x; x[5] = q1.y; x[6] = q1.z; x[7] = q1.w;
        }
        if constexpr (IN_GN) {
            const float *af = affine + ((size_t)img * CIN + c0) * 2;
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const float4 q = *reinterpret_cast<const float4 *>(af + e * 2);
                x[e] = fmaxf(fmaf(x[e], q.x, q.y), 0.0f);
                x[e + 1] = fmaxf(fmaf(x[e + 1], q.z, q.w), 0.0f);
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            _Float16 h, l;
            split_h(x[e], h, l);
            ah[ks][e] = h;
            al[ks][e] = l;
        }
    }
    const int Hout = 2 * Hin, Wout = 2 * Win;
    unsigned x3m = 0u;
    for (int nb = 0; nb < 4 * nbp; ++nb) {
        dacc16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        const dh8 *wf = wimg + (size_t)nb * KS * 2 * 64 + lane;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const dh8 bh = wf[(ks * 2 + 0) * 64], bl = wf[(ks * 2 + 1) * 64];
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ks], bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bh, acc, 0, 0, 0);
        }
        const int dydx = nb / nbp, co = (nb % nbp) * 32 + li;
        if (co >= cout) continue;
        const float bv = bias ? bias[co] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = 8 * (r >> 2) + 4 * kg + (r & 3);
            const int q = p0 + m, y = q / Win, x = q % Win;
            out[(((size_t)img * Hout + 2 * y + (dydx >> 1)) * Wout + 2 * x + (dydx & 1)) * out_cstride + co] = acc[r] + bv;
            x3m = max(x3m, x3_mag(acc[r] + bv));
        }
    }
    x3_raise(range_flag, x3m);
}

// Last layer: GroupNorm + ReLU of decoder3's second convolution, the descriptor map as NCHW fp32 (models/decoder.py:98: the clone that
// becomes `featmap`) and the 3x3 head on the same values (:99-100), one 8 x 32 pixel tile per workgroup.  The head is 288 fp32 fmaf per
// pixel in a fixed order (channel-major inside a tap, taps in row order): plain VALU, exact fp32 products.
__global__ __launch_bounds__(256) void dec_final_kernel(const float *__restrict__ in, const float *__restrict__ affine, int H, int W,
                                                        const float *__restrict__ head_w, const float *__restrict__ head_b,
                                                        float *__restrict__ featmap, float *__restrict__ logits)
{
    constexpr int TH = 8, TW = 32, HW_ = TW + 2, NPIX = (TH + 2) * HW_;   // 340 halo pixels
    __shared__ float v[NPIX * 33];
    __shared__ float hw[288];
    const int t = threadIdx.x;
    const int tiles_x = W / TW;
    const int img = blockIdx.y, y0 = (blockIdx.x / tiles_x) * TH, x0 = (blockIdx.x % tiles_x) * TW;
    for (int i = t; i < 288; i += 256) hw[(i % 9) * 32 + i / 9] = head_w[i];       // [tap][channel] from torch's [1, 32, 3, 3]
    const int p8 = t >> 3, cq = t & 7;
    const float *af = affine + ((size_t)img * 32 + cq * 4) * 2;
    const float4 q0 = *reinterpret_cast<const float4 *>(af), q1 = *reinterpret_cast<const float4 *>(af + 4);
#pragma unroll
    for (int pp = 0; pp < 11; ++pp) {
        const int pix = pp * 32 + p8;
        if (pix >= NPIX) continue;
        const int gy = y0 - 1 + pix / HW_, gx = x0 - 1 + pix % HW_;
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
            x = *reinterpret_cast<const float4 *>(in + (((size_t)img * H + gy) * W + gx) * 32 + cq * 4);
            x.x = fmaxf(fmaf(x.x, q0.x, q0.y), 0.0f);
            x.y = fmaxf(fmaf(x.y, q0.z, q0.w), 0.0f);
            x.z = fmaxf(fmaf(x.z, q1.x, q1.y), 0.0f);
            x.w = fmaxf(fmaf(x.w, q1.z, q1.w), 0.0f);
        }
        float *d = v + pix * 33 + cq * 4;
        d[0] = x.x; d[1] = x.y; d[2] = x.z; d[3] = x.w;
    }
    __syncthreads();
    const int row = t >> 5, col = t & 31;
    const int gy = y0 + row, gx = x0 + col;
    const float *ctr = v + ((row + 1) * HW_ + col + 1) * 33;
#pragma unroll 8
    for (int c = 0; c < 32; ++c) featmap[(((size_t)img * 32 + c) * H + gy) * W + gx] = ctr[c];
    float acc = head_b[0];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const float *src = v + ((row + tap / 3) * HW_ + col + tap % 3) * 33;
#pragma unroll 8
        for (int c = 0; c < 32; ++c) acc = fmaf(src[c], hw[tap * 32 + c], acc);
    }
    logits[((size_t)img * H + gy) * W + gx] = acc;
}

// ------------------------------------------------------------------------------------------------------------------------------------
// a4  the two convolutions of ImageTextFusion on its 24 x 24 maps (models/fusion.py:562 / :595-600 conv1 = 7x7 on the cost volume, 80 -> 128;
//     :566-570 / :614-615 guidance_projection = 3x3 + ReLU on the Swin map, 512 -> 128), same arithmetic as dec_conv3x3_kernel.  The maps are
//     too small to tile (576 pixels = 18 M blocks): one workgroup = one WHOLE image x 64 output channels, nine waves of two M blocks, the
//     zero-padded halo map of a 32-channel slab as hi / lo planes in LDS (108 KB for 3x3, 144 KB for 7x7), weights per (tap, k-step) straight
//     from global memory one tap ahead (all nine waves ask for the same fragments: L1 hits).  NHWC in, NHWC out.
struct FusConv {
    const float *in;          // [n, 24, 24, cin]
    const dh8 *wimg;          // fragments ((((slab * T + tap) * 2 + kstep) * (cout / 32) + nb) * 2 + part)
    const float *bias;        // [cout] or NULL
    float *out;               // [n, 24, 24, cout]
    int cin, cout, relu;
    unsigned *range_flag;     // per-device flag word (common.h), or NULL
};

template <int KS>
__global__ __launch_bounds__(576) void fus_conv24_kernel(const FusConv a)
{
    constexpr int S = 24, PAD = KS / 2, HWP = S + 2 * PAD, HP = HWP * HWP, T = KS * KS;
    constexpr int PLANE = HP * DEC_PSTRIDE;
    extern __shared__ __attribute__((aligned(16))) char flds[];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, li = lane & 31, kg = lane >> 5;
    const int groups = a.cout / 64, nbt = a.cout / 32;
    const int img = blockIdx.x / groups, nb0 = (blockIdx.x % groups) * 2;
    const int slabs = (a.cin + 31) / 32;
    dacc16 acc[2][2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.0f;
    int a_off[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        const int p = (wave * 2 + mb) * 32 + li;
        a_off[mb] = ((p / S) * HWP + p % S) * DEC_PSTRIDE + kg * 16;        // halo coordinates of the pixel at tap (0, 0)
    }
    const dh8 *wf = a.wimg + lane;
    auto frag = [&](int step, int nb, int part) { return wf[(((size_t)step * nbt + nb0 + nb) * 2 + part) * 64]; };
    for (int s = 0; s < slabs; ++s) {
        __syncthreads();
        {
            const int p8 = t >> 3, cq = t & 7;
            const bool cok = s * 32 + cq * 4 < a.cin;
            constexpr int PASSES = (HP + 71) / 72;
#pragma unroll 1
            for (int p0 = 0; p0 < PASSES; p0 += 5) {
                float4 v[5];
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    const int pix = (p0 + j) * 72 + p8;
                    const int gy = pix / HWP - PAD, gx = pix % HWP - PAD;
                    v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (p0 + j < PASSES && pix < HP && cok && gy >= 0 && gy < S && gx >= 0 && gx < S)
                        v[j] = *reinterpret_cast<const float4 *>(a.in + (((size_t)img * S + gy) * S + gx) * a.cin + s * 32 + cq * 4);
                }
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    const int pix = (p0 + j) * 72 + p8;
                    if (p0 + j >= PASSES || pix >= HP) continue;
                    dh4 hi, lo;
                    _Float16 h, l;
                    split_h(v[j].x, h, l); hi[0] = h; lo[0] = l;
                    split_h(v[j].y, h, l); hi[1] = h; lo[1] = l;
                    split_h(v[j].z, h, l); hi[2] = h; lo[2] = l;
                    split_h(v[j].w, h, l); hi[3] = h; lo[3] = l;
                    *reinterpret_cast<dh4 *>(flds + pix * DEC_PSTRIDE + cq * 8) = hi;
                    *reinterpret_cast<dh4 *>(flds + PLANE + pix * DEC_PSTRIDE + cq * 8) = lo;
                }
            }
        }
        __syncthreads();
        dh8 cur[2][2][2], nxt[2][2][2];                        // [k-step][N block][hi | lo] of one tap
        const int step0 = s * T * 2;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                cur[ks][nb][0] = frag(step0 + ks, nb, 0);
                cur[ks][nb][1] = frag(step0 + ks, nb, 1);
            }
#pragma unroll 1
        for (int tap = 0; tap < T; ++tap) {
            // the next tap's fragments (the image is padded by one tap: the last prefetch of the last slab reads zeros nobody uses)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    nxt[ks][nb][0] = frag(step0 + (tap + 1) * 2 + ks, nb, 0);
                    nxt[ks][nb][1] = frag(step0 + (tap + 1) * 2 + ks, nb, 1);
                }
            const int toff = ((tap / KS) * HWP + tap % KS) * DEC_PSTRIDE;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                dh8 ah[2], al[2];
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    ah[mb] = *reinterpret_cast<const dh8 *>(flds + a_off[mb] + toff + ks * 32);
                    al[mb] = *reinterpret_cast<const dh8 *>(flds + PLANE + a_off[mb] + toff + ks * 32);
                }
#pragma unroll
                for (int part = 0; part < 3; ++part)
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                        for (int nb = 0; nb < 2; ++nb)
                            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(part == 0 ? al[mb] : ah[mb], cur[ks][nb][part == 1 ? 1 : 0],
                                                                                 acc[mb][nb], 0, 0, 0);
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    cur[ks][nb][0] = nxt[ks][nb][0];
                    cur[ks][nb][1] = nxt[ks][nb][1];
                }
        }
    }
    unsigned x3m = 0u;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int co = (nb0 + nb) * 32 + li;
        const float bv = a.bias ? a.bias[co] : 0.0f;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int p = (wave * 2 + mb) * 32 + 8 * (r >> 2) + 4 * kg + (r & 3);
                float v = acc[mb][nb][r] + bv;
                x3m = max(x3m, x3_mag(v));
                if (a.relu) v = fmaxf(v, 0.0f);
                a.out[((size_t)img * S * S + p) * a.cout + co] = v;
            }
    }
    if (a.range_flag) x3_raise(a.range_flag, x3m);
}

}  // namespace oryon

using namespace oryon;

// ------------------------------------------------------------------------------------------------ host side
struct oryon_decoder {
    int device = 0;
    char *blob = nullptr;                   // one allocation: packed weight images + fp32 vectors
    // weight images
    const dh8 *gp_img[2] = {nullptr, nullptr};
    const dh8 *up_img[3] = {nullptr, nullptr, nullptr};
    const dh8 *c1_img[3] = {nullptr, nullptr, nullptr};
    const dh8 *c2_img[3] = {nullptr, nullptr, nullptr};
    // fp32 vectors (copies: the handle does not keep the caller's tensors alive)
    const float *gp_b[2], *up_b[3], *n1_g[3], *n1_b[3], *n2_g[3], *n2_b[3], *head_w, *head_b;
};

namespace {
constexpr int D_IN = 128;                                 // input_dim (models/decoder.py:125)
constexpr int D_CAT[3] = {128, 64, 32};                   // channels after cat (= the Up block's in_channels)
constexpr int D_G[3] = {32, 16, 0};                       // projected guidance channels
constexpr int D_OUT[3] = {64, 32, 32};                    // decoder_dims + the extra upsampling block
constexpr int D_GIN[2] = {256, 128};                      // Swin guidance channels

inline int64_t conv_img_halves(int cin, int NB) { return ((int64_t)(cin / 32) * 18 + DEC_RING) * NB * 2 * 64 * 8; }   // + RING steps of zeros
inline int64_t up_img_halves(int cin, int nbp) { return (int64_t)(4 * nbp) * (cin / 16) * 2 * 64 * 8; }
inline int nb_of(int cout) { return (cout + 31) / 32; }
inline int64_t align256(int64_t x) { return (x + 255) / 256 * 256; }

struct WsLayout {
    int64_t R[3], stats, affine[6], total;
};
WsLayout ws_layout(int n, int h, int w)
{
    WsLayout L;
    const int64_t big = align256((int64_t)n * (8 * h) * (8 * w) * 32 * 4);
    int64_t off = 0;
    for (int i = 0; i < 3; ++i) { L.R[i] = off; off += big; }
    L.stats = off; off += align256((int64_t)n * ((8 * h / 16) * (8 * w / 16)) * 4 * 2 * 4);
    for (int i = 0; i < 6; ++i) { L.affine[i] = off; off += align256((int64_t)n * 64 * 2 * 4); }
    L.total = off;
    return L;
}

template <int NB, bool IN_NCHW, bool IN_GN>
void launch_conv(hipStream_t st, const DecConv &a, int n)
{
    const int tiles = (a.H / 16) * (a.W / 16), total = tiles * n;
    if constexpr (NB == 1 && !IN_NCHW) {
        if (a.cin == 32 && a.cout == 32 && total > 512) {   // single slab, one full 32-channel output block: persistent workgroups, weights in registers
            hipLaunchKernelGGL((dec_conv3x3_kernel<1, false, IN_GN, true>), dim3(512), dim3(256), 0, st, a, tiles, total);
            return;
        }
    }
    hipLaunchKernelGGL((dec_conv3x3_kernel<NB, IN_NCHW, IN_GN, false>), dim3(total), dim3(256), 0, st, a, tiles, total);
}
}  // namespace

extern "C" {

int oryon_decoder_create(const oryon_decoder_weights_t *w, oryon_decoder_t **out, void *stream)
{
    ORYON_CHECK_ARG(w != nullptr && out != nullptr);
    for (int i = 0; i < 2; ++i) ORYON_CHECK_ARG(w->gp_w[i] && w->gp_b[i]);
    for (int i = 0; i < 3; ++i)
        ORYON_CHECK_ARG(w->up_w[i] && w->up_b[i] && w->c1_w[i] && w->n1_g[i] && w->n1_b[i] && w->c2_w[i] && w->n2_g[i] && w->n2_b[i]);
    ORYON_CHECK_ARG(w->head_w && w->head_b);
    hipStream_t st = as_stream(stream);
    auto *d = new oryon_decoder();
    (void)hipGetDevice(&d->device);
    // blob layout
    int64_t off = 0;
    int64_t o_gp[2], o_up[3], o_c1[3], o_c2[3];
    for (int i = 0; i < 2; ++i) { o_gp[i] = off; off += align256(conv_img_halves(D_GIN[i], 1) * 2); }
    for (int i = 0; i < 3; ++i) {
        const int cup = D_CAT[i] - D_G[i];
        o_up[i] = off; off += align256(up_img_halves(i == 0 ? D_IN : D_OUT[i - 1], nb_of(cup)) * 2);
        o_c1[i] = off; off += align256(conv_img_halves(D_CAT[i], nb_of(D_OUT[i])) * 2);
        o_c2[i] = off; off += align256(conv_img_halves(D_OUT[i], nb_of(D_OUT[i])) * 2);
    }
    const int64_t o_vec = off;
    off += 4096 * 4;                                       // all fp32 vectors together are < 4096 floats
    hipError_t e = hipMalloc(&d->blob, off);
    if (e != hipSuccess) {
        delete d;
        set_error("oryon_decoder_create: hipMalloc(%lld) failed: %s", (long long)off, hipGetErrorString(e));
        return ORYON_ERR_HIP;
    }
    auto pack_conv = [&](const float *src, int cout, int cin, int64_t o) {
        const int NB = nb_of(cout);
        const int64_t total = conv_img_halves(cin, NB);
        hipLaunchKernelGGL(dec_pack_conv3x3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, src, cout, cin, NB,
                           reinterpret_cast<_Float16 *>(d->blob + o), total);
        return reinterpret_cast<const dh8 *>(d->blob + o);
    };
    for (int i = 0; i < 2; ++i) d->gp_img[i] = pack_conv(w->gp_w[i], D_G[i], D_GIN[i], o_gp[i]);
    float *vec = reinterpret_cast<float *>(d->blob + o_vec);
    int vo = 0;
    auto keep = [&](const float *src, int nfl) {
        float *dst = vec + vo;
        vo += (nfl + 3) / 4 * 4;
        (void)hipMemcpyAsync(dst, src, (size_t)nfl * 4, hipMemcpyDeviceToDevice, st);
        return (const float *)dst;
    };
    for (int i = 0; i < 2; ++i) d->gp_b[i] = keep(w->gp_b[i], D_G[i]);
    for (int i = 0; i < 3; ++i) {
        const int cin = i == 0 ? D_IN : D_OUT[i - 1], cup = D_CAT[i] - D_G[i], nbp = nb_of(cup);
        const int64_t total = up_img_halves(cin, nbp);
        hipLaunchKernelGGL(dec_pack_upconv_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w->up_w[i], cin, cup, nbp,
                           reinterpret_cast<_Float16 *>(d->blob + o_up[i]), total);
        d->up_img[i] = reinterpret_cast<const dh8 *>(d->blob + o_up[i]);
        d->up_b[i] = keep(w->up_b[i], cup);
        d->c1_img[i] = pack_conv(w->c1_w[i], D_OUT[i], D_CAT[i], o_c1[i]);
        d->c2_img[i] = pack_conv(w->c2_w[i], D_OUT[i], D_OUT[i], o_c2[i]);
        d->n1_g[i] = keep(w->n1_g[i], D_OUT[i]);
        d->n1_b[i] = keep(w->n1_b[i], D_OUT[i]);
        d->n2_g[i] = keep(w->n2_g[i], D_OUT[i]);
        d->n2_b[i] = keep(w->n2_b[i], D_OUT[i]);
    }
    d->head_w = keep(w->head_w, 288);
    d->head_b = keep(w->head_b, 1);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);      // the caller's weight tensors may go away after this call
    if (e != hipSuccess) {
        (void)hipFree(d->blob);
        delete d;
        set_error("oryon_decoder_create: packing the weights failed: %s", hipGetErrorString(e));
        return ORYON_ERR_HIP;
    }
    *out = d;
    return ORYON_OK;
}

void oryon_decoder_destroy(oryon_decoder_t *d)
{
    if (!d) return;
    (void)hipFree(d->blob);
    delete d;
}

int64_t oryon_decoder_workspace_bytes(int n_img, int h, int w)
{
    if (n_img <= 0 || h <= 0 || w <= 0 || (h % 8) || (w % 8)) return 0;
    return ws_layout(n_img, h, w).total;
}

int oryon_decoder_workspace_layout(int n_img, int h, int w, int64_t *offsets3)
{
    ORYON_CHECK_ARG(offsets3 != nullptr && n_img > 0 && h > 0 && w > 0 && h % 8 == 0 && w % 8 == 0);
    const WsLayout L = ws_layout(n_img, h, w);
    for (int i = 0; i < 3; ++i) offsets3[i] = L.R[i];
    return ORYON_OK;
}

int oryon_decoder_forward(const oryon_decoder_t *d, const float *x, const float *g2, const float *g3, int n_img, int h, int w, void *workspace,
                          int64_t workspace_bytes, float *featmap, float *logits, int guidance_layout, int stop_after, void *stream)
{
    ORYON_CHECK_ARG(guidance_layout == ORYON_LAYOUT_NCHW || guidance_layout == ORYON_LAYOUT_NHWC);
    ORYON_CHECK_ARG(d != nullptr && x != nullptr && g2 != nullptr && g3 != nullptr && workspace != nullptr);
    ORYON_CHECK_ARG(n_img > 0 && n_img < 65536 && h > 0 && w > 0 && h % 8 == 0 && w % 8 == 0);
    ORYON_CHECK_ARG(featmap != nullptr && logits != nullptr);
    const WsLayout L = ws_layout(n_img, h, w);
    ORYON_CHECK_ARG(workspace_bytes >= L.total);
    hipStream_t st = as_stream(stream);
    unsigned *rflag = x3_range_flag(st);
    if (!rflag) return ORYON_ERR_HIP;
    char *ws = reinterpret_cast<char *>(workspace);
    float *R[3] = {reinterpret_cast<float *>(ws + L.R[0]), reinterpret_cast<float *>(ws + L.R[1]), reinterpret_cast<float *>(ws + L.R[2])};
    float *stats = reinterpret_cast<float *>(ws + L.stats);
    float *aff[6];
    for (int i = 0; i < 6; ++i) aff[i] = reinterpret_cast<float *>(ws + L.affine[i]);
    const float *guid[2] = {g2, g3};
    const float *prev = x;                                  // the block's input: x (NCHW) or the previous block's second conv (raw, NHWC)
    const float *prev_aff = nullptr;
    int H = h, W = w;
    for (int i = 0; i < 3; ++i) {
        const int cin = i == 0 ? D_IN : D_OUT[i - 1], ccat = D_CAT[i], cup = ccat - D_G[i], cout = D_OUT[i];
        const int nbp = nb_of(cup);
        float *cat = R[0], *a1 = R[1], *b1 = R[2];
        // ConvTranspose2d 2x2 s2 (+ the previous block's GroupNorm + ReLU on its input) -> channels 0 .. cup - 1 of the cat buffer
        {
            const dim3 grid((unsigned)ceil_div(H * W / 32, 4), n_img);
            if (i == 0)
                hipLaunchKernelGGL((dec_upconv_kernel<8, true, false>), grid, dim3(256), 0, st, prev, prev_aff, H, W, cin, d->up_img[i], d->up_b[i],
                                   cat, ccat, cup, nbp, rflag);
            else if (i == 1)
                hipLaunchKernelGGL((dec_upconv_kernel<4, false, true>), grid, dim3(256), 0, st, prev, prev_aff, H, W, cin, d->up_img[i], d->up_b[i],
                                   cat, ccat, cup, nbp, rflag);
            else
                hipLaunchKernelGGL((dec_upconv_kernel<2, false, true>), grid, dim3(256), 0, st, prev, prev_aff, H, W, cin, d->up_img[i], d->up_b[i],
                                   cat, ccat, cup, nbp, rflag);
        }
        H *= 2;
        W *= 2;
        const int tiles = (H / 16) * (W / 16);
        if (D_G[i] > 0) {
            // guidance projection: conv3x3 + bias + ReLU of the Swin map (NCHW, read in place) -> channels cup .. of the cat buffer
            DecConv a{};
            a.range_flag = rflag;
            a.in = guid[i]; a.affine = nullptr; a.wimg = d->gp_img[i]; a.bias = d->gp_b[i]; a.out = cat; a.stats = nullptr;
            a.H = H; a.W = W; a.cin = D_GIN[i]; a.in_cstride = D_GIN[i]; a.in_coff = 0; a.out_cstride = ccat; a.out_coff = cup; a.cout = D_G[i]; a.relu = 1;
            if (guidance_layout == ORYON_LAYOUT_NHWC) launch_conv<1, false, false>(st, a, n_img);
            else launch_conv<1, true, false>(st, a, n_img);
        }
        ORYON_CHECK_LAUNCH();
        if (stop_after == 3 * i + 1) return ORYON_OK;
        // conv1 (raw output + GroupNorm partial sums)
        {
            DecConv a{};
            a.range_flag = rflag;
            a.in = cat; a.affine = nullptr; a.wimg = d->c1_img[i]; a.bias = nullptr; a.out = a1; a.stats = stats;
            a.H = H; a.W = W; a.cin = ccat; a.in_cstride = ccat; a.in_coff = 0; a.out_cstride = cout; a.out_coff = 0; a.cout = cout; a.relu = 0;
            if (cout == 64) launch_conv<2, false, false>(st, a, n_img);
            else launch_conv<1, false, false>(st, a, n_img);
            hipLaunchKernelGGL(dec_gn_affine_kernel, dim3(n_img * (cout / 16)), dim3(64), 0, st, stats, n_img, tiles, cout / 16,
                               d->n1_g[i], d->n1_b[i], 1e-5, 1.0 / ((double)H * W * 16), aff[2 * i]);
        }
        ORYON_CHECK_LAUNCH();
        if (stop_after == 3 * i + 2) return ORYON_OK;
        // conv2 on relu(GN(conv1))
        {
            DecConv a{};
            a.range_flag = rflag;
            a.in = a1; a.affine = aff[2 * i]; a.wimg = d->c2_img[i]; a.bias = nullptr; a.out = b1; a.stats = stats;
            a.H = H; a.W = W; a.cin = cout; a.in_cstride = cout; a.in_coff = 0; a.out_cstride = cout; a.out_coff = 0; a.cout = cout; a.relu = 0;
            if (cout == 64) launch_conv<2, false, true>(st, a, n_img);
            else launch_conv<1, false, true>(st, a, n_img);
            hipLaunchKernelGGL(dec_gn_affine_kernel, dim3(n_img * (cout / 16)), dim3(64), 0, st, stats, n_img, tiles, cout / 16,
                               d->n2_g[i], d->n2_b[i], 1e-5, 1.0 / ((double)H * W * 16), aff[2 * i + 1]);
        }
        ORYON_CHECK_LAUNCH();
        if (stop_after == 3 * i + 3) return ORYON_OK;
        prev = b1;
        prev_aff = aff[2 * i + 1];
    }
    hipLaunchKernelGGL(dec_final_kernel, dim3((H / 8) * (W / 32), n_img), dim3(256), 0, st, prev, prev_aff, H, W, d->head_w, d->head_b, featmap,
                       logits);
    ORYON_CHECK_LAUNCH();
    return ORYON_OK;
}

static inline int64_t conv24_image_halves(int cout, int cin, int ksize)
{
    return ((int64_t)((cin + 31) / 32) * ksize * ksize + 1) * 2 * (cout / 32) * 2 * 64 * 8;          // + one tap of zeros (the prefetch reads ahead)
}

int64_t oryon_conv24_image_bytes(int cout, int cin, int ksize)
{
    if (cout <= 0 || cout % 64 || cin <= 0 || cin % 4 || (ksize != 3 && ksize != 7)) return 0;
    return conv24_image_halves(cout, cin, ksize) * 2;
}

int oryon_conv24_pack_f16x3(const float *w, int cout, int cin, int ksize, void *image, void *stream)
{
    ORYON_CHECK_ARG(w != nullptr && image != nullptr && oryon_conv24_image_bytes(cout, cin, ksize) > 0);
    const int64_t total = conv24_image_halves(cout, cin, ksize);
    hipLaunchKernelGGL(dec_pack_conv3x3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), w, cout, cin, cout / 32,
                       reinterpret_cast<_Float16 *>(image), total, ksize * ksize);
    ORYON_CHECK_LAUNCH();
    return ORYON_OK;
}

int oryon_conv24_f16x3(const float *x, int n, int cin, const void *image, const float *bias, int cout, int ksize, int relu, float *y, void *stream)
{
    ORYON_CHECK_ARG(x != nullptr && image != nullptr && y != nullptr && n >= 0 && oryon_conv24_image_bytes(cout, cin, ksize) > 0);
    ORYON_CHECK_ARG((int64_t)n * (cout / 64) < 2147483647LL);
    ORYON_CHECK_ARG((((uintptr_t)x | (uintptr_t)y | (uintptr_t)image) & 15) == 0);
    if (n == 0) return ORYON_OK;
    FusConv a{};
    a.range_flag = x3_range_flag(as_stream(stream));
    if (!a.range_flag) return ORYON_ERR_HIP;
    a.in = x; a.wimg = reinterpret_cast<const dh8 *>(image); a.bias = bias; a.out = y; a.cin = cin; a.cout = cout; a.relu = relu ? 1 : 0;
    const dim3 grid((unsigned)(n * (cout / 64)));
    if (ksize == 3) {
        constexpr int dyn = 2 * 26 * 26 * DEC_PSTRIDE;
        allow_dynamic_lds(reinterpret_cast<const void *>(&fus_conv24_kernel<3>), dyn);
        hipLaunchKernelGGL(fus_conv24_kernel<3>, grid, dim3(576), dyn, as_stream(stream), a);
    } else {
        constexpr int dyn = 2 * 30 * 30 * DEC_PSTRIDE;
        allow_dynamic_lds(reinterpret_cast<const void *>(&fus_conv24_kernel<7>), dyn);
        hipLaunchKernelGGL(fus_conv24_kernel<7>, grid, dim3(576), dyn, as_stream(stream), a);
    }
    ORYON_CHECK_LAUNCH();
    return ORYON_OK;
}

}  // extern "C"
