// K0 family: masks -> ROI lists -> (subsample) -> gathered, L2-normalised descriptor rows.
// Replaces utils/pcd.py:184-193, losses.py:58-59, pipeline.py:408-411 of the reference (see include/oryon_hip.h).
// All of this is HBM-bound integer/gather work: coalesced row-major scans, wave ballots for the ordered
// compaction, an LDS transpose so the [N,C] descriptor rows are written in 128-byte runs.
#include <hip/hip_fp16.h>
#include <stdlib.h>
#include "common.h"

namespace oryon {

constexpr int SCAN_THREADS = 1024;
constexpr int SCAN_WAVES = SCAN_THREADS / 64;

// Ordered (stable) block compaction step: every thread contributes `flag`; returns this thread's output
// slot (valid when flag) relative to the chunk start and the chunk total through `total`.
__device__ __forceinline__ int block_rank(bool flag, int *s_wave /*[SCAN_WAVES]*/, int &total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long b = __ballot(flag);
    const int before = __popcll(b & ((1ull << lane) - 1ull));
    __syncthreads();                       // previous user of s_wave is done
    if (lane == 0) s_wave[wave] = __popcll(b);
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SCAN_WAVES; ++w) {
        const int c = s_wave[w];
        base += (w < wave) ? c : 0;
        tot += c;
    }
    total = tot;
    return base + before;
}

// Ordered block scan of per-thread COUNTS (exclusive): this thread's offset inside the chunk, the chunk total through `total`.
__device__ __forceinline__ int block_scan_counts(int cnt, int *s_wave /*[SCAN_WAVES]*/, int &total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        incl += (lane >= o) ? v : 0;
    }
    __syncthreads();                       // previous user of s_wave is done
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SCAN_WAVES; ++w) {
        const int c = s_wave[w];
        base += (w < wave) ? c : 0;
        tot += c;
    }
    total = tot;
    return base + incl - cnt;
}

// One workgroup per map, EIGHT consecutive pixels per thread and scan step (two 16-byte loads): 7 block scans for a 224 x 224 map
// instead of 49 - the kernel is bound by the latency of its scan steps (55 -> ~12 us per launch at 64 maps), not by the 200 KB it reads.
__global__ __launch_bounds__(SCAN_THREADS) void roi_compact_kernel(const int32_t *__restrict__ mask, int HW,
                                                                    int32_t *__restrict__ roi, int32_t *__restrict__ count)
{
    __shared__ int s_wave[SCAN_WAVES];
    const int m = blockIdx.x;
    const int32_t *mk = mask + (size_t)m * HW;
    int32_t *out = roi + (size_t)m * HW;
    const bool vec = (HW % 4 == 0) && ((reinterpret_cast<uintptr_t>(mk) & 15) == 0);
    int base = 0;
    for (int p0 = 0; p0 < HW; p0 += SCAN_THREADS * 8) {
        const int p = p0 + (int)threadIdx.x * 8;
        int v[8];
        if (vec && p + 8 <= HW) {
            const int4 a = *reinterpret_cast<const int4 *>(mk + p), b = *reinterpret_cast<const int4 *>(mk + p + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (p + e < HW) ? mk[p + e] : 0;
        }
        unsigned bits = 0u;
#pragma unroll
        for (int e = 0; e < 8; ++e) bits |= (v[e] == 1 ? 1u : 0u) << e;
        int total;
        int o = base + block_scan_counts(__popc(bits), s_wave, total);
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (bits & (1u << e)) out[o++] = p + e;
        base += total;
    }
    if (threadIdx.x == 0) count[m] = base;
}

__global__ void mask_from_logits_kernel(const float *__restrict__ logits, int64_t n, float thr, int32_t *__restrict__ mask)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float s = 1.0f / (1.0f + expf(-logits[i]));
        mask[i] = s > thr ? 1 : 0;
    }
}

__global__ void mask_resize_nearest_kernel(const uint8_t *__restrict__ in, int HI, int WI, int HO, int WO,
                                           int32_t *__restrict__ out)
{
    const int m = blockIdx.y;
    const float sy = (float)HI / (float)HO, sx = (float)WI / (float)WO;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < HO * WO; p += gridDim.x * blockDim.x) {
        const int y = p / WO, x = p % WO;
        int ys = (int)floorf((float)y * sy), xs = (int)floorf((float)x * sx);
        ys = ys < HI - 1 ? ys : HI - 1;
        xs = xs < WI - 1 ? xs : WI - 1;
        out[(size_t)m * HO * WO + p] = (int32_t)in[((size_t)m * HI + ys) * WI + xs];
    }
}

// Subsample without replacement: keep the max_keep smallest 32-bit keys (ties by index), in ROI order.
__global__ __launch_bounds__(SCAN_THREADS) void roi_subsample_kernel(int32_t *__restrict__ roi, int32_t *__restrict__ count,
                                                                      int roi_stride, int max_keep, uint64_t seed,
                                                                      const int64_t *__restrict__ map_key)
{
    __shared__ int s_wave[SCAN_WAVES];
    __shared__ unsigned s_hist[256];
    __shared__ unsigned s_prefix, s_remaining;
    const int m = blockIdx.x;
    const int n = count[m];
    if (n <= max_keep) return;
    const uint64_t key = map_key ? (uint64_t)map_key[m] : (uint64_t)m;
    int32_t *r = roi + (size_t)m * roi_stride;

    if (threadIdx.x == 0) { s_prefix = 0u; s_remaining = (unsigned)max_keep; }
    // radix select, most significant byte first
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        if (threadIdx.x < 256) s_hist[threadIdx.x] = 0u;
        __syncthreads();
        const unsigned prefix = s_prefix;
        const unsigned hi_mask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
        for (int i = threadIdx.x; i < n; i += SCAN_THREADS) {
            const unsigned k = rng_u32(seed, key, 0u, (uint32_t)i);
            if ((k & hi_mask) == prefix) atomicAdd(&s_hist[(k >> shift) & 0xFFu], 1u);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned rem = s_remaining, cum = 0u;
            int b = 0;
            for (; b < 256; ++b) {
                if (cum + s_hist[b] >= rem) break;
                cum += s_hist[b];
            }
            s_remaining = rem - cum;
            s_prefix = prefix | ((unsigned)b << shift);
        }
        __syncthreads();
    }
    const unsigned T = s_prefix;
    const int ties_to_take = (int)s_remaining;
    int kept = 0, ties_seen = 0;
    // eight consecutive entries per thread and scan step (as roi_compact_kernel): 2 steps for 12.5 k entries instead of 13
    for (int i0 = 0; i0 < n; i0 += SCAN_THREADS * 8) {
        const int i = i0 + (int)threadIdx.x * 8;
        int32_t v[8];
        unsigned below = 0u, tie = 0u;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const bool in = i + e < n;
            const unsigned k = in ? rng_u32(seed, key, 0u, (uint32_t)(i + e)) : 0xFFFFFFFFu;
            v[e] = in ? r[i + e] : 0;
            below |= (in && k < T ? 1u : 0u) << e;
            tie |= (in && k == T ? 1u : 0u) << e;
        }
        int tie_total;
        int tie_rank = ties_seen + block_scan_counts(__popc(tie), s_wave, tie_total);
        unsigned keep = below;
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (tie & (1u << e)) { if (tie_rank < ties_to_take) keep |= 1u << e; ++tie_rank; }
        int keep_total;
        int pos = kept + block_scan_counts(__popc(keep), s_wave, keep_total);
        // all reads of this chunk (v) happened before the scans' barriers -> in-place write is safe
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (keep & (1u << e)) r[pos++] = v[e];
        kept += keep_total;
        ties_seen += tie_total;
    }
    if (threadIdx.x == 0) count[m] = kept;
}

// Gather + normalise.  One thread per ROI row for the (canonical, k-ordered) norm; the scaled values go
// through an LDS transpose so global stores are 128-byte runs of one descriptor row.
constexpr int GN_ROWS = 256;
constexpr int GN_KT = 32;
__global__ __launch_bounds__(GN_ROWS) void gather_normalise_kernel(const float *__restrict__ feat, int C, int HW,
                                                                    const int32_t *__restrict__ roi, int roi_stride,
                                                                    const int32_t *__restrict__ count, int rows_cap,
                                                                    int Cp, float *__restrict__ out,
                                                                    __half *__restrict__ out16)
{
    __shared__ float tile[GN_ROWS * (GN_KT + 1)];
    const int m = blockIdx.y;
    const int n = count[m];
    const int row0 = blockIdx.x * GN_ROWS;
    if (row0 >= n) return;                       // only the last partially filled block zero-fills
    const int t = threadIdx.x;
    const int row = row0 + t;
    const bool live = row < n;
    const float *f = feat + (size_t)m * C * HW;
    const int pix = live ? roi[(size_t)m * roi_stride + row] : 0;
    float n2 = 0.0f;
    if (live)
        for (int k = 0; k < C; ++k) {
            const float v = f[(size_t)k * HW + pix];
            n2 = __fmaf_rn(v, v, n2);
        }
    float d = sqrt_rn(n2);
    d = d < 1e-8f ? 1e-8f : d;
    float *o = out + ((size_t)m * rows_cap + row0) * Cp;
    for (int k0 = 0; k0 < Cp; k0 += GN_KT) {
#pragma unroll 8
        for (int kk = 0; kk < GN_KT; ++kk) {
            const float v = (live && k0 + kk < C) ? __fdiv_rn(f[(size_t)(k0 + kk) * HW + pix], d) : 0.0f;
            tile[t * (GN_KT + 1) + kk] = v;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < (GN_ROWS * GN_KT / 4) / GN_ROWS; ++i) {
            const int fidx = t + GN_ROWS * i;
            const int r = fidx / (GN_KT / 4), c4 = fidx % (GN_KT / 4);
            // k-permuted row layout consumed by K1: position 8g + 4h + j holds k = 8g + 2j + h  (c4 = 2g + h)
            const int kb = 8 * (c4 >> 1) + (c4 & 1);
            float4 v;
            v.x = tile[r * (GN_KT + 1) + kb + 0];
            v.y = tile[r * (GN_KT + 1) + kb + 2];
            v.z = tile[r * (GN_KT + 1) + kb + 4];
            v.w = tile[r * (GN_KT + 1) + kb + 6];
            *reinterpret_cast<float4 *>(o + (size_t)r * Cp + k0 + c4 * 4) = v;
        }
        if (out16) {
            // fp16 copy of the same unit rows in NATURAL k order (operand of the screening pass K1s): 8 halves per store
            __half *o16 = out16 + ((size_t)m * rows_cap + row0) * Cp;
#pragma unroll
            for (int i = 0; i < (GN_ROWS * GN_KT / 8) / GN_ROWS; ++i) {
                const int fidx = t + GN_ROWS * i;
                const int r = fidx / (GN_KT / 8), c8 = fidx % (GN_KT / 8);
                union { __half h[8]; uint4 u; } pk;
#pragma unroll
                for (int e = 0; e < 8; ++e) pk.h[e] = __float2half_rn(tile[r * (GN_KT + 1) + c8 * 8 + e]);
                *reinterpret_cast<uint4 *>(o16 + (size_t)r * Cp + k0 + c8 * 8) = pk.u;
            }
        }
        __syncthreads();
    }
}

// Second-generation gather + normalise (C_pad <= 512).  A workgroup owns ROWS ROI rows of one map and keeps their raw channel
// values in LDS ([k][row], one float of padding per k), so the channel-planar map is read from HBM exactly once:
//   load    all 4 waves, lane = row: every channel is one 128/256-byte run (each in its own DRAM page - this phase runs at
//           ~3.7 TB/s on its own and bounds the kernel), NL loads in flight per lane
//   norm    wave 0 runs the canonical k-ordered fmaf chain per row out of LDS (bit-exact vs the oracle)
//   store   all waves divide and write FULL 128-byte lines: fp32 rows k-permuted in 16-byte chunks, fp16 rows in natural
//           order (64-byte partial-line stores cost 3x per byte on this memory system, measured)
// Phases of different workgroups overlap on a CU (4 workgroups at C_pad = 256); a register-staged software pipeline inside a
// persistent workgroup (next tile's loads in flight under norm + stores) measured only 5-7 % faster on the same GPU and was
// not kept.
// (callers pad row capacities to 256: the zero-fill granularity)
template <int NL, int ROWS>        // NL: channel loads in flight per lane; ROWS: ROI rows per workgroup (32 or 64)
__global__ __launch_bounds__(256) void gather_normalise_v2_kernel(const float *__restrict__ feat, int C, int HW,
                                                                   const int32_t *__restrict__ roi, int roi_stride,
                                                                   const int32_t *__restrict__ count, int rows_cap, int Cp,
                                                                   float *__restrict__ out, __half *__restrict__ out16)
{
    constexpr int LD = ROWS + 1;
    constexpr int CPW = 64 / ROWS;             // channels one wave instruction covers (1 or 2)
    constexpr int KSTEP = 4 * CPW;             // channel stride between a lane's consecutive values
    constexpr int NRG = ROWS / 8;              // 8-row groups; waves beyond NRG split the k range instead
    constexpr int KSPLIT = NRG >= 4 ? 1 : 4 / NRG;
    extern __shared__ float raw[];             // [Cp][LD] raw values, then sd[ROWS] norms
    float *sd = raw + (size_t)Cp * LD;
    const int m = blockIdx.y;
    const int n = count[m];
    const int row0 = blockIdx.x * ROWS;
    // zero-fill contract: rows [n, round_up(n, 256)) must be written as zeros
    const int n_fill = (n + 255) / 256 * 256;
    if (row0 >= n_fill) return;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float *f = feat + (size_t)m * C * HW;
    const int lrow = lane & (ROWS - 1);
    const int kfirst = wave * CPW + lane / ROWS;
    const int my_row = row0 + lrow;
    const bool live = my_row < n;
    const int pix = live ? roi[(size_t)m * roi_stride + my_row] : 0;
    for (int kb = kfirst; kb < Cp; kb += KSTEP * NL) {
        float v[NL];
#pragma unroll
        for (int u = 0; u < NL; ++u) {
            const int k = kb + KSTEP * u;
            v[u] = (live && k < C) ? f[(size_t)k * HW + pix] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < NL; ++u) {
            const int k = kb + KSTEP * u;
            if (k < Cp) raw[k * LD + lrow] = v[u];
        }
    }
    __syncthreads();
    if (wave == 0 && lane < ROWS) {
        float n2 = 0.0f;
        for (int k0 = 0; k0 < C; k0 += 16) {
            float x[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) x[u] = (k0 + u < C) ? raw[(k0 + u) * LD + lane] : 0.0f;
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (k0 + u < C) n2 = __fmaf_rn(x[u], x[u], n2);
        }
        float d = sqrt_rn(n2);
        sd[lane] = d < 1e-8f ? 1e-8f : d;
    }
    __syncthreads();
    float *o = out + ((size_t)m * rows_cap + row0) * Cp;
    // fp32, k-permuted: lane -> (row_sub 0..7, chunk c4 0..7) of a 32-wide k group; 8 rows x 128 bytes per instruction
    for (int rg = wave % NRG; rg < NRG; rg += 4) {
        const int r = rg * 8 + (lane >> 3), c4 = lane & 7;
        const float d = sd[r];
        const int kb = 8 * (c4 >> 1) + (c4 & 1);
        for (int k0 = 32 * (wave / NRG); k0 < Cp; k0 += 32 * KSPLIT) {
            float4 q;
            q.x = __fdiv_rn(raw[(k0 + kb + 0) * LD + r], d);
            q.y = __fdiv_rn(raw[(k0 + kb + 2) * LD + r], d);
            q.z = __fdiv_rn(raw[(k0 + kb + 4) * LD + r], d);
            q.w = __fdiv_rn(raw[(k0 + kb + 6) * LD + r], d);
            *reinterpret_cast<float4 *>(o + (size_t)r * Cp + k0 + c4 * 4) = q;
        }
    }
    if (out16) {
        __half *o16 = out16 + ((size_t)m * rows_cap + row0) * Cp;
        if (Cp >= 64) {
            // fp16, natural order: lane -> (row_sub 0..7, chunk c8 0..7) of a 64-wide k group
            for (int rg = wave % NRG; rg < NRG; rg += 4) {
                const int r = rg * 8 + (lane >> 3), c8 = lane & 7;
                const float d = sd[r];
                for (int k0 = 64 * (wave / NRG); k0 < Cp; k0 += 64 * KSPLIT) {
                    if (k0 + c8 * 8 >= Cp) continue;
                    union { __half h[8]; uint4 u; } pk;
#pragma unroll
                    for (int e = 0; e < 8; ++e) pk.h[e] = __float2half_rn(__fdiv_rn(raw[(k0 + c8 * 8 + e) * LD + r], d));
                    *reinterpret_cast<uint4 *>(o16 + (size_t)r * Cp + k0 + c8 * 8) = pk.u;
                }
            }
        } else {
            // 64-byte rows are adjacent in memory: lane -> (row_sub 0..15, chunk c8 0..3), one contiguous 1 KB run
            constexpr int NRG16 = ROWS / 16;
            const int r = (wave % NRG16) * 16 + (lane >> 2), c8 = lane & 3;
            const float d = sd[r];
            if (wave < NRG16) {
                union { __half h[8]; uint4 u; } pk;
#pragma unroll
                for (int e = 0; e < 8; ++e) pk.h[e] = __float2half_rn(__fdiv_rn(raw[(c8 * 8 + e) * LD + r], d));
                *reinterpret_cast<uint4 *>(o16 + (size_t)r * Cp + c8 * 8) = pk.u;
            }
        }
    }
}

// K0 with the int8 copies for K1s8 (C_pad 256 / 512, 32-row tiles): everything gather_normalise_v2_kernel<.,32> writes, plus
//   out8   [n_maps, rows_cap, Cp] int8:  q = rint(x^ * 2^E), |q| <= 127, with ONE exponent E per 16-row slice of the screening
//          kernel's accumulator layout (rows {0-3,8-11,16-19,24-27} + 4*h of each 32-row block, h = 0/1), so that the
//          integer maximum over a slice is also its score maximum
//   scale  [n_maps, rows_cap/16] fp32 = 2^-E per slice (slice id = (row / 32) * 2 + h)
//   eps_max[n_maps] fp32 = max over the map's slices of 2^-(E+1), the per-element quantisation bound (atomic max on the bits)
// x^ is the canonical fp32 value (fdiv), so |q * 2^-E - x^| <= 2^-(E+1) exactly as match16.hip's error analysis needs.
template <int NL>
__global__ __launch_bounds__(256) void gather_normalise_q8_kernel(const float *__restrict__ feat, int C, int HW,
                                                                   const int32_t *__restrict__ roi, int roi_stride,
                                                                   const int32_t *__restrict__ count, int rows_cap, int Cp,
                                                                   float *__restrict__ out, __half *__restrict__ out16,
                                                                   int8_t *__restrict__ out8, float *__restrict__ scale,
                                                                   unsigned *__restrict__ eps_max)
{
    constexpr int ROWS = 32, LD = ROWS + 1, CPW = 2, KSTEP = 8;
    extern __shared__ float raw[];             // [Cp][LD] raw values -> unit values, then sd[ROWS] norms, smax[2] slice maxima
    float *sd = raw + (size_t)Cp * LD;
    unsigned *smax = reinterpret_cast<unsigned *>(sd + ROWS);
    const int m = blockIdx.y;
    const int n = count[m];
    const int row0 = blockIdx.x * ROWS;
    const int n_fill = (n + 255) / 256 * 256;
    if (row0 >= n_fill) return;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float *f = feat + (size_t)m * C * HW;
    const int lrow = lane & (ROWS - 1);
    const int kfirst = wave * CPW + lane / ROWS;
    const int my_row = row0 + lrow;
    const bool live = my_row < n;
    const int pix = live ? roi[(size_t)m * roi_stride + my_row] : 0;
    if (t < 2) smax[t] = 0u;
    for (int kb = kfirst; kb < Cp; kb += KSTEP * NL) {
        float v[NL];
#pragma unroll
        for (int u = 0; u < NL; ++u) {
            const int k = kb + KSTEP * u;
            v[u] = (live && k < C) ? f[(size_t)k * HW + pix] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < NL; ++u) {
            const int k = kb + KSTEP * u;
            if (k < Cp) raw[k * LD + lrow] = v[u];
        }
    }
    __syncthreads();
    if (wave == 0 && lane < ROWS) {
        float n2 = 0.0f;
        for (int k0 = 0; k0 < C; k0 += 16) {
            float x[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) x[u] = (k0 + u < C) ? raw[(k0 + u) * LD + lane] : 0.0f;
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (k0 + u < C) n2 = __fmaf_rn(x[u], x[u], n2);
        }
        float d = sqrt_rn(n2);
        sd[lane] = d < 1e-8f ? 1e-8f : d;
    }
    __syncthreads();
    float *o = out + ((size_t)m * rows_cap + row0) * Cp;
    // fp32, k-permuted, and the unit value written back in place: lane -> (row_sub 0..7, chunk c4 0..7) of a 32-wide k group,
    // wave = 8-row group.  The slice maximum of |x^| rides along.
    {
        const int r = wave * 8 + (lane >> 3), c4 = lane & 7;
        const float d = sd[r];
        const int kb = 8 * (c4 >> 1) + (c4 & 1);
        float mx = 0.0f;
        for (int k0 = 0; k0 < Cp; k0 += 32) {
            float4 q;
            q.x = __fdiv_rn(raw[(k0 + kb + 0) * LD + r], d);
            q.y = __fdiv_rn(raw[(k0 + kb + 2) * LD + r], d);
            q.z = __fdiv_rn(raw[(k0 + kb + 4) * LD + r], d);
            q.w = __fdiv_rn(raw[(k0 + kb + 6) * LD + r], d);
            raw[(k0 + kb + 0) * LD + r] = q.x;
            raw[(k0 + kb + 2) * LD + r] = q.y;
            raw[(k0 + kb + 4) * LD + r] = q.z;
            raw[(k0 + kb + 6) * LD + r] = q.w;
            mx = fmaxf(fmaxf(mx, fmaxf(fabsf(q.x), fabsf(q.y))), fmaxf(fabsf(q.z), fabsf(q.w)));
            *reinterpret_cast<float4 *>(o + (size_t)r * Cp + k0 + c4 * 4) = q;
        }
        // a wave covers rows 8w..8w+7: lanes 0-31 belong to slice h = 0, lanes 32-63 to h = 1 -> reduce inside each half, one
        // LDS atomic per half-wave (non-negative floats order like their bit patterns)
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
        if ((lane & 31) == 0) atomicMax(&smax[lane >> 5], __float_as_uint(mx));
    }
    __syncthreads();
    // slice exponents: 2^E * max|x^| <= 127
    float sc[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float mxs = __uint_as_float(smax[h]);
        int E = mxs > 0.0f ? ilogbf(127.0f / mxs) : 30;
        E = E > 30 ? 30 : (E < 0 ? 0 : E);
        sc[h] = ldexpf(1.0f, E);
        if (t == h) {
            scale[(size_t)m * (rows_cap / 16) + (row0 / 32) * 2 + h] = ldexpf(1.0f, -E);
            if (row0 < n) atomicMax(&eps_max[m], __float_as_uint(ldexpf(1.0f, -E - 1)));
        }
    }
    {
        // fp16 (optional), natural order: lane -> (row_sub 0..7, chunk c8 0..7) of a 64-wide k group: 8 rows x one 128-byte line
        __half *o16 = out16 + ((size_t)m * rows_cap + row0) * Cp;
        const int r = wave * 8 + (lane >> 3), c8 = lane & 7;
        if (out16)
        for (int k0 = 0; k0 < Cp; k0 += 64) {
            union { __half h[8]; uint4 u; } pk;
#pragma unroll
            for (int e = 0; e < 8; ++e) pk.h[e] = __float2half_rn(raw[(k0 + c8 * 8 + e) * LD + r]);
            *reinterpret_cast<uint4 *>(o16 + (size_t)r * Cp + k0 + c8 * 8) = pk.u;
        }
        // int8, natural order: lane -> (row_sub 0..7, chunk c16 0..7) of a 128-wide k group: 8 rows x one 128-byte line
        int8_t *o8 = out8 + ((size_t)m * rows_cap + row0) * Cp;
        const float s8 = sc[(r >> 2) & 1];
        for (int k0 = 0; k0 < Cp; k0 += 128) {
            union { int8_t b[16]; uint4 u; } pk;
#pragma unroll
            for (int e = 0; e < 16; ++e) pk.b[e] = (int8_t)(int)rintf(raw[(k0 + c8 * 16 + e) * LD + r] * s8);
            *reinterpret_cast<uint4 *>(o8 + (size_t)r * Cp + k0 + c8 * 16) = pk.u;
        }
    }
}

}  // namespace oryon

using namespace oryon;

extern "C" int oryon_roi_compact(const int32_t *mask, int n_maps, int HW, int32_t *roi, int32_t *count, void *stream)
{
    ORYON_CHECK_ARG(mask && roi && count && n_maps >= 0 && HW > 0);
    if (n_maps == 0) return ORYON_OK;
    hipLaunchKernelGGL(roi_compact_kernel, dim3(n_maps), dim3(SCAN_THREADS), 0, as_stream(stream), mask, HW, roi, count);
    ORYON_CHECK_LAUNCH();
    return ORYON_OK;
}

extern "C" int oryon_mask_from_logits(const float *logits, int64_t n, float threshold, int32_t *mask, void *stream)
{
    ORYON_CHECK_ARG(logits && mask && n >= 0);
    if (n == 0) return ORYON_OK;
    const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(mask_from_logits_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), logits, n, threshold, mask);
    ORYON_CHECK_LAUNCH();
    return ORYON_OK;
}

extern "C" int oryon_mask_resize_nearest(const uint8_t *mask_in, int n_maps, int HI, int WI, int HO, int WO,
                                         int32_t *mask_out, void *stream)
{
    ORYON_CHECK_ARG(mask_in && mask_out && n_maps >= 0 && HI > 0 && WI > 0 && HO > 0 && WO > 0);
    if (n_maps == 0) return ORYON_OK;
    const int bx = ceil_div(HO * WO, 256) < 64 ? ceil_div(HO * WO, 256) : 64;
    hipLaunchKernelGGL(mask_resize_nearest_kernel, dim3(bx, n_maps), dim3(256), 0, as_stream(stream), mask_in, HI, WI, HO,
                       WO, mask_out);
    ORYON_CHECK_LAUNCH();
    return ORYON_OK;
}

extern "C" int oryon_roi_subsample(int32_t *roi, int32_t *count, int n_maps, int roi_stride, int max_keep, uint64_t seed,
                                   const int64_t *map_key, void *stream)
{
    ORYON_CHECK_ARG(roi && count && n_maps >= 0 && roi_stride > 0 && max_keep > 0);
    if (n_maps == 0) return ORYON_OK;
    hipLaunchKernelGGL(roi_subsample_kernel, dim3(n_maps), dim3(SCAN_THREADS), 0, as_stream(stream), roi, count, roi_stride,
                       max_keep, seed, map_key);
    ORYON_CHECK_LAUNCH();
    return ORYON_OK;
}

extern "C" int oryon_gather_normalise_q8(const float *feat, int n_maps, int C, int HW, const int32_t *roi, int roi_stride,
                                         const int32_t *count, int rows_cap, int C_pad, float *out, void *out_f16, int8_t *out_i8,
                                         float *slice_scale, float *eps_max, void *stream)
{
    ORYON_CHECK_ARG(feat && roi && count && out && out_i8 && slice_scale && eps_max);      // out_f16 may be NULL
    ORYON_CHECK_ARG(n_maps >= 0 && C > 0 && HW > 0 && roi_stride > 0 && C_pad >= C && (C_pad == 256 || C_pad == 512));
    ORYON_CHECK_ARG(rows_cap > 0 && rows_cap % 256 == 0);
    if (n_maps == 0) return ORYON_OK;
    hipStream_t st = as_stream(stream);
    ORYON_CHECK_HIP(hipMemsetAsync(eps_max, 0, (size_t)n_maps * sizeof(float), st));
    const size_t sh = ((size_t)C_pad * 33 + 32 + 8) * sizeof(float);
    allow_dynamic_lds(reinterpret_cast<const void *>(gather_normalise_q8_kernel<16>), 160 * 1024);   // 8 / 16 / 32 loads in flight measure the same
    hipLaunchKernelGGL((gather_normalise_q8_kernel<16>), dim3(rows_cap / 32, n_maps), dim3(256), sh, st, feat, C, HW, roi, roi_stride, count,
                       rows_cap, C_pad, out, static_cast<__half *>(out_f16), out_i8, slice_scale, reinterpret_cast<unsigned *>(eps_max));
    ORYON_CHECK_LAUNCH();
    return ORYON_OK;
}

extern "C" int oryon_gather_normalise_f32(const float *feat, int n_maps, int C, int HW, const int32_t *roi, int roi_stride,
                                          const int32_t *count, int rows_cap, int C_pad, float *out, void *out_f16,
                                          void *stream)
{
    ORYON_CHECK_ARG(feat && roi && count && out && n_maps >= 0 && C > 0 && HW > 0 && roi_stride > 0);
    ORYON_CHECK_ARG(C_pad >= C && C_pad % GN_KT == 0);
    ORYON_CHECK_ARG(rows_cap > 0 && rows_cap % GN_ROWS == 0);
    if (n_maps == 0) return ORYON_OK;
    if (C_pad <= 512) {
        static const int rows_env = dev_env_int("ORYON_GATHER_ROWS", 0);
        const int rows = rows_env ? rows_env : 32;
        const size_t sh = ((size_t)C_pad * (rows + 1) + 64) * sizeof(float);
#define LAUNCH_G2(NLV, RV)                                                                                                 \
    do {                                                                                                                   \
        allow_dynamic_lds(reinterpret_cast<const void *>(gather_normalise_v2_kernel<NLV, RV>), 160 * 1024);              \
        hipLaunchKernelGGL((gather_normalise_v2_kernel<NLV, RV>), dim3(rows_cap / RV, n_maps), dim3(256), sh, as_stream(stream), feat, \
                           C, HW, roi, roi_stride, count, rows_cap, C_pad, out, static_cast<__half *>(out_f16));          \
    } while (0)
        if (rows == 64) LAUNCH_G2(16, 64); else LAUNCH_G2(16, 32);
#undef LAUNCH_G2
    } else {
        hipLaunchKernelGGL(gather_normalise_kernel, dim3(rows_cap / GN_ROWS, n_maps), dim3(GN_ROWS), 0, as_stream(stream), feat,
                           C, HW, roi, roi_stride, count, rows_cap, C_pad, out, static_cast<__half *>(out_f16));
    }
    ORYON_CHECK_LAUNCH();
    return ORYON_OK;
}
