// K0v3: ROI rows of a descriptor map -> the operands of the int8 screening matcher, moving only the bytes the path needs.
// Replaces utils/pcd.py:192-193 (`feats[:, roi[:,0], roi[:,1]].T`) + the normalisation inside torch's cosine_similarity
// (utils/pcd.py:28-29) for the K1s8 matcher.  Per ROI row it reads the C raw fp32 channel values ONCE and writes
//     out8  [n_maps, rows_cap, CP] int8   q = rint(x^ * 2^E), one exponent E per 16-row slice of the screening kernel's accumulator
//                                         layout (rows {0-3,8-11,16-19,24-27} + 4h of a 32-row block, h = 0/1)
//     scale [n_maps, rows_cap/16]  fp32   2^-E of the slice  (slice id = (row / 32) * 2 + h)
//     eps   [n_maps]               fp32   max over the map's live slices of 2^-(E+1)   (atomic max on the bits; caller zeroes)
//     norm  [n_maps, rows_cap]     fp32   d = max(sqrt(sum_k x_k^2), 1e-8) in the canonical k-ordered fmaf chain: with it the exact
//                                         re-scoring pass recovers the canonical unit value x_k / d of ANY row from the raw map
//     out32 [n_maps, rows_cap, CP] fp32   (optional, anchors and fall-back only) the canonical unit rows x_k / d, k-permuted for K1
// i.e. 4*C bytes in and C + 4.25 bytes out per row instead of 4*C in and 5*C out (round 1 wrote an fp32 copy of every query row
// although the re-scoring pass touches about one candidate per anchor).
//
// Structure.  A WAVE owns ROWS = 64 / LPR consecutive ROI rows and keeps their raw values in registers (KPL = CP / LPR per lane):
// lane -> (row, channel segment).  NCHW maps: for a fixed channel the lanes of a segment read one run of ROWS consecutive
// pixels (a single 256-byte / 128-byte request when the ROI is contiguous), up to 64 such loads in flight per wave, no LDS and no
// barrier on the read side.  The canonical norm is a serial chain in k, so a lane simply runs it over its own registers (LPR = 2:
// the second segment's lanes restart the chain from the first segment's result).  NHWC (channels_last) maps: rows are contiguous, so
// they are loaded 4 rows x 256 bytes per instruction and transposed to lane = row through the wave's LDS staging buffer, 64 channels
// at a time.  Outputs leave through the same 16 KB-per-wave staging buffer so that every global store instruction writes whole
// 128-byte lines.  Waves never synchronise with each other.
//
// Quantisation bound (used by match_decide_kernel): q = rint(x * rs), rs = RN(2^E / d), so |q - x^ 2^E| <= 1/2 + 127 * 3 * 2^-24:
// |q 2^-E - x^| <= 2^-(E+1) * (1 + 4.6e-5).  2^E * max|x^| <= 127 by the choice of E, so |q| <= 127.
#include <hip/hip_fp16.h>
#include <stdlib.h>
#include "common.h"

namespace oryon {

constexpr int G8_STAGE_BYTES = 16384;          // per wave

// Ordering point between LDS writes and LDS reads of ONE wave through its private staging buffer.  The LDS executes a wave's
// DS instructions in issue order, so all that is needed is that the compiler keeps the program order: a scheduling barrier plus a
// compiler-level memory clobber.  (A memory fence would also work but drains vmcnt, i.e. it waits for the prefetched global loads of
// the NEXT chunk: measured 1.8 -> 1.08 ms on the channels_last path.)
#define WAVE_LDS_ORDER()                    \
    do {                                    \
        __builtin_amdgcn_wave_barrier();    \
        asm volatile("" ::: "memory");      \
    } while (0)

typedef float f32x16g __attribute__((ext_vector_type(16)));
typedef float f32x32g __attribute__((ext_vector_type(32)));
typedef unsigned u32x6g __attribute__((ext_vector_type(6)));

template <int N, int FMT>
struct RowRegs {
    float v[N];
    __device__ __forceinline__ float get(int k) const { return v[k]; }
    __device__ __forceinline__ void set(int k, float x) { v[k] = x; }
};
template <int N>
struct RowRegs<N, 1> {
    f32x16g t[N / 16];
    __device__ __forceinline__ float get(int k) const { return t[((k >> 5) << 1) | (k & 1)][(k & 31) >> 1]; }
    __device__ __forceinline__ void set(int k, float x) { t[((k >> 5) << 1) | (k & 1)][(k & 31) >> 1] = x; }
};
template <int N>
struct RowRegs<N, 3> : RowRegs<N, 1> {};          // FMT = 3 (mx6 slots + hi / lo half rows in one pass) converts to fp6 as FMT = 1 does

template <int N, class RR>
__device__ __forceinline__ float chain_sq(const RR &r, float acc)
{
#pragma unroll
    for (int i = 0; i < N; ++i) acc = __fmaf_rn(r.get(i), r.get(i), acc);
    return acc;
}

// staging-buffer address of 16-byte slot `s` of tile-lane `t` (row length RB bytes, RB >= 256 a multiple of 256): slots are
// XOR-swizzled inside each 256-byte line so that lane-per-row writes and slot-per-lane reads both spread over the banks
__device__ __forceinline__ unsigned stage_addr(int t, int s, int RB)
{
    const int L = RB >= 256 ? 16 : RB / 16;                  // slots that share one swizzle group (a 256-byte line, or a shorter row)
    return (unsigned)(t * RB + (s & ~(L - 1)) * 16 + (((s & (L - 1)) ^ (t & (L - 1))) << 4));
}

// FMT = 1 (round 3): MX-fp6 screening operands instead of int8 rows, same row size.  A row's 32-channel block b becomes one 32-byte
// slot: 24 bytes = 32 fp6 e2m3 codes (element t at bits [6t, 6t+6)), byte 24 = the block's E8M0 exponent (value 2^(e-127)), rest 0 -
// exactly what a lane of v_mfma_scale_f32_32x32x64_f8f6f4 consumes for its (row, 32 k) share (tools/probe_mfma_mx.hip).  The block
// exponent puts the block's largest |x^| into (3.75, 7.5]; codes come from v_cvt_scalef32_2xpk16_fp6_f32 (RN-even, saturating; its
// two 16-float operands interleave: result element 2i = a[i], 2i+1 = b[i] - tools/probe_cvt_fp6.hip), the decoded values from
// v_cvt_scalef32_pk32_f32_fp6, so the row's quantisation error e = x^ - dequant is MEASURED: eps_max[m] receives the largest
// |e|_2 over the map's live rows (float bits, atomic max) - the bound of the mx6 screen is |s6 - a^.q^| <= |ea| + |eq| + |ea||eq|.
// `scale` is not written in this mode.
template <int CP, int LPR, bool NHWC, int FMT>
__device__ __forceinline__ void gather_q8_v3_tile(
    const int bid, char *__restrict__ stage_all, const float *__restrict__ feat, int C, int HW, const int32_t *__restrict__ roi, int roi_stride,
    const int32_t *__restrict__ count, const int32_t *__restrict__ map_enable, int rows_cap, int n_maps, int chunk_tiles,
    int chunks_per_map, int8_t *__restrict__ out8, float *__restrict__ scale, unsigned *__restrict__ eps_max,
    float *__restrict__ norm, float *__restrict__ out32, int round_f16, void *__restrict__ aux)
{
    constexpr int ROWS = 64 / LPR;             // ROI rows per wave
    constexpr int KPL = CP / LPR;              // channels per lane
    constexpr int WG_ROWS = 4 * ROWS;
    static_assert(KPL % 64 == 0 && (LPR == 1 || LPR == 2), "geometry");
    // XCD-aware unit map: a unit is a contiguous chunk of a map's tiles and lives on ONE XCD (block b runs on XCD b % 8), so tiles that
    // share a 128-byte line at their common border share it through that XCD's L2 and a map's channel planes are walked in order
    const int xcd = bid & 7, pos = bid >> 3;
    const int unit = (pos / chunk_tiles) * 8 + xcd;
    if (unit >= n_maps * chunks_per_map) return;
    const int m = unit / chunks_per_map;
    const int wg_tile = (unit % chunks_per_map) * chunk_tiles + pos % chunk_tiles;
    if (map_enable && !map_enable[m]) return;
    const int n = count[m];
    const int n_fill = (n + 255) / 256 * 256;                 // rows [n, n_fill) are written as zero rows
    const int wg_row0 = wg_tile * WG_ROWS;
    if (wg_row0 >= n_fill || wg_row0 >= rows_cap) return;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char *stage = stage_all + wave * G8_STAGE_BYTES;
    const int row0 = wg_row0 + wave * ROWS;
    if (row0 >= n_fill) return;                               // waves are independent: no workgroup barrier anywhere below
    const int lrow = lane % ROWS, seg = lane / ROWS;
    const int my_row = row0 + lrow;
    const bool live = my_row < n;
    const int pix = live ? roi[(size_t)m * roi_stride + my_row] : 0;
    const bool tile_full = (row0 + ROWS <= n) && (C == CP);   // wave-uniform

    // the row's raw values.  FMT = 0: a plain register array.  FMT = 1: sixteen-float register tuples, tuple 2b = the even and tuple
    // 2b + 1 the odd channels of block b - the operand pairs of the fp6 conversion instruction, which takes 16 CONTIGUOUS registers
    // each (built from a scalar array the tuples have to be copied together while all 256 values are live: ~100 scratch spills)
    RowRegs<KPL, FMT> R;
#define VG(k) R.get(k)
#define VS(k, x) R.set(k, x)
    if constexpr (!NHWC) {
        // channel-planar map: value (k, pix) at feat[m][k][pix]
        const char *fb = reinterpret_cast<const char *>(feat + (size_t)m * C * HW);
        const unsigned voff = (unsigned)pix * 4u + (unsigned)seg * (unsigned)KPL * (unsigned)HW * 4u;
        if (tile_full) {
            // buffer loads: the map as a raw buffer (base in SGPRs), the lane's pixel offset in ONE VGPR, the plane offset in an SGPR - no
            // per-load address arithmetic on the VALU (as flat global loads the 256 addresses cost 256 v_lshl_add_u64 per tile)
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(fb), 0, (int)((unsigned)C * (unsigned)HW * 4u), 0x00020000);
#pragma unroll
            for (int i = 0; i < KPL; ++i) VS(i, __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)voff, i * HW * 4, 0)));
        } else {
            // ragged tile (last rows of a map, or C < CP): every lane loads from a clamped, valid address; dead rows / channels are
            // zeroed afterwards.  The bounds go through opaque copies so that the compiler neither shares the 256 plane addresses
            // with the fast path nor keeps 256 predicate masks alive across the loads (it spilled 434 SGPRs doing that).
            int hw_b = HW, c_b = C;
            asm volatile("" : "+s"(hw_b), "+s"(c_b));
            if constexpr (LPR == 1) {
                // narrow maps (the reference's own C = 32 zero-padded to the 256-channel rows): 32-channel blocks beyond C are not loaded
                // at all - a wave-uniform scalar branch per block, no per-lane predicate - instead of 224 clamped re-reads of plane C - 1
#pragma unroll
                for (int b = 0; b < KPL / 32; ++b) {
                    if (32 * b < c_b) {
#pragma unroll
                        for (int i = 0; i < 32; ++i) {
                            int k = 32 * b + i;
                            k = k < c_b ? k : c_b - 1;
                            VS(32 * b + i, *reinterpret_cast<const float *>(fb + (size_t)k * hw_b * 4 + (unsigned)pix * 4u));
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < 32; ++i) VS(32 * b + i, 0.0f);
                    }
                }
            } else {
#pragma unroll
            for (int i = 0; i < KPL; ++i) {
                int k = seg * KPL + i;
                k = k < c_b ? k : c_b - 1;
                VS(i, *reinterpret_cast<const float *>(fb + (size_t)k * hw_b * 4 + (unsigned)pix * 4u));
            }
            }
            int c_c = C;
            asm volatile("" : "+s"(c_c));
#pragma unroll
            for (int i = 0; i < KPL; ++i) VS(i, (live && (seg * KPL + i < c_c)) ? VG(i) : 0.0f);
        }
    } else {
        // channels_last map: value (k, pix) at feat[m][pix][k].  64 channels of the wave's 64 / LPR rows at a time: 4 rows x 256 bytes
        // per load instruction -> staging buffer -> lane = (row, segment).  The loads of chunk c+1 are in flight while chunk c goes
        // through the staging buffer (two register sets of 16 x 16 bytes).
        const float *fb = feat + (size_t)m * C * HW;
        const int sub = lane >> 4, slot = lane & 15;          // load role: row sub-index, 16-byte slot of a 256-byte run
        constexpr int NCH = KPL / 64;
        // pixel of the 16 rows this lane loads from (tile-lane t = j*4 + sub -> row t % ROWS), -1 = dead row
        int px[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int rr = row0 + (j * 4 + sub) % ROWS;
            px[j] = rr < n ? roi[(size_t)m * roi_stride + rr] : -1;
        }
        auto load_chunk = [&](int c, float4 (&x)[16]) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int sg = (j * 4 + sub) / ROWS;
                const int k0 = sg * KPL + 64 * c + slot * 4;
                float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
                if (tile_full) {
                    val = *reinterpret_cast<const float4 *>(fb + (size_t)px[j] * C + k0);
                } else if (px[j] >= 0) {
                    const float *src = fb + (size_t)px[j] * C;
                    val.x = k0 + 0 < C ? src[k0 + 0] : 0.f; val.y = k0 + 1 < C ? src[k0 + 1] : 0.f;
                    val.z = k0 + 2 < C ? src[k0 + 2] : 0.f; val.w = k0 + 3 < C ? src[k0 + 3] : 0.f;
                }
                x[j] = val;
            }
        };
        float4 xa[16], xb[16];
        load_chunk(0, xa);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            float4 (&cur)[16] = (c & 1) ? xb : xa;
            float4 (&nxt)[16] = (c & 1) ? xa : xb;
            if (c + 1 < NCH) load_chunk(c + 1, nxt);
            WAVE_LDS_ORDER();                                           // previous chunk's staging reads precede these writes
#pragma unroll
            for (int j = 0; j < 16; ++j) *reinterpret_cast<float4 *>(stage + stage_addr(j * 4 + sub, slot, 256)) = cur[j];
            WAVE_LDS_ORDER();
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const float4 q = *reinterpret_cast<const float4 *>(stage + stage_addr(lane, s, 256));
                VS(64 * c + 4 * s + 0, q.x); VS(64 * c + 4 * s + 1, q.y); VS(64 * c + 4 * s + 2, q.z); VS(64 * c + 4 * s + 3, q.w);
            }
        }
    }

    // the reference's half-descriptor branch (utils/pcd.py:195-197: feats.half() before pdist): every raw value is rounded to the
    // nearest float16 first; everything downstream runs unchanged on the rounded values
    if (round_f16) {
#pragma unroll
        for (int i = 0; i < KPL; ++i) VS(i, __half2float(__float2half_rn(VG(i))));
    }

    // canonical norm: ONE k-ordered fmaf chain per row.  LPR = 2: lanes of segment 1 continue from segment 0's result.
    float n2 = chain_sq<KPL>(R, 0.0f);
    if constexpr (LPR == 2) {
        const float lo = __shfl(n2, lrow);                    // segment 0's partial chain of the same row
        const float full = chain_sq<KPL>(R, lo);              // meaningful on segment-1 lanes
        n2 = __shfl(full, ROWS + lrow);
    }
    float d = sqrt_rn(n2);
    d = d < 1e-8f ? 1e-8f : d;
    float rs = 0.0f;
    if constexpr (FMT == 0) {
    float mx = 0.0f;
#pragma unroll
    for (int i = 0; i < KPL; ++i) mx = fmaxf(mx, fabsf(VG(i)));
    if constexpr (LPR == 2) mx = fmaxf(mx, __shfl_xor(mx, 32));
    float mxs = __fdiv_rn(mx, d);                             // = max_k |x_k / d| (IEEE division is monotone)
    // slice = 16 rows {0-3,8-11,16-19,24-27} + 4h of a 32-row block: the lanes that differ in row bits 0,1,3,4
    mxs = fmaxf(mxs, __shfl_xor(mxs, 1));
    mxs = fmaxf(mxs, __shfl_xor(mxs, 2));
    mxs = fmaxf(mxs, __shfl_xor(mxs, 8));
    mxs = fmaxf(mxs, __shfl_xor(mxs, 16));
    int E = mxs > 0.0f ? ilogbf(127.0f / mxs) : 30;
    E = E > 30 ? 30 : (E < 0 ? 0 : E);
    const float sc = ldexpf(1.0f, E);
    rs = __fdiv_rn(sc, d);
    const int h = (lrow >> 2) & 1;
    if (seg == 0 && (lrow & 27) == 0) scale[(size_t)m * (rows_cap / 16) + (my_row >> 5) * 2 + h] = ldexpf(1.0f, -E);
    {
        // per-map quantisation bound: slices that hold at least one live row
        float e = live ? ldexpf(1.0f, -E - 1) : 0.0f;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) e = fmaxf(e, __shfl_xor(e, off));
        if (lane == 0 && e > 0.0f) atomicMax(&eps_max[m], __float_as_uint(e));
    }
    }
    if (seg == 0 && norm) norm[(size_t)m * rows_cap + my_row] = d;

    if constexpr (FMT == 2 || FMT == 3) {
        // error-compensated fp16 operands of the canonical unit row u = RN(x / d): hi = half(u), lo = half(u - hi) (22 significant
        // bits together), natural k order, as two row arrays [n_maps, rows_cap, CP] of halves (out8 = hi rows, aux = lo rows): the
        // operands of match_x3_scan_kernel.
        static_assert(LPR == 1, "the hi / lo rows are written with one lane per row");
        // FMT = 2: out8 = hi rows, aux = lo rows.  FMT = 3 (round 4: the engine's K0 pass when recent steps needed the second level - one
        // read of the maps instead of two): out8 keeps the mx6 slots, aux = hi rows followed by the lo rows ([2][n_maps, rows_cap, CP]
        // halves), `scale` (unused by the mx6 format) receives the largest |u - hi|^2 per map, eps_max the mx6 error norm as in FMT = 1.
        constexpr int HB = CP * 2;                              // bytes per half row
        float lo2 = 0.0f;                                       // |lo|^2 of the lane's row: K1x3's first sweep multiplies hi parts only
        // the unit values replace the raw ones in place first (computed inside the staging passes, all 256 quotients were hoisted and
        // lived beside the 256 raw values: ~100 registers in scratch)
#pragma unroll
        for (int i = 0; i < KPL; ++i) {
            VS(i, __fdiv_rn(VG(i), d));
            if ((i & 15) == 15) __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int part = 0; part < KPL / 64; ++part) {           // 64 channels per staging pass: slots 0-7 = hi halves, 8-15 = lo halves
            // chunks that hold none of the map's C channels are not written at all (their values are zero and every reader of the hi / lo
            // arrays skips them: match_x3.hip takes ceil(C / 64) live chunks) - a C = 32 map writes 128 + 128 bytes per row, not 1 KB
            if (part * 64 >= C) continue;
            const int c0 = part * 64;
            WAVE_LDS_ORDER();
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                unsigned wh[4], wl[4];
                float sq[4];                                    // per-slot partial sums: ONE long fmaf chain over all 256 values made the
                                                                // scheduler hoist every conversion in front of it (~100 registers in scratch)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float u0 = VG(c0 + 8 * s + 2 * j), u1 = VG(c0 + 8 * s + 2 * j + 1);
                    const __half h0 = __float2half_rn(u0), h1 = __float2half_rn(u1);
                    const float d0 = u0 - __half2float(h0), d1 = u1 - __half2float(h1);          // exact: hi is u rounded to 11 bits
                    const __half l0 = __float2half_rn(d0), l1 = __float2half_rn(d1);
                    wh[j] = (unsigned)__half_as_ushort(h0) | ((unsigned)__half_as_ushort(h1) << 16);
                    wl[j] = (unsigned)__half_as_ushort(l0) | ((unsigned)__half_as_ushort(l1) << 16);
                    sq[j] = fmaf(d0, d0, d1 * d1);                                                // |lo| <= |d| (1 + 2^-11)
                }
                lo2 += (sq[0] + sq[1]) + (sq[2] + sq[3]);
                *reinterpret_cast<uint4 *>(stage + stage_addr(lane, s, 256)) = make_uint4(wh[0], wh[1], wh[2], wh[3]);
                *reinterpret_cast<uint4 *>(stage + stage_addr(lane, s + 8, 256)) = make_uint4(wl[0], wl[1], wl[2], wl[3]);
            }
            WAVE_LDS_ORDER();
            // Row layout of the hi / lo arrays (round 4): tiles of 32 rows, inside a tile the four 64-channel chunks one after the other -
            // byte (row q, b) of a map at (q >> 5) * 16384 + (b >> 7) * 4096 + (q & 31) * 128 + (b & 127).  The second sweep of K1x3 streams
            // a tile chunk by chunk: each is now one contiguous 4 KB run per part instead of 32 pieces of 128 bytes 512 bytes apart.
            char *oh = (FMT == 3 ? reinterpret_cast<char *>(aux) : reinterpret_cast<char *>(out8)) + ((size_t)m * rows_cap + row0) * HB + part * 4096;
            char *ol = (FMT == 3 ? reinterpret_cast<char *>(aux) + (size_t)n_maps * rows_cap * HB : reinterpret_cast<char *>(aux)) +
                       ((size_t)m * rows_cap + row0) * HB + part * 4096;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int f = j * 64 + lane, t = f / 16, s = f % 16;
                const uint4 q = *reinterpret_cast<const uint4 *>(stage + stage_addr(t, s, 256));
                *reinterpret_cast<uint4 *>((s < 8 ? oh : ol) + (size_t)(t >> 5) * (32 * HB) + (t & 31) * 128 + (s & 7) * 16) = q;
            }
        }
        unsigned *lo_max = FMT == 3 ? reinterpret_cast<unsigned *>(scale) : eps_max;
        if (lo_max) {                                           // largest |u - hi|_2^2 of the map's rows (|lo| <= |u - hi| (1 + 2^-11))
            // No wave reduction here: any cross-lane operation at this point (shuffles, DPP) cost ~100 registers of scratch in this
            // instantiation.  Every lane compares with the map's current maximum (a cached broadcast load; stale values only cost an extra
            // atomic) and only lanes that would raise it issue the atomic: a few per map after the first waves.  Dead rows are zero rows.
            // The SQUARE goes out (the consumer takes the root).
            if (lo2 > 0.0f && __float_as_uint(lo2) > __atomic_load_n(&lo_max[m], __ATOMIC_RELAXED)) atomicMax(&lo_max[m], __float_as_uint(lo2));
        }
    }
    if constexpr (FMT == 0) {
    // int8 rows: magic-number rounding (RN-even, like rintf) - the low byte of (x*rs + 1.5*2^23) is the two's complement of the integer
        constexpr int RB8 = KPL;                              // bytes per tile-lane row
#pragma unroll
        for (int s = 0; s < KPL / 16; ++s) {
            unsigned w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                unsigned b[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) b[e] = __float_as_uint(__fmaf_rn(VG(16 * s + 4 * j + e), rs, 12582912.0f));
                const unsigned lo = __builtin_amdgcn_perm(b[1], b[0], 0x0c0c0400u);      // bytes: b0[0], b1[0], 0, 0
                const unsigned hi2 = __builtin_amdgcn_perm(b[3], b[2], 0x04000c0cu);     // bytes: 0, 0, b2[0], b3[0]
                w[j] = lo | hi2;
            }
            *reinterpret_cast<uint4 *>(stage + stage_addr(lane, s, RB8)) = make_uint4(w[0], w[1], w[2], w[3]);
        }
        WAVE_LDS_ORDER();
        constexpr int SPR = RB8 / 16;                         // 16-byte slots per tile-lane row
        char *o8 = reinterpret_cast<char *>(out8) + ((size_t)m * rows_cap + row0) * CP;
#pragma unroll
        for (int j = 0; j < SPR; ++j) {
            const int f = j * 64 + lane, t = f / SPR, s = f % SPR;
            const uint4 q = *reinterpret_cast<const uint4 *>(stage + stage_addr(t, s, RB8));
            *reinterpret_cast<uint4 *>(o8 + (size_t)(t % ROWS) * CP + (t / ROWS) * KPL + s * 16) = q;
        }
    }

    if (out32) {
        // canonical unit rows, k-permuted inside groups of 8 (position 8g + 4h + j holds k = 8g + 2j + h): 64 channels per pass
        float *o32 = out32 + ((size_t)m * rows_cap + row0) * CP;
#pragma unroll
        for (int c = 0; c < KPL / 64; ++c) {
            WAVE_LDS_ORDER();
#pragma unroll
            for (int g = 0; g < 8; ++g)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    float4 q;
                    q.x = __fdiv_rn(VG(64 * c + 8 * g + 0 + hh), d);
                    q.y = __fdiv_rn(VG(64 * c + 8 * g + 2 + hh), d);
                    q.z = __fdiv_rn(VG(64 * c + 8 * g + 4 + hh), d);
                    q.w = __fdiv_rn(VG(64 * c + 8 * g + 6 + hh), d);
                    *reinterpret_cast<float4 *>(stage + stage_addr(lane, 2 * g + hh, 256)) = q;
                }
            WAVE_LDS_ORDER();
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int t = j * 4 + (lane >> 4), s = lane & 15;
                const float4 q = *reinterpret_cast<const float4 *>(stage + stage_addr(t, s, 256));
                *reinterpret_cast<float4 *>(o32 + (size_t)(t % ROWS) * CP + (t / ROWS) * KPL + 64 * c + s * 4) = q;
            }
        }
    }
    if constexpr (FMT == 1 || FMT == 3) {
        // MX-fp6 slots (see the kernel's header): one 32-channel block = staging slots 2b (code dwords 0-3) and 2b + 1 (code dwords
        // 4-5, exponent byte, zero)
        constexpr int RB8 = KPL;
        float err2 = 0.0f;
        const float rd = __fdiv_rn(1.0f, d);                             // x * rd = x / d up to 2 ulp: inside the bound's slack
#pragma unroll
        for (int b = 0; b < KPL / 32; ++b) {
            float bm = 0.0f;
#pragma unroll
            for (int i = 0; i < 32; ++i) bm = fmaxf(bm, fabsf(VG(32 * b + i)));
            // block exponent e: bm / (d * 2^e) in (3.75, 7.5]; an all-zero block keeps e = -40.  (FMT = 3: the registers already hold the
            // canonical unit values x / d - the hi / lo pass above divided in place)
            const float r = (FMT == 3 ? bm : __fdiv_rn(bm, d)) * (1.0f / 7.5f);
            int e = r > 0.0f ? ilogbf(r) + 1 : -40;
            if (r > 0.0f && ldexpf(1.0f, e - 1) >= r) e -= 1;          // r an exact power of two: 2^(e-1) == r is enough
            e = e < -40 ? -40 : (e > 8 ? 8 : e);
            // values in code units: x / (d 2^e) (rd 2^-e is exact, a power of two times rd).  The conversions run with the constant
            // scale 1: a per-lane scale operand gave wrong codes (every row has its own exponent here)
            const float rde = (FMT == 3 ? 1.0f : rd) * ldexpf(1.0f, -e);
            R.t[2 * b] *= rde;                                          // in place: the raw values are not needed any more (out32 is done)
            R.t[2 * b + 1] *= rde;
            const f32x16g ev = R.t[2 * b], od = R.t[2 * b + 1];
            const u32x6g c = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(ev, od, 1.0f);
            const f32x32g back = __builtin_amdgcn_cvt_scalef32_pk32_f32_fp6(c, 1.0f);    // the codes' values: what the MFMA multiplies (x 2^e)
            float be = 0.0f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float d0 = ev[i] - back[2 * i], d1 = od[i] - back[2 * i + 1];
                be = __fmaf_rn(d0, d0, be);
                be = __fmaf_rn(d1, d1, be);
            }
            err2 = __fmaf_rn(be, ldexpf(1.0f, 2 * e), err2);
            *reinterpret_cast<uint4 *>(stage + stage_addr(lane, 2 * b, RB8)) = make_uint4(c[0], c[1], c[2], c[3]);
            *reinterpret_cast<uint4 *>(stage + stage_addr(lane, 2 * b + 1, RB8)) = make_uint4(c[4], c[5], (unsigned)(e + 127), 0u);
            // one block at a time: left to itself the scheduler interleaves all eight blocks' conversions (each needs 2 x 16 + 32
            // contiguous registers) on top of the 256 live raw values and spills ~100 registers to scratch
            __builtin_amdgcn_sched_barrier(0);
        }
        // |e|_2 of the row (already in unit terms), rounded up: the fp32 sum of <= 512 squares is within 3e-5 relative (1.00002 after
        // the square root), + 3e-7 covers x * rd against the canonical x / d
        if constexpr (LPR == 2) err2 += __shfl_xor(err2, 32);
        float er = live ? sqrt_rn(err2) * 1.00002f + 3e-7f : 0.0f;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) er = fmaxf(er, __shfl_xor(er, off));
        if (lane == 0 && er > 0.0f) atomicMax(&eps_max[m], __float_as_uint(er));
        WAVE_LDS_ORDER();
        constexpr int SPR = RB8 / 16;
        char *o8 = reinterpret_cast<char *>(out8) + ((size_t)m * rows_cap + row0) * CP;
#pragma unroll
        for (int j = 0; j < SPR; ++j) {
            const int f = j * 64 + lane, t = f / SPR, s = f % SPR;
            const uint4 q = *reinterpret_cast<const uint4 *>(stage + stage_addr(t, s, RB8));
            *reinterpret_cast<uint4 *>(o8 + (size_t)(t % ROWS) * CP + (t / ROWS) * KPL + s * 16) = q;
        }
    }
#undef VG
#undef VS
}

// one workgroup tile per block (the K0 passes proper)
template <int CP, int LPR, bool NHWC, int FMT = 0>
__global__ __launch_bounds__(256, (CP / LPR > 128 ? 1 : 2)) void gather_q8_v3_kernel(
    const float *__restrict__ feat, int C, int HW, const int32_t *__restrict__ roi, int roi_stride,
    const int32_t *__restrict__ count, const int32_t *__restrict__ map_enable, int rows_cap, int n_maps, int chunk_tiles,
    int chunks_per_map, int8_t *__restrict__ out8, float *__restrict__ scale, unsigned *__restrict__ eps_max,
    float *__restrict__ norm, float *__restrict__ out32, int round_f16, void *__restrict__ aux)
{
    extern __shared__ __attribute__((aligned(256))) char stage_all[];
    gather_q8_v3_tile<CP, LPR, NHWC, FMT>((int)blockIdx.x, stage_all, feat, C, HW, roi, roi_stride, count, map_enable, rows_cap, n_maps, chunk_tiles,
                                           chunks_per_map, out8, scale, eps_max, norm, out32, round_f16, aux);
}

// The matcher's device-gated passes (fp32 query rows for pairs with ambiguous anchors, hi / lo rows for K1x3, overflow rows): on most steps
// no map is enabled, and a grid of one block per tile then dispatches ~40 k workgroups only to let them exit (15-35 us per launch on the
// matcher's stream, three launches per step).  Here a fixed grid walks the tiles (stride = grid size, a multiple of 8: a block keeps its XCD).
template <int CP, int LPR, bool NHWC, int FMT>
__global__ __launch_bounds__(256, (CP / LPR > 128 ? 1 : 2)) void gather_q8_v3_gated_kernel(
    int total_blocks, const float *__restrict__ feat, int C, int HW, const int32_t *__restrict__ roi, int roi_stride,
    const int32_t *__restrict__ count, const int32_t *__restrict__ map_enable, int rows_cap, int n_maps, int chunk_tiles,
    int chunks_per_map, int8_t *__restrict__ out8, float *__restrict__ scale, unsigned *__restrict__ eps_max,
    float *__restrict__ norm, float *__restrict__ out32, int round_f16, void *__restrict__ aux)
{
    extern __shared__ __attribute__((aligned(256))) char stage_all[];
    // nothing enabled (the usual case): one parallel look at the gates instead of a dependent load per tile
    {
        bool any = false;
        for (int m = threadIdx.x & 63; m < n_maps; m += 64) any |= map_enable[m] != 0;
        if (__ballot(any) == 0ull) return;
    }
    for (int bid = (int)blockIdx.x; bid < total_blocks; bid += (int)gridDim.x) {
        WAVE_LDS_ORDER();                                         // the previous tile's staging reads precede this tile's writes
        gather_q8_v3_tile<CP, LPR, NHWC, FMT>(bid, stage_all, feat, C, HW, roi, roi_stride, count, map_enable, rows_cap, n_maps, chunk_tiles,
                                               chunks_per_map, out8, scale, eps_max, norm, out32, round_f16, aux);
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// K0v4 (round 5): the MX-fp6 pass over channel-planar maps with the 256 channels of a 64-row tile SPLIT OVER THE FOUR WAVES of a
// workgroup.  K0v3 keeps a row's 256 raw values in one lane: 512 registers, one wave per SIMD, ~800 v_accvgpr moves per tile, and a
// wave's loads, its 255-step norm chain and its conversions run one after the other with nobody to hide them.  Here wave w owns
// channels [64 w, 64 w + 64) = two 32-channel blocks of the tile's 64 rows (lane = row): 64 raw values per lane in the four
// 16-float tuples the fp6 conversion consumes, < 128 registers, four workgroups (16 waves) per CU, so one workgroup's loads fly
// under another's arithmetic.
//   * canonical norm: the k-ordered fmaf chain is handed from wave to wave through LDS (wave w continues from wave w - 1's partial
//     sum): the accumulation order is k = 0 ... 255 as before, so d - and with it every fp32 unit row - is bit for bit K0v3's;
//   * block exponents, codes, the measured error: per block exactly the arithmetic of K0v3 (FMT = 1); the row's error sum is formed
//     from the per-block (sum, exponent) pairs in block order by one wave, i.e. the same fmaf chain;
//   * stores: the tile's 64 x 256-byte mx6 rows are assembled in a 16 KB LDS stage (the same XOR swizzle) and leave as ONE
//     contiguous 16 KB run; the fp32 anchor rows (WANT32) go through wave-private stages as in K0v3.
// Reference: utils/pcd.py:184-193 (the gather), :28-29 (the normalisation inside cosine_similarity).
constexpr int G4_STAGE = 16384;                 // 64 rows x 256 B
constexpr int G4_HAND = 4 * 64 * 4;             // chain hand-off [wave][row]
constexpr int G4_TERMS = 2 * 8 * 64 * 4;        // (block error sum, block exponent) [block][row]
constexpr int G4_LDS = G4_STAGE + G4_HAND + G4_TERMS;
constexpr int G4_STAGE32 = 4096;              // per wave, WANT32 only: 64 rows x 64 B

template <bool WANT32, bool X3 = false>
__global__ __launch_bounds__(256, X3 ? 3 : 4) void gather_mx6_v4_kernel(
    const float *__restrict__ feat, int C, int HW, const int32_t *__restrict__ roi, int roi_stride, const int32_t *__restrict__ count,
    int rows_cap, int n_maps, int chunk_tiles, int chunks_per_map, uint8_t *__restrict__ out8, unsigned *__restrict__ eps_max,
    float *__restrict__ norm, float *__restrict__ out32, int round_f16, void *__restrict__ aux = nullptr, unsigned *__restrict__ lo_max = nullptr)
{
    static_assert(!(WANT32 && X3), "the hi / lo rows are a query-side product");
    extern __shared__ __attribute__((aligned(256))) char lds4[];
    char *stage = lds4;
    float *hand = reinterpret_cast<float *>(lds4 + G4_STAGE);
    float *term_be = reinterpret_cast<float *>(lds4 + G4_STAGE + G4_HAND);
    int *term_e = reinterpret_cast<int *>(lds4 + G4_STAGE + G4_HAND + G4_TERMS / 2);

    const int bid = (int)blockIdx.x;
    const int xcd = bid & 7, pos = bid >> 3;
    const int unit = (pos / chunk_tiles) * 8 + xcd;
    if (unit >= n_maps * chunks_per_map) return;
    const int m = unit / chunks_per_map;
    const int tile = (unit % chunks_per_map) * chunk_tiles + pos % chunk_tiles;
    const int n = count[m];
    const int n_fill = (n + 255) / 256 * 256;                 // rows [n, n_fill) are written as zero rows
    const int row0 = tile * 64;
    if (row0 >= n_fill || row0 >= rows_cap) return;           // workgroup-uniform: no barrier is skipped by part of a workgroup

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int my_row = row0 + lane;
    const bool live = my_row < n;
    const int pix = live ? roi[(size_t)m * roi_stride + my_row] : 0;
    const bool tile_full = (row0 + 64 <= n) && (C == 256);    // workgroup-uniform

    RowRegs<64, 1> R;                                         // local channel i = global channel 64 wave + i
    const char *fb = reinterpret_cast<const char *>(feat + (size_t)m * C * HW);
    if (tile_full) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(fb), 0, (int)((unsigned)C * (unsigned)HW * 4u), 0x00020000);
        const int s0 = wave * 64 * HW * 4;
#pragma unroll
        for (int i = 0; i < 64; ++i) R.set(i, __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, pix * 4, s0 + i * HW * 4, 0)));
    } else {
        int hw_b = HW, c_b = C;
        asm volatile("" : "+s"(hw_b), "+s"(c_b));
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int kb = wave * 64 + 32 * b;
            if (kb < c_b) {
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    int k = kb + i;
                    k = k < c_b ? k : c_b - 1;
                    R.set(32 * b + i, *reinterpret_cast<const float *>(fb + (size_t)k * hw_b * 4 + (unsigned)pix * 4u));
                }
            } else {
#pragma unroll
                for (int i = 0; i < 32; ++i) R.set(32 * b + i, 0.0f);
            }
        }
        int c_c = C;
        asm volatile("" : "+s"(c_c));
#pragma unroll
        for (int i = 0; i < 64; ++i) R.set(i, (live && (wave * 64 + i < c_c)) ? R.get(i) : 0.0f);
    }
    if (round_f16) {
#pragma unroll
        for (int i = 0; i < 64; ++i) R.set(i, __half2float(__float2half_rn(R.get(i))));
    }

    // canonical norm: the chain passes through the four waves in channel order
    float acc = 0.0f;
#pragma unroll 1
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
            if (w > 0) acc = hand[(w - 1) * 64 + lane];
            acc = chain_sq<64>(R, acc);
            hand[w * 64 + lane] = acc;
        }
        __syncthreads();
    }
    float d = sqrt_rn(hand[3 * 64 + lane]);
    d = d < 1e-8f ? 1e-8f : d;
    if (wave == 3 && norm) norm[(size_t)m * rows_cap + my_row] = d;

    if constexpr (WANT32) {
        // canonical unit rows, k-permuted inside groups of 8 (position 8g + 4h + j holds k = 8g + 2j + h): this wave's 64 channels in four
        // passes of 16 through a wave-private 4 KB stage (64 rows x 64 B), out as 16 rows x 64 bytes per store instruction.  (A 16 KB
        // stage per wave - K0v3's - would leave room for two workgroups per CU only.)
        // (two passes of 128 bytes per row since the end of round 5: rows 0-31 through this wave's quarter of the still unused mx6 stage,
        // rows 32-63 through its private 4 KB; 8 rows x 128 B per store instruction instead of 16 x 64 B)
        char *stA = stage + wave * 4096, *stB = lds4 + G4_LDS + wave * G4_STAGE32;
        char *my_st = (lane < 32 ? stA : stB) + (lane & 31) * 128;
        float *o32 = out32 + ((size_t)m * rows_cap + row0) * 256 + 64 * wave;
#pragma unroll
        for (int P = 0; P < 2; ++P) {
            WAVE_LDS_ORDER();
#pragma unroll
            for (int sl = 0; sl < 8; ++sl) {
                const int g = 4 * P + (sl >> 1), hh = sl & 1;
                float4 q;
                q.x = __fdiv_rn(R.get(8 * g + 0 + hh), d);
                q.y = __fdiv_rn(R.get(8 * g + 2 + hh), d);
                q.z = __fdiv_rn(R.get(8 * g + 4 + hh), d);
                q.w = __fdiv_rn(R.get(8 * g + 6 + hh), d);
                *reinterpret_cast<float4 *>(my_st + ((sl ^ (lane & 7)) << 4)) = q;
            }
            WAVE_LDS_ORDER();
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int t = j * 8 + (lane >> 3), sl = lane & 7;
                const float4 q = *reinterpret_cast<const float4 *>((t < 32 ? stA : stB) + (t & 31) * 128 + ((sl ^ (t & 7)) << 4));
                *reinterpret_cast<float4 *>(o32 + (size_t)t * 256 + 32 * P + sl * 4) = q;
            }
        }
        __syncthreads();                                              // the mx6 stage below overlaps the other waves' transposition areas
    }

    if constexpr (X3) {
        // FMT = 3 (the engine's K0 pass when recent steps needed the second level): besides the mx6 slots, the canonical unit row
        // u = RN(x / d) as error-compensated float16 halves hi = half(u), lo = half(u - hi) for K1x3 (match_x3.hip) - layout of K0v3:
        // tiles of 32 rows, inside a tile the four 64-channel chunks one after the other, hi array then lo array in `aux`.  This
        // wave's chunk is chunk `wave`; it leaves in two passes (hi, lo) of 128 bytes per row (below).  The unit values replace the raw ones in place (the fp6
        // conversion below then works on them, as K0v3's FMT = 3 does).
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            R.set(i, __fdiv_rn(R.get(i), d));
            if ((i & 15) == 15) __builtin_amdgcn_sched_barrier(0);
        }
        float lo2 = 0.0f;
        if (wave * 64 < C) {                                            // chunks beyond the map's channels are not written (readers skip them)
            constexpr size_t HB = 512;                                  // bytes per half row
            // two passes (hi, lo) of 128 bytes per row: rows 0-31 through this wave's quarter of the (still unused) mx6 stage, rows
            // 32-63 through its private 4 KB, 16-byte slots XOR-swizzled with the row; out as 8 rows x 128 B = 1 KB contiguous per store
            // instruction (a 32-row tile of one chunk is 4 KB contiguous in `aux`).  Four passes of 64-byte pieces before: 1.48 -> 1.16 ms for the
            // query pass of cfg2 (64 x 36 864 rows), outputs unchanged - the half-filled 128-byte lines were what held the pass at 4 TB/s
            char *stA = stage + wave * 4096, *stB = lds4 + G4_LDS + wave * G4_STAGE32;
            char *my_st = (lane < 32 ? stA : stB) + (lane & 31) * 128;
            char *ob[2];
            ob[0] = reinterpret_cast<char *>(aux) + ((size_t)m * rows_cap + row0) * HB + wave * 4096;
            ob[1] = ob[0] + (size_t)n_maps * rows_cap * HB;
#pragma unroll
            for (int lo_pass = 0; lo_pass < 2; ++lo_pass) {
                WAVE_LDS_ORDER();
#pragma unroll
                for (int sl = 0; sl < 8; ++sl) {                        // channels 8 sl .. 8 sl + 7 of this wave's 64
                    unsigned w[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float u0 = R.get(8 * sl + 2 * j), u1 = R.get(8 * sl + 2 * j + 1);
                        const __half h0 = __float2half_rn(u0), h1 = __float2half_rn(u1);
                        if (lo_pass) {
                            const float d0 = u0 - __half2float(h0), d1 = u1 - __half2float(h1);      // exact: hi is u rounded to 11 bits
                            const __half l0 = __float2half_rn(d0), l1 = __float2half_rn(d1);
                            w[j] = (unsigned)__half_as_ushort(l0) | ((unsigned)__half_as_ushort(l1) << 16);
                            lo2 += fmaf(d0, d0, d1 * d1);                                               // |lo| <= |d| (1 + 2^-11)
                        } else {
                            w[j] = (unsigned)__half_as_ushort(h0) | ((unsigned)__half_as_ushort(h1) << 16);
                        }
                    }
                    *reinterpret_cast<uint4 *>(my_st + ((sl ^ (lane & 7)) << 4)) = make_uint4(w[0], w[1], w[2], w[3]);
                }
                WAVE_LDS_ORDER();
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int t = j * 8 + (lane >> 3), sl = lane & 7;     // row t of the tile, 16-byte slot sl of its 128 bytes
                    const uint4 q = *reinterpret_cast<const uint4 *>((t < 32 ? stA : stB) + (t & 31) * 128 + ((sl ^ (t & 7)) << 4));
                    *reinterpret_cast<uint4 *>(ob[lo_pass] + (size_t)(t >> 5) * (32 * HB) + (t & 31) * 128 + sl * 16) = q;
                }
            }
        }
        reinterpret_cast<float *>(lds4 + G4_LDS + 4 * G4_STAGE32)[wave * 64 + lane] = lo2;       // per-wave partial |u - hi|^2 of the row
        __syncthreads();                                                // the mx6 stage below overlaps the other waves' transposition areas
    }

    // MX-fp6 slots of this wave's two blocks (arithmetic of K0v3, FMT = 1; FMT = 3: on the unit values)
    const float rd = __fdiv_rn(1.0f, d);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int gb = 2 * wave + b;
        float bm = 0.0f;
#pragma unroll
        for (int i = 0; i < 32; ++i) bm = fmaxf(bm, fabsf(R.get(32 * b + i)));
        const float r = (X3 ? bm : __fdiv_rn(bm, d)) * (1.0f / 7.5f);
        int e = r > 0.0f ? ilogbf(r) + 1 : -40;
        if (r > 0.0f && ldexpf(1.0f, e - 1) >= r) e -= 1;
        e = e < -40 ? -40 : (e > 8 ? 8 : e);
        const float rde = (X3 ? 1.0f : rd) * ldexpf(1.0f, -e);
        R.t[2 * b] *= rde;
        R.t[2 * b + 1] *= rde;
        const f32x16g ev = R.t[2 * b], od = R.t[2 * b + 1];
        const u32x6g c = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(ev, od, 1.0f);
        const f32x32g back = __builtin_amdgcn_cvt_scalef32_pk32_f32_fp6(c, 1.0f);
        float be = 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float d0 = ev[i] - back[2 * i], d1 = od[i] - back[2 * i + 1];
            be = __fmaf_rn(d0, d0, be);
            be = __fmaf_rn(d1, d1, be);
        }
        term_be[gb * 64 + lane] = be;
        term_e[gb * 64 + lane] = e;
        *reinterpret_cast<uint4 *>(stage + stage_addr(lane, 2 * gb, 256)) = make_uint4(c[0], c[1], c[2], c[3]);
        *reinterpret_cast<uint4 *>(stage + stage_addr(lane, 2 * gb + 1, 256)) = make_uint4(c[4], c[5], (unsigned)(e + 127), 0u);
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    {
        char *o8 = reinterpret_cast<char *>(out8) + ((size_t)m * rows_cap + row0) * 256;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f = j * 256 + (int)threadIdx.x;
            const uint4 q = *reinterpret_cast<const uint4 *>(stage + stage_addr(f >> 4, f & 15, 256));
            *reinterpret_cast<uint4 *>(o8 + (size_t)f * 16) = q;
        }
    }
    if (wave == 0) {
        // |e|_2 of the row: the per-block sums in block order, the very chain K0v3 runs in one lane
        float err2 = 0.0f;
#pragma unroll
        for (int gb = 0; gb < 8; ++gb) err2 = __fmaf_rn(term_be[gb * 64 + lane], ldexpf(1.0f, 2 * term_e[gb * 64 + lane]), err2);
        float er = live ? sqrt_rn(err2) * 1.00002f + 3e-7f : 0.0f;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) er = fmaxf(er, __shfl_xor(er, off));
        if (lane == 0 && er > 0.0f) atomicMax(&eps_max[m], __float_as_uint(er));
        if constexpr (X3) {
            // largest |u - hi|_2^2 of the map's rows (K1x3's first sweep multiplies hi parts only); dead rows are zero rows
            const float *lp = reinterpret_cast<const float *>(lds4 + G4_LDS + 4 * G4_STAGE32);
            const float lo2 = ((lp[lane] + lp[64 + lane]) + lp[128 + lane]) + lp[192 + lane];
            if (lo_max && lo2 > 0.0f && __float_as_uint(lo2) > __atomic_load_n(&lo_max[m], __ATOMIC_RELAXED)) atomicMax(&lo_max[m], __float_as_uint(lo2));
        }
    }
}

}  // namespace oryon

using namespace oryon;

namespace {
template <int CP, int LPR, bool NHWC, int FMT = 0>
void launch_g8(hipStream_t st, const float *feat, int n_maps, int C, int HW, const int32_t *roi, int roi_stride, const int32_t *count,
               const int32_t *map_enable, int rows_cap, int8_t *out8, float *scale, float *eps, float *norm, float *out32, int round_f16,
               void *aux = nullptr)
{
    constexpr int WG_ROWS = 4 * (64 / LPR);
    const int T = (rows_cap + WG_ROWS - 1) / WG_ROWS;                 // workgroup tiles per map
    const int chunks_per_map = n_maps >= 8 ? 1 : (8 + n_maps - 1) / n_maps;
    const int chunk_tiles = (T + chunks_per_map - 1) / chunks_per_map;
    const int units = n_maps * chunks_per_map;
    const int groups = ((units + 7) / 8) * 8 * chunk_tiles;
    // (the hi / lo pass of K1x3 stays one tile per block: it does real work on every step of smooth inputs and the tile loop costs it 20 %:
    // 1.53 against 1.27 ms; the fp32-row passes are rare fall-backs)
    if (map_enable && FMT != 2 && FMT != 3) {
        auto gk = gather_q8_v3_gated_kernel<CP, LPR, NHWC, FMT>;
        allow_dynamic_lds(reinterpret_cast<const void *>(gk), 4 * G8_STAGE_BYTES);
        const int grid = groups < 2048 ? groups : 2048;
        hipLaunchKernelGGL(gk, dim3(grid), dim3(256), 4 * G8_STAGE_BYTES, st, groups, feat, C, HW, roi, roi_stride, count, map_enable, rows_cap,
                           n_maps, chunk_tiles, chunks_per_map, out8, scale, reinterpret_cast<unsigned *>(eps), norm, out32, round_f16, aux);
        return;
    }
    auto kern = gather_q8_v3_kernel<CP, LPR, NHWC, FMT>;
    allow_dynamic_lds(reinterpret_cast<const void *>(kern), 4 * G8_STAGE_BYTES);
    hipLaunchKernelGGL(kern, dim3(groups), dim3(256), 4 * G8_STAGE_BYTES, st, feat, C, HW, roi, roi_stride, count, map_enable, rows_cap,
                       n_maps, chunk_tiles, chunks_per_map, out8, scale, reinterpret_cast<unsigned *>(eps), norm, out32, round_f16, aux);
}
}  // namespace

namespace oryon {
// internal entry (also used by the matcher's lazy fp32 fall-back): map_enable gates whole maps on the device; zero_eps = false when
// the call must not touch eps_max (fall-back pass)
int gather_q8_launch(const float *feat, int n_maps, int C, int HW, int layout, const int32_t *roi, int roi_stride, const int32_t *count,
                     const int32_t *map_enable, int rows_cap, int C_pad, int8_t *out8, float *scale, float *eps, float *norm,
                     float *out32, int lanes_per_row, int round_f16, hipStream_t st, int fmt, void *aux)
{
    const int lpr = C_pad == 512 ? 2 : (lanes_per_row == 2 ? 2 : 1);
    // K0v4 (round 5): the MX-fp6 passes over channel-planar maps of up to 256 channels - narrow maps (the reference's C = 32) included:
    // waves whose 64 channels lie beyond C load nothing and contribute zero blocks (smooth C = 32 @ 192 x 192 step 2.89 -> 2.77 ms);
    // channels_last, C_pad 512 and the device-gated fall-back passes stay on K0v3
    static const int v4 = dev_env_int("ORYON_K0V4", 1);
    if ((fmt == 1 || (fmt == 3 && aux && scale && !out32)) && v4 && C_pad == 256 && layout == ORYON_LAYOUT_NCHW && !map_enable &&
        (lanes_per_row != 2 || fmt == 3)) {
        const int T = (rows_cap + 63) / 64;
        const int chunks_per_map = n_maps >= 8 ? 1 : (8 + n_maps - 1) / n_maps;
        const int chunk_tiles = (T + chunks_per_map - 1) / chunks_per_map;
        const int units = n_maps * chunks_per_map;
        const int groups = ((units + 7) / 8) * 8 * chunk_tiles;
        if (fmt == 3) {
            auto k4 = gather_mx6_v4_kernel<false, true>;
            constexpr int X3_LDS = G4_LDS + 4 * G4_STAGE32 + 1024;
            hipLaunchKernelGGL(k4, dim3(groups), dim3(256), X3_LDS, st, feat, C, HW, roi, roi_stride, count, rows_cap, n_maps, chunk_tiles,
                               chunks_per_map, reinterpret_cast<uint8_t *>(out8), reinterpret_cast<unsigned *>(eps), norm, out32, round_f16, aux,
                               reinterpret_cast<unsigned *>(scale));
        } else if (out32) {
            auto k4 = gather_mx6_v4_kernel<true>;
            allow_dynamic_lds(reinterpret_cast<const void *>(k4), G4_LDS + 4 * G4_STAGE32);
            hipLaunchKernelGGL(k4, dim3(groups), dim3(256), G4_LDS + 4 * G4_STAGE32, st, feat, C, HW, roi, roi_stride, count, rows_cap, n_maps,
                               chunk_tiles, chunks_per_map, reinterpret_cast<uint8_t *>(out8), reinterpret_cast<unsigned *>(eps), norm, out32, round_f16,
                               nullptr, nullptr);
        } else {
            auto k4 = gather_mx6_v4_kernel<false>;
            hipLaunchKernelGGL(k4, dim3(groups), dim3(256), G4_LDS, st, feat, C, HW, roi, roi_stride, count, rows_cap, n_maps, chunk_tiles,
                               chunks_per_map, reinterpret_cast<uint8_t *>(out8), reinterpret_cast<unsigned *>(eps), norm, out32, round_f16,
                               nullptr, nullptr);
        }
        return hipGetLastError() == hipSuccess ? ORYON_OK : ORYON_ERR_HIP;
    }
#define G8(CPV, LPRV, FMTV)                                                                                                    \
    do {                                                                                                                       \
        if (layout == ORYON_LAYOUT_NHWC) launch_g8<CPV, LPRV, true, FMTV>(st, feat, n_maps, C, HW, roi, roi_stride, count, map_enable, rows_cap, out8, scale, eps, norm, out32, round_f16); \
        else launch_g8<CPV, LPRV, false, FMTV>(st, feat, n_maps, C, HW, roi, roi_stride, count, map_enable, rows_cap, out8, scale, eps, norm, out32, round_f16); \
    } while (0)
    if (fmt == 3) {
        // mx6 slots + hi / lo half rows in one pass (C_pad 256 only): out8 = mx6 rows, aux = hi rows | lo rows, scale = |u - hi|^2 maxima
        if (C_pad != 256 || !aux || !scale || out32) return ORYON_ERR_INVALID_ARG;
        if (layout == ORYON_LAYOUT_NHWC) launch_g8<256, 1, true, 3>(st, feat, n_maps, C, HW, roi, roi_stride, count, map_enable, rows_cap, out8, scale, eps, norm, out32, round_f16, aux);
        else launch_g8<256, 1, false, 3>(st, feat, n_maps, C, HW, roi, roi_stride, count, map_enable, rows_cap, out8, scale, eps, norm, out32, round_f16, aux);
    } else if (fmt == 2) {
        // hi / lo half rows for the fp16x3 scan (C_pad 256 only): out8 = hi rows, aux = lo rows
        if (C_pad != 256 || !aux) return ORYON_ERR_INVALID_ARG;
        if (layout == ORYON_LAYOUT_NHWC) launch_g8<256, 1, true, 2>(st, feat, n_maps, C, HW, roi, roi_stride, count, map_enable, rows_cap, out8, scale, eps, norm, out32, round_f16, aux);
        else launch_g8<256, 1, false, 2>(st, feat, n_maps, C, HW, roi, roi_stride, count, map_enable, rows_cap, out8, scale, eps, norm, out32, round_f16, aux);
    } else if (fmt == 1) {
        if (C_pad == 512) G8(512, 2, 1);
        else if (lpr == 2) G8(256, 2, 1);
        else G8(256, 1, 1);
    } else if (C_pad == 512) G8(512, 2, 0);
    else if (lpr == 2) G8(256, 2, 0);
    else G8(256, 1, 0);
#undef G8
    return hipGetLastError() == hipSuccess ? ORYON_OK : ORYON_ERR_HIP;
}
}  // namespace oryon

extern "C" int oryon_gather_q8(const float *feat, int n_maps, int C, int HW, int layout, const int32_t *roi, int roi_stride,
                               const int32_t *count, int rows_cap, int C_pad, int8_t *out_i8, float *slice_scale, float *eps_max,
                               float *row_norm, float *out_f32, int round_f16, void *stream)
{
    ORYON_CHECK_ARG(feat && roi && count && out_i8 && slice_scale && eps_max);                 // row_norm, out_f32 may be NULL
    ORYON_CHECK_ARG(n_maps >= 0 && C > 0 && HW > 0 && roi_stride > 0 && C_pad >= C && (C_pad == 256 || C_pad == 512));
    ORYON_CHECK_ARG(layout == ORYON_LAYOUT_NCHW || layout == ORYON_LAYOUT_NHWC);
    ORYON_CHECK_ARG(rows_cap > 0 && rows_cap % 256 == 0 && (size_t)C * (size_t)HW * 4u < (1ull << 32));
    if (n_maps == 0) return ORYON_OK;
    hipStream_t st = as_stream(stream);
    ORYON_CHECK_HIP(hipMemsetAsync(eps_max, 0, (size_t)n_maps * sizeof(float), st));
    static const int lpr_env = dev_env_int("ORYON_GATHER8_LPR", 1);
    const int rc = gather_q8_launch(feat, n_maps, C, HW, layout, roi, roi_stride, count, nullptr, rows_cap, C_pad, out_i8, slice_scale,
                                    eps_max, row_norm, out_f32, lpr_env, round_f16, st, 0, nullptr);
    if (rc) { set_error("oryon_gather_q8: launch failed"); return rc; }
    return ORYON_OK;
}

extern "C" int oryon_gather_mx6(const float *feat, int n_maps, int C, int HW, int layout, const int32_t *roi, int roi_stride,
                                const int32_t *count, int rows_cap, int C_pad, uint8_t *out_mx6, float *err_max, float *row_norm,
                                float *out_f32, int round_f16, void *stream)
{
    ORYON_CHECK_ARG(feat && roi && count && out_mx6 && err_max);                               // row_norm, out_f32 may be NULL
    ORYON_CHECK_ARG(n_maps >= 0 && C > 0 && HW > 0 && roi_stride > 0 && C_pad >= C && (C_pad == 256 || C_pad == 512));
    ORYON_CHECK_ARG(layout == ORYON_LAYOUT_NCHW || layout == ORYON_LAYOUT_NHWC);
    ORYON_CHECK_ARG(rows_cap > 0 && rows_cap % 256 == 0 && (size_t)C * (size_t)HW * 4u < (1ull << 32));
    if (n_maps == 0) return ORYON_OK;
    hipStream_t st = as_stream(stream);
    ORYON_CHECK_HIP(hipMemsetAsync(err_max, 0, (size_t)n_maps * sizeof(float), st));
    static const int lpr_env = dev_env_int("ORYON_GATHER8_LPR", 1);
    const int rc = gather_q8_launch(feat, n_maps, C, HW, layout, roi, roi_stride, count, nullptr, rows_cap, C_pad,
                                    reinterpret_cast<int8_t *>(out_mx6), nullptr, err_max, row_norm, out_f32, lpr_env, round_f16, st, 1, nullptr);
    if (rc) { set_error("oryon_gather_mx6: launch failed"); return rc; }
    return ORYON_OK;
}

extern "C" int oryon_gather_mx6_x3(const float *feat, int n_maps, int C, int HW, int layout, const int32_t *roi, int roi_stride,
                                   const int32_t *count, int rows_cap, int C_pad, uint8_t *out_mx6, float *err_max, float *row_norm,
                                   void *hi_lo_f16, float *lo_sq_max, int round_f16, void *stream)
{
    ORYON_CHECK_ARG(feat && roi && count && out_mx6 && err_max && hi_lo_f16 && lo_sq_max);                 // row_norm may be NULL
    ORYON_CHECK_ARG(n_maps >= 0 && C > 0 && HW > 0 && roi_stride > 0 && C_pad == 256 && C <= C_pad);
    ORYON_CHECK_ARG(layout == ORYON_LAYOUT_NCHW || layout == ORYON_LAYOUT_NHWC);
    ORYON_CHECK_ARG(rows_cap > 0 && rows_cap % 256 == 0 && (size_t)C * (size_t)HW * 4u < (1ull << 32));
    if (n_maps == 0) return ORYON_OK;
    hipStream_t st = as_stream(stream);
    ORYON_CHECK_HIP(hipMemsetAsync(err_max, 0, (size_t)n_maps * sizeof(float), st));
    ORYON_CHECK_HIP(hipMemsetAsync(lo_sq_max, 0, (size_t)n_maps * sizeof(float), st));
    const int rc = gather_q8_launch(feat, n_maps, C, HW, layout, roi, roi_stride, count, nullptr, rows_cap, C_pad,
                                    reinterpret_cast<int8_t *>(out_mx6), lo_sq_max, err_max, row_norm, nullptr, 1, round_f16, st, 3, hi_lo_f16);
    if (rc) { set_error("oryon_gather_mx6_x3: launch failed"); return rc; }
    return ORYON_OK;
}
