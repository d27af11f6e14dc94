// K1x3: fp32-GRADE scan of a few anchors per pair on the fp16 matrix pipe (round 3) - the second level behind the low-precision screens.
//
// Replaces, for the sampled anchors whose best and second-best match no 6- / 8-bit screen can separate (smooth descriptor fields: cosine gaps
// of 1e-4), the exact fp32-MFMA scan of utils/pcd.py:202-204's argmin (match_f32_regb_kernel on a compacted list: 5.9 ms per cfg2 step of
// such inputs, 157 TFLOP/s pipe) by an error-compensated fp16 contraction: every unit value u is split hi = half(u), lo = half(u - hi) (22
// significant bits; K0 FMT = 2 writes the query rows that way, match_x3_split_anchors_kernel the anchors) and
//     s3 = sum_k  al.qh + ah.ql + ah.qh      (three v_mfma_f32_32x32x16_f16 per 16 channels, fp32 accumulate)
// approximates the canonical dot product within DELTA3 = 6.5e-5 (C <= 256: split error <= 1.4e-6, dropped lo.lo term <= 2.4e-7, fp32
// accumulation of 768 terms here <= 4.6e-5 and of 256 terms in the canonical chain <= 1.5e-5, all worst case).  TWO sweeps over a workgroup's
// share of the query rows: the first multiplies the hi parts only (a third of the MFMAs) and keeps the maximum per anchor column - with
//     |s_hi - s| <= (|al|_2 + |ql|_2)(1 + 2^-10) + |al||ql| + 3.1e-5
// (|al| per anchor from match_x3_split_anchors_kernel, the largest |ql| of the pair from K0 FMT = 2: ~1.4e-4 each on unit rows of 256) that is a
// lower bound of the anchor's true maximum; the second sweep computes s3 and appends (index, score) to the anchor's candidate list whenever
// s3 >= bound - DELTA3, a FIXED limit, so every exact minimiser is listed and the lists hold what lies within ~4e-4 of the maximum (tens of
// rows on the smoothest fields probed).  match_x3_rescore_kernel then keeps the entries within MARGIN3 = 2 DELTA3 + slack of the best listed
// score and runs the canonical fp32 chain on x_k / d from the raw map for them (typically 1-5 rows): distance, first index of the minimum,
// validity - bit for bit what the exact scan returns.  A list that overflows (crowds of near-duplicates) sends its anchor to the exact scan
// (device-side list; the caller materialises fp32 rows for such pairs only).
//
// History: the first version kept a RUNNING maximum in one sweep and listed everything within MARGIN3 of it - scanning row-major towards a
// smooth peak nearly every row is a new running maximum, the lists overflowed and the exact fall-back ran anyway (hard step 14.0 ms against
// 10.9 without K1x3); its first list writer used a returning atomicAdd per entry (9 ms per launch; now one writer per list, no atomics).
#include <hip/hip_fp16.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include <type_traits>
#include "common.h"
#include "match_common.h"

namespace oryon {

typedef _Float16 half8x __attribute__((ext_vector_type(8)));

constexpr int X3_CAPH = 128;               // candidate slots per (pair, query split, anchor, lane half): each list has ONE writer, no atomics
constexpr float X3_MARGIN = 1.32e-4f;      // 2 * DELTA3 (6.5e-5) + 2e-6
constexpr float X3_MARGIN_R = 3.4e-5f;     // 2 * DELTA_R (1.63e-5: refined fp64 score against the canonical fp32 chain) + slack
constexpr int X3_MAX_TILES = 256;          // tile flags per wave of the scan (a workgroup's share of the query rows: cap_q / 32 / S tiles)
// byte offset of (query row q, byte b of its 512-byte half row) inside a map's hi (or lo) array: tiles of 32 rows, chunk-major inside a tile
// (gather8.hip, FMT = 2 / 3)
__host__ __device__ __forceinline__ size_t x3_q_off(int q, int b) { return (size_t)(q >> 5) * 16384 + (size_t)(b >> 7) * 4096 + (size_t)(q & 31) * 128 + (size_t)(b & 127); }
constexpr int X3_JOB_TILES = 16;           // tiles per sweep-2 job (one reload of the 64 anchors' operands per job: 64 KB against 512 KB of tiles)
constexpr int X3_SURV = 256;               // survivors of the first filter kept per anchor (more: exact-scan route)

// fp32 anchor rows (k-permuted inside groups of 8: position 8g + 4h + j holds k = 8g + 2j + h) -> hi / lo half rows in natural k order
// + al_norm [B, cap_s]: an upper bound of |lo|_2 of every row (Cp = 256: the 32 groups of a row are 32 consecutive lanes)
__global__ __launch_bounds__(256) void match_x3_split_anchors_kernel(const float *__restrict__ a_c, int Cp, int cap_s,
                                                                      const int32_t *__restrict__ n_c, __half *__restrict__ ah,
                                                                      __half *__restrict__ al, float *__restrict__ al_norm)
{
    const int p = blockIdx.y;
    const int n = n_c[p] < cap_s ? n_c[p] : cap_s;
    const int n_fill = (n + 255) / 256 * 256 < cap_s ? (n + 255) / 256 * 256 : cap_s;      // the scan reads whole 256-anchor panels: zero-fill
    const int groups = n_fill * (Cp / 8);
    const int groups_up = (groups + 255) / 256 * 256;          // whole workgroups stay in the loop: the row reduction below uses shuffles
    for (int g = blockIdx.x * 256 + threadIdx.x; g < groups_up; g += gridDim.x * 256) {
        const int row = g / (Cp / 8), gi = g % (Cp / 8);
        union { __half h[8]; uint4 u; } hi, lo;
        float l2 = 0.0f;
        if (g >= groups) {
            hi.u = make_uint4(0, 0, 0, 0);
            lo.u = make_uint4(0, 0, 0, 0);
        } else if (row < n) {
            const float4 *src = reinterpret_cast<const float4 *>(a_c + ((size_t)p * cap_s + row) * Cp) + 2 * gi;
            const float4 x = src[0], y = src[1];                // x = k 8g+{0,2,4,6}, y = k 8g+{1,3,5,7}
            const float v[8] = {x.x, y.x, x.y, y.y, x.z, y.z, x.w, y.w};
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                hi.h[i] = __float2half_rn(v[i]);
                lo.h[i] = __float2half_rn(v[i] - __half2float(hi.h[i]));
                l2 = fmaf(__half2float(lo.h[i]), __half2float(lo.h[i]), l2);
            }
        } else {
            hi.u = make_uint4(0, 0, 0, 0);
            lo.u = make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) l2 += __shfl_xor(l2, off);          // Cp / 8 = 32 lanes per row
        if (g < groups) {
            reinterpret_cast<uint4 *>(ah + ((size_t)p * cap_s + row) * Cp)[gi] = hi.u;
            reinterpret_cast<uint4 *>(al + ((size_t)p * cap_s + row) * Cp)[gi] = lo.u;
            if (gi == 0) al_norm[(size_t)p * cap_s + row] = sqrtf(l2) * 1.0001f + 1e-12f;
        }
    }
}

// Seed of the scan's running maxima (round 4): s_hi of the 16 query rows of the anchor's WINNING SLICE of the low-precision screen
// (sid_final: slice (blk, half) = rows blk * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, r = 0..15).  On smooth fields the screen cannot tell
// which row of the neighbourhood wins, but its winning slice sits at the peak: the best hi.hi score there is within ~1e-3 of the anchor's
// true maximum.  It is the score of a real row, so it is a valid start for the running maximum of EVERY query split of the anchor:
// sweep 1's tile flags (tested against the running maximum) then single out the few tiles around the peak from the first tile on instead
// of flagging every tile on the way up to it (hard cfg2 step: 46 % of the (wave, tile) pairs flagged without the seed), and the splits
// that do not hold the peak list almost nothing.  One wave per compacted anchor; 16 lanes x 4 segments of 64 channels.
__global__ __launch_bounds__(256) void match_x3_seed_kernel(const __half *__restrict__ ah, const __half *__restrict__ qh, int Cp, int cap_s,
                                                             int cap_q, const int32_t *__restrict__ n_c, const int32_t *__restrict__ n_q,
                                                             const int32_t *__restrict__ orig_idx, int orig_stride,
                                                             const int32_t *__restrict__ sid_final, int cap_a, int kc, float *__restrict__ seed)
{
    const int p = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + wave;
    const int nc = n_c[p] < cap_s ? n_c[p] : cap_s;
    if (row >= nc) return;
    const int nq = n_q[p];
    const int sid = sid_final[(size_t)p * cap_a + orig_idx[(size_t)p * orig_stride + row]];
    const int half = sid & 1, blk = sid >> 1;
    const int r = lane >> 2, seg = lane & 3;
    const int q = blk * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
    float s = 0.0f;
    if (q < nq && seg < kc) {                                   // (chunks beyond the map's channels are never written: zero by definition)
        const uint4 *ar = reinterpret_cast<const uint4 *>(ah + ((size_t)p * cap_s + row) * Cp) + seg * (Cp / 32);
        const uint4 *qr = reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(qh + (size_t)p * cap_q * Cp) + x3_q_off(q, seg * 128));
        for (int i = 0; i < Cp / 32; ++i) {
            const uint4 av = ar[i], qv = qr[i];
            const __half2 *a2 = reinterpret_cast<const __half2 *>(&av), *q2 = reinterpret_cast<const __half2 *>(&qv);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s = fmaf(__low2float(a2[e]), __low2float(q2[e]), s);
                s = fmaf(__high2float(a2[e]), __high2float(q2[e]), s);
            }
        }
    }
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    float mx = (q < nq) ? s : -INFINITY;
#pragma unroll
    for (int off = 4; off < 64; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    if (lane == 0) seed[(size_t)p * cap_s + row] = mx;
}

// grid: units (pair, query split) dealt to the 8 XCDs, T = cap_s / 256 anchor panels per unit; 4 waves x 64 anchors (two B-operand sets of hi +
// lo rows = 256 registers: one workgroup per CU with the 512-register budget - with 32 anchors per wave every pair of ds_read_b128 fed only
// 3 MFMAs and the LDS, not the matrix pipe, set the pace: 5.1 ms); tiles of 32 query rows (hi part 16 KB + lo part 16 KB, double-buffered)
template <int CP, int WAVES, bool NARROW>
__device__ __forceinline__ void match_x3_scan_item(const int xcd, const int slot, char *__restrict__ smem, const __half *__restrict__ ah,
                                                   const __half *__restrict__ al, const __half *__restrict__ qh, const __half *__restrict__ ql, int B,
                                                   int cap_s, int cap_q, const int32_t *__restrict__ n_c, const int32_t *__restrict__ n_q, int T, int S,
                                                   const float *__restrict__ al_norm, const float *__restrict__ ql_max,
                                                   const float *__restrict__ seed, float *__restrict__ smax, unsigned short *__restrict__ tl,
                                                   uint2 *__restrict__ jobs, int32_t *__restrict__ njobs, int kc, int32_t *__restrict__ dbg,
                                                   long long *__restrict__ dbg_wg)
{
    constexpr int RB = CP * 2;                 // bytes per half row
    constexpr int ROWS = 32;
    constexpr int PART = ROWS * RB;            // 16 KB at CP = 256
    constexpr int NKS = CP / 16;
    constexpr int NI = PART / (1024 * WAVES);  // 1 KB DMA instructions per wave and tile
    constexpr int LPR = RB / 256;
    constexpr int PANEL = 64 * WAVES;          // anchors of a workgroup

    const int unit = (slot / T) * 8 + xcd;
    if (unit >= B * S) return;
    const int panel = slot % T;
    const int p = unit / S, split = unit % S;
    const int nc = n_c[p] < cap_s ? n_c[p] : cap_s, nq = n_q[p];
    constexpr int NAB = 2;
    // (measured, not kept: dealing the 32-anchor blocks round-robin over panels / waves so that the few blocks whose matches lie in this
    // split's band do not share a wave - each wave then covers two bands and flags 40 % more tiles: scan 1.69 instead of 1.60 ms)
    const int a0 = panel * PANEL;
    if (a0 >= nc) return;
#define X3_ANCHOR(ab_) (a0 + wave * 64 + (ab_) * 32 + l31)
    const int nqt = (nq + ROWS - 1) / ROWS;
    const int qt_per = (nqt + S - 1) / S;
    const int qt_begin = split * qt_per;
    const int qt_end = (qt_begin + qt_per < nqt) ? qt_begin + qt_per : nqt;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, hi = lane >> 5;
    long long tk0 = dbg ? wall_clock64() : 0;
    half8x bh[NAB][NKS];                                    // sweep 1 multiplies hi parts only
#pragma unroll
    for (int ab = 0; ab < NAB; ++ab) {
        const int ar = X3_ANCHOR(ab);
        const int arc = ar < cap_s ? ar : cap_s - 1;
        const __half *rh = ah + ((size_t)p * cap_s + arc) * CP + 8 * hi;
#pragma unroll
        for (int s = 0; s < NKS; ++s) bh[ab][s] = *reinterpret_cast<const half8x *>(rh + 16 * s);
    }
    unsigned dma_off[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int line = (wave * NI + j) * 4 + (lane >> 4), sl = lane & 15;
        const int row = line / LPR;
        const int cc = sl ^ (row & 15);
        dma_off[j] = (unsigned)x3_q_off(row, ((line % LPR) * 16 + cc) * 16);        // source of the 16 bytes that land at LDS (row, slot (line % LPR) * 16 + sl)
    }
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const char *qhp = reinterpret_cast<const char *>(qh + (size_t)p * cap_q * CP);
    unsigned koff[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) koff[c] = (unsigned)(l31 * RB) + ((((unsigned)(hi ^ (l31 & 15))) ^ (2u * c)) << 4);
    auto rd = [&](int part, int s, unsigned tile) -> half8x {
        return *reinterpret_cast<const half8x *>(smem + koff[s & 7] + tile + (unsigned)(part * PART + (s >> 3) * 256));
    };

    // Round 4 (measured with ORYON_X3_DEBUG's phase clocks: sweep 1 took 1.0 us per tile against 0.5 us of MFMAs, sweep 2 2.3 us per visited
    // tile with 1.2 of the 4 waves multiplying on average - both were waiting, not computing):
    //  * sweep 1 moves hi parts only, so the 128 KB ring holds EIGHT 16 KB tiles: requests run 7 tiles ahead instead of 3;
    //  * sweep 2 is wave-private.  A wave's anchors cover a band of the image, so the tiles that can hold its candidates are mostly not
    //    the ones its neighbours need: with workgroup-wide tiles and a barrier per tile every wave waited for whichever wave had work.
    //    Now each wave walks ITS OWN flagged tiles through ITS OWN quarter of the LDS (32 KB = four 8 KB chunks of 64 channels, hi + lo
    //    parts of the 32 rows; chunk c of every tile lives in slot c, requested three chunks ahead), with no barrier at all.
#define X3_WAIT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
    constexpr int NS1 = 8;                                       // sweep 1: ring slots of PART bytes
    auto issue1 = [&](int qt, int slot1) {
        const char *qb = qhp + (size_t)qt * PART;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            char *dst = smem + slot1 * PART + (wave_u * NI + j) * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(qb + dma_off[j]),
                                             (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        }
    };
    // |s_hi - s| <= e_hi = (|al| + |ql|)(1 + 2^-10) + |al||ql| + 3.1e-5 (Cauchy-Schwarz on the measured norms; fp32 accumulation of 256
    // products here and in the canonical chain)
    float e_hi[NAB];
    const float qlm = sqrtf(ql_max[p]) * 1.002f;            // K0 FMT = 2 hands over the largest |u - hi|^2 of the pair's query rows
#pragma unroll
    for (int ab = 0; ab < NAB; ++ab) {
        const int a = X3_ANCHOR(ab);
        const float aln = al_norm[(size_t)p * cap_s + (a < cap_s ? a : cap_s - 1)];
        e_hi[ab] = (aln + qlm) * 1.001f + aln * qlm + 3.1e-5f;
    }
    // Tile flags: sweep 2 lists a row only if s3 >= max - e_hi - DELTA3, and s3 <= s_hi + e_hi + DELTA3, so a tile none of whose rows
    // has s_hi >= max_hi - 2 e_hi - 2 DELTA3 for any of the wave's anchors holds no candidate for this wave.  Sweep 1 flags the tiles that
    // pass against the RUNNING maximum (a superset of those that pass against the final one) - started from the seed, i.e. close to the
    // final maximum from the first tile on; sweep 2 multiplies only flagged tiles.  The anchors arrive in image order
    // (match_list_sampled_amb_kernel), so a wave's 64 anchors - and their matches - cover a band of the image (hard cfg2 step: 19 % of
    // the (wave, tile) pairs flagged with the seed, 46 % without).
    constexpr int FLAG_BASE = NS1 * PART;                       // behind the ring: 4 x X3_MAX_TILES flag bytes, 4 x X3_MAX_TILES list entries
    unsigned char *tile_flag = reinterpret_cast<unsigned char *>(smem + FLAG_BASE) + wave * X3_MAX_TILES;
    unsigned short *my_list = reinterpret_cast<unsigned short *>(smem + FLAG_BASE + WAVES * X3_MAX_TILES) + wave * X3_MAX_TILES;
    for (int i = t; i < WAVES * X3_MAX_TILES; i += 64 * WAVES) reinterpret_cast<unsigned char *>(smem + FLAG_BASE)[i] = 0;
    const int ntl = qt_end - qt_begin;
    const bool flags_ok = ntl <= X3_MAX_TILES;
    long long tk1 = dbg ? wall_clock64() : 0;
    // ---- sweep 1: hi.hi only (a third of the MFMAs, half of the tile bytes) -> a lower bound of every anchor's maximum over this split's rows
    float runmax[NAB];
#pragma unroll
    for (int ab = 0; ab < NAB; ++ab) {
        // start from the seed (match_x3_seed_kernel: the hi.hi score of a real row of this anchor - whichever split holds it)
        const int a = X3_ANCHOR(ab);
        runmax[ab] = (seed && a < nc) ? seed[(size_t)p * cap_s + a] : -INFINITY;
    }
    // The A fragments of a tile are read into registers one tile AHEAD (two sets of NKS fragments): as first written, every k-step was a
    // ds_read followed at once by the two MFMAs that need it - with one wave per SIMD the ~140-cycle LDS latency of each of the 16 reads
    // lay bare between 64-cycle MFMA pairs (measured 1.1 us per tile against 0.5 us of MFMAs).  Now tile it + 1's sixteen reads are issued
    // as a block and return under tile it's 32 MFMAs.
    auto wait_tiles = [&](int n) {           // at most n tiles of this wave's DMA still in flight
        switch (n) {
            case 0: X3_WAIT(0); break;
            case 1: X3_WAIT(NI); break;
            case 2: X3_WAIT(2 * NI); break;
            case 3: X3_WAIT(3 * NI); break;
            case 4: X3_WAIT(4 * NI); break;
            case 5: X3_WAIT(5 * NI); break;
            default: X3_WAIT(6 * NI); break;
        }
    };
    auto load_frag = [&](half8x (&f)[NKS], int it) {
        const unsigned tile = (unsigned)((it % NS1) * PART);
#pragma unroll
        for (int s = 0; s < NKS; ++s) f[s] = rd(0, s, tile);
    };
    auto step1 = [&](const half8x (&cur)[NKS], half8x (&nxt)[NKS], int it) {
        if (it + 1 < ntl) {
            const int later = ntl - 2 - it;                  // tiles behind tile it + 1
            wait_tiles(later < NS1 - 3 ? later : NS1 - 3);   // tile it + 1 (this wave's share) has landed
        }
        __syncthreads();                     // every wave's share of tile it + 1 is visible; every wave has consumed tile it - 1
        const int ahead = it + NS1 - 1;
        if (ahead < ntl) issue1(qt_begin + ahead, ahead % NS1);          // into the slot tile it - 1 has left
        if (it + 1 < ntl) load_frag(nxt, it + 1);
        __builtin_amdgcn_sched_barrier(0);   // the reads above are ISSUED before the MFMAs below (the compiler would sink them to their first use)
        const int qt = qt_begin + it;
        f32x16 acc[NAB];
#pragma unroll
        for (int ab = 0; ab < NAB; ++ab)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ab][r] = 0.0f;
#pragma unroll
        for (int s = 0; s < NKS; ++s)
            if (!NARROW || s < 4 * kc) {                        // NARROW: k-steps of never-written chunks are skipped (zero by definition)
#pragma unroll
                for (int ab = 0; ab < NAB; ++ab) acc[ab] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[s], bh[ab][s], acc[ab], 0, 0, 0);
            }
        const int q0 = qt * ROWS + 4 * hi;
        int fl = 0;
#pragma unroll
        for (int ab = 0; ab < NAB; ++ab) {
            float x = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) x = fmaxf(x, (q0 + (r & 3) + 8 * (r >> 2) < nq) ? acc[ab][r] : -INFINITY);
            runmax[ab] = fmaxf(runmax[ab], x);
            // dead anchor columns (beyond the pair's count: zero rows, every score 0 = their own running maximum) must not flag anything -
            // until round 4 the wave that holds the list's tail (500 sampled anchors in 512 slots) flagged every tile through them
            const bool h = (X3_ANCHOR(ab) < nc) && x >= runmax[ab] - 2.0f * e_hi[ab] - X3_MARGIN;
            fl |= (__ballot(h) != 0ull) ? (1 << ab) : 0;
        }
        if (fl && flags_ok && lane == 0) tile_flag[it] = (unsigned char)fl;          // bit ab: anchor block ab of this wave needs the tile
    };
    __syncthreads();
#pragma unroll
    for (int d = 0; d < NS1 - 1; ++d)
        if (d < ntl) issue1(qt_begin + d, d);
    if constexpr (WAVES == 4) {
        half8x fa[NKS], fb[NKS];
        if (ntl > 0) {
            wait_tiles(ntl - 1 < NS1 - 2 ? ntl - 1 : NS1 - 2);
            __syncthreads();
            load_frag(fa, 0);
        }
        for (int it = 0; it < ntl; it += 2) {
            step1(fa, fb, it);
            if (it + 1 < ntl) step1(fb, fa, it + 1);
        }
    } else {
        // eight waves (two per SIMD, <= 256 registers each): the other wave of the SIMD covers a wave's LDS latency, so the fragments are
        // read a k-step ahead only and a tile is consumed in the iteration it is waited for
        for (int it = 0; it < ntl; ++it) {
            const int rem = ntl - 1 - it;
            wait_tiles(rem < NS1 - 2 ? rem : NS1 - 2);
            __syncthreads();                 // every wave's share of tile `it` is visible; every wave is done with tile it - 1
            const int ahead = it + NS1 - 1;
            if (ahead < ntl) issue1(qt_begin + ahead, ahead % NS1);
            const unsigned tile = (unsigned)((it % NS1) * PART);
            f32x16 acc[NAB];
#pragma unroll
            for (int ab = 0; ab < NAB; ++ab)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ab][r] = 0.0f;
            half8x xh = rd(0, 0, tile);
#pragma unroll
            for (int s = 0; s < NKS; ++s) {
                half8x nh = xh;
                if (s + 1 < NKS) nh = rd(0, s + 1, tile);
                if (!NARROW || s < 4 * kc) {                   // NARROW: channels beyond the live chunks are zero on both sides (wave-uniform)
#pragma unroll
                    for (int ab = 0; ab < NAB; ++ab) acc[ab] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, bh[ab][s], acc[ab], 0, 0, 0);
                }
                xh = nh;
            }
            const int q0 = (qt_begin + it) * ROWS + 4 * hi;
            int fl = 0;
#pragma unroll
            for (int ab = 0; ab < NAB; ++ab) {
                float x = -INFINITY;
#pragma unroll
                for (int r = 0; r < 16; ++r) x = fmaxf(x, (q0 + (r & 3) + 8 * (r >> 2) < nq) ? acc[ab][r] : -INFINITY);
                runmax[ab] = fmaxf(runmax[ab], x);
                const bool h = (X3_ANCHOR(ab) < nc) && x >= runmax[ab] - 2.0f * e_hi[ab] - X3_MARGIN;
                fl |= (__ballot(h) != 0ull) ? (1 << ab) : 0;
            }
            if (fl && flags_ok && lane == 0) tile_flag[it] = (unsigned char)fl;
        }
    }
    __syncthreads();                         // the ring is free: sweep 2's private regions overlay it
    long long tk2 = dbg ? wall_clock64() : 0;
    if (dbg && lane == 0) {                                  // ORYON_X3_DEBUG: flagged / all (wave, tile) pairs
        int f = 0;
        for (int i = 0; i < ntl && i < X3_MAX_TILES; ++i) f += tile_flag[i] != 0;
        atomicAdd(&dbg[0], f);
        atomicAdd(&dbg[1], ntl);
    }
    // this wave's flagged tiles, in order (every tile when the split is too long for the flag array)
    int n_t2 = 0;
    if (flags_ok) {
        for (int b0 = 0; b0 < ntl; b0 += 64) {
            const int i = b0 + lane;
            const bool f = i < ntl && tile_flag[i];
            const unsigned long long m = __ballot(f);
            if (f) my_list[n_t2 + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)i;
            n_t2 += __popcll(m);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        n_t2 = ntl;
    }
    // ---- hand-over to sweep 2 (match_x3_sweep2_kernel): this split's maxima, the wave's tile list, jobs of <= X3_JOB_TILES tiles each.
    // One wave of this kernel often has a whole band of tiles to multiply while its neighbours have none (its anchors' matches lie in this
    // split's rows, theirs do not), and a lone wave streaming tiles is latency-bound (3.6 us per tile measured): as a sweep inside this
    // kernel the slowest wave set the pace of its workgroup (up to 800 us against 176 us of sweep 1).  As jobs of a second kernel the same
    // work spreads over every SIMD of the chip.
#pragma unroll
    for (int ab = 0; ab < NAB; ++ab) {
        runmax[ab] = fmaxf(runmax[ab], __shfl_xor(runmax[ab], 32));
        const int a = X3_ANCHOR(ab);
        if (hi == 0 && a < nc) smax[((size_t)p * S + split) * cap_s + a] = runmax[ab];
    }
    {
        const int owner = ((p * S + split) * T + panel) * WAVES + wave;      // = unit * (groups per unit) + 64-anchor group
        if (flags_ok) {
            unsigned short *gl = tl + (size_t)owner * X3_MAX_TILES;
            for (int i = lane; i < n_t2; i += 64) {
                const int ti = my_list[i];
                gl[i] = (unsigned short)(ti | ((int)tile_flag[ti] << 14));      // bits 14-15: which anchor block needs the tile
            }
        }
        const int nj = (n_t2 + X3_JOB_TILES - 1) / X3_JOB_TILES;
        if (nj > 0) {
            int base = 0;
            if (lane == 0) base = atomicAdd(njobs, nj);
            base = __shfl(base, 0);
            for (int j = lane; j < nj; j += 64) {
                const int first = j * X3_JOB_TILES;
                const int c = n_t2 - first < X3_JOB_TILES ? n_t2 - first : X3_JOB_TILES;
                jobs[base + j] = make_uint2((unsigned)owner, (flags_ok ? 0u : 0x80000000u) | ((unsigned)first << 8) | (unsigned)c);
            }
        }
    }
#undef X3_WAIT
    if (dbg && t == 0) {                                     // phase times (100 MHz ticks) summed over workgroups, sweep-2 tiles visited
        const long long tk3 = wall_clock64();
        atomicAdd(&dbg[2], (int)(tk1 - tk0));
        atomicAdd(&dbg[3], (int)(tk2 - tk1));
        atomicAdd(&dbg[4], (int)(tk3 - tk2));
        atomicAdd(&dbg[5], n_t2);
        atomicAdd(&dbg[6], 1);
        atomicMin(reinterpret_cast<unsigned long long *>(dbg + 8), (unsigned long long)tk0);
        atomicMax(reinterpret_cast<unsigned long long *>(dbg + 10), (unsigned long long)tk3);
        // slowest workgroup
        atomicMax(&dbg[7], (int)(tk3 - tk0));
        if (dbg_wg) {
            dbg_wg[(size_t)(slot * 8 + xcd) * 4 + 0] = tk0;
            dbg_wg[(size_t)(slot * 8 + xcd) * 4 + 1] = tk3;
            dbg_wg[(size_t)(slot * 8 + xcd) * 4 + 2] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_REG_HW_ID
            dbg_wg[(size_t)(slot * 8 + xcd) * 4 + 3] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 20);       // HW_REG_XCC_ID
        }
    }
}

#undef X3_ANCHOR
// Persistent launch (round 4).  Launched one block per work item (1024 blocks of ~220 us, one per CU at a time: 512 registers per lane,
// 130 KB of LDS) the scan took 2.4 ms although its workgroups' own times summed to 0.9 ms per CU: ORYON_X3_DEBUG's per-workgroup clocks
// showed the dispatcher leaving most CUs empty after the first round (256 running, then 30-180).  So the grid is one workgroup per CU and
// the workgroups pull items themselves: one queue per XCD (an item's two anchor panels and its query rows stay in that XCD's L2), in the
// order the one-block-per-item grid used; a workgroup whose XCD has run dry takes items of the others.
template <int CP, int WAVES, bool NARROW>
__global__ __launch_bounds__(64 * WAVES, 1) void match_x3_scan_kernel(const __half *__restrict__ ah, const __half *__restrict__ al,
                                                               const __half *__restrict__ qh, const __half *__restrict__ ql, int B, int cap_s,
                                                               int cap_q, const int32_t *__restrict__ n_c, const int32_t *__restrict__ n_q, int T,
                                                               int S, const float *__restrict__ al_norm, const float *__restrict__ ql_max,
                                                               const float *__restrict__ seed, float *__restrict__ smax,
                                                               unsigned short *__restrict__ tl, uint2 *__restrict__ jobs,
                                                               int32_t *__restrict__ njobs, int32_t *__restrict__ queue /*[8], zeroed*/,
                                                               int items_per_xcd, int kc, int32_t *__restrict__ dbg, long long *__restrict__ dbg_wg)
{
    extern __shared__ __attribute__((aligned(256))) char smem[];
    __shared__ int item_s;
    {
        // nothing listed in any pair (the usual step): one parallel look at the counts instead of a walk through the queues
        bool any = false;
        for (int m = threadIdx.x; m < B; m += 64 * WAVES) any |= n_c[m] > 0;
        if (!__syncthreads_or(any)) return;
    }
    const int my_xcd = blockIdx.x & 7;
    for (;;) {
        __syncthreads();                                  // everyone is done with the previous item (its LDS, item_s)
        if (threadIdx.x == 0) {
            int it = -1;
            for (int k = 0; k < 8 && it < 0; ++k) {       // own XCD's queue first, then the others'
                const int x = (my_xcd + k) & 7;
                if (__atomic_load_n(&queue[x], __ATOMIC_RELAXED) < items_per_xcd) {
                    const int got = atomicAdd(&queue[x], 1);
                    if (got < items_per_xcd) it = got * 8 + x;
                }
            }
            item_s = it;
        }
        __syncthreads();
        const int it = item_s;
        if (it < 0) return;
        match_x3_scan_item<CP, WAVES, NARROW>(it & 7, it >> 3, smem, ah, al, qh, ql, B, cap_s, cap_q, n_c, n_q, T, S, al_norm, ql_max, seed, smax, tl, jobs, njobs, kc, dbg, dbg_wg);
    }
}

// Sweep 2 (round 4: its own kernel).  Independent WAVES, four per CU (512 registers: the 64 anchors' hi + lo operands; 32 KB of LDS each:
// four 8 KB chunks of 64 channels, hi + lo parts of a tile's 32 rows, requested three chunks ahead), pulling jobs - (64-anchor group of a
// split, <= 16 of its flagged tiles) - from the queue sweep 1 filled: hi / lo compensated products, rows scoring within the limit of the
// anchor's maximum appended to its candidate list.  The limit comes from the maximum over ALL splits (sweep 1 has finished), so splits
// away from the peak list nothing; several jobs can append to one list, so a lane reserves its entries with one atomic per tile.
template <int CP, bool NARROW>
__global__ __launch_bounds__(256, 1) void match_x3_sweep2_kernel(const __half *__restrict__ ah, const __half *__restrict__ al,
                                                                const __half *__restrict__ qh, const __half *__restrict__ ql, int cap_s, int cap_q,
                                                                const int32_t *__restrict__ n_c, const int32_t *__restrict__ n_q, int G, int S,
                                                                const float *__restrict__ al_norm, const float *__restrict__ ql_max,
                                                                const float *__restrict__ smax, const unsigned short *__restrict__ tl,
                                                                const uint2 *__restrict__ jobs, const int32_t *__restrict__ njobs,
                                                                int32_t *__restrict__ next_job, int32_t *__restrict__ cnt, uint2 *__restrict__ cand,
                                                                int kc /* live 64-channel chunks: ceil(C_true / 64); the rest of a row is zero */,
                                                                int32_t *__restrict__ dbg)
{
    constexpr int RB = CP * 2, ROWS = 32, PART = ROWS * RB, NKS = CP / 16, NAB = 2;
    const long long tw0 = dbg ? wall_clock64() : 0;
    long long t_load = 0;
    int n_tiles_done = 0, n_jobs_done = 0;
    constexpr int CH = 64;                                  // channels per chunk
    constexpr int NCH = CP / CH;                            // 4 chunks per tile = the 4 slots of the region
    constexpr int CHB = 2 * ROWS * CH * 2;                  // 8 KB: [part][row][128 B]
    constexpr int NJ = ROWS * CH * 2 / 1024;                // 4 DMA instructions per part and chunk (8 rows x 128 B each)
    constexpr int PERC = 2 * NJ;                            // DMA instructions per chunk
    // four independent waves per workgroup, each with its own 32 KB region and its own job stream (no barrier anywhere): one workgroup per
    // CU instead of four one-wave workgroups - a quarter of the blocks to dispatch when nothing was posted (the usual step)
    extern __shared__ __attribute__((aligned(256))) char smem2[];
    char *reg = smem2 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * (NCH * CHB);
    const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
    if constexpr (!NARROW) kc = CP / 64;                        // every chunk is live: the compiler sees a constant
    const int n_jobs = __atomic_load_n(njobs, __ATOMIC_RELAXED);
    // DMA source: lane L of instruction j lands at LDS (row j*8 + L/8, 16-byte slot L%8); slots are XOR-swizzled with (row >> 1) & 7 so
    // that the 16 lanes of a ds_read_b128 phase (rows r .. r+15 at one logical slot) cover all 64 banks
    unsigned src2[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int row = j * 8 + (lane >> 3), sl = lane & 7;
        src2[j] = (unsigned)(row * 128 + ((sl ^ ((row >> 1) & 7)) << 4));          // inside the chunk's contiguous 4 KB (32 rows x 128 bytes)
    }
    unsigned ko2[CH / 16];
#pragma unroll
    for (int s4 = 0; s4 < CH / 16; ++s4) ko2[s4] = (unsigned)(l31 * 128 + (((2 * s4 + hi) ^ ((l31 >> 1) & 7)) << 4));
#define X3_WAIT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
    for (;;) {
        int jx = 0;
        if (lane == 0) jx = atomicAdd(next_job, 1);
        jx = __builtin_amdgcn_readfirstlane(jx);
        if (jx >= n_jobs) {
            if (dbg && lane == 0) {                          // ORYON_X3_DEBUG: per-wave totals
                const long long tw1 = wall_clock64();
                atomicAdd(&dbg[12], (int)(tw1 - tw0));       // busy ticks of this wave (start to its last job's end)
                atomicAdd(&dbg[13], (int)t_load);            // of which: operand loads
                atomicAdd(&dbg[14], n_tiles_done);
                atomicAdd(&dbg[15], n_jobs_done);
                atomicMax(&dbg[16], (int)(tw1 - tw0));
                atomicAdd(&dbg[17], 1);
            }
            return;
        }
        const long long tj0 = dbg ? wall_clock64() : 0;
        const uint2 jb = jobs[jx];
        const int owner = (int)jb.x, first = (int)((jb.y >> 8) & 0x7fffffu), n_t = (int)(jb.y & 0xffu);
        const bool dense = (jb.y >> 31) != 0;
        const int a_base = (owner % G) * 64, ps = owner / G, split = ps % S, p = ps / S;      // G = 64-anchor groups per (pair, split)
        const int nc = n_c[p] < cap_s ? n_c[p] : cap_s, nq = n_q[p];
        const int nqt = (nq + ROWS - 1) / ROWS, qt_per = (nqt + S - 1) / S, qt_begin = split * qt_per;
        const unsigned short *gl = tl + (size_t)owner * X3_MAX_TILES + first;
        const char *qhp = reinterpret_cast<const char *>(qh + (size_t)p * cap_q * CP), *qlp = reinterpret_cast<const char *>(ql + (size_t)p * cap_q * CP);
        half8x bh[NAB][NKS], bl[NAB][NKS];
        float lim[NAB], run3[NAB];
        const float qlm = sqrtf(ql_max[p]) * 1.002f;
#pragma unroll
        for (int ab = 0; ab < NAB; ++ab) {
            const int ar = a_base + ab * 32 + l31;
            const int arc = ar < cap_s ? ar : cap_s - 1;
            const __half *rh = ah + ((size_t)p * cap_s + arc) * CP + 8 * hi, *rl = al + ((size_t)p * cap_s + arc) * CP + 8 * hi;
#pragma unroll
            for (int s = 0; s < NKS; ++s) {
                bh[ab][s] = *reinterpret_cast<const half8x *>(rh + 16 * s);
                bl[ab][s] = *reinterpret_cast<const half8x *>(rl + 16 * s);
            }
            // max_j s_j >= (largest sweep-1 maximum of any split) - e_hi, and every exact maximiser has s3 >= max_j s_j - DELTA3: a FIXED
            // emission limit per anchor column
            const float aln = al_norm[(size_t)p * cap_s + arc];
            const float e_hi = (aln + qlm) * 1.001f + aln * qlm + 3.1e-5f;
            float gm = -INFINITY;
            if (ar < nc)
                for (int s2 = 0; s2 < S; ++s2) gm = fmaxf(gm, smax[((size_t)p * S + s2) * cap_s + ar]);
            lim[ab] = gm - e_hi - 0.5f * X3_MARGIN;
            run3[ab] = -INFINITY;
        }
        if (dbg) { t_load += wall_clock64() - tj0; n_tiles_done += n_t; n_jobs_done += 1; }
        auto tile_at = [&](int i) { return qt_begin + (dense ? first + i : (int)(gl[i] & 0x3fffu)); };
        auto issue2 = [&](int g) {                               // chunk g = (tile g / kc of the job, k-chunk g % kc), into slot g % 4
            const int ti = g / kc, c = g - ti * kc;
            const int qt = tile_at(ti);
#pragma unroll
            for (int part = 0; part < 2; ++part) {
                const char *qb = (part ? qlp : qhp) + (size_t)qt * PART + c * 4096;
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(qb + src2[j]),
                                                     (__attribute__((address_space(3))) void *)(reg + (g & 3) * CHB + part * (CHB / 2) + j * 1024), 16, 0, 0);
            }
        };
        const int n_g = n_t * kc;                                 // narrow descriptors (C = 32 zero-padded to 256): only the live chunks travel
#pragma unroll
        for (int d = 0; d < NCH - 1; ++d)
            if (d < n_g) issue2(d);
        for (int i = 0; i < n_t; ++i) {
            const int qt = tile_at(i);
            const int fmask = dense ? 3 : (int)(gl[i] >> 14);     // wave-uniform: which anchor blocks need this tile
            f32x16 acc[NAB];
#pragma unroll
            for (int ab = 0; ab < NAB; ++ab)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ab][r] = 0.0f;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                if (c < kc) {                                  // (wave-uniform; the dead chunks of a narrow row are zero)
                const int g = i * kc + c;
                const int rem = n_g - 1 - g;
                // chunk g has landed when at most min(2, chunks left) later chunks are in flight (the candidate stores / atomics of earlier
                // tiles only make the count conservative)
                if (rem >= 2) X3_WAIT(2 * PERC); else if (rem == 1) X3_WAIT(PERC); else X3_WAIT(0);
                __builtin_amdgcn_wave_barrier();
                // slot (g + 3) % 4 held chunk g - 1, whose operand reads completed before its MFMAs were issued
                if (g + NCH - 1 < n_g) issue2(g + NCH - 1);
                const char *cb = reg + (g & 3) * CHB;
#pragma unroll
                for (int s4 = 0; s4 < CH / 16; ++s4) {
                    const half8x xh = *reinterpret_cast<const half8x *>(cb + ko2[s4]);
                    const half8x xl = *reinterpret_cast<const half8x *>(cb + CHB / 2 + ko2[s4]);
                    const int s = c * (CH / 16) + s4;
                    // small terms first (as the PointDSC fp16x3 kernels do): lo.hi, hi.lo, hi.hi; the two anchor blocks alternate
                    if (fmask == 3) {
#pragma unroll
                        for (int ab = 0; ab < NAB; ++ab) acc[ab] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl, bh[ab][s], acc[ab], 0, 0, 0);
#pragma unroll
                        for (int ab = 0; ab < NAB; ++ab) acc[ab] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, bl[ab][s], acc[ab], 0, 0, 0);
#pragma unroll
                        for (int ab = 0; ab < NAB; ++ab) acc[ab] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, bh[ab][s], acc[ab], 0, 0, 0);
                    } else if (fmask == 1) {                  // one anchor block only: half the MFMAs
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl, bh[0][s], acc[0], 0, 0, 0);
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, bl[0][s], acc[0], 0, 0, 0);
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, bh[0][s], acc[0], 0, 0, 0);
                    } else {
                        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl, bh[1][s], acc[1], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, bl[1][s], acc[1], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, bh[1][s], acc[1], 0, 0, 0);
                    }
                }
                }
            }
            const int q0 = qt * ROWS + 4 * hi;
#pragma unroll
            for (int ab = 0; ab < NAB; ++ab) {
                if (!((fmask >> ab) & 1)) continue;           // this block's accumulators were not computed: none of its anchors can list a row here
                const int a = a_base + ab * 32 + l31;
                float x = -INFINITY;
#pragma unroll
                for (int r = 0; r < 16; ++r) x = fmaxf(x, (q0 + (r & 3) + 8 * (r >> 2) < nq) ? acc[ab][r] : -INFINITY);
                // the s3 scores themselves tighten the limit as they come in (the maximum is at least every s3 - DELTA3): rows behind the
                // peak that the fixed limit alone would still list are dropped
                run3[ab] = fmaxf(run3[ab], fmaxf(x, __shfl_xor(x, 32)));
                lim[ab] = fmaxf(lim[ab], run3[ab] - X3_MARGIN);
                if (a < nc && x >= lim[ab]) {
                    int n = 0;
#pragma unroll
                    for (int r = 0; r < 16; ++r) n += (acc[ab][r] >= lim[ab] && q0 + (r & 3) + 8 * (r >> 2) < nq) ? 1 : 0;   // zero-padded rows of the last tile are not candidates
                    const size_t lid = (((size_t)p * S + split) * cap_s + a) * 2 + hi;
                    int pos = atomicAdd(&cnt[lid], n);          // other jobs of this split may hold other tiles of the same anchor
                    uint2 *list = cand + lid * X3_CAPH;
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (acc[ab][r] >= lim[ab] && q0 + (r & 3) + 8 * (r >> 2) < nq) {
                            if (pos < X3_CAPH) list[pos] = make_uint2((unsigned)(q0 + (r & 3) + 8 * (r >> 2)), __float_as_uint(acc[ab][r]));
                            ++pos;
                        }
                }
            }
        }
    }
#undef X3_WAIT
}

// one wave per compacted anchor: final maximum over the lists of all query splits, stale entries dropped, canonical fp32 chain on the raw
// map for the rest (as resolve_anchor in match16.hip), result -> md_c / am_c / va_c; overflowed lists -> the pair's overflow list
template <bool NHWC>
__global__ __launch_bounds__(256) void match_x3_rescore_kernel(const float *__restrict__ a_c, int Cp, int cap_s, const int32_t *__restrict__ n_c,
                                                                const float *__restrict__ feat_q, int C_true, int HW,
                                                                const int32_t *__restrict__ roi_q, int roi_stride, const float *__restrict__ norm_q,
                                                                int cap_q, const int32_t *__restrict__ n_q, int S, float thr,
                                                                const int32_t *__restrict__ cnt, const uint2 *__restrict__ cand,
                                                                const __half *__restrict__ ah, const __half *__restrict__ al,
                                                                const __half *__restrict__ qh, const __half *__restrict__ ql,
                                                                int round_f16, float *__restrict__ md_c,
                                                                int32_t *__restrict__ am_c, uint8_t *__restrict__ va_c,
                                                                int32_t *__restrict__ n_ovf, int32_t *__restrict__ ovf_idx)
{
    extern __shared__ float lds_x3[];
    const int p = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + wave;
    const int nc = n_c[p] < cap_s ? n_c[p] : cap_s;
    if (row >= nc) return;
    const int nq = n_q[p];
    float *A = lds_x3 + wave * (2 * Cp + 2 * X3_SURV), *Q = A + Cp;
    int *sj = reinterpret_cast<int *>(Q + Cp);              // survivors of the first filter: row index, refined score
    float *sref = reinterpret_cast<float *>(sj + X3_SURV);
    // the 2 S list lengths at once (lane l < 2 S: list (query split l >> 1, lane half l & 1)); most lists are empty - a workgroup of the scan
    // only lists what comes within ~4e-4 of ITS split's maximum - so the loops below visit the non-empty ones only (walking all 16 lists with a
    // dependent load each cost 0.9 ms per cfg2 step of smooth inputs)
    int my_cnt = 0;
    if (lane < 2 * S) my_cnt = cnt[(((size_t)p * S + (lane >> 1)) * cap_s + row) * 2 + (lane & 1)];
    const bool overflow = __ballot(my_cnt > X3_CAPH) != 0ull;
    const unsigned long long nonempty = __ballot(my_cnt > 0);
    float m1 = -INFINITY;
    for (unsigned long long m = nonempty; m; m &= m - 1) {
        const int l = __ffsll((long long)m) - 1;
        const int c = __shfl(my_cnt, l);
        const size_t o = (((size_t)p * S + (l >> 1)) * cap_s + row) * 2 + (l & 1);
        for (int e = lane; e < c && e < X3_CAPH; e += 64) m1 = fmaxf(m1, __uint_as_float(cand[o * X3_CAPH + e].y));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m1 = fmaxf(m1, __shfl_xor(m1, off));
    const size_t crow = (size_t)p * cap_s + row;
    if (overflow) {                                        // wave-uniform: every lane read the same counts
        if (lane == 0) ovf_idx[(size_t)p * cap_s + atomicAdd(&n_ovf[p], 1)] = row;
        return;
    }
    for (int pos = lane; pos < Cp; pos += 64) {            // anchor row -> natural k order
        const int g = pos >> 3, hh = (pos >> 2) & 1, jj = pos & 3;
        A[8 * g + 2 * jj + hh] = a_c[crow * Cp + pos];
    }
    // second filter: entries within MARGIN3 of the best listed s3 get a REFINED score from the hi / lo rows - the exact products
    // (ah + al)(qh + ql) summed in fp64 - which is within DELTA_R = 1.63e-5 of the canonical fp32 chain (256 roundings of the chain:
    // 1.53e-5; split residuals, lo parts being subnormal halves: 9.6e-7; the score's own rounding to float), a quarter of the s3 bound
    // (whose two 256- / 768-term fp32 accumulations cost it 6e-5).  Rows are read as two coalesced
    // 512-byte loads per candidate; only entries within 2 DELTA_R of the best refined score go on to the canonical chain, whose raw-map
    // gather (one 64-byte sector per channel in NCHW: 16 KB per candidate) is what this kernel's time goes into.
    double a4[4];
    {
        const uint2 h = reinterpret_cast<const uint2 *>(ah + crow * Cp)[lane], l = reinterpret_cast<const uint2 *>(al + crow * Cp)[lane];
        const __half2 h0 = *reinterpret_cast<const __half2 *>(&h.x), h1 = *reinterpret_cast<const __half2 *>(&h.y);
        const __half2 l0 = *reinterpret_cast<const __half2 *>(&l.x), l1 = *reinterpret_cast<const __half2 *>(&l.y);
        a4[0] = (double)__low2float(h0) + (double)__low2float(l0);
        a4[1] = (double)__high2float(h0) + (double)__high2float(l0);
        a4[2] = (double)__low2float(h1) + (double)__low2float(l1);
        a4[3] = (double)__high2float(h1) + (double)__high2float(l1);
    }
    int ns = 0;
    float ref_max = -INFINITY;
    for (unsigned long long m = nonempty; m; m &= m - 1) {
        const int l = __ffsll((long long)m) - 1;
        const int n = __shfl(my_cnt, l);
        const size_t o = (((size_t)p * S + (l >> 1)) * cap_s + row) * 2 + (l & 1);
        for (int e0 = 0; e0 < n; e0 += 64) {
            const int e_ = e0 + lane;
            const uint2 ent = e_ < n ? cand[o * X3_CAPH + e_] : make_uint2(0u, 0u);
            const int qi = (int)ent.x;
            const bool hit = e_ < n && qi < nq && __uint_as_float(ent.y) >= m1 - X3_MARGIN;
            unsigned long long hits = __ballot(hit);
            while (hits) {
                const int src = __ffsll((long long)hits) - 1;
                hits &= hits - 1;
                const int jj = __shfl(qi, src);
                const size_t qoff = (size_t)p * cap_q * Cp * 2 + x3_q_off(jj, lane * 8);          // bytes: lane l holds channels 4 l .. 4 l + 3
                const bool live_chunk = lane * 4 < C_true;                   // chunks beyond the map's channels are never written
                const uint2 h = live_chunk ? *reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(qh) + qoff) : make_uint2(0u, 0u);
                const uint2 lo_ = live_chunk ? *reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(ql) + qoff) : make_uint2(0u, 0u);
                const __half2 h0 = *reinterpret_cast<const __half2 *>(&h.x), h1 = *reinterpret_cast<const __half2 *>(&h.y);
                const __half2 l0 = *reinterpret_cast<const __half2 *>(&lo_.x), l1 = *reinterpret_cast<const __half2 *>(&lo_.y);
                double v = a4[0] * ((double)__low2float(h0) + (double)__low2float(l0));
                v = fma(a4[1], (double)__high2float(h0) + (double)__high2float(l0), v);
                v = fma(a4[2], (double)__low2float(h1) + (double)__low2float(l1), v);
                v = fma(a4[3], (double)__high2float(h1) + (double)__high2float(l1), v);
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
                const float r = (float)v;
                ref_max = fmaxf(ref_max, r);
                if (ns < X3_SURV && lane == 0) { sj[ns] = jj; sref[ns] = r; }
                ++ns;
            }
        }
    }
    if (ns > X3_SURV) {                                     // a crowd of near-duplicates: the exact scan takes this anchor (wave-uniform)
        if (lane == 0) ovf_idx[(size_t)p * cap_s + atomicAdd(&n_ovf[p], 1)] = row;
        return;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float d = INFINITY;
    int j = 0x7fffffff;
    const float *fq = feat_q + (size_t)p * C_true * HW;
    for (int i = 0; i < ns; ++i) {
        if (!(sref[i] >= ref_max - X3_MARGIN_R)) continue;  // wave-uniform (LDS broadcast)
        const int jj = sj[i];
        {
            const int pix = roi_q[(size_t)p * roi_stride + jj];
            const float dq = norm_q[(size_t)p * cap_q + jj];
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (int k = lane; k < Cp; k += 64) {
                float x = 0.0f;
                if (k < C_true) x = NHWC ? fq[(size_t)pix * C_true + k] : fq[(size_t)k * HW + pix];
                if (round_f16) x = __half2float(__float2half_rn(x));
                Q[k] = __fdiv_rn(x, dq);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            float dot = 0.0f;                               // every lane runs the same canonical chain on broadcast LDS reads
            for (int k = 0; k < C_true; k += 8) {
                float av[8], qv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) { av[e] = A[k + e]; qv[e] = Q[k + e]; }
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (k + e < C_true) dot = __fmaf_rn(av[e], qv[e], dot);
            }
            lex_min(d, j, __fmaf_rn(-0.5f, dot, 0.5f), jj);
        }
    }
    if (lane == 0) {
        md_c[crow] = d;
        am_c[crow] = j == 0x7fffffff ? 0 : j;
        va_c[crow] = (d < thr) ? 1 : 0;
    }
}

// results of the exact fall-back for overflowed anchors -> their compact rows
__global__ __launch_bounds__(256) void match_x3_scatter_ovf_kernel(int cap_s, const int32_t *__restrict__ n_ovf, const int32_t *__restrict__ ovf_idx,
                                                                    const float *__restrict__ md_o, const int32_t *__restrict__ am_o,
                                                                    const uint8_t *__restrict__ va_o, float *__restrict__ md_c,
                                                                    int32_t *__restrict__ am_c, uint8_t *__restrict__ va_c)
{
    const int p = blockIdx.y, sl = blockIdx.x * 256 + threadIdx.x;
    if (sl >= n_ovf[p] || sl >= cap_s) return;
    const size_t src = (size_t)p * cap_s + sl, dst = (size_t)p * cap_s + ovf_idx[(size_t)p * cap_s + sl];
    md_c[dst] = md_o[src];
    am_c[dst] = am_o[src];
    va_c[dst] = va_o[src];
}

int gather_q8_launch(const float *feat, int n_maps, int C, int HW, int layout, const int32_t *roi, int roi_stride, const int32_t *count,
                     const int32_t *map_enable, int rows_cap, int C_pad, int8_t *out8, float *scale, float *eps, float *norm,
                     float *out32, int lanes_per_row, int round_f16, hipStream_t st, int fmt, void *aux);

// jobs a (split, 64-anchor group) can post: its tiles in chunks of X3_JOB_TILES
static int x3_jobs_per_owner(int cap_q, int S)
{
    const int ntl_max = ((cap_q + 31) / 32 + S - 1) / S + 1;
    return (ntl_max + X3_JOB_TILES - 1) / X3_JOB_TILES;
}

size_t match_x3_scratch_bytes(int B, int cap_s, int S, int cap_q)
{
    const size_t owners = (size_t)B * S * ((cap_s + 511) / 512) * 8;
    return (size_t)B * S * cap_s * 2 * (sizeof(int32_t) + X3_CAPH * sizeof(uint2)) + (size_t)B * (3 * cap_s + 3) * sizeof(int32_t) + 8192 + 4096 +
           (size_t)B * S * cap_s * sizeof(float) /* smax */ + owners * X3_MAX_TILES * sizeof(unsigned short) /* tile lists */ +
           owners * x3_jobs_per_owner(cap_q, S) * sizeof(uint2) /* jobs */ + 1024;
}

// a_c [B, cap_s, 256] fp32 compact anchor rows (k-permuted), n_c [B] -> md_c / am_c / va_c [B, cap_s]; n_ovf / ovf_idx: anchors whose
// lists overflowed (to be redone by the exact scan).  qh / ql: room for [B, cap_q, 256] halves each; ah / al: [B, cap_s, 256] halves;
// scratch: match_x3_scratch_bytes.  Every launch is gated on the device by n_c.
int match_x3_resolve(const float *a_c, const int32_t *n_c, int cap_s, const float *feat_q, int C_true, int HW, int layout,
                     const int32_t *roi_q, int roi_stride_q, const float *q_norm, const int32_t *n_q, int B, int cap_q, float threshold,
                     int round_f16, __half *qh, __half *ql, __half *ah, __half *al, void *scratch, float *md_c, int32_t *am_c, uint8_t *va_c,
                     int32_t **n_ovf_out, int32_t **ovf_idx_out, const int32_t *orig_idx, int orig_stride, const int32_t *sid_final, int cap_a,
                     const __half *q_hi_lo_pre, const float *q_lo_sq_max_pre, hipStream_t st)
{
    constexpr int CP = 256;
    // sweep 1 as 8-wave workgroups (512 anchors, two waves per SIMD) unless ORYON_X3_WAVES=4 (256 anchors, one wave per SIMD)
    static const int x3_waves = (dev_env_int("ORYON_X3_WAVES", 8) == 4) ? 4 : 8;
    const int T = (cap_s + 64 * x3_waves - 1) / (64 * x3_waves);
    const int G = T * x3_waves;                                  // 64-anchor groups per (pair, split)
    const int S = 8;
    char *sp = static_cast<char *>(scratch);
    int32_t *cnt = reinterpret_cast<int32_t *>(sp);
    size_t off = ((size_t)B * S * cap_s * 2 * sizeof(int32_t) + 255) / 256 * 256;
    uint2 *cand = reinterpret_cast<uint2 *>(sp + off);
    off += ((size_t)B * S * cap_s * 2 * X3_CAPH * sizeof(uint2) + 255) / 256 * 256;
    int32_t *n_ovf = reinterpret_cast<int32_t *>(sp + off);             // n_ovf | ql_max: adjacent, zeroed by ONE memset
    const size_t cnt_block = ((size_t)B * sizeof(int32_t) + 255) / 256 * 256;
    off += cnt_block;
    float *ql_max = reinterpret_cast<float *>(sp + off);
    off += cnt_block;
    int32_t *queue = reinterpret_cast<int32_t *>(sp + off);            // sweep 1's per-XCD item counters [0..7], njobs [8], next_job [9]
    int32_t *njobs = queue + 8, *next_job = queue + 9;                  // (zeroed by the same memset)
    off += 256;
    int32_t *ovf_idx = reinterpret_cast<int32_t *>(sp + off);
    off += ((size_t)B * cap_s * sizeof(int32_t) + 255) / 256 * 256;
    float *al_norm = reinterpret_cast<float *>(sp + off);
    off += ((size_t)B * cap_s * sizeof(float) + 255) / 256 * 256;
    float *seed = reinterpret_cast<float *>(sp + off);
    off += ((size_t)B * cap_s * sizeof(float) + 255) / 256 * 256 + 8192;                    // + the debug counters' slack
    float *smax = reinterpret_cast<float *>(sp + off);
    off += ((size_t)B * S * cap_s * sizeof(float) + 255) / 256 * 256;
    const size_t owners = (size_t)B * S * G;
    unsigned short *tl = reinterpret_cast<unsigned short *>(sp + off);
    off += (owners * X3_MAX_TILES * sizeof(unsigned short) + 255) / 256 * 256;
    uint2 *jobs = reinterpret_cast<uint2 *>(sp + off);
    *n_ovf_out = n_ovf;
    *ovf_idx_out = ovf_idx;
    if (hipMemsetAsync(n_ovf, 0, 2 * cnt_block + 256, st) != hipSuccess) return ORYON_ERR_HIP;
    if (hipMemsetAsync(cnt, 0, (size_t)B * S * cap_s * 2 * sizeof(int32_t), st) != hipSuccess) return ORYON_ERR_HIP;     // sweep 2 appends with atomics
    // query rows as hi / lo halves, only for pairs that have listed anchors (the gather's per-map gate reads the counts themselves) -
    // unless the caller's K0 pass already wrote them for every pair (oryon_gather_mx6_x3: the step engine does when recent steps came here)
    int rc = ORYON_OK;
    if (q_hi_lo_pre && q_lo_sq_max_pre) {
        qh = const_cast<__half *>(q_hi_lo_pre);
        ql = qh + (size_t)B * cap_q * CP;
        ql_max = const_cast<float *>(q_lo_sq_max_pre);
    } else {
        rc = gather_q8_launch(feat_q, B, C_true, HW, layout, roi_q, roi_stride_q, n_q, n_c, cap_q, CP, reinterpret_cast<int8_t *>(qh), nullptr,
                              ql_max, nullptr, nullptr, 1, round_f16, st, 2, ql);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(match_x3_split_anchors_kernel, dim3(64, B), dim3(256), 0, st, a_c, CP, cap_s, n_c, ah, al, al_norm);
    const int groups = ((B * S + 7) / 8) * 8 * T;
    static const bool dbg = dev_env_set("ORYON_X3_DEBUG");          // development aid: list statistics of this call on stderr
    int32_t *dbg_dev = nullptr;
    if (dbg) {
        dbg_dev = reinterpret_cast<int32_t *>(seed + (size_t)B * cap_s + 64);         // in the scratch's 8 KB of slack
        (void)hipMemsetAsync(dbg_dev, 0, 96, st);
        (void)hipMemsetAsync(dbg_dev + 8, 0xff, 8, st);
    }
    // seeds of the running maxima from the screen's winning slices (ORYON_X3_SEED=0: the round-3 scan, for A/B timing; same results)
    static const bool use_seed = dev_env_int("ORYON_X3_SEED", 1) != 0;
    const float *seed_arg = nullptr;
    if (use_seed && orig_idx && sid_final) {
        hipLaunchKernelGGL(match_x3_seed_kernel, dim3((cap_s + 3) / 4, B), dim3(256), 0, st, ah, qh, CP, cap_s, cap_q, n_c, n_q, orig_idx, orig_stride,
                           sid_final, cap_a, (C_true + 63) / 64, seed);
        seed_arg = seed;
    }
    static long long *dbg_wg = nullptr;
    if (dbg && !dbg_wg) (void)hipMalloc(&dbg_wg, (size_t)65536 * 4 * sizeof(long long));
    if (dbg && dbg_wg) (void)hipMemsetAsync(dbg_wg, 0, (size_t)(groups < 65536 ? groups : 65536) * 4 * sizeof(long long), st);
    constexpr int X3_SCAN_LDS = 4 * 2 * 32 * CP * 2 + 8 * X3_MAX_TILES + 8 * 2 * X3_MAX_TILES + 64;   // 128 KB of tile ring + tile flags + the waves' tile lists
    allow_dynamic_lds(reinterpret_cast<const void *>(&match_x3_scan_kernel<CP, 4, false>), X3_SCAN_LDS);
    allow_dynamic_lds(reinterpret_cast<const void *>(&match_x3_scan_kernel<CP, 8, false>), X3_SCAN_LDS);
    allow_dynamic_lds(reinterpret_cast<const void *>(&match_x3_scan_kernel<CP, 4, true>), X3_SCAN_LDS);
    allow_dynamic_lds(reinterpret_cast<const void *>(&match_x3_scan_kernel<CP, 8, true>), X3_SCAN_LDS);
    // one workgroup per CU, items pulled from per-XCD queues (see the kernel)
    static int n_cus = 0;
    if (!n_cus) {
        int dev_ = 0, v = 0;
        (void)hipGetDevice(&dev_);
        n_cus = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev_) == hipSuccess && v > 0) ? v : 256;
    }
    const int grid = groups < n_cus ? groups : n_cus / 8 * 8;
    const int kc = (C_true + 63) / 64;                          // live 64-channel chunks of a row (narrow maps are zero-padded to 256)
#define X3_SCAN_LAUNCH(WV, NR)                                                                                                   \
    hipLaunchKernelGGL((match_x3_scan_kernel<CP, WV, NR>), dim3(grid), dim3(64 * WV), X3_SCAN_LDS, st, ah, al, qh, ql, B, cap_s, cap_q, n_c, n_q, T, \
                       S, al_norm, ql_max, seed_arg, smax, tl, jobs, njobs, queue, groups / 8, kc, dbg_dev, dbg_wg)
    if (x3_waves == 8) { if (kc < 4) X3_SCAN_LAUNCH(8, true); else X3_SCAN_LAUNCH(8, false); }
    else { if (kc < 4) X3_SCAN_LAUNCH(4, true); else X3_SCAN_LAUNCH(4, false); }
#undef X3_SCAN_LAUNCH
    // sweep 2: four independent waves per CU pull the jobs sweep 1 posted (none posted: they exit at once)
#define X3_SWEEP2_LAUNCH(NR)                                                                                                     \
    hipLaunchKernelGGL((match_x3_sweep2_kernel<CP, NR>), dim3(n_cus), dim3(256), 4 * 32768, st, ah, al, qh, ql, cap_s, cap_q, n_c, n_q, G, S, al_norm, \
                       ql_max, smax, tl, jobs, njobs, next_job, cnt, cand, kc, dbg_dev)
    allow_dynamic_lds(reinterpret_cast<const void *>(&match_x3_sweep2_kernel<CP, true>), 4 * 32768);
    allow_dynamic_lds(reinterpret_cast<const void *>(&match_x3_sweep2_kernel<CP, false>), 4 * 32768);
    if (kc < 4) X3_SWEEP2_LAUNCH(true); else X3_SWEEP2_LAUNCH(false);
#undef X3_SWEEP2_LAUNCH
    const size_t lds = (size_t)4 * (2 * CP + 2 * X3_SURV) * sizeof(float);
    if (layout == ORYON_LAYOUT_NHWC)
        hipLaunchKernelGGL((match_x3_rescore_kernel<true>), dim3(cap_s / 4, B), dim3(256), lds, st, a_c, CP, cap_s, n_c, feat_q, C_true, HW, roi_q,
                           roi_stride_q, q_norm, cap_q, n_q, S, threshold, cnt, cand, ah, al, qh, ql, round_f16, md_c, am_c, va_c, n_ovf, ovf_idx);
    else
        hipLaunchKernelGGL((match_x3_rescore_kernel<false>), dim3(cap_s / 4, B), dim3(256), lds, st, a_c, CP, cap_s, n_c, feat_q, C_true, HW, roi_q,
                           roi_stride_q, q_norm, cap_q, n_q, S, threshold, cnt, cand, ah, al, qh, ql, round_f16, md_c, am_c, va_c, n_ovf, ovf_idx);
    if (dbg) {
        (void)hipStreamSynchronize(st);
        std::vector<int32_t> hc((size_t)B * S * cap_s * 2), hn(B), hov(B);
        std::vector<float> hq(B), ha((size_t)B * cap_s);
        (void)hipMemcpy(hc.data(), cnt, hc.size() * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(hn.data(), n_c, (size_t)B * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(hov.data(), n_ovf, (size_t)B * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(hq.data(), ql_max, (size_t)B * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(ha.data(), al_norm, ha.size() * 4, hipMemcpyDeviceToHost);
        long tot = 0, nl = 0, over = 0, mx = 0, anchors = 0, novf = 0;
        double amax = 0, qmax = 0;
        for (int p = 0; p < B; ++p) {
            const int n = hn[p] < cap_s ? hn[p] : cap_s;
            anchors += n; novf += hov[p];
            qmax = sqrtf(hq[p]) > qmax ? sqrtf(hq[p]) : qmax;
            for (int a = 0; a < n; ++a) {
                amax = ha[(size_t)p * cap_s + a] > amax ? ha[(size_t)p * cap_s + a] : amax;
                for (int sp = 0; sp < S; ++sp)
                    for (int h = 0; h < 2; ++h) {
                        const int c = hc[((((size_t)p * S + sp) * cap_s + a) * 2) + h];
                        tot += c; nl += 1; over += c > X3_CAPH; mx = c > mx ? c : mx;
                    }
            }
        }
        int32_t hd[24] = {0};
        (void)hipMemcpy(hd, dbg_dev, 96, hipMemcpyDeviceToHost);
        if (hd[17] > 0)
            fprintf(stderr, "[x3] sweep 2: %d waves, mean alive %.1f us (longest %.1f us), operand loads %.1f us per wave, %d jobs, %d tiles (%.2f us of wave time per tile)\n",
                    hd[17], hd[12] * 0.01 / hd[17], hd[16] * 0.01, hd[13] * 0.01 / hd[17], hd[15], hd[14], hd[14] ? (hd[12] - hd[13]) * 0.01 / hd[14] : 0.0);
        {
            unsigned long long t_lo, t_hi;
            memcpy(&t_lo, hd + 8, 8);
            memcpy(&t_hi, hd + 10, 8);
            fprintf(stderr, "[x3] scan span %.1f us (first start to last end), slowest workgroup %.1f us\n", (double)(t_hi - t_lo) * 0.01, hd[7] * 0.01);
        }
        fprintf(stderr, "[x3] sweep-2 (wave, tile) pairs flagged: %d of %d\n", hd[0], hd[1]);
        if (dbg_wg && groups <= 65536) {
            std::vector<long long> w((size_t)groups * 4);
            (void)hipMemcpy(w.data(), dbg_wg, w.size() * sizeof(long long), hipMemcpyDeviceToHost);
            long long t0 = -1;
            for (int g = 0; g < groups; ++g) if (w[4 * g + 1] && (t0 < 0 || w[4 * g] < t0)) t0 = w[4 * g];
            std::vector<unsigned long long> ids;
            for (int g = 0; g < groups; ++g) if (w[4 * g + 1]) ids.push_back(((unsigned long long)(w[4 * g + 3] & 0xf) << 32) | (unsigned long long)(w[4 * g + 2] & 0xff00));
            std::sort(ids.begin(), ids.end());
            const size_t distinct = std::unique(ids.begin(), ids.end()) - ids.begin();
            fprintf(stderr, "[x3] workgroups ran on %zu distinct (xcc, se, sh, cu) places\n", distinct);
            // concurrency profile: workgroups running at 10 sample times, and the start time of every 128th workgroup in launch order
            long long t1 = 0;
            for (int g = 0; g < groups; ++g) if (w[4 * g + 1] > t1) t1 = w[4 * g + 1];
            fprintf(stderr, "[x3] running at 5%%..95%% of the span:");
            for (int k = 0; k < 10; ++k) {
                const long long ts = t0 + (t1 - t0) * (2 * k + 1) / 20;
                int r = 0;
                for (int g = 0; g < groups; ++g) r += w[4 * g + 1] && w[4 * g] <= ts && ts < w[4 * g + 1];
                fprintf(stderr, " %d", r);
            }
            fprintf(stderr, "\n[x3] start (us) of block 0, 128, 256, ...:");
            for (int g = 0; g < groups; g += 128) fprintf(stderr, " %.0f", w[4 * g + 1] ? (w[4 * g] - t0) * 0.01 : -1.0);
            fprintf(stderr, "\n[x3] raw ids of blocks 0..11 (hw_id, xcc_id):");
            for (int g = 0; g < 12 && g < groups; ++g) fprintf(stderr, " (%llx,%llx)", (unsigned long long)w[4 * g + 2], (unsigned long long)w[4 * g + 3]);
            fprintf(stderr, "\n");
        }
        if (hd[6] > 0)
            fprintf(stderr, "[x3] per workgroup (%d ran): operand load %.1f us, sweep 1 %.1f us, list + sweep 2 %.1f us; sweep-2 tiles visited %.1f of %.1f\n", hd[6],
                    hd[2] * 0.01 / hd[6], hd[3] * 0.01 / hd[6], hd[4] * 0.01 / hd[6], (double)hd[5] / hd[6], (double)hd[1] / (4.0 * hd[6]));
        fprintf(stderr, "[x3] anchors %ld, lists %ld, entries %ld (%.1f per anchor), longest %ld, overflowed lists %ld, overflowed anchors %ld, max|al| %.3g, max|ql| %.3g\n",
                anchors, nl, tot, anchors ? (double)tot / anchors : 0.0, mx, over, novf, amax, qmax);
    }
    return hipGetLastError() == hipSuccess ? ORYON_OK : ORYON_ERR_HIP;
}

void match_x3_scatter_ovf(int B, int cap_s, const int32_t *n_ovf, const int32_t *ovf_idx, const float *md_o, const int32_t *am_o,
                          const uint8_t *va_o, float *md_c, int32_t *am_c, uint8_t *va_c, hipStream_t st)
{
    hipLaunchKernelGGL(match_x3_scatter_ovf_kernel, dim3((cap_s + 255) / 256, B), dim3(256), 0, st, cap_s, n_ovf, ovf_idx, md_o, am_o, va_o, md_c,
                       am_c, va_c);
}

}  // namespace oryon
