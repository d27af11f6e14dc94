// K1: cosine nearest-neighbour matcher on the fp32 matrix cores of gfx950.
//
// Replaces utils/pcd.py:202-205 of the reference:
//     dist = 0.5 * (1 - cosine_similarity(A[:,None], B[None], dim=2));  amin / argmin over dim 1;  min_dist < th
// The reference materialises a [N1,N2,C] temporary; here nothing of size N1xN2 ever exists.
//
// Formulation:  S = Q^ * A^T  (rows = query descriptors, columns = anchor descriptors), both operands
// already gathered + normalised into k-contiguous [N, C] rows by K0.  One workgroup owns a 128-anchor
// column panel and streams the query rows past it in 128-row tiles; each of its 4 waves holds a 64x64
// block of S in four 32x32 v_mfma_f32_32x32x2_f32 accumulators.  In that instruction's C/D layout a lane
// owns ONE column (= one anchor) and 16 rows (= 16 queries), so the running (min, argmin) over queries is
// lane-local: no cross-lane traffic until the very end (lane l <-> l+32, then the two waves sharing a
// column panel through LDS).
//
// Numerics: the f32-input MFMA is an exact k-ordered fmaf chain (guide §3), so
//     dot = fma(a_{C-1} q_{C-1}, ... fma(a_0 q_0, 0)),   dist = fma(-0.5, dot, 0.5) = 0.5*(1-dot) bitwise,
// which is what oracle/oryon_oracle.c computes: min_dist, argmin (first index on ties) and valid are
// bit-identical to that oracle.
//
// Bound: 2*N1*N2*C flops against 157.3 TF/s (fp32 MFMA); HBM traffic is (N1+N2)*C*4 B per pair because all
// column panels of a pair run on ONE XCD (blockIdx -> (pair, panel) map below) and share the query stream
// through that XCD's L2.
#include <stdlib.h>
#include "common.h"
#include "match_common.h"

namespace oryon {

constexpr int MT = ORYON_MATCH_TILE;  // 128 queries x 128 anchors per workgroup step
constexpr int BK = 32;                // k-tile
constexpr int LD = BK + 1;            // padded LDS row: ds_read_b32 of a 32-row column is conflict-free
constexpr int MATCH_THREADS = 256;
constexpr int MAX_SPLIT = 16;
constexpr int TILE_FLOATS = MT * LD;

__global__ __launch_bounds__(MATCH_THREADS, 2) void match_f32_kernel(
    const float *__restrict__ a_hat, const float *__restrict__ q_hat, int B, int Cp, int cap_a, int cap_q,
    const int32_t *__restrict__ n_a, const int32_t *__restrict__ n_q, float thr, int T, int S,
    float *__restrict__ min_dist, int32_t *__restrict__ argmin, uint8_t *__restrict__ valid,
    float *__restrict__ ws_dist, int32_t *__restrict__ ws_idx, const int32_t *__restrict__ panel_flag,
    const uint8_t *__restrict__ row_flag)
{
    __shared__ float smem[4 * TILE_FLOATS];  // [buf][Q|A][128][33]

    // blockIdx -> (pair, anchor panel, query split).  Blocks are dealt to XCDs round-robin (b % 8), so
    // giving XCD x the pairs {x, x+8, ...} keeps every panel of a pair on one L2.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int per_pair = T * S;
    const int p = (slot / per_pair) * 8 + xcd;
    if (p >= B) return;
    const int rem = slot % per_pair;
    const int panel = rem / S, split = rem % S;
    const int na = n_a[p], nq = n_q[p];
    const int a0 = panel * MT;
    if (a0 >= na) return;
    if (panel_flag && !panel_flag[(size_t)p * T + panel]) return;   // K1s fallback mode: only the flagged panels are recomputed

    const int nqt = (nq + MT - 1) / MT;
    const int qt_per = (nqt + S - 1) / S;
    const int qt_begin = split * qt_per;
    const int qt_end = (qt_begin + qt_per < nqt) ? qt_begin + qt_per : nqt;
    const int KT = Cp / BK;
    const int nit = (qt_end > qt_begin) ? (qt_end - qt_begin) * KT : 0;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;  // wave grid: 2 (query halves) x 2 (anchor halves)
    const int l31 = lane & 31, hi = lane >> 5;

    const float *ap = a_hat + ((size_t)p * cap_a + a0) * Cp;
    const float *qp = q_hat + (size_t)p * cap_q * Cp;

    float4 rq[4], ra[4];
    auto gload = [&](int it) {
        const int qt = qt_begin + it / KT, kt = it % KT;
        const float *qb = qp + (size_t)qt * MT * Cp + kt * BK;
        const float *ab = ap + kt * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = t + MATCH_THREADS * i, row = f >> 3, c4 = f & 7;
            rq[i] = *reinterpret_cast<const float4 *>(qb + (size_t)row * Cp + c4 * 4);
            ra[i] = *reinterpret_cast<const float4 *>(ab + (size_t)row * Cp + c4 * 4);
        }
    };
    auto lstore = [&](int buf) {
        float *Qs = smem + buf * 2 * TILE_FLOATS, *As = Qs + TILE_FLOATS;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = t + MATCH_THREADS * i, row = f >> 3, c4 = f & 7;
            float *q = Qs + row * LD + c4 * 4, *a = As + row * LD + c4 * 4;
            q[0] = rq[i].x; q[1] = rq[i].y; q[2] = rq[i].z; q[3] = rq[i].w;
            a[0] = ra[i].x; a[1] = ra[i].y; a[2] = ra[i].z; a[3] = ra[i].w;
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;

    float best[2] = {INFINITY, INFINITY};
    int bidx[2] = {0x7fffffff, 0x7fffffff};

    if (nit > 0) {
        gload(0);
        lstore(0);
    }
    __syncthreads();

    for (int it = 0; it < nit; ++it) {
        const int cur = it & 1;
        const bool more = it + 1 < nit;
        if (more) gload(it + 1);

        // rows hold k in the permuted order written by K0 (position 8g+4h+j <- k = 8g+2j+h): MFMA step ks
        // (k = 2ks + hi) sits at position 8*(ks/4) + 4*hi + ks%4
        const float *Qs = smem + cur * 2 * TILE_FLOATS + (wm * 64 + l31) * LD + 4 * hi;
        const float *As = smem + cur * 2 * TILE_FLOATS + TILE_FLOATS + (wn * 64 + l31) * LD + 4 * hi;
#pragma unroll
        for (int ks = 0; ks < BK / 2; ++ks) {
            const int kp = 8 * (ks >> 2) + (ks & 3);
            const float q0 = Qs[kp], q1 = Qs[32 * LD + kp];
            const float b0 = As[kp], b1 = As[32 * LD + kp];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(q0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(q0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(q1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(q1, b1, acc[1][1], 0, 0, 0);
        }

        if ((it % KT) == KT - 1) {
            // dot products of this query tile are complete: fold them into the running (min, argmin).
            const int qt = qt_begin + it / KT;
            const int qlane = qt * MT + wm * 64 + 4 * hi;
            const bool full = (qt + 1) * MT <= nq;
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int q = qlane + mi * 32 + (r & 3) + 8 * (r >> 2);
                        float d = __fmaf_rn(-0.5f, acc[mi][ni][r], 0.5f);
                        if (!full) d = (q < nq) ? d : INFINITY;
                        const bool better = d < best[ni];   // strict: first (smallest) query index wins ties
                        best[ni] = better ? d : best[ni];
                        bidx[ni] = better ? q : bidx[ni];
                        acc[mi][ni][r] = 0.0f;
                    }
        }
        if (more) lstore(cur ^ 1);
        __syncthreads();
    }

    // lane l and l+32 hold the same anchor column (different query rows)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const float od = __shfl_xor(best[ni], 32);
        const int oi = __shfl_xor(bidx[ni], 32);
        lex_min(best[ni], bidx[ni], od, oi);
    }
    // waves (wm=0, wn) and (wm=1, wn) hold the same columns: merge through LDS (tiles are free now)
    float *sd = smem;
    int *si = reinterpret_cast<int *>(smem + 2 * MT);
    if (hi == 0) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            sd[wm * MT + wn * 64 + ni * 32 + l31] = best[ni];
            si[wm * MT + wn * 64 + ni * 32 + l31] = bidx[ni];
        }
    }
    __syncthreads();
    if (t < MT && a0 + t < na) {
        float d = sd[t];
        int i = si[t];
        lex_min(d, i, sd[MT + t], si[MT + t]);
        if (S == 1) {
            const size_t o = (size_t)p * cap_a + a0 + t;
            if (!row_flag || row_flag[o]) {
                min_dist[o] = d;
                argmin[o] = (i == 0x7fffffff) ? 0 : i;
                valid[o] = (d < thr) ? 1 : 0;
            }
        } else {
            const size_t o = ((size_t)p * S + split) * cap_a + a0 + t;
            ws_dist[o] = d;
            ws_idx[o] = i;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Anchor-stationary kernel (C_pad in {32,64,128,256}).  Every wave keeps ITS 32 anchor rows - the MFMA B
// operand for the whole K range - in registers for the life of the workgroup, so the only thing that moves
// is the query stream:  HBM/L2 -> LDS by LDS-DMA (global_load_lds, 16 B per lane, no staging VGPRs),
// LDS -> MFMA A operand by ds_read_b128.
//   * LDS tile = 32 KB = ROWS query rows x KH floats, double-buffered (64 KB -> 2 workgroups per CU);
//     ROWS = 64 and KH = 128 for C_pad >= 128 (C_pad = 256 takes two k-parts per query tile).  A wave always
//     runs 128 MFMAs (8192 cycles) per barrier, on NACC = ROWS/32 >= 2 accumulators issued round-robin so that
//     back-to-back MFMAs are independent.
//   * The k-permuted row layout written by K0 (position 8g+4h+j holds k = 8g+2j+h) makes one 16-byte access
//     deliver what lane (i, h) feeds to four consecutive MFMA steps, while the accumulation order stays the
//     natural k order (bit-exact vs the oracle).
//   * LDS image: 256-byte lines, 16-byte slots XOR-swizzled with the row index: slot = chunk ^ (row & 15).
//     The DMA writes lane-linearly (hardware constraint) from a per-lane pre-swizzled SOURCE address; the
//     ds_read_b128 of a 16-lane group then hits 16 distinct slots (rows distinct mod 16): conflict-free.
//   * Work units: (pair, query split) units are dealt round-robin to the 8 XCDs (blockIdx & 7) and all anchor
//     panels of a unit sit on ONE XCD, so a unit's query stream is fetched once per XCD and shared through
//     that XCD's L2; several splits per pair keep the tail of the launch short.
template <int CP, int VAR = 0>   // VAR != 0: timing ablations only (ORYON_MATCH_VARIANT), results are then meaningless
__global__ __launch_bounds__(MATCH_THREADS, 2) void match_f32_regb_kernel(
    const float *__restrict__ a_hat, const float *__restrict__ q_hat, int B, int cap_a, int cap_q,
    const int32_t *__restrict__ n_a, const int32_t *__restrict__ n_q, float thr, int T, int S,
    float *__restrict__ min_dist, int32_t *__restrict__ argmin, uint8_t *__restrict__ valid,
    float *__restrict__ ws_dist, int32_t *__restrict__ ws_idx, const int32_t *__restrict__ panel_flag,
    const uint8_t *__restrict__ row_flag)
{
    constexpr int RB = CP * 4;                          // source row bytes
    constexpr int ROWS = CP >= 128 ? 64 : 8192 / CP;    // query rows per LDS tile
    constexpr int KH = 8192 / ROWS;                     // k extent of one LDS tile
    constexpr int KP = CP / KH;                         // k-parts per query tile (1 or 2)
    constexpr int NACC = ROWS / 32;                     // 32x32 accumulators per wave
    constexpr int NGT = KH / 8;                         // 8-wide k groups per LDS tile
    constexpr int TILE_BYTES = 32768;
    constexpr int NI = 8;                               // DMA wave-instructions per wave per tile (1 KB each)
    constexpr int RG = NACC >= 4 ? 2 : 3;               // fragment ring depth, in k groups
    constexpr bool NARROW = KH < 64;                    // two 128-byte rows share one 256-byte line
    constexpr int LPR = NARROW ? 1 : KH * 4 / 256;      // 256-byte lines per tile row
    static_assert(NACC * NGT == 32 && KP * NGT * 8 == CP, "tile geometry");
    __shared__ __attribute__((aligned(256))) char smem[2 * TILE_BYTES];

    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int unit = (slot / T) * 8 + xcd;              // (pair, split) unit; all T panels of a unit share an XCD
    if (unit >= B * S) return;
    const int panel = slot % T;
    const int p = unit / S, split = unit % S;
    const int na = n_a[p], nq = n_q[p];
    const int a0 = panel * MT;
    if (a0 >= na) return;
    if (panel_flag && !panel_flag[(size_t)p * T + panel]) return;   // flagged-recompute mode: untouched panels exit
    const int nqt = (nq + ROWS - 1) / ROWS;
    const int qt_per = (nqt + S - 1) / S;
    const int qt_begin = split * qt_per;
    const int qt_end = (qt_begin + qt_per < nqt) ? qt_begin + qt_per : nqt;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, hi = lane >> 5;
    const char *qp = reinterpret_cast<const char *>(q_hat + (size_t)p * cap_q * CP);

    // B operand: this lane's anchor row, positions 8g + 4*hi .. +3  (k = 8g + hi, +2, +4, +6)
    float4 breg[CP / 8];
    {
        const float *arow = a_hat + ((size_t)p * cap_a + a0 + wave * 32 + l31) * CP + 4 * hi;
#pragma unroll
        for (int g = 0; g < CP / 8; ++g) breg[g] = *reinterpret_cast<const float4 *>(arow + 8 * g);
    }

    // LDS-DMA source offsets (bytes relative to the tile's first row / k-part), one per wave-instruction
    unsigned dma_off[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int line = (wave * NI + j) * 4 + (lane >> 4), sl = lane & 15;
        if (NARROW) {
            const int cc = sl ^ (line & 15);
            dma_off[j] = (unsigned)((line * 2 + (cc >> 3)) * RB + (cc & 7) * 16);
        } else {
            const int row = line / LPR;
            const int cc = sl ^ (row & 15);
            dma_off[j] = (unsigned)(row * RB + ((line % LPR) * 16 + cc) * 16);
        }
    }
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto issue = [&](int qt, int kp, int buf) {
        const char *qb = qp + (size_t)qt * ROWS * RB + kp * KH * 4;   // wave-uniform
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            char *dst = smem + buf * TILE_BYTES + (wave_u * NI + j) * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(qb + dma_off[j]),
                                             (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        }
    };

    // fragment (g, m): tile row m*32 + l31, chunk 2g + hi.
    //   wide  : byte = row*KH*4 + (chunk>>4)*256 + (((chunk&15) ^ (row&15)) << 4),   row&15 == l31&15
    //   narrow: line = row>>1 = m*16 + (l31>>1), byte = line*256 + ((((l31&1)*8 + chunk) ^ (line&15)) << 4)
    const unsigned rd_base = NARROW ? (unsigned)((l31 >> 1) * 256) : (unsigned)(l31 * KH * 4);
    const unsigned rd_key = NARROW ? (unsigned)(((l31 & 1) * 8 + hi) ^ ((l31 >> 1) & 15)) : (unsigned)(hi ^ (l31 & 15));
    auto rd = [&](int g, int m, unsigned tile) -> float4 {
        unsigned key = rd_key;
        asm volatile("" : "+v"(key));      // keep the 2-VALU address arithmetic here instead of 32 hoisted VGPRs
        const unsigned imm = NARROW ? (unsigned)(m * 4096) : (unsigned)(m * 32 * KH * 4 + (g >> 3) * 256);
        const unsigned off = rd_base + ((key ^ (2u * (g & 7))) << 4) + imm + tile;
        return *reinterpret_cast<const float4 *>(smem + off);
    };

    f32x16 acc[NACC];
#pragma unroll
    for (int m = 0; m < NACC; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.0f;
    float best = INFINITY;
    int bidx = 0x7fffffff;

    if (qt_end > qt_begin) issue(qt_begin, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int buf = 0;
    for (int qt = qt_begin; qt < qt_end; ++qt) {
#pragma unroll
        for (int kp = 0; kp < KP; ++kp) {
            if (!(VAR & 2)) {
                if (kp + 1 < KP) issue(qt, kp + 1, buf ^ 1);
                else if (qt + 1 < qt_end) issue(qt + 1, 0, buf ^ 1);
            }
            const unsigned tile = buf * TILE_BYTES;
            float4 ring[RG][NACC];
#pragma unroll
            for (int g = 0; g < RG - 1; ++g)
#pragma unroll
                for (int m = 0; m < NACC; ++m) ring[g][m] = (VAR & 4) ? make_float4(1.f, 2.f, 3.f, 4.f) : rd(g, m, tile);
#pragma unroll
            for (int g = 0; g < NGT; ++g) {
                // reads of group g+RG-1 go out first (their ring slot was freed by group g-1), then the 4*NACC MFMAs of
                // group g, accumulators round-robin so consecutive MFMAs are independent
                if (!(VAR & 4) && g + RG - 1 < NGT) {
#pragma unroll
                    for (int m = 0; m < NACC; ++m) ring[(g + RG - 1) % RG][m] = rd(g + RG - 1, m, tile);
                }
                const float4 bv = breg[kp * NGT + g];
#pragma unroll
                for (int m = 0; m < NACC; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(ring[g % RG][m].x, bv.x, acc[m], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < NACC; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(ring[g % RG][m].y, bv.y, acc[m], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < NACC; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(ring[g % RG][m].z, bv.z, acc[m], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < NACC; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(ring[g % RG][m].w, bv.w, acc[m], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (kp == KP - 1) {
                if (VAR & 1) {
#pragma unroll
                    for (int m = 0; m < NACC; ++m)
#pragma unroll
                        for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(acc[m][r]));
                    if (qt == qt_end - 1) best = acc[0][0];
                } else {
                    const int qlane = qt * ROWS + 4 * hi;
                    const bool full = (qt + 1) * ROWS <= nq;
                    // dist = fma(-0.5, dot, 0.5) is monotone in dot, so the tile's smallest distance is the image of its
                    // largest dot product: one max tree + one fma decide whether ANY lane of the wave improves.  Only then
                    // (ever rarer as the running minimum settles) is the exact first-index search below executed.
                    float mx = acc[0][0];
#pragma unroll
                    for (int m = 0; m < NACC; ++m)
#pragma unroll
                        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, acc[m][r]);
                    const bool improves = __fmaf_rn(-0.5f, mx, 0.5f) < best;
                    if (!full || __any(improves)) {
#pragma unroll
                        for (int m = 0; m < NACC; ++m)
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int q = qlane + m * 32 + (r & 3) + 8 * (r >> 2);
                                float d = __fmaf_rn(-0.5f, acc[m][r], 0.5f);
                                if (!full) d = (q < nq) ? d : INFINITY;
                                const bool better = d < best;     // strict: first (smallest) query index wins ties
                                best = better ? d : best;
                                bidx = better ? q : bidx;
                            }
                    }
#pragma unroll
                    for (int m = 0; m < NACC; ++m)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[m][r] = 0.0f;
                }
            }
            if (!(VAR & 2)) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                buf ^= 1;
            }
        }
    }
    {
        const float od = __shfl_xor(best, 32);
        const int oi = __shfl_xor(bidx, 32);
        lex_min(best, bidx, od, oi);
    }
    const int a = a0 + wave * 32 + l31;
    if (hi == 0 && a < na && (!row_flag || row_flag[(size_t)p * cap_a + a])) {
        if (S == 1) {
            const size_t o = (size_t)p * cap_a + a;
            min_dist[o] = best;
            argmin[o] = (bidx == 0x7fffffff) ? 0 : bidx;
            valid[o] = (best < thr) ? 1 : 0;
        } else {
            const size_t o = ((size_t)p * S + split) * cap_a + a;
            ws_dist[o] = best;
            ws_idx[o] = bidx;
        }
    }
}

__global__ void match_merge_kernel(const float *__restrict__ ws_dist, const int32_t *__restrict__ ws_idx, int B, int S,
                                   int cap_a, const int32_t *__restrict__ n_a, float thr, float *__restrict__ min_dist,
                                   int32_t *__restrict__ argmin, uint8_t *__restrict__ valid)
{
    const int p = blockIdx.y;
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n_a[p]) return;
    float d = INFINITY;
    int i = 0x7fffffff;
    for (int s = 0; s < S; ++s) {
        const size_t o = ((size_t)p * S + s) * cap_a + a;
        lex_min(d, i, ws_dist[o], ws_idx[o]);
    }
    const size_t o = (size_t)p * cap_a + a;
    min_dist[o] = d;
    argmin[o] = (i == 0x7fffffff) ? 0 : i;
    valid[o] = (d < thr) ? 1 : 0;
}

static int pick_split(int B, int T)
{
    // aim for >= ~16 rounds of 512 resident workgroups so the launch tail is short and (pair, split) units balance
    // over the 8 XCDs
    int S = (8192 + B * T - 1) / (B * T);
    if (S < 1) S = 1;
    if (S > MAX_SPLIT) S = MAX_SPLIT;
    return S;
}

int match_f32_flagged(const float *a_hat, const float *q_hat, int B, int C, int cap_a, int cap_q, const int32_t *n_a,
                      const int32_t *n_q, float threshold, float *min_dist, int32_t *argmin, uint8_t *valid,
                      const int32_t *panel_flag, const uint8_t *row_flag, void *stream)
{
    const int T = cap_a / MT;
    const int groups = ((B + 7) / 8) * 8 * T;      // S = 1: one unit per pair
    hipStream_t st = as_stream(stream);
#define LAUNCH_FLAGGED(CPV)                                                                                               \
    hipLaunchKernelGGL((match_f32_regb_kernel<CPV>), dim3(groups), dim3(MATCH_THREADS), 0, st, a_hat, q_hat, B, cap_a, cap_q, \
                       n_a, n_q, threshold, T, 1, min_dist, argmin, valid, nullptr, nullptr, panel_flag, row_flag)
    if (C == 128) LAUNCH_FLAGGED(128);
    else if (C == 256) LAUNCH_FLAGGED(256);
    else   // wide descriptors: the LDS-staged kernel, one unit per pair
        hipLaunchKernelGGL(match_f32_kernel, dim3(groups), dim3(MATCH_THREADS), 0, st, a_hat, q_hat, B, C, cap_a, cap_q, n_a, n_q,
                           threshold, T, 1, min_dist, argmin, valid, nullptr, nullptr, panel_flag, row_flag);
#undef LAUNCH_FLAGGED
    ORYON_CHECK_LAUNCH();
    return ORYON_OK;
}

}  // namespace oryon

using namespace oryon;

extern "C" size_t oryon_match_workspace_bytes(int B, int cap_a)
{
    if (B <= 0 || cap_a <= 0) return 0;
    const int S = pick_split(B, cap_a / MT);
    return S == 1 ? 0 : (size_t)B * S * cap_a * (sizeof(float) + sizeof(int32_t));
}

extern "C" int oryon_match_f32(const float *a_hat, const float *q_hat, int B, int C, int cap_a, int cap_q,
                               const int32_t *n_a, const int32_t *n_q, float threshold, float *min_dist, int32_t *argmin,
                               uint8_t *valid, void *workspace, size_t workspace_bytes, void *stream)
{
    ORYON_CHECK_ARG(a_hat && q_hat && n_a && n_q && min_dist && argmin && valid);
    ORYON_CHECK_ARG(B >= 0 && C > 0 && C % BK == 0 && cap_a > 0 && cap_a % MT == 0 && cap_q > 0 && cap_q % MT == 0);
    if (B == 0) return ORYON_OK;
    const int T = cap_a / MT;
    const int S = pick_split(B, T);
    float *ws_dist = nullptr;
    int32_t *ws_idx = nullptr;
    if (S > 1) {
        const size_t need = (size_t)B * S * cap_a * (sizeof(float) + sizeof(int32_t));
        if (!workspace || workspace_bytes < need) {
            set_error("oryon_match_f32: workspace too small (%zu < %zu)", workspace_bytes, need);
            return ORYON_ERR_WORKSPACE;
        }
        ws_dist = static_cast<float *>(workspace);
        ws_idx = reinterpret_cast<int32_t *>(ws_dist + (size_t)B * S * cap_a);
    }
    int groups = ((B + 7) / 8) * 8 * T * S;                         // staged kernel: pairs dealt to XCDs
    const int groups_regb = ((B * S + 7) / 8) * 8 * T;                // anchor-stationary kernel: (pair, split) units
    hipStream_t st = as_stream(stream);
#define LAUNCH_REGB(CPV)                                                                                                  \
    hipLaunchKernelGGL((match_f32_regb_kernel<CPV>), dim3(groups_regb), dim3(MATCH_THREADS), 0, st, a_hat, q_hat, B, cap_a, cap_q, \
                       n_a, n_q, threshold, T, S, min_dist, argmin, valid, ws_dist, ws_idx, nullptr, nullptr)
    profile_begin(st, C == 32 ? "match_f32_regb_kernel<32>" : C == 64 ? "match_f32_regb_kernel<64>" : C == 128 ? "match_f32_regb_kernel<128>" :
                      C == 256 ? "match_f32_regb_kernel<256>" : "match_f32_kernel (LDS-staged, wide descriptors)");
    if (C == 32) LAUNCH_REGB(32);
    else if (C == 64) LAUNCH_REGB(64);
    else if (C == 128) LAUNCH_REGB(128);
    else if (C == 256) {
        static const int var = dev_env_int("ORYON_MATCH_VARIANT", 0);
#define LAUNCH_VAR(V) hipLaunchKernelGGL((match_f32_regb_kernel<256, V>), dim3(groups_regb), dim3(MATCH_THREADS), 0, st, a_hat, q_hat, B, cap_a, cap_q, n_a, n_q, threshold, T, S, min_dist, argmin, valid, ws_dist, ws_idx, nullptr, nullptr)
        switch (var) {
            case 1: LAUNCH_VAR(1); break; case 2: LAUNCH_VAR(2); break; case 3: LAUNCH_VAR(3); break; case 4: LAUNCH_VAR(4); break;
            case 5: LAUNCH_VAR(5); break; case 6: LAUNCH_VAR(6); break; case 7: LAUNCH_VAR(7); break; default: LAUNCH_REGB(256);
        }
#undef LAUNCH_VAR
    }
    else   // wide descriptors (e.g. C = 512): operands do not fit the register file, stage both through LDS
        hipLaunchKernelGGL(match_f32_kernel, dim3(groups), dim3(MATCH_THREADS), 0, st, a_hat, q_hat, B, C, cap_a, cap_q, n_a, n_q,
                           threshold, T, S, min_dist, argmin, valid, ws_dist, ws_idx, nullptr, nullptr);
#undef LAUNCH_REGB
    profile_end(st);
    ORYON_CHECK_LAUNCH();
    if (S > 1) {
        hipLaunchKernelGGL(match_merge_kernel, dim3(cap_a / 256 + 1, B), dim3(256), 0, as_stream(stream), ws_dist, ws_idx, B,
                           S, cap_a, n_a, threshold, min_dist, argmin, valid);
        ORYON_CHECK_LAUNCH();
    }
    return ORYON_OK;
}
