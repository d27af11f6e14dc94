// K1s: cosine nearest neighbour with an fp16-MFMA SCREENING pass and an exact fp32 re-scoring pass.
//
// Same contract as K1 (utils/pcd.py:202-205: dist = 0.5*(1-cos), row amin/argmin, < threshold) for every
// anchor row that can possibly be valid, at a fraction of the fp32-MFMA cost:
//
//   pass 0  s16_ij = a16_i . q16_j on v_mfma_f32_32x32x16_f16 (IEEE-half copies of the unit rows, fp32 accumulate);
//           per anchor only max_j s16_ij is kept.
//   pass 1  the same products again; every j with s16_ij >= max_i - MARGIN is appended to anchor i's candidate list.
//           Rows with max_i < (1 - 2*threshold) - DELTA can never pass the threshold: no candidates, valid = 0.
//   pass 2  candidates are re-scored with the canonical fp32 fmaf chain of K1 (bit-exact vs the oracle) and the first
//           index of the smallest distance wins.  Lists that overflow (duplicate-heavy inputs) flag their anchor panel,
//           which is then recomputed by the exact fp32 kernel (match_f32_regb_kernel) - still no host round trip.
//
// Why this is exact.  For unit rows |a16.q16 - a.q| <= DELTA with
//   DELTA = (2^-10 + 2^-22) * sum|a_k q_k|   (two half roundings per product, sum|a_k q_k| <= |a||q| = 1)
//         + 2 * C * 2^-24                     (fp32 accumulation of the MFMA and of the canonical chain)  ~= 1.04e-3  (C <= 512).
// If j* minimises the exact distance then a.q_j* >= max_j a.q_j - 2^-23 (dist is a monotone rounding of the dot), hence
// s16_ij* >= max_j s16_ij - 2*DELTA - 2^-23: every exact minimiser - including every tied one - is in the list, and pass 2
// returns exactly what the full fp32 scan returns.  MARGIN = 2.2e-3 > 2*DELTA + 2^-23.
// Subnormal half inputs (|x| < 2^-14, common in unit rows of 256+ channels) are honoured by v_mfma_f32_32x32x16_f16 on gfx950
// (tools/probe_mfma_f16_denorm.hip, measured), and K0's float->half conversion keeps them, so they round with an absolute
// error <= 2^-25 each: <= 2 * 2^-25 * sqrt(C) ~= 1.3e-6 in the dot product, inside the slack of SCREEN_DELTA.
#include <hip/hip_fp16.h>
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "match_common.h"

namespace oryon {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

constexpr float SCREEN_DELTA = 1.05e-3f;
constexpr float SCREEN_MARGIN = 2.2e-3f;
constexpr int SCREEN_CAP = 64;          // candidate slots per anchor
constexpr int MT16 = 256;               // anchors per workgroup (4 waves x 2 blocks of 32)
constexpr int screen_tile_bytes(int CP) { return CP == 512 ? 65536 : 32768; }

// Out-of-line candidate append (pass 1 slow path, taken by a few % of the tiles): keeping it out of the kernel body
// keeps the hot loop's register allocation identical to pass 0.  vals: this lane's NV scores of one anchor column.
template <int NV>
__device__ __noinline__ void emit_candidates(const float *vals, float thr, int qlane, size_t arow, int32_t *cnt, int32_t *cand)
{
    for (int e = 0; e < NV; ++e)
        if (vals[e] >= thr) {
            const int r = e & 15, qb = e >> 4;
            const int sl = atomicAdd(&cnt[arow], 1);
            if (sl < SCREEN_CAP) cand[arow * SCREEN_CAP + sl] = qlane + qb * 32 + (r & 3) + 8 * (r >> 2);
        }
}

template <int CP, int MODE, int VAR = 0>   // VAR != 0: timing ablations only (ORYON_MATCH16_VARIANT)
__global__ __launch_bounds__(256, CP == 512 ? 1 : 2) void match_f16_screen_kernel(
    const __half *__restrict__ a16, const __half *__restrict__ q16, int B, int cap_a, int cap_q,
    const int32_t *__restrict__ n_a, const int32_t *__restrict__ n_q, int T, int S, float valid_cut,
    float *__restrict__ ws_max /*[B,S_thr|S,cap_a]*/, int32_t *__restrict__ cnt /*[B,cap_a]*/, int32_t *__restrict__ cand,
    int S_thr, const int32_t *__restrict__ row_map, int32_t *__restrict__ ws_i1, float *__restrict__ ws_m2)
{
    constexpr int RB = CP * 2;                   // row bytes
    constexpr int TILE_BYTES = screen_tile_bytes(CP);   // 32 KB; 64 KB at C=512 (one workgroup per CU, 512 registers per lane)
    constexpr int ROWS = TILE_BYTES / RB;        // query rows per LDS tile (64 at C=256 and C=512)
    constexpr int NQB = ROWS / 32;               // query blocks per tile
    constexpr int NAB = 2;                       // anchor blocks per wave
    constexpr int NKS = CP / 16;                 // MFMA k-steps
    constexpr int NI = TILE_BYTES / 4096;        // 1 KB DMA instructions per wave and tile
    constexpr int LPR = RB / 256;                // 256-byte lines per row
    static_assert(NQB * NKS * 1024 == TILE_BYTES && NQB >= 1 && LPR >= 1, "tile geometry");
    char *smem;
    if constexpr (2 * TILE_BYTES > 65536) {      // beyond the static LDS limit: dynamic, sized by the launcher
        extern __shared__ __attribute__((aligned(256))) char smem_dyn[];
        smem = smem_dyn;
    } else {
        __shared__ __attribute__((aligned(256))) char smem_st[2 * TILE_BYTES];
        smem = smem_st;
    }

    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int unit = (slot / T) * 8 + xcd;
    if (unit >= B * S) return;
    const int panel = slot % T;
    const int p = unit / S, split = unit % S;
    const int na = n_a[p], nq = n_q[p];
    const int a0 = panel * MT16;
    if (a0 >= na) return;
    const int nqt = (nq + ROWS - 1) / ROWS;
    const int qt_per = (nqt + S - 1) / S;
    const int qt_begin = split * qt_per;
    const int qt_end = (qt_begin + qt_per < nqt) ? qt_begin + qt_per : nqt;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, hi = lane >> 5;
    const char *qp = reinterpret_cast<const char *>(q16 + (size_t)p * cap_q * CP);

    // stationary B operand: anchors a0 + wave*64 + ab*32 + l31, k-step s -> halves 16s + 8hi .. +7
    half8 breg[NAB][NKS];
#pragma unroll
    for (int ab = 0; ab < NAB; ++ab) {
        const __half *arow = a16 + ((size_t)p * cap_a + a0 + wave * 64 + ab * 32 + l31) * CP + 8 * hi;
#pragma unroll
        for (int s = 0; s < NKS; ++s) breg[ab][s] = *reinterpret_cast<const half8 *>(arow + 16 * s);
    }

    unsigned dma_off[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int line = (wave * NI + j) * 4 + (lane >> 4), sl = lane & 15;
        const int row = line / LPR;
        const int cc = sl ^ (row & 15);
        dma_off[j] = (unsigned)(row * RB + ((line % LPR) * 16 + cc) * 16);
    }
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto issue = [&](int qt, int buf) {
        const char *qb = qp + (size_t)qt * TILE_BYTES;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            char *dst = smem + buf * TILE_BYTES + (wave_u * NI + j) * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(qb + dma_off[j]),
                                             (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        }
    };
    // swizzled operand addresses: 8 per-lane offsets in registers, the rest immediates / one add per tile (VALU and MFMA issue
    // serialise on a SIMD: every VALU instruction taken out of the loop is MFMA time)
    unsigned koff[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) koff[c] = (unsigned)(l31 * RB) + ((((unsigned)(hi ^ (l31 & 15))) ^ (2u * c)) << 4);
    auto rd = [&](int s, int qb, unsigned tile) -> half8 {
        return *reinterpret_cast<const half8 *>(smem + koff[s & 7] + tile + (unsigned)(qb * 32 * RB + (s >> 3) * 256));
    };

    f32x16 acc[NQB][NAB];
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
        for (int ab = 0; ab < NAB; ++ab)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[qb][ab][r] = 0.0f;

    float runmax[NAB], thr[NAB], run2[NAB];
    int runidx[NAB];
#pragma unroll
    for (int ab = 0; ab < NAB; ++ab) {
        runmax[ab] = -INFINITY;
        run2[ab] = -INFINITY;
        runidx[ab] = 0;
        thr[ab] = INFINITY;
        if (MODE == 1) {
            const int a = a0 + wave * 64 + ab * 32 + l31;
            float m = -INFINITY;
            for (int s = 0; s < S_thr; ++s) m = fmaxf(m, ws_max[((size_t)p * S_thr + s) * cap_a + a]);
            thr[ab] = (a < na && m >= valid_cut) ? m - SCREEN_MARGIN : INFINITY;
        }
    }

    if (qt_end > qt_begin) issue(qt_begin, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int buf = 0;
    for (int qt = qt_begin; qt < qt_end; ++qt) {
        if (!(VAR & 2) && qt + 1 < qt_end) issue(qt + 1, buf ^ 1);
        const unsigned tile = buf * TILE_BYTES;
        half8 ring[2][NQB];
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) ring[0][qb] = rd(0, qb, tile);
#pragma unroll
        for (int s = 0; s < NKS; ++s) {
            if (s + 1 < NKS) {
#pragma unroll
                for (int qb = 0; qb < NQB; ++qb) ring[(s + 1) & 1][qb] = (VAR & 4) ? ring[s & 1][qb] : rd(s + 1, qb, tile);
            }
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
                for (int ab = 0; ab < NAB; ++ab)
                    acc[qb][ab] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[s & 1][qb], breg[ab][s], acc[qb][ab], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (VAR & 1) {
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
                for (int ab = 0; ab < NAB; ++ab)
#pragma unroll
                    for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(acc[qb][ab][r]));
            if (qt == qt_end - 1) runmax[0] = acc[0][0][0];
            if (!(VAR & 2)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); buf ^= 1; }
            continue;
        }
        // epilogue: lane owns anchor column (ab, l31); rows of the C/D block are queries.
        // Rows >= n_q of the last tile need no masking: K0 zero-fills them, so they score exactly 0, which can neither
        // reach valid_cut (> 0 for thresholds < 0.5) nor a candidate threshold; pass 2 ignores indices >= n_q anyway.
        const int qlane = qt * ROWS + 4 * hi;
#pragma unroll
        for (int ab = 0; ab < NAB; ++ab) {
            float m = -INFINITY;
            if (MODE != 2) {
#pragma unroll
                for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) m = fmaxf(m, acc[qb][ab][r]);
            }
            if (MODE == 0) {
                runmax[ab] = fmaxf(runmax[ab], m);
            } else if (MODE == 2) {
                // per (tile, query block) SLICE maxima only: running (m1, slice id of m1, m2 = best slice maximum other than
                // m1's slice).  16 rows of one lane half form a slice; what happens INSIDE the winning slice is resolved by
                // match_decide_kernel, which re-scores just those 16 rows.
#pragma unroll
                for (int qb = 0; qb < NQB; ++qb) {
                    float x = acc[qb][ab][0];
#pragma unroll
                    for (int r = 1; r < 16; ++r) x = fmaxf(x, acc[qb][ab][r]);
                    const bool improved = x > runmax[ab];
                    run2[ab] = fmaxf(fminf(runmax[ab], x), run2[ab]);
                    runmax[ab] = fmaxf(runmax[ab], x);
                    runidx[ab] = improved ? ((qt * NQB + qb) * 2 + hi) : runidx[ab];
                }
            } else if (!(VAR & 8) && __any(m >= thr[ab])) {
                float vals[NQB * 16];
#pragma unroll
                for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) vals[qb * 16 + r] = acc[qb][ab][r];
                const int acol = a0 + wave * 64 + ab * 32 + l31;
                const int aout = (row_map && acol < na) ? row_map[(size_t)p * cap_a + acol] : acol;
                emit_candidates<NQB * 16>(vals, thr[ab], qlane, (size_t)p * cap_a + aout, cnt, cand);
            }
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[qb][ab][r] = 0.0f;
        }
        if (!(VAR & 2)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            buf ^= 1;
        }
    }
    if (MODE == 0) {
#pragma unroll
        for (int ab = 0; ab < NAB; ++ab) {
            const float m = fmaxf(runmax[ab], __shfl_xor(runmax[ab], 32));
            const int a = a0 + wave * 64 + ab * 32 + l31;
            if (hi == 0) ws_max[((size_t)p * S + split) * cap_a + a] = m;
        }
    }
    if (MODE == 2) {
#pragma unroll
        for (int ab = 0; ab < NAB; ++ab) {
            const float om1 = __shfl_xor(runmax[ab], 32), om2 = __shfl_xor(run2[ab], 32);
            const int oi1 = __shfl_xor(runidx[ab], 32);
            const float m1 = fmaxf(runmax[ab], om1);
            const float m2 = fmaxf(fminf(runmax[ab], om1), fmaxf(run2[ab], om2));
            const int i1 = (om1 > runmax[ab]) ? oi1 : runidx[ab];
            const int a = a0 + wave * 64 + ab * 32 + l31;
            if (hi == 0) {
                const size_t o = ((size_t)p * S + split) * cap_a + a;
                ws_max[o] = m1;
                ws_i1[o] = i1;
                ws_m2[o] = m2;
            }
        }
    }
}

// Single-pass strategy, step 2 (one wave per anchor): merge the per-split (m1, slice of m1, m2) triples and decide
//   m1 < valid_cut        -> can never be valid                                        (no candidates)
//   m1 - m2 > MARGIN      -> every index within MARGIN of m1 lies in m1's 16-row slice: those 16 rows are re-scored here
//                            from the fp16 rows (fp32 accumulate) and the ones within MARGIN (+ recomputation slack) become
//                            the candidates
//   otherwise             -> ambiguous: appended to the pair's list; a second, compacted screening pass collects every
//                            index within MARGIN for these anchors only.
template <int ROWS_TILE>
__global__ __launch_bounds__(256) void match_decide_kernel(const __half *__restrict__ a16, const __half *__restrict__ q16, int Cp,
                                                            int cap_a, int cap_q, const int32_t *__restrict__ n_a,
                                                            const int32_t *__restrict__ n_q, int S, float valid_cut,
                                                            const float *__restrict__ ws_m1, const int32_t *__restrict__ ws_i1,
                                                            const float *__restrict__ ws_m2, float *__restrict__ m_final,
                                                            int32_t *__restrict__ cnt, int32_t *__restrict__ cand,
                                                            int32_t *__restrict__ n_amb, int32_t *__restrict__ amb_idx,
                                                            const float *__restrict__ a_scale8, const float *__restrict__ eps_a8,
                                                            const float *__restrict__ eps_q8, float cut0, float sqrt_c, float c_true,
                                                            const int8_t *__restrict__ a8, const int8_t *__restrict__ q8,
                                                            const float *__restrict__ q_scale8)
{
    // a_scale8 != nullptr: the (m1, slice, m2) triples come from the INT8 screening pass (K1s8) in units of 2^-E_a per anchor
    // slice; margin and validity cut then follow the per-anchor int8 bound DELTA8 (see the header of the int8 kernel).
    constexpr int NQB = ROWS_TILE / 32;
    // 16 anchors per wave: an empty launch (the fp16 stage behind K1s8 usually has nothing to do) costs 5 k workgroups, not 80 k
    const int p = blockIdx.y, lane = threadIdx.x & 63;
    const int n_anchors = n_a[p];
    const float valid_cut_in = valid_cut;
    auto one_anchor = [&](const int a) {
    float valid_cut = valid_cut_in;
    const size_t arow = (size_t)p * cap_a + a;
    float m1 = -INFINITY, m2 = -INFINITY;
    int sid = 0;
    for (int s = 0; s < S; ++s) {
        const size_t o = ((size_t)p * S + s) * cap_a + a;
        const float x1 = ws_m1[o], x2 = ws_m2[o];
        m2 = fmaxf(fminf(m1, x1), fmaxf(m2, x2));
        if (x1 > m1) { m1 = x1; sid = ws_i1[o]; }
    }
    float margin = SCREEN_MARGIN, sa = 1.0f;
    if (a_scale8) {
        sa = a_scale8[(size_t)p * (cap_a / 16) + (a >> 5) * 2 + ((a >> 2) & 1)];   // 2^-E of the anchor's slice
        m1 *= sa;                                             // exact: power of two
        m2 *= sa;
        // K0's rounding: |q 2^-E - x^| <= 2^-(E+1) * (1 + 4.6e-5)  (gather8.hip: q = rint(x * RN(2^E / d)))
        const float ea = 0.50003f * sa, eq = 1.00006f * eps_q8[p];
        (void)eps_a8;
        // |s8 - a^.q^| <= ea*|q^|_1 + eq*|a^|_1 + C*ea*eq <= (ea + eq)*sqrt(C) + C*ea*eq   (+ fp32 accumulation slack of the exact scan)
        const float delta = (ea + eq) * sqrt_c + c_true * ea * eq + 4e-5f;
        margin = 2.0f * delta + 2e-7f;
        valid_cut = cut0 - delta - 1e-6f;
        if (!(delta < 0.2f)) { margin = INFINITY; valid_cut = -INFINITY; }     // degenerate scales: leave it to the fp16 pass
    }
    if (lane == 0) m_final[arow] = m1;
    if (!(m1 >= valid_cut)) {
        if (a_scale8 && lane == 0) cnt[arow] = -1;            // int8 mode: "cannot be valid" is carried by the count
        return;
    }
    if (!(m1 - m2 > margin)) {
        if (lane == 0) {
            const int sl = atomicAdd(&n_amb[p], 1);
            amb_idx[(size_t)p * cap_a + sl] = a;
        }
        return;
    }
    // slice sid = (tile*NQB + qb)*2 + half: rows tile*ROWS + qb*32 + (r&3) + 8*(r>>2) + 4*half, r = 0..15
    const int half = sid & 1, qb = (sid >> 1) % NQB, tile = (sid >> 1) / NQB;
    const int r = lane >> 2, seg = lane & 3;                   // 4 lanes per row, each a quarter of K
    const int q = tile * ROWS_TILE + qb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
    const int nq = n_q[p];
    if (a_scale8) {
        // int8 mode: the slice is re-scored from the int8 rows (the very integers the screening pass accumulated, so the slice
        // maximum reproduces m1 exactly); every row within the int8 margin of it goes to the exact fp32 re-scoring
        int idot = 0;
        if (q < nq) {
            const uint4 *ar = reinterpret_cast<const uint4 *>(a8 + arow * Cp) + seg * (Cp / 64);
            const uint4 *qr = reinterpret_cast<const uint4 *>(q8 + ((size_t)p * cap_q + q) * Cp) + seg * (Cp / 64);
            for (int i0 = 0; i0 < Cp / 64; i0 += 4) {
                uint4 av[4], qv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { av[u] = ar[i0 + u]; qv[u] = qr[i0 + u]; }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    idot = __builtin_amdgcn_sdot4((int)av[u].x, (int)qv[u].x, idot, false);
                    idot = __builtin_amdgcn_sdot4((int)av[u].y, (int)qv[u].y, idot, false);
                    idot = __builtin_amdgcn_sdot4((int)av[u].z, (int)qv[u].z, idot, false);
                    idot = __builtin_amdgcn_sdot4((int)av[u].w, (int)qv[u].w, idot, false);
                }
            }
        }
        idot += __shfl_xor(idot, 1);
        idot += __shfl_xor(idot, 2);
        const float s8 = (float)idot * q_scale8[(size_t)p * (cap_q / 16) + (sid >> 1) * 2 + half] * sa;
        const bool hit8 = (seg == 0) && (q < nq) && (s8 >= m1 - margin);
        const unsigned long long b8 = __ballot(hit8);
        if (hit8) cand[arow * SCREEN_CAP + __popcll(b8 & ((1ull << lane) - 1ull))] = q;
        if (lane == 0) cnt[arow] = __popcll(b8);
        return;
    }
    float sdot = 0.0f;
    if (q < nq) {
        const uint4 *ar = reinterpret_cast<const uint4 *>(a16 + arow * Cp) + seg * (Cp / 32);
        const uint4 *qr = reinterpret_cast<const uint4 *>(q16 + ((size_t)p * cap_q + q) * Cp) + seg * (Cp / 32);
        for (int i0 = 0; i0 < Cp / 32; i0 += 4) {
            uint4 av[4], qv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { av[u] = ar[i0 + u]; qv[u] = qr[i0 + u]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const __half2 *ah = reinterpret_cast<const __half2 *>(&av[u]), *qh = reinterpret_cast<const __half2 *>(&qv[u]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float2 x = __half22float2(ah[e]), y = __half22float2(qh[e]);
                    sdot = fmaf(x.x, y.x, sdot);
                    sdot = fmaf(x.y, y.y, sdot);
                }
            }
        }
    }
    sdot += __shfl_xor(sdot, 1);
    sdot += __shfl_xor(sdot, 2);
    const bool hit = (seg == 0) && (q < nq) && (sdot >= m1 - SCREEN_MARGIN - 4e-5f);
    const unsigned long long b = __ballot(hit);
    if (hit) cand[arow * SCREEN_CAP + __popcll(b & ((1ull << lane) - 1ull))] = q;
    if (lane == 0) cnt[arow] = __popcll(b);
    };
    for (int g = 0; g < 16; ++g) {
        const int a = (blockIdx.x * 16 + g) * 4 + (threadIdx.x >> 6);
        if (a >= n_anchors) break;
        one_anchor(a);
    }
}

// gather the fp16 rows (and thresholds) of the ambiguous anchors into a dense panel layout for the second pass
__global__ __launch_bounds__(256) void match_compact_kernel(const __half *__restrict__ a16, int Cp, int cap_a,
                                                             const int32_t *__restrict__ n_amb, const int32_t *__restrict__ amb_idx,
                                                             const float *__restrict__ m_final, __half *__restrict__ a16c,
                                                             float *__restrict__ amb_max)
{
    const int p = blockIdx.y, lane = threadIdx.x & 63;
    const int n_sl = n_amb[p];
    for (int g = 0; g < 16; ++g) {
    const int sl = (blockIdx.x * 16 + g) * 4 + (threadIdx.x >> 6);
    if (sl >= n_sl) break;
    const int a = amb_idx[(size_t)p * cap_a + sl];
    const uint4 *src = reinterpret_cast<const uint4 *>(a16 + ((size_t)p * cap_a + a) * Cp);
    uint4 *dst = reinterpret_cast<uint4 *>(a16c + ((size_t)p * cap_a + sl) * Cp);
    for (int i = lane; i < Cp / 8; i += 64) dst[i] = src[i];
    if (lane == 0) amb_max[(size_t)p * cap_a + sl] = m_final[(size_t)p * cap_a + a];
    }
}

// pass 2: L lanes per anchor row, candidates strided over them; canonical fp32 chain on the k-permuted fp32 rows.  Almost
// every anchor has ONE candidate and the chain is a serial 256-step fmaf per (anchor, candidate): few lanes per anchor keep
// more lanes of a wave busy.
template <int L>
__global__ __launch_bounds__(256) void match_rescore_kernel(const float *__restrict__ a_hat, const float *__restrict__ q_hat,
                                                             int Cp, int cap_a, int cap_q, const int32_t *__restrict__ n_a,
                                                             const int32_t *__restrict__ n_q, int S, float thr, float valid_cut,
                                                             const float *__restrict__ ws_max, const float *__restrict__ m_final,
                                                             const int32_t *__restrict__ cnt,
                                                             const int32_t *__restrict__ cand, float *__restrict__ min_dist,
                                                             int32_t *__restrict__ argmin, uint8_t *__restrict__ valid,
                                                             uint8_t *__restrict__ row_flag, int32_t *__restrict__ panel_flag)
{
    const int p = blockIdx.y;
    const int a = blockIdx.x * (256 / L) + (threadIdx.x / L), sub = threadIdx.x % L;
    const bool live = a < n_a[p];
    const size_t arow = (size_t)p * cap_a + (live ? a : 0);
    float m16 = -INFINITY;
    if (live) {
        if (m_final) m16 = m_final[arow];
        else
            for (int s = 0; s < S; ++s) m16 = fmaxf(m16, ws_max[((size_t)p * S + s) * cap_a + a]);
    }
    const int c_raw = live ? cnt[arow] : 0;
    const bool possible = live && (m16 >= valid_cut) && c_raw >= 0;      // count -1: ruled out by the int8 stage
    const int c = possible ? c_raw : 0;
    const bool overflow = c > SCREEN_CAP;
    float d = INFINITY;
    int j = 0x7fffffff;
    if (!overflow) {
        const float *ar = a_hat + arow * Cp;
        const int nq_p = n_q[p];
        for (int ci = sub; ci < c; ci += L) {
            const int jj = cand[arow * SCREEN_CAP + ci];
            if (jj >= nq_p) continue;                     // zero-padded query rows can never be the answer
            const float *qr = q_hat + ((size_t)p * cap_q + jj) * Cp;
            float dot = 0.0f;
            for (int g = 0; g < Cp; g += 8) {
                const float4 a0 = *reinterpret_cast<const float4 *>(ar + g), a1 = *reinterpret_cast<const float4 *>(ar + g + 4);
                const float4 q0 = *reinterpret_cast<const float4 *>(qr + g), q1 = *reinterpret_cast<const float4 *>(qr + g + 4);
                // positions 0..3 hold k = 8g+0,2,4,6 and 4..7 hold k = 8g+1,3,5,7: accumulate in natural k order
                dot = __fmaf_rn(a0.x, q0.x, dot); dot = __fmaf_rn(a1.x, q1.x, dot);
                dot = __fmaf_rn(a0.y, q0.y, dot); dot = __fmaf_rn(a1.y, q1.y, dot);
                dot = __fmaf_rn(a0.z, q0.z, dot); dot = __fmaf_rn(a1.z, q1.z, dot);
                dot = __fmaf_rn(a0.w, q0.w, dot); dot = __fmaf_rn(a1.w, q1.w, dot);
            }
            lex_min(d, j, __fmaf_rn(-0.5f, dot, 0.5f), jj);
        }
    }
#pragma unroll
    for (int off = L / 2; off > 0; off >>= 1) {
        const float od = __shfl_xor(d, off);
        const int oj = __shfl_xor(j, off);
        lex_min(d, j, od, oj);
    }
    if (!live || sub != 0) return;
    if (!possible) {                    // cannot reach the threshold: report the screening estimate, valid = 0
        min_dist[arow] = __fmaf_rn(-0.5f, m16, 0.5f);
        argmin[arow] = 0;
        valid[arow] = 0;
    } else if (overflow) {              // list overflow: the exact fp32 kernel recomputes this anchor's panel
        row_flag[arow] = 1;
        panel_flag[(size_t)p * (cap_a / ORYON_MATCH_TILE) + a / ORYON_MATCH_TILE] = 1;
    } else {
        min_dist[arow] = d;
        argmin[arow] = j;
        valid[arow] = (d < thr) ? 1 : 0;
    }
}

// pass 2 for the K1s8 path fed by K0v3 (gather8.hip), which writes NO fp32 copy of the query rows: a candidate's canonical unit
// values are recovered on the fly as x_k / d from the raw descriptor map and the row norm d K0 stored (the same IEEE division K0's
// fp32 rows come from, so the chain below is bit for bit the one match_rescore_kernel runs on materialised rows).  Anchor rows are
// the materialised, k-permuted fp32 rows (5000 per pair).  Candidates of neighbouring anchors are neighbouring query pixels on real
// (and synthetic) rigid pairs, so the strided 4-byte reads of an NCHW map share their 64-byte sectors across a wave.
template <int L, bool NHWC>
__global__ __launch_bounds__(256) void match_rescore_raw_kernel(
    const float *__restrict__ a_hat, const float *__restrict__ feat_q, int C_true, int HW, const int32_t *__restrict__ roi_q,
    int roi_stride, const float *__restrict__ norm_q, int Cp, int cap_a, int cap_q, const int32_t *__restrict__ n_a,
    const int32_t *__restrict__ n_q, float thr, const float *__restrict__ m_final, const int32_t *__restrict__ cnt,
    const int32_t *__restrict__ cand, float *__restrict__ min_dist, int32_t *__restrict__ argmin, uint8_t *__restrict__ valid,
    uint8_t *__restrict__ row_flag, int32_t *__restrict__ panel_flag, int32_t *__restrict__ need_f32, int round_f16)
{
    const int p = blockIdx.y;
    const int a = blockIdx.x * (256 / L) + (threadIdx.x / L), sub = threadIdx.x % L;
    const bool live = a < n_a[p];
    const size_t arow = (size_t)p * cap_a + (live ? a : 0);
    const int c_raw = live ? cnt[arow] : 0;
    const bool possible = live && c_raw >= 0;                 // count -1: ruled out by the int8 stage
    const int c = possible ? c_raw : 0;
    const bool overflow = c > SCREEN_CAP;
    float d = INFINITY;
    int j = 0x7fffffff;
    if (!overflow) {
        const float *ar = a_hat + arow * Cp;
        const int nq_p = n_q[p];
        const float *fq = feat_q + (size_t)p * C_true * HW;
        for (int ci = sub; ci < c; ci += L) {
            const int jj = cand[arow * SCREEN_CAP + ci];
            if (jj >= nq_p) continue;
            const int pix = roi_q[(size_t)p * roi_stride + jj];
            const float dq = norm_q[(size_t)p * cap_q + jj];
            float dot = 0.0f;
            for (int g = 0; g < C_true; g += 8) {
                // anchor positions 0..3 of a group hold k = 8g+0,2,4,6 and 4..7 hold k = 8g+1,3,5,7
                const float4 a0 = *reinterpret_cast<const float4 *>(ar + g), a1 = *reinterpret_cast<const float4 *>(ar + g + 4);
                float x[8];
                if constexpr (NHWC) {
                    if (g + 8 <= C_true) {
                        const float4 q0 = *reinterpret_cast<const float4 *>(fq + (size_t)pix * C_true + g);
                        const float4 q1 = *reinterpret_cast<const float4 *>(fq + (size_t)pix * C_true + g + 4);
                        x[0] = q0.x; x[1] = q0.y; x[2] = q0.z; x[3] = q0.w; x[4] = q1.x; x[5] = q1.y; x[6] = q1.z; x[7] = q1.w;
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) x[e] = g + e < C_true ? fq[(size_t)pix * C_true + g + e] : 0.0f;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] = g + e < C_true ? fq[(size_t)(g + e) * HW + pix] : 0.0f;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = __fdiv_rn(round_f16 ? __half2float(__float2half_rn(x[e])) : x[e], dq);
                dot = __fmaf_rn(a0.x, x[0], dot); dot = __fmaf_rn(a1.x, x[1], dot);
                dot = __fmaf_rn(a0.y, x[2], dot); dot = __fmaf_rn(a1.y, x[3], dot);
                dot = __fmaf_rn(a0.z, x[4], dot); dot = __fmaf_rn(a1.z, x[5], dot);
                dot = __fmaf_rn(a0.w, x[6], dot); dot = __fmaf_rn(a1.w, x[7], dot);
            }
            // channels C_true .. Cp-1 are zero on both sides: fma(0, 0, dot) leaves dot unchanged, nothing to add
            lex_min(d, j, __fmaf_rn(-0.5f, dot, 0.5f), jj);
        }
    }
#pragma unroll
    for (int off = L / 2; off > 0; off >>= 1) {
        const float od = __shfl_xor(d, off);
        const int oj = __shfl_xor(j, off);
        lex_min(d, j, od, oj);
    }
    if (!live || sub != 0) return;
    if (!possible) {
        min_dist[arow] = __fmaf_rn(-0.5f, m_final[arow], 0.5f);
        argmin[arow] = 0;
        valid[arow] = 0;
    } else if (overflow) {
        row_flag[arow] = 1;
        panel_flag[(size_t)p * (cap_a / ORYON_MATCH_TILE) + a / ORYON_MATCH_TILE] = 1;
        need_f32[p] = 1;
    } else {
        min_dist[arow] = d;
        argmin[arow] = j;
        valid[arow] = (d < thr) ? 1 : 0;
    }
}

// pairs that need materialised fp32 query rows: undecided anchors (fp16 stage) or an overflowed candidate list (exact panel scan)
__global__ void match_need_f32_kernel(int B, const int32_t *__restrict__ n_amb, int32_t *__restrict__ need_f32)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < B && n_amb[p] > 0) need_f32[p] = 1;
}

// ------------------------------------------------------------------------------------------------ K1s8: int8 pre-screen
// The same single-pass (m1, slice, m2) screening as MODE 2 above on v_mfma_i32_32x32x32_i8 - twice the fp16 matrix rate on
// gfx950 (3.4 POP/s sustained vs 1.7 PFLOP/s, tools/probe_mfma_rates.hip), half the operand bytes, 128 query rows per 32 KB
// tile.  K0 writes q = rint(x^ * 2^E) with one exponent per 16-row slice, so inside a slice the integer maximum IS the score
// maximum and the epilogue costs what the fp16 one costs: integer max tree, one convert, one multiply by the slice's 2^-E.
// Accumulation is exact (|sum| <= 512 * 127^2 < 2^24).
//
// Bound: with ea = 2^-(E_a+1), eq = max over the pair's query slices of 2^-(E_q+1):
//   |s8_ij - a^_i.q^_j| <= ea*|q^_j|_1 + eq*|a^_i|_1 + C*ea*eq <= (ea + eq)*sqrt(C) + C*ea*eq =: DELTA8_i   (unit rows: |x|_1 <= sqrt(C)).
// match_decide_kernel then decides per anchor exactly as for fp16, with DELTA8_i in place of DELTA:
//   m1 < (1-2thr) - DELTA8      -> cannot be valid
//   m1 - m2 > 2*DELTA8          -> every exact minimiser lies in m1's slice: its 16 rows are re-scored in fp16, candidates -> exact
//                                  fp32 re-scoring (unchanged)
//   otherwise                   -> the anchor is handed to the fp16 screening (compacted set, complete K1s pipeline), so the int8
//                                  stage can only ever lose time, never exactness.
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
constexpr int screen8_tile_bytes(int CP) { return CP * 128; }     // 128 query rows per tile

template <int CP>
__global__ __launch_bounds__(256, CP == 512 ? 1 : 2) void match_i8_screen_kernel(
    const int8_t *__restrict__ a8, const int8_t *__restrict__ q8, const float *__restrict__ q_scale, int B, int cap_a, int cap_q,
    const int32_t *__restrict__ n_a, const int32_t *__restrict__ n_q, int T, int S, float *__restrict__ ws_max,
    int32_t *__restrict__ ws_i1, float *__restrict__ ws_m2)
{
    constexpr int RB = CP;                       // row bytes
    constexpr int TILE_BYTES = screen8_tile_bytes(CP);
    constexpr int ROWS = 128, NQB = 4, NAB = 2;
    constexpr int NKS = CP / 32;                 // MFMA k-steps
    constexpr int NI = TILE_BYTES / 4096;
    constexpr int LPR = RB / 256;
    static_assert(LPR >= 1, "rows are whole 256-byte lines");
    char *smem;
    if constexpr (2 * TILE_BYTES > 65536) {
        extern __shared__ __attribute__((aligned(256))) char smem_dyn8[];
        smem = smem_dyn8;
    } else {
        __shared__ __attribute__((aligned(256))) char smem_st8[2 * TILE_BYTES];
        smem = smem_st8;
    }
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int unit = (slot / T) * 8 + xcd;
    if (unit >= B * S) return;
    const int panel = slot % T;
    const int p = unit / S, split = unit % S;
    const int na = n_a[p], nq = n_q[p];
    const int a0 = panel * MT16;
    if (a0 >= na) return;
    const int nqt = (nq + ROWS - 1) / ROWS;
    const int qt_per = (nqt + S - 1) / S;
    const int qt_begin = split * qt_per;
    const int qt_end = (qt_begin + qt_per < nqt) ? qt_begin + qt_per : nqt;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, hi = lane >> 5;
    const char *qp = reinterpret_cast<const char *>(q8) + (size_t)p * cap_q * RB;
    const float2 *qs = reinterpret_cast<const float2 *>(q_scale + (size_t)p * (cap_q / 16));    // (h = 0, h = 1) per 32-row block

    // stationary B operand: anchors a0 + wave*64 + ab*32 + l31, k-step s -> bytes 32s + 16hi .. +15
    i32x4 breg[NAB][NKS];
#pragma unroll
    for (int ab = 0; ab < NAB; ++ab) {
        const char *arow = reinterpret_cast<const char *>(a8) + ((size_t)p * cap_a + a0 + wave * 64 + ab * 32 + l31) * RB + 16 * hi;
#pragma unroll
        for (int s = 0; s < NKS; ++s) breg[ab][s] = *reinterpret_cast<const i32x4 *>(arow + 32 * s);
    }
    unsigned dma_off[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int line = (wave * NI + j) * 4 + (lane >> 4), sl = lane & 15;
        const int row = line / LPR;
        const int cc = sl ^ (row & 15);
        dma_off[j] = (unsigned)(row * RB + ((line % LPR) * 16 + cc) * 16);
    }
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto issue = [&](int qt, int buf) {
        const char *qb = qp + (size_t)qt * TILE_BYTES;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            char *dst = smem + buf * TILE_BYTES + (wave_u * NI + j) * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(qb + dma_off[j]),
                                             (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        }
    };
    // swizzled operand addresses: 8 per-lane offsets (one per 16-byte chunk pair of a 256-byte line) live in registers, everything
    // else (query block, line, tile) is an immediate or one add per tile - VALU and MFMA issue serialise on a SIMD of this chip
    // (tools/probe_mfma_valu_overlap.hip), so every VALU instruction taken out of the loop is MFMA time
    unsigned koff[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) koff[c] = (unsigned)(l31 * RB) + ((((unsigned)(hi ^ (l31 & 15))) ^ (2u * c)) << 4);
    auto rd = [&](int s, int qb, unsigned tile) -> i32x4 {
        return *reinterpret_cast<const i32x4 *>(smem + koff[s & 7] + tile + (unsigned)(qb * 32 * RB + (s >> 3) * 256));
    };

    i32x16 acc[NQB][NAB];
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
        for (int ab = 0; ab < NAB; ++ab)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[qb][ab][r] = 0;
    float runmax[NAB], run2[NAB];
    int runidx[NAB];
#pragma unroll
    for (int ab = 0; ab < NAB; ++ab) { runmax[ab] = -INFINITY; run2[ab] = -INFINITY; runidx[ab] = 0; }

    if (qt_end > qt_begin) issue(qt_begin, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int buf = 0;
    for (int qt = qt_begin; qt < qt_end; ++qt) {
        if (qt + 1 < qt_end) issue(qt + 1, buf ^ 1);
        float2 sc2[NQB];
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) sc2[qb] = qs[qt * NQB + qb];
        const unsigned tile = buf * TILE_BYTES;
        i32x4 ring[2][NQB];
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) ring[0][qb] = rd(0, qb, tile);
#pragma unroll
        for (int s = 0; s < NKS; ++s) {
            if (s + 1 < NKS) {
#pragma unroll
                for (int qb = 0; qb < NQB; ++qb) ring[(s + 1) & 1][qb] = rd(s + 1, qb, tile);
            }
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
                for (int ab = 0; ab < NAB; ++ab)
                    acc[qb][ab] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ring[s & 1][qb], breg[ab][s], acc[qb][ab], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // epilogue: slice maxima.  Zero-padded rows score exactly 0 and cannot reach any cut (> 0).
#pragma unroll
        for (int ab = 0; ab < NAB; ++ab) {
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb) {
                int xi = acc[qb][ab][0];
#pragma unroll
                for (int r = 1; r < 16; ++r) xi = max(xi, acc[qb][ab][r]);
                const float x = (float)xi * (hi ? sc2[qb].y : sc2[qb].x);
                const bool improved = x > runmax[ab];
                run2[ab] = fmaxf(fminf(runmax[ab], x), run2[ab]);
                runmax[ab] = fmaxf(runmax[ab], x);
                runidx[ab] = improved ? ((qt * NQB + qb) * 2 + hi) : runidx[ab];
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[qb][ab][r] = 0;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        buf ^= 1;
    }
#pragma unroll
    for (int ab = 0; ab < NAB; ++ab) {
        const float om1 = __shfl_xor(runmax[ab], 32), om2 = __shfl_xor(run2[ab], 32);
        const int oi1 = __shfl_xor(runidx[ab], 32);
        const float m1 = fmaxf(runmax[ab], om1);
        const float m2 = fmaxf(fminf(runmax[ab], om1), fmaxf(run2[ab], om2));
        const int i1 = (om1 > runmax[ab]) ? oi1 : runidx[ab];
        const int a = a0 + wave * 64 + ab * 32 + l31;
        if (hi == 0) {
            const size_t o = ((size_t)p * S + split) * cap_a + a;
            ws_max[o] = m1;
            ws_i1[o] = i1;
            ws_m2[o] = m2;
        }
    }
}

// Round-2 restructuring of the int8 screening loop (same operands, same outputs, same tiles as match_i8_screen_kernel).
// Round 1 ran all 8 (query block, anchor block) accumulators of a tile through the k loop together, then reduced them: ~180 VALU +
// 128 accumulator zeroings per 64 MFMAs, none of which overlapped matrix work (PMC: 68 % MFMA-pipe utilisation).  Here a query
// block's TWO anchor-block chains (8 k-steps each, alternating so no MFMA waits on its predecessor's accumulator) run to completion
// before the next query block starts, so
//   * the slice-maximum epilogue of query block qb-1 (2 x {v_max3 tree, convert, scale, running (m1, slice, m2) update}) is issued
//     in the shadow of query block qb's 16 MFMAs - the matrix pipe executes 32 cycles per MFMA, the wave issues one every ~32;
//   * accumulators start from the inline constant 0 (first MFMA of a chain takes C = 0): no zeroing;
//   * 4 accumulators are live instead of 8, which pays for a double-buffered A operand (the 8 ds_read_b128 of query block qb+1
//     also issue under qb's MFMAs).
// sched_group_barrier pins the interleave (1 MFMA : 2 VALU : <=1 LDS read) in the emitted code.
// WAVES = 8: 512 anchors per workgroup - the eight waves share ONE query-tile stream (half the L2 -> LDS bytes, DMA issues and LDS footprint per
// anchor of two 4-wave workgroups on a CU); T is then the number of 512-anchor panels.
template <int CP, int VAR = 0, int WAVES = 4>     // VAR != 0: timing ablations only (ORYON_SCREEN8_ABLATE): 1 no epilogue, 2 no DMA / barrier, 8 DMA spread over two blocks
__global__ __launch_bounds__(64 * WAVES, CP == 512 ? 1 : 2) void match_i8_screen_v2_kernel(
    const int8_t *__restrict__ a8, const int8_t *__restrict__ q8, const float *__restrict__ q_scale, int B, int cap_a, int cap_q,
    const int32_t *__restrict__ n_a, const int32_t *__restrict__ n_q, int T, int S, float *__restrict__ ws_max,
    int32_t *__restrict__ ws_i1, float *__restrict__ ws_m2)
{
    constexpr int RB = CP;
    constexpr int TILE_BYTES = screen8_tile_bytes(CP);
    constexpr int ROWS = 128, NQB = 4, NAB = 2;
    constexpr int NKS = CP / 32;
    constexpr int NI = TILE_BYTES / (1024 * WAVES);
    constexpr int LPR = RB / 256;
    char *smem;
    if constexpr (2 * TILE_BYTES > 65536) {
        extern __shared__ __attribute__((aligned(256))) char smem_dyn8b[];
        smem = smem_dyn8b;
    } else {
        __shared__ __attribute__((aligned(256))) char smem_st8b[2 * TILE_BYTES];
        smem = smem_st8b;
    }
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int unit = (slot / T) * 8 + xcd;
    if (unit >= B * S) return;
    const int panel = slot % T;
    const int p = unit / S, split = unit % S;
    const int na = n_a[p], nq = n_q[p];
    const int a0 = panel * (64 * WAVES);
    if (a0 >= na) return;
    const int nqt = (nq + ROWS - 1) / ROWS;
    const int qt_per = (nqt + S - 1) / S;
    const int qt_begin = split * qt_per;
    const int qt_end = (qt_begin + qt_per < nqt) ? qt_begin + qt_per : nqt;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, hi = lane >> 5;
    const unsigned hi_mask = 0u - (unsigned)hi;
    const char *qp = reinterpret_cast<const char *>(q8) + (size_t)p * cap_q * RB;
    const float2 *qs = reinterpret_cast<const float2 *>(q_scale + (size_t)p * (cap_q / 16));

    i32x4 breg[NAB][NKS];
#pragma unroll
    for (int ab = 0; ab < NAB; ++ab) {
        const int arow_i = a0 + wave * 64 + ab * 32 + l31;              // WAVES = 8: the last panel may reach past cap_a (a multiple of 256)
        const char *arow = reinterpret_cast<const char *>(a8) + ((size_t)p * cap_a + (arow_i < cap_a ? arow_i : cap_a - 1)) * RB + 16 * hi;
#pragma unroll
        for (int s = 0; s < NKS; ++s) breg[ab][s] = *reinterpret_cast<const i32x4 *>(arow + 32 * s);
    }
    unsigned dma_off[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int line = (wave * NI + j) * 4 + (lane >> 4), sl = lane & 15;
        const int row = line / LPR;
        const int cc = sl ^ (row & 15);
        dma_off[j] = (unsigned)(row * RB + ((line % LPR) * 16 + cc) * 16);
    }
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto issue_one = [&](int qt, int buf, int j) {
        const char *qb = qp + (size_t)qt * TILE_BYTES;
        char *dst = smem + buf * TILE_BYTES + (wave_u * NI + j) * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(qb + dma_off[j]),
                                         (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
    };
    unsigned koff[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) koff[c] = (unsigned)(l31 * RB) + ((((unsigned)(hi ^ (l31 & 15))) ^ (2u * c)) << 4);
    auto rd = [&](int s, int qb, unsigned tile) -> i32x4 {
        return *reinterpret_cast<const i32x4 *>(smem + koff[s & 7] + tile + (unsigned)(qb * 32 * RB + (s >> 3) * 256));
    };

    float runmax[NAB], run2[NAB];
    int runidx[NAB];
#pragma unroll
    for (int ab = 0; ab < NAB; ++ab) { runmax[ab] = -INFINITY; run2[ab] = -INFINITY; runidx[ab] = 0; }
    // slice epilogue of one finished 32x32 block: integer maximum of the lane's 16 rows (exact: one exponent per slice), one
    // convert, one multiply, running (best, slice of best, best other slice)
    auto reduce_block = [&](const i32x16 &c, float sc, int sid, int ab) {
        int m0 = max(max(c[0], c[1]), c[2]), m1 = max(max(c[3], c[4]), c[5]), m2 = max(max(c[6], c[7]), c[8]);
        int m3 = max(max(c[9], c[10]), c[11]), m4 = max(max(c[12], c[13]), c[14]);
        int xi = max(max(max(m0, m1), m2), max(max(m3, m4), c[15]));
        const float x = (float)xi * sc;
        const bool improved = x > runmax[ab];
        run2[ab] = fmaxf(fminf(runmax[ab], x), run2[ab]);
        runmax[ab] = fmaxf(runmax[ab], x);
        runidx[ab] = improved ? sid : runidx[ab];
    };
    const i32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    if (qt_end > qt_begin) {
#pragma unroll
        for (int j = 0; j < NI; ++j) issue_one(qt_begin, 0, j);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    i32x4 areg[NKS];
#pragma unroll
    for (int s = 0; s < NKS; ++s) areg[s] = rd(s, 0, 0u);
    // finished accumulators of the previous query block, reduced under the next block's MFMAs.  The steady-state loop has no branch:
    // before the first block `prev` is a dummy that can never win (score -2^30), and the last tile re-issues its own DMA into the idle
    // buffer instead of testing for "one more tile".
    i32x16 prev[NAB];
#pragma unroll
    for (int ab = 0; ab < NAB; ++ab)
#pragma unroll
        for (int r = 0; r < 16; ++r) prev[ab][r] = -(1 << 30);
    float prev_sc = 1.0f;
    int prev_sid = 0;
    int buf = 0;
    for (int qt = qt_begin; qt < qt_end; ++qt) {
        const unsigned tile = buf * TILE_BYTES;
        const int qt_next = qt + 1 < qt_end ? qt + 1 : qt;
        float2 sc2[NQB];
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) sc2[qb] = qs[qt * NQB + qb];
        // ONE A-operand buffer: after both anchor blocks have consumed k-step s of query block qb, areg[s] is re-loaded with k-step s of
        // query block qb+1 (needed 16 MFMAs ~ 500 cycles later)
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) {
            i32x16 acc[NAB];
            // the whole next tile is requested under the FIRST query block's MFMAs: ~1500 cycles before the wait at the tile's end
            // (spreading the 8 requests over the four blocks left the last ones ~500 cycles, less than an L2 miss: +4 %)
            if (!(VAR & 2) && (((VAR & 8) && qb < 2) || (!(VAR & 8) && qb == 0))) {
#pragma unroll
                for (int j = ((VAR & 8) ? qb * (NI / 2) : 0); j < ((VAR & 8) ? (qb + 1) * (NI / 2) : NI); ++j) issue_one(qt_next, buf ^ 1, j);
            }
#pragma unroll
            for (int s = 0; s < NKS; ++s) {
#pragma unroll
                for (int ab = 0; ab < NAB; ++ab)
                    acc[ab] = __builtin_amdgcn_mfma_i32_32x32x32_i8(areg[s], breg[ab][s], s == 0 ? zero16 : acc[ab], 0, 0, 0);
                if (qb + 1 < NQB) areg[s] = rd(s, qb + 1, tile);
            }
            if (!(VAR & 1)) {
#pragma unroll
                for (int ab = 0; ab < NAB; ++ab) reduce_block(prev[ab], prev_sc, prev_sid, ab);
            } else {
#pragma unroll
                for (int ab = 0; ab < NAB; ++ab)
#pragma unroll
                    for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(prev[ab][r]));
            }
            // pin the interleave: per MFMA pair two VALU of the previous block's epilogue and one LDS read of the next A operand
#pragma unroll
            for (int i = 0; i < NKS; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                if (qb + 1 < NQB) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
#pragma unroll
            for (int ab = 0; ab < NAB; ++ab) prev[ab] = acc[ab];
            // lane half hi picks .x / .y with bit masks: written as `hi ? .y : .x` the compiler indexes the float2 array dynamically,
            // moves it to LDS (one 32-byte slot per thread) and reads it back with 8-way bank-conflicting ds_read_b32 - 1.1e8
            // SQ_LDS_BANK_CONFLICT cycles per launch in the round-1 / early round-2 counters
            {
                const unsigned ux = __builtin_bit_cast(unsigned, sc2[qb].x), uy = __builtin_bit_cast(unsigned, sc2[qb].y);
                prev_sc = __builtin_bit_cast(float, (ux & ~hi_mask) | (uy & hi_mask));
            }
            prev_sid = (qt * NQB + qb) * 2 + hi;
        }
        if (!(VAR & 2)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            buf ^= 1;
        }
        // first A operand of the next tile (its DMA has landed: barrier above)
#pragma unroll
        for (int s = 0; s < NKS; ++s) areg[s] = rd(s, 0, buf * TILE_BYTES);
    }
#pragma unroll
    for (int ab = 0; ab < NAB; ++ab) reduce_block(prev[ab], prev_sc, prev_sid, ab);
#pragma unroll
    for (int ab = 0; ab < NAB; ++ab) {
        const float om1 = __shfl_xor(runmax[ab], 32), om2 = __shfl_xor(run2[ab], 32);
        const int oi1 = __shfl_xor(runidx[ab], 32);
        const float m1 = fmaxf(runmax[ab], om1);
        const float m2 = fmaxf(fminf(runmax[ab], om1), fmaxf(run2[ab], om2));
        const int i1 = (om1 > runmax[ab]) ? oi1 : runidx[ab];
        const int a = a0 + wave * 64 + ab * 32 + l31;
        if (hi == 0 && a < cap_a) {
            const size_t o = ((size_t)p * S + split) * cap_a + a;
            ws_max[o] = m1;
            ws_i1[o] = i1;
            ws_m2[o] = m2;
        }
    }
}


// K1s6, the MX-fp6 screen (round 3), lives in screen_mx6.hip (own translation unit: compiled with -fno-honor-nans)

// fp32 rows (k permuted inside groups of 8: position 8g+4h+j holds k = 8g+2j+h) -> fp16 rows in natural k order, the values K0's
// fp16 output would hold.  One lane per group of 8.
__device__ __forceinline__ uint4 half_group_from_permuted(const float4 lo, const float4 hi4)
{
    union { __half h[8]; uint4 u; } pk;
    pk.h[0] = __float2half_rn(lo.x); pk.h[1] = __float2half_rn(hi4.x);
    pk.h[2] = __float2half_rn(lo.y); pk.h[3] = __float2half_rn(hi4.y);
    pk.h[4] = __float2half_rn(lo.z); pk.h[5] = __float2half_rn(hi4.z);
    pk.h[6] = __float2half_rn(lo.w); pk.h[7] = __float2half_rn(hi4.w);
    return pk.u;
}

// query-side fp16 copies for the fp16 pipeline, made only for pairs that have undecided anchors (flag-gated on the device)
__global__ __launch_bounds__(256) void match_make_q16_kernel(const float *__restrict__ q_hat, int Cp, int cap_q,
                                                              const int32_t *__restrict__ n_q, const int32_t *__restrict__ n_amb,
                                                              __half *__restrict__ q16)
{
    const int p = blockIdx.y;
    if (n_amb[p] == 0) return;
    const int n_fill = (n_q[p] + 255) / 256 * 256;
    const size_t groups = (size_t)n_fill * (Cp / 8);
    const float4 *src = reinterpret_cast<const float4 *>(q_hat + (size_t)p * cap_q * Cp);
    uint4 *dst = reinterpret_cast<uint4 *>(q16 + (size_t)p * cap_q * Cp);
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < groups; g += (size_t)gridDim.x * 256)
        dst[g] = half_group_from_permuted(src[2 * g], src[2 * g + 1]);
}

// gather the rows of the anchors the int8 stage could not decide into dense panels for the fp16 pipeline (fp32 copy + fp16 copy)
__global__ __launch_bounds__(256) void match_compact8_kernel(const float *__restrict__ a_hat, const __half *__restrict__ a16, int Cp,
                                                              int cap_a, const int32_t *__restrict__ n_amb,
                                                              const int32_t *__restrict__ amb_idx, float *__restrict__ a_hat_c,
                                                              __half *__restrict__ a16_c)
{
    const int p = blockIdx.y, lane = threadIdx.x & 63;
    const int n_fill = (n_amb[p] + 255) / 256 * 256;               // the fp16 kernels read whole 256-anchor panels: zero-fill the tail
    for (int g = 0; g < 16; ++g) {
    const int sl = (blockIdx.x * 16 + g) * 4 + (threadIdx.x >> 6);
    if (sl >= n_fill) break;
    uint4 *d32 = reinterpret_cast<uint4 *>(a_hat_c + ((size_t)p * cap_a + sl) * Cp);
    uint4 *d16 = reinterpret_cast<uint4 *>(a16_c + ((size_t)p * cap_a + sl) * Cp);
    if (sl >= n_amb[p]) {
        for (int i = lane; i < Cp / 4; i += 64) d32[i] = make_uint4(0, 0, 0, 0);
        for (int i = lane; i < Cp / 8; i += 64) d16[i] = make_uint4(0, 0, 0, 0);
        continue;
    }
    const int a = amb_idx[(size_t)p * cap_a + sl];
    const uint4 *s32 = reinterpret_cast<const uint4 *>(a_hat + ((size_t)p * cap_a + a) * Cp);
    (void)a16;
    for (int i = lane; i < Cp / 4; i += 64) d32[i] = s32[i];
    const float4 *f32 = reinterpret_cast<const float4 *>(s32);
    for (int g = lane; g < Cp / 8; g += 64) d16[g] = half_group_from_permuted(f32[2 * g], f32[2 * g + 1]);
    }
}

__global__ __launch_bounds__(256) void match_scatter8_kernel(int cap_a, const int32_t *__restrict__ n_amb,
                                                              const int32_t *__restrict__ amb_idx, const float *__restrict__ md_c,
                                                              const int32_t *__restrict__ am_c, const uint8_t *__restrict__ va_c,
                                                              float *__restrict__ min_dist, int32_t *__restrict__ argmin,
                                                              uint8_t *__restrict__ valid)
{
    const int p = blockIdx.y, sl = blockIdx.x * 256 + threadIdx.x;
    if (sl >= n_amb[p]) return;
    const size_t src = (size_t)p * cap_a + sl, dst = (size_t)p * cap_a + amb_idx[src];
    min_dist[dst] = md_c[src];
    argmin[dst] = am_c[src];
    valid[dst] = va_c[src];
}

static int pick_split16(int B, int T)
{
    static const int target = dev_env_int("ORYON_SCREEN_WGS", 8192);     // timing experiments only
    int S = (target + B * T - 1) / (B * T);
    if (S < 1) S = 1;
    if (S > 16) S = 16;
    return S;
}

struct ScreenWs {
    float *ws_max, *ws_m2, *m_final, *amb_max;
    int32_t *ws_i1, *cnt, *cand, *panel_flag, *n_amb, *amb_idx;
    __half *a16c;
    uint8_t *row_flag;
    size_t bytes, zero_off, zero_bytes;
};

static ScreenWs carve_screen(void *base, int B, int C, int cap_a, int S)
{
    ScreenWs w;
    char *p = static_cast<char *>(base);
    size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off = (off + n + 255) / 256 * 256; return o; };
    const size_t o_max = take((size_t)B * S * cap_a * sizeof(float));
    const size_t o_m2 = take((size_t)B * S * cap_a * sizeof(float));
    const size_t o_i1 = take((size_t)B * S * cap_a * sizeof(int32_t));
    const size_t o_mf = take((size_t)B * cap_a * sizeof(float));
    const size_t o_am = take((size_t)B * cap_a * sizeof(float));
    const size_t o_ai = take((size_t)B * cap_a * sizeof(int32_t));
    const size_t o_a16 = take((size_t)B * cap_a * C * sizeof(__half));
    const size_t o_cand = take((size_t)B * cap_a * SCREEN_CAP * sizeof(int32_t));
    w.zero_off = off;
    const size_t o_cnt = take((size_t)B * cap_a * sizeof(int32_t));
    const size_t o_rf = take((size_t)B * cap_a);
    const size_t o_pf = take((size_t)B * (cap_a / ORYON_MATCH_TILE) * sizeof(int32_t));
    const size_t o_na = take((size_t)B * sizeof(int32_t));
    w.zero_bytes = off - w.zero_off;
    w.bytes = off;
    w.ws_m2 = base ? reinterpret_cast<float *>(p + o_m2) : nullptr;
    w.ws_i1 = base ? reinterpret_cast<int32_t *>(p + o_i1) : nullptr;
    w.m_final = base ? reinterpret_cast<float *>(p + o_mf) : nullptr;
    w.amb_max = base ? reinterpret_cast<float *>(p + o_am) : nullptr;
    w.amb_idx = base ? reinterpret_cast<int32_t *>(p + o_ai) : nullptr;
    w.a16c = base ? reinterpret_cast<__half *>(p + o_a16) : nullptr;
    w.n_amb = base ? reinterpret_cast<int32_t *>(p + o_na) : nullptr;
    w.ws_max = base ? reinterpret_cast<float *>(p + o_max) : nullptr;
    w.cand = base ? reinterpret_cast<int32_t *>(p + o_cand) : nullptr;
    w.cnt = base ? reinterpret_cast<int32_t *>(p + o_cnt) : nullptr;
    w.row_flag = base ? reinterpret_cast<uint8_t *>(p + o_rf) : nullptr;
    w.panel_flag = base ? reinterpret_cast<int32_t *>(p + o_pf) : nullptr;
    return w;
}

}  // namespace oryon

using namespace oryon;

extern "C" size_t oryon_match_screened_workspace_bytes(int B, int C, int cap_a)
{
    if (B <= 0 || C <= 0 || cap_a <= 0 || cap_a % MT16) return 0;
    return carve_screen(nullptr, B, C, cap_a, pick_split16(B, cap_a / MT16)).bytes;
}

namespace {
// one launcher per descriptor width; C = 512 needs 2 x 64 KB of dynamic LDS (opt-in above 64 KB)
template <int CP, int MODE>
void launch_screen(int groups, hipStream_t st, const __half *a16, const __half *q16, int B, int cap_a, int cap_q, const int32_t *n_a,
                   const int32_t *n_q, int T, int S, float valid_cut, float *ws_max, int32_t *cnt, int32_t *cand, int S_thr,
                   const int32_t *row_map, int32_t *ws_i1, float *ws_m2)
{
    constexpr size_t dyn = 2 * screen_tile_bytes(CP) > 65536 ? 2 * screen_tile_bytes(CP) : 0;
    if (dyn) allow_dynamic_lds(reinterpret_cast<const void *>(&match_f16_screen_kernel<CP, MODE, 0>), (int)dyn);
    hipLaunchKernelGGL((match_f16_screen_kernel<CP, MODE, 0>), dim3(groups), dim3(256), dyn, st, a16, q16, B, cap_a, cap_q, n_a, n_q, T,
                       S, valid_cut, ws_max, cnt, cand, S_thr, row_map, ws_i1, ws_m2);
}
}  // namespace

extern "C" int oryon_match_screened(const float *a_hat, const float *q_hat, const void *a_f16, const void *q_f16, int B, int C,
                                    int cap_a, int cap_q, const int32_t *n_a, const int32_t *n_q, float threshold, float *min_dist,
                                    int32_t *argmin, uint8_t *valid, void *workspace, size_t workspace_bytes, void *stream)
{
    ORYON_CHECK_ARG(a_hat && q_hat && a_f16 && q_f16 && n_a && n_q && min_dist && argmin && valid);
    ORYON_CHECK_ARG(B >= 0 && (C == 128 || C == 256 || C == 512) && cap_a > 0 && cap_a % MT16 == 0 && cap_q > 0 && cap_q % 256 == 0);
    ORYON_CHECK_ARG(threshold > 0.0f && threshold <= 0.5f);
    if (B == 0) return ORYON_OK;
    const int T = cap_a / MT16;
    const int S = pick_split16(B, T);
    ScreenWs w = carve_screen(workspace, B, C, cap_a, S);
    if (!workspace || workspace_bytes < w.bytes) {
        set_error("oryon_match_screened: workspace too small (%zu < %zu)", workspace_bytes, w.bytes);
        return ORYON_ERR_WORKSPACE;
    }
    hipStream_t st = as_stream(stream);
    ORYON_CHECK_HIP(hipMemsetAsync(static_cast<char *>(workspace) + w.zero_off, 0, w.zero_bytes, st));
    // rows whose best fp16 score is below this can never satisfy 0.5*(1-dot) < threshold
    const float valid_cut = (1.0f - 2.0f * threshold) - SCREEN_DELTA - 1e-6f;
    const int groups = ((B * S + 7) / 8) * 8 * T;
    const __half *a16 = static_cast<const __half *>(a_f16), *q16 = static_cast<const __half *>(q_f16);
#define LAUNCH16(CPV, MODEV)                                                                                              \
    launch_screen<CPV, MODEV>(groups, st, a16, q16, B, cap_a, cap_q, n_a, n_q, T, S, valid_cut, w.ws_max, w.cnt, w.cand, S, nullptr,   \
                              w.ws_i1, w.ws_m2)
#define LAUNCH16_AMB(CPV)                                                                                                 \
    launch_screen<CPV, 1>(groups, st, w.a16c, q16, B, cap_a, cap_q, w.n_amb, n_q, T, S, valid_cut, w.amb_max, w.cnt, w.cand, 1,        \
                          w.amb_idx, nullptr, nullptr)
    static const int var16 = dev_env_int("ORYON_MATCH16_VARIANT", 0);
    static const bool two_pass = dev_env_set("ORYON_SCREEN_TWOPASS");
    const float *m_final = nullptr;
    if (two_pass || var16) {
#define LAUNCH16V(V) hipLaunchKernelGGL((match_f16_screen_kernel<256, 0, V>), dim3(groups), dim3(256), 0, st, a16, q16, B, cap_a, cap_q, n_a, n_q, T, S, valid_cut, w.ws_max, w.cnt, w.cand, S, nullptr, w.ws_i1, w.ws_m2)
        if (C == 256 && var16 == 8) {
            LAUNCH16(256, 0);
            hipLaunchKernelGGL((match_f16_screen_kernel<256, 1, 8>), dim3(groups), dim3(256), 0, st, a16, q16, B, cap_a, cap_q, n_a, n_q, T, S, valid_cut, w.ws_max, w.cnt, w.cand, S, nullptr, w.ws_i1, w.ws_m2);
        } else if (C == 256 && var16) {
            switch (var16) { case 1: LAUNCH16V(1); break; case 2: LAUNCH16V(2); break; case 3: LAUNCH16V(3); break; case 4: LAUNCH16V(4); break;
                             case 5: LAUNCH16V(5); break; case 6: LAUNCH16V(6); break; default: LAUNCH16V(7); break; }
            LAUNCH16(256, 1);
        } else if (C == 256) { LAUNCH16(256, 0); LAUNCH16(256, 1); }
        else if (C == 512) { LAUNCH16(512, 0); LAUNCH16(512, 1); }
        else { LAUNCH16(128, 0); LAUNCH16(128, 1); }
#undef LAUNCH16V
    } else {
        // single screening pass keeping (max, argmax, second max) per anchor; anchors whose runner-up is within MARGIN of the
        // maximum (duplicates, smooth descriptor fields) go through a second, compacted candidate pass
        profile_begin(st, C == 256 ? "match_f16_screen_kernel<256, 2>" : C == 512 ? "match_f16_screen_kernel<512, 2>" : "match_f16_screen_kernel<128, 2>");
        if (C == 256) LAUNCH16(256, 2); else if (C == 512) LAUNCH16(512, 2); else LAUNCH16(128, 2);
        profile_end(st);
        ORYON_CHECK_LAUNCH();
        if (C >= 256)
            hipLaunchKernelGGL((match_decide_kernel<64>), dim3(cap_a / 64, B), dim3(256), 0, st, a16, q16, C, cap_a, cap_q, n_a, n_q, S,
                               valid_cut, w.ws_max, w.ws_i1, w.ws_m2, w.m_final, w.cnt, w.cand, w.n_amb, w.amb_idx, nullptr, nullptr, nullptr, 0.f, 0.f, 0.f, nullptr, nullptr, nullptr);
        else
            hipLaunchKernelGGL((match_decide_kernel<128>), dim3(cap_a / 64, B), dim3(256), 0, st, a16, q16, C, cap_a, cap_q, n_a, n_q, S,
                               valid_cut, w.ws_max, w.ws_i1, w.ws_m2, w.m_final, w.cnt, w.cand, w.n_amb, w.amb_idx, nullptr, nullptr, nullptr, 0.f, 0.f, 0.f, nullptr, nullptr, nullptr);
        hipLaunchKernelGGL(match_compact_kernel, dim3(cap_a / 64, B), dim3(256), 0, st, a16, C, cap_a, w.n_amb, w.amb_idx, w.m_final,
                           w.a16c, w.amb_max);
        if (C == 256) LAUNCH16_AMB(256); else if (C == 512) LAUNCH16_AMB(512); else LAUNCH16_AMB(128);
        m_final = w.m_final;
    }
#undef LAUNCH16
#undef LAUNCH16_AMB
    ORYON_CHECK_LAUNCH();
    // 4 lanes per anchor: 16 -> 394 us, 8 -> 230, 4 -> 184, 2 -> 175, 1 -> 200 us at cfg2 (almost every anchor has one candidate)
    hipLaunchKernelGGL((match_rescore_kernel<4>), dim3(cap_a / 64, B), dim3(256), 0, st, a_hat, q_hat, C, cap_a, cap_q, n_a, n_q, S,
                       threshold, valid_cut, w.ws_max, m_final, w.cnt, w.cand, min_dist, argmin, valid, w.row_flag, w.panel_flag);
    ORYON_CHECK_LAUNCH();
    // exact recomputation of the (rare) panels whose candidate lists overflowed; exits immediately elsewhere
    return match_f32_flagged(a_hat, q_hat, B, C, cap_a, cap_q, n_a, n_q, threshold, min_dist, argmin, valid, w.panel_flag,
                             w.row_flag, stream);
}


// ------------------------------------------------------------------------------------------------ K1s8 entry points
namespace {
struct Screen8Ws {
    ScreenWs top;
    __half *q16;
    float *a_hat_c, *md_c;
    int32_t *am_c;
    uint8_t *va_c;
    void *nested;
    size_t nested_bytes, bytes;
};

Screen8Ws carve_screen8(void *base, int B, int C, int cap_a, int cap_q, int S)
{
    Screen8Ws w;
    w.top = carve_screen(base, B, C, cap_a, S);
    char *p = static_cast<char *>(base);
    size_t off = (w.top.bytes + 255) / 256 * 256;
    auto take = [&](size_t n) { size_t o = off; off = (off + n + 255) / 256 * 256; return o; };
    const size_t o_q16 = take((size_t)B * cap_q * C * sizeof(__half));
    const size_t o_ah = take((size_t)B * cap_a * C * sizeof(float));
    const size_t o_md = take((size_t)B * cap_a * sizeof(float));
    const size_t o_am = take((size_t)B * cap_a * sizeof(int32_t));
    const size_t o_va = take((size_t)B * cap_a);
    w.nested_bytes = carve_screen(nullptr, B, C, cap_a, S).bytes;
    const size_t o_ne = take(w.nested_bytes);
    w.bytes = off;
    w.q16 = base ? reinterpret_cast<__half *>(p + o_q16) : nullptr;
    w.a_hat_c = base ? reinterpret_cast<float *>(p + o_ah) : nullptr;
    w.md_c = base ? reinterpret_cast<float *>(p + o_md) : nullptr;
    w.am_c = base ? reinterpret_cast<int32_t *>(p + o_am) : nullptr;
    w.va_c = base ? reinterpret_cast<uint8_t *>(p + o_va) : nullptr;
    w.nested = base ? static_cast<void *>(p + o_ne) : nullptr;
    return w;
}

// the kernel launch_screen8<CP> dispatches under the current development switches (for oryon_dominant_kernel)
template <int CP>
const char *screen8_name()
{
    const int variant = dev_env_int("ORYON_SCREEN8_VARIANT", 2);
    const int ablate = dev_env_int("ORYON_SCREEN8_ABLATE", 0);
    const int waves = dev_env_int("ORYON_SCREEN8_WAVES", 8);
    if (variant == 1) return CP == 256 ? "match_i8_screen_kernel<256>" : "match_i8_screen_kernel<512>";
    if (ablate && CP == 256) return "match_i8_screen_v2_kernel<256, ABLATED> (timing ablation: results are wrong)";
    if (waves == 8 && CP == 256) return "match_i8_screen_v2_kernel<256, 0, 8>";
    return CP == 256 ? "match_i8_screen_v2_kernel<256, 0, 4>" : "match_i8_screen_v2_kernel<512, 0, 4>";
}

template <int CP>
void launch_screen8(int groups, hipStream_t st, const int8_t *a8, const int8_t *q8, const float *q_scale, int B, int cap_a, int cap_q,
                    const int32_t *n_a, const int32_t *n_q, int T, int S, float *ws_max, int32_t *ws_i1, float *ws_m2)
{
    constexpr size_t dyn = 2 * screen8_tile_bytes(CP) > 65536 ? 2 * screen8_tile_bytes(CP) : 0;
    static const int variant = dev_env_int("ORYON_SCREEN8_VARIANT", 2);
    if (variant == 1) {                 // round-1 loop (kept for A/B timing)
        if (dyn) allow_dynamic_lds(reinterpret_cast<const void *>(&match_i8_screen_kernel<CP>), (int)dyn);
        hipLaunchKernelGGL((match_i8_screen_kernel<CP>), dim3(groups), dim3(256), dyn, st, a8, q8, q_scale, B, cap_a, cap_q, n_a, n_q, T, S,
                           ws_max, ws_i1, ws_m2);
        return;
    }
    static const int ablate = dev_env_int("ORYON_SCREEN8_ABLATE", 0);
    if (ablate && CP == 256) {
#define ABL(V) case V: hipLaunchKernelGGL((match_i8_screen_v2_kernel<256, V>), dim3(groups), dim3(256), 0, st, a8, q8, q_scale, B, cap_a, cap_q, n_a, n_q, T, S, ws_max, ws_i1, ws_m2); break
        switch (ablate) { ABL(1); ABL(2); ABL(3); default: ABL(8); }
#undef ABL
        return;
    }
    static const int waves = dev_env_int("ORYON_SCREEN8_WAVES", 8);
    if (waves == 8 && CP == 256) {
        const int T8 = (cap_a + 511) / 512;
        hipLaunchKernelGGL((match_i8_screen_v2_kernel<256, 0, 8>), dim3(groups / T * T8), dim3(512), 0, st, a8, q8, q_scale, B, cap_a, cap_q, n_a,
                           n_q, T8, S, ws_max, ws_i1, ws_m2);
        return;
    }
    if (dyn) allow_dynamic_lds(reinterpret_cast<const void *>(&match_i8_screen_v2_kernel<CP>), (int)dyn);
    hipLaunchKernelGGL((match_i8_screen_v2_kernel<CP>), dim3(groups), dim3(256), dyn, st, a8, q8, q_scale, B, cap_a, cap_q, n_a, n_q, T, S,
                       ws_max, ws_i1, ws_m2);
}
}  // namespace


extern "C" size_t oryon_match_screened8_workspace_bytes(int B, int C, int cap_a, int cap_q)
{
    if (B <= 0 || C <= 0 || cap_a <= 0 || cap_a % MT16 || cap_q <= 0) return 0;
    return carve_screen8(nullptr, B, C, cap_a, cap_q, pick_split16(B, cap_a / MT16)).bytes;
}

extern "C" int oryon_match_screened8(const float *a_hat, const float *q_hat, const int8_t *a_i8, const int8_t *q_i8, const float *a_scale, const float *q_scale, const float *q_eps_max, int B,
                                     int C_true, int C, int cap_a, int cap_q, const int32_t *n_a, const int32_t *n_q, float threshold,
                                     float *min_dist, int32_t *argmin, uint8_t *valid, int32_t *n_undecided, void *workspace,
                                     size_t workspace_bytes, void *stream)
{
    ORYON_CHECK_ARG(a_hat && q_hat && a_i8 && q_i8 && a_scale && q_scale && q_eps_max && n_a && n_q);
    ORYON_CHECK_ARG(min_dist && argmin && valid && B >= 0 && (C == 256 || C == 512) && C_true > 0 && C_true <= C);
    ORYON_CHECK_ARG(cap_a > 0 && cap_a % MT16 == 0 && cap_q > 0 && cap_q % 256 == 0 && threshold > 0.0f && threshold <= 0.5f);
    if (B == 0) return ORYON_OK;
    const int T = cap_a / MT16;
    const int S = pick_split16(B, T);
    Screen8Ws w8 = carve_screen8(workspace, B, C, cap_a, cap_q, S);
    if (!workspace || workspace_bytes < w8.bytes) {
        set_error("oryon_match_screened8: workspace too small (%zu < %zu)", workspace_bytes, w8.bytes);
        return ORYON_ERR_WORKSPACE;
    }
    ScreenWs &w = w8.top;
    hipStream_t st = as_stream(stream);
    ORYON_CHECK_HIP(hipMemsetAsync(static_cast<char *>(workspace) + w.zero_off, 0, w.zero_bytes, st));
    const float cut0 = 1.0f - 2.0f * threshold;
    const float valid_cut16 = cut0 - SCREEN_DELTA - 1e-6f;
    const int groups = ((B * S + 7) / 8) * 8 * T;
    profile_begin(st, C == 256 ? screen8_name<256>() : screen8_name<512>());
    if (C == 256) launch_screen8<256>(groups, st, a_i8, q_i8, q_scale, B, cap_a, cap_q, n_a, n_q, T, S, w.ws_max, w.ws_i1, w.ws_m2);
    else launch_screen8<512>(groups, st, a_i8, q_i8, q_scale, B, cap_a, cap_q, n_a, n_q, T, S, w.ws_max, w.ws_i1, w.ws_m2);
    profile_end(st);
    ORYON_CHECK_LAUNCH();
    hipLaunchKernelGGL((match_decide_kernel<128>), dim3(cap_a / 64, B), dim3(256), 0, st, static_cast<const __half *>(nullptr),
                       static_cast<const __half *>(nullptr), C, cap_a, cap_q, n_a, n_q, S, valid_cut16,
                       w.ws_max, w.ws_i1, w.ws_m2, w.m_final, w.cnt, w.cand, w.n_amb, w.amb_idx, a_scale, nullptr, q_eps_max, cut0,
                       sqrtf((float)C_true), (float)C_true, a_i8, q_i8, q_scale);
    ORYON_CHECK_LAUNCH();
    hipLaunchKernelGGL((match_rescore_kernel<4>), dim3(cap_a / 64, B), dim3(256), 0, st, a_hat, q_hat, C, cap_a, cap_q, n_a, n_q, S,
                       threshold, -INFINITY, w.ws_max, w.m_final, w.cnt, w.cand, min_dist, argmin, valid, w.row_flag, w.panel_flag);
    ORYON_CHECK_LAUNCH();
    int rc = match_f32_flagged(a_hat, q_hat, B, C, cap_a, cap_q, n_a, n_q, threshold, min_dist, argmin, valid, w.panel_flag, w.row_flag,
                               stream);
    if (rc) return rc;
    if (n_undecided) ORYON_CHECK_HIP(hipMemcpyAsync(n_undecided, w.n_amb, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    // anchors the int8 stage could not decide: complete fp16 pipeline on the compacted set, results scattered back
    // (their fp16 operands are made here, and only for pairs that have such anchors: K0 does not write fp16 rows for this path)
    hipLaunchKernelGGL(match_compact8_kernel, dim3(cap_a / 64, B), dim3(256), 0, st, a_hat, static_cast<const __half *>(nullptr), C, cap_a,
                       w.n_amb, w.amb_idx, w8.a_hat_c, w.a16c);
    hipLaunchKernelGGL(match_make_q16_kernel, dim3(64, B), dim3(256), 0, st, q_hat, C, cap_q, n_q, w.n_amb, w8.q16);
    ORYON_CHECK_LAUNCH();
    rc = oryon_match_screened(w8.a_hat_c, q_hat, w.a16c, w8.q16, B, C, cap_a, cap_q, w.n_amb, n_q, threshold, w8.md_c, w8.am_c, w8.va_c,
                              w8.nested, w8.nested_bytes, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(match_scatter8_kernel, dim3(cap_a / 256, B), dim3(256), 0, st, cap_a, w.n_amb, w.amb_idx, w8.md_c, w8.am_c, w8.va_c,
                       min_dist, argmin, valid);
    ORYON_CHECK_LAUNCH();
    return ORYON_OK;
}

// ------------------------------------------------------------------------------------------------ K1s8 on K0v3 operands (no fp32 query rows)
namespace oryon {
int gather_q8_launch(const float *feat, int n_maps, int C, int HW, int layout, const int32_t *roi, int roi_stride, const int32_t *count,
                     const int32_t *map_enable, int rows_cap, int C_pad, int8_t *out8, float *scale, float *eps, float *norm,
                     float *out32, int lanes_per_row, int round_f16, hipStream_t st, int fmt = 0, void *aux = nullptr);
}

namespace {
struct Screen8RawWs {
    Screen8Ws base;
    float *q_hat, *scale_scratch, *eps_scratch;
    int8_t *q8_scratch;
    int32_t *need_f32;
    size_t bytes;
};

Screen8RawWs carve_screen8_raw(void *base, int B, int C, int cap_a, int cap_q, int S)
{
    Screen8RawWs w;
    w.base = carve_screen8(base, B, C, cap_a, cap_q, S);
    char *p = static_cast<char *>(base);
    size_t off = (w.base.bytes + 255) / 256 * 256;
    auto take = [&](size_t n) { size_t o = off; off = (off + n + 255) / 256 * 256; return o; };
    const size_t o_qh = take((size_t)B * cap_q * C * sizeof(float));
    const size_t o_q8 = take((size_t)B * cap_q * C);                       // the fall-back pass rewrites the same int8 rows here
    const size_t o_sc = take((size_t)B * (cap_q / 16) * sizeof(float));
    const size_t o_ep = take((size_t)B * sizeof(float));
    const size_t o_nf = take((size_t)B * sizeof(int32_t));
    w.bytes = off;
    w.q_hat = base ? reinterpret_cast<float *>(p + o_qh) : nullptr;
    w.q8_scratch = base ? reinterpret_cast<int8_t *>(p + o_q8) : nullptr;
    w.scale_scratch = base ? reinterpret_cast<float *>(p + o_sc) : nullptr;
    w.eps_scratch = base ? reinterpret_cast<float *>(p + o_ep) : nullptr;
    w.need_f32 = base ? reinterpret_cast<int32_t *>(p + o_nf) : nullptr;
    return w;
}
}  // namespace

extern "C" size_t oryon_match_screened8_raw_workspace_bytes(int B, int C, int cap_a, int cap_q)
{
    if (B <= 0 || C <= 0 || cap_a <= 0 || cap_a % MT16 || cap_q <= 0) return 0;
    return carve_screen8_raw(nullptr, B, C, cap_a, cap_q, pick_split16(B, cap_a / MT16)).bytes;
}

extern "C" int oryon_match_screened8_raw(const float *a_hat, const int8_t *a_i8, const float *a_scale, const float *feat_q, int C_true,
                                         int HW, int layout, const int32_t *roi_q, int roi_stride, const float *q_norm,
                                         const int8_t *q_i8, const float *q_scale, const float *q_eps_max, int B, int C, int cap_a,
                                         int cap_q, const int32_t *n_a, const int32_t *n_q, float threshold, float *min_dist,
                                         int32_t *argmin, uint8_t *valid, int32_t *n_undecided, int round_f16, void *workspace,
                                         size_t workspace_bytes, void *stream)
{
    ORYON_CHECK_ARG(a_hat && a_i8 && a_scale && feat_q && roi_q && q_norm && q_i8 && q_scale && q_eps_max && n_a && n_q);
    ORYON_CHECK_ARG(min_dist && argmin && valid && B >= 0 && (C == 256 || C == 512) && C_true > 0 && C_true <= C && HW > 0);
    ORYON_CHECK_ARG(layout == ORYON_LAYOUT_NCHW || layout == ORYON_LAYOUT_NHWC);
    ORYON_CHECK_ARG(cap_a > 0 && cap_a % MT16 == 0 && cap_q > 0 && cap_q % 256 == 0 && threshold > 0.0f && threshold <= 0.5f);
    if (B == 0) return ORYON_OK;
    const int T = cap_a / MT16;
    const int S = pick_split16(B, T);
    Screen8RawWs wr = carve_screen8_raw(workspace, B, C, cap_a, cap_q, S);
    if (!workspace || workspace_bytes < wr.bytes) {
        set_error("oryon_match_screened8_raw: workspace too small (%zu < %zu)", workspace_bytes, wr.bytes);
        return ORYON_ERR_WORKSPACE;
    }
    Screen8Ws &w8 = wr.base;
    ScreenWs &w = w8.top;
    hipStream_t st = as_stream(stream);
    ORYON_CHECK_HIP(hipMemsetAsync(static_cast<char *>(workspace) + w.zero_off, 0, w.zero_bytes, st));
    ORYON_CHECK_HIP(hipMemsetAsync(wr.need_f32, 0, (size_t)B * sizeof(int32_t), st));
    const float cut0 = 1.0f - 2.0f * threshold;
    const float valid_cut16 = cut0 - SCREEN_DELTA - 1e-6f;
    const int groups = ((B * S + 7) / 8) * 8 * T;
    profile_begin(st, C == 256 ? screen8_name<256>() : screen8_name<512>());
    if (C == 256) launch_screen8<256>(groups, st, a_i8, q_i8, q_scale, B, cap_a, cap_q, n_a, n_q, T, S, w.ws_max, w.ws_i1, w.ws_m2);
    else launch_screen8<512>(groups, st, a_i8, q_i8, q_scale, B, cap_a, cap_q, n_a, n_q, T, S, w.ws_max, w.ws_i1, w.ws_m2);
    profile_end(st);
    ORYON_CHECK_LAUNCH();
    hipLaunchKernelGGL((match_decide_kernel<128>), dim3(cap_a / 64, B), dim3(256), 0, st, static_cast<const __half *>(nullptr),
                       static_cast<const __half *>(nullptr), C, cap_a, cap_q, n_a, n_q, S, valid_cut16,
                       w.ws_max, w.ws_i1, w.ws_m2, w.m_final, w.cnt, w.cand, w.n_amb, w.amb_idx, a_scale, nullptr, q_eps_max, cut0,
                       sqrtf((float)C_true), (float)C_true, a_i8, q_i8, q_scale);
    ORYON_CHECK_LAUNCH();
    static const int resc_l = dev_env_int("ORYON_RESCORE_LANES", 2);
#define RESCORE_RAW(LV, NHWCV)                                                                                                 \
    hipLaunchKernelGGL((match_rescore_raw_kernel<LV, NHWCV>), dim3(cap_a / (256 / LV), B), dim3(256), 0, st, a_hat, feat_q, C_true, HW, \
                       roi_q, roi_stride, q_norm, C, cap_a, cap_q, n_a, n_q, threshold, w.m_final, w.cnt, w.cand, min_dist, argmin,  \
                       valid, w.row_flag, w.panel_flag, wr.need_f32, round_f16)
    if (layout == ORYON_LAYOUT_NHWC) { if (resc_l == 1) RESCORE_RAW(1, true); else if (resc_l == 4) RESCORE_RAW(4, true); else RESCORE_RAW(2, true); }
    else { if (resc_l == 1) RESCORE_RAW(1, false); else if (resc_l == 4) RESCORE_RAW(4, false); else RESCORE_RAW(2, false); }
#undef RESCORE_RAW
    ORYON_CHECK_LAUNCH();
    if (n_undecided) ORYON_CHECK_HIP(hipMemcpyAsync(n_undecided, w.n_amb, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    // Fall-backs (both rare, both gated per pair on the device): pairs with undecided anchors or an overflowed candidate list get their
    // canonical fp32 query rows materialised now - the price round 1 paid for EVERY pair - and then take the round-1 route.
    hipLaunchKernelGGL(match_need_f32_kernel, dim3((B + 255) / 256), dim3(256), 0, st, B, w.n_amb, wr.need_f32);
    int rc = gather_q8_launch(feat_q, B, C_true, HW, layout, roi_q, roi_stride, n_q, wr.need_f32, cap_q, C, wr.q8_scratch, wr.scale_scratch,
                              wr.eps_scratch, nullptr, wr.q_hat, 1, round_f16, st);
    if (rc) { set_error("oryon_match_screened8_raw: fall-back gather launch failed"); return rc; }
    rc = match_f32_flagged(a_hat, wr.q_hat, B, C, cap_a, cap_q, n_a, n_q, threshold, min_dist, argmin, valid, w.panel_flag, w.row_flag,
                           stream);
    if (rc) return rc;
    hipLaunchKernelGGL(match_compact8_kernel, dim3(cap_a / 64, B), dim3(256), 0, st, a_hat, static_cast<const __half *>(nullptr), C, cap_a,
                       w.n_amb, w.amb_idx, w8.a_hat_c, w.a16c);
    hipLaunchKernelGGL(match_make_q16_kernel, dim3(64, B), dim3(256), 0, st, wr.q_hat, C, cap_q, n_q, w.n_amb, w8.q16);
    ORYON_CHECK_LAUNCH();
    rc = oryon_match_screened(w8.a_hat_c, wr.q_hat, w.a16c, w8.q16, B, C, cap_a, cap_q, w.n_amb, n_q, threshold, w8.md_c, w8.am_c, w8.va_c,
                              w8.nested, w8.nested_bytes, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(match_scatter8_kernel, dim3(cap_a / 256, B), dim3(256), 0, st, cap_a, w.n_amb, w.amb_idx, w8.md_c, w8.am_c, w8.va_c,
                       min_dist, argmin, valid);
    ORYON_CHECK_LAUNCH();
    return ORYON_OK;
}

// ------------------------------------------------------------------------------------------------ lazy tail: K1s8 -> sampled correspondences
// The batched engine consumes the matcher through oryon_select_corrs only: it needs the VALID FLAG of every anchor and the argmin of the
// <= max_corrs anchors that get sampled (utils/pcd.py:205-214).  The int8 bound decides validity outright for almost every anchor:
//     m1 - DELTA8 > 1 - 2 thr   =>  the exact distance is below the threshold   (valid, whatever the argmin)
//     m1 + DELTA8 < 1 - 2 thr   =>  it is not                                   (as before)
// so candidate generation (16 int8 rows per anchor) and exact re-scoring (256 strided reads per candidate) are deferred to the sampled
// anchors - 500 per pair instead of 5000.  Anchors whose validity the bound cannot settle are resolved (exactly) before the sampling;
// a pair in which a possibly-valid anchor is AMBIGUOUS (runner-up slice within the int8 margin: its argmin needs the fp16 stage) takes
// the eager route of oryon_match_screened8_raw for all of its anchors.  Outputs are what select(match_screened8_raw(...)) gives,
// bit for bit: same valid set, same sampled rows (the sampling keys depend on the valid set only), same argmin for every sampled row.
namespace oryon {
int select_corrs_launch(const int32_t *roi_a, const int32_t *roi_q, int roi_stride_a, int roi_stride_q, const int32_t *n_a,
                        const int32_t *n_q, const int32_t *argmin, const uint8_t *valid, int cap_a, int B, int W, int max_corrs,
                        int corr_rows, uint64_t seed, const int64_t *pair_key, int32_t *scratch, int32_t *corrs, int32_t *n_valid,
                        int32_t *n_sel, int32_t *status, int32_t *sel_rows, const int32_t *pair_eager, hipStream_t st);

// INVALID / VALID: settled by the int8 bound.  UNCERTAIN: unambiguous winner slice, validity not settled (resolved from that slice before
// the sampling).  AMB_VALID: validity settled (valid) but the runner-up slice is within the int8 margin - the exact argmin is computed
// only if the row gets sampled.  AMB_UNCERTAIN: neither settled - resolved before the sampling.  Both AMB kinds are resolved by the EXACT
// fp32 scan (K1) on a compacted list of just those anchor rows against the pair's materialised fp32 query rows.
constexpr uint8_t LZ_INVALID = 0, LZ_VALID = 1, LZ_UNCERTAIN = 2, LZ_AMB_VALID = 3, LZ_RESOLVED = 4, LZ_AMB_UNCERTAIN = 5;

// one THREAD per anchor: merge the per-split (m1, slice, m2) triples and classify
__global__ __launch_bounds__(256) void match_decide_lite_kernel(
    int cap_a, const int32_t *__restrict__ n_a, int S, const float *__restrict__ ws_m1, const int32_t *__restrict__ ws_i1,
    const float *__restrict__ ws_m2, const float *__restrict__ a_scale8, const float *__restrict__ eps_q8, float cut0, float sqrt_c,
    float c_true, int force_eager, int fmt, int x3, float *__restrict__ m_final, int32_t *__restrict__ sid_final, float *__restrict__ margin_out,
    uint8_t *__restrict__ state, uint8_t *__restrict__ valid, float *__restrict__ min_dist, int32_t *__restrict__ argmin,
    int32_t *__restrict__ pair_eager, int32_t *__restrict__ n_unc, int32_t *__restrict__ unc_idx, int32_t *__restrict__ n_ambu,
    int32_t *__restrict__ ambu_idx, int32_t *__restrict__ need_f32_lazy, int32_t *__restrict__ n_amb_total)
{
    const int p = blockIdx.y, a = blockIdx.x * 256 + threadIdx.x;
    if (a >= n_a[p]) return;
    const size_t arow = (size_t)p * cap_a + a;
    float m1 = -INFINITY, m2 = -INFINITY;
    int sid = 0;
    for (int s = 0; s < S; ++s) {
        const size_t o = ((size_t)p * S + s) * cap_a + a;
        const float x1 = ws_m1[o], x2 = ws_m2[o];
        m2 = fmaxf(fminf(m1, x1), fmaxf(m2, x2));
        if (x1 > m1) { m1 = x1; sid = ws_i1[o]; }
    }
    float delta;
    if (fmt == 1) {
        // mx6 screen: scores are dequantised dot products; a_scale8 / eps_q8 hold the pair's largest measured row error |e|_2 of the
        // anchor / query rows (K0, FMT = 1): |s6 - a^.q^| <= |ea| + |eq| + |ea||eq| + fp32 accumulation slack
        const float ea = a_scale8[p], eq = eps_q8[p];
        delta = ea + eq + ea * eq + 1.2e-4f;       // + fp32 accumulation of <= 512 products in the MFMA and in the canonical chain (<= 7e-5)
    } else {
        const float sa = a_scale8[(size_t)p * (cap_a / 16) + (a >> 5) * 2 + ((a >> 2) & 1)];
        m1 *= sa;
        m2 *= sa;
        const float ea = 0.50003f * sa, eq = 1.00006f * eps_q8[p];
        delta = (ea + eq) * sqrt_c + c_true * ea * eq + 4e-5f;
    }
    const bool usable = delta < 0.2f;
    const float margin = usable ? 2.0f * delta + 2e-7f : INFINITY;
    m_final[arow] = m1;
    sid_final[arow] = sid;
    margin_out[arow] = margin;
    uint8_t st;
    const bool certain_valid = usable && m1 > cut0 + delta + 1e-5f;
    if (usable && !(m1 >= cut0 - delta - 1e-6f)) st = LZ_INVALID;
    else if (!(m1 - m2 > margin)) st = certain_valid ? LZ_AMB_VALID : LZ_AMB_UNCERTAIN;
    else if (certain_valid) st = LZ_VALID;
    else st = LZ_UNCERTAIN;
    state[arow] = st;
    // provisional outputs: the distance is the screening estimate until (unless) the row is resolved exactly
    min_dist[arow] = __fmaf_rn(-0.5f, m1, 0.5f);
    argmin[arow] = 0;
    valid[arow] = (st == LZ_VALID || st == LZ_AMB_VALID) ? 1 : 0;
    if (force_eager) { pair_eager[p] = 1; return; }
    if (st == LZ_UNCERTAIN) unc_idx[(size_t)p * cap_a + atomicAdd(&n_unc[p], 1)] = a;
    if (st == LZ_AMB_UNCERTAIN) ambu_idx[(size_t)p * cap_a + atomicAdd(&n_ambu[p], 1)] = a;
    if (st == LZ_AMB_VALID || st == LZ_AMB_UNCERTAIN) {
        // this pair's fp32 query rows get materialised (device-gated launch) - with the fp16x3 second level (x3) only when an
        // ambiguous anchor's VALIDITY is open too: the sampled valid ones are then resolved from hi / lo half rows made later
        if (st == LZ_AMB_UNCERTAIN || !x3) need_f32_lazy[p] = 1;
        atomicAdd(&n_amb_total[p], 1);
    }
}

__global__ void match_mask_counts_kernel(int B, const int32_t *__restrict__ n_a, const int32_t *__restrict__ pair_eager,
                                         int32_t *__restrict__ n_a_eager, int32_t *__restrict__ n_a_lazy)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= B) return;
    n_a_eager[p] = pair_eager[p] ? n_a[p] : 0;
    n_a_lazy[p] = pair_eager[p] ? 0 : n_a[p];
}

__global__ void match_cascade_counts_kernel(int B, const int32_t *__restrict__ n_amb_total, const int32_t *__restrict__ n_after,
                                            const int32_t *__restrict__ n_before, int32_t *__restrict__ out)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= B) return;
    const long long tot = n_amb_total[p], nb = n_before[p], na = n_after[p];
    out[p] = nb > 0 ? (int32_t)(tot * na / nb) : (int32_t)tot;
}

__global__ void match_sum_counts_kernel(int B, const int32_t *__restrict__ a, const int32_t *__restrict__ b, int32_t *__restrict__ out)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < B) out[p] = a[p] + b[p];
}

// fp32 anchor rows of a work list -> dense panels for the exact scan (rows [count, round_up(count, 128)) zero-filled)
__global__ __launch_bounds__(256) void match_compact_f32_kernel(const float *__restrict__ a_hat, int Cp, int cap_a, int cap_c,
                                                                 const int32_t *__restrict__ count, const int32_t *__restrict__ idx,
                                                                 int idx_stride, float *__restrict__ a_c)
{
    const int p = blockIdx.y, lane = threadIdx.x & 63;
    const int n = count[p] < cap_c ? count[p] : cap_c;
    const int n_fill = (n + 127) / 128 * 128;
    for (int g = 0; g < 16; ++g) {
        const int sl = (blockIdx.x * 16 + g) * 4 + (threadIdx.x >> 6);
        if (sl >= n_fill || sl >= cap_c) break;
        uint4 *d = reinterpret_cast<uint4 *>(a_c + ((size_t)p * cap_c + sl) * Cp);
        if (sl >= n) {
            for (int i = lane; i < Cp / 4; i += 64) d[i] = make_uint4(0, 0, 0, 0);
            continue;
        }
        const uint4 *src = reinterpret_cast<const uint4 *>(a_hat + ((size_t)p * cap_a + idx[(size_t)p * idx_stride + sl]) * Cp);
        for (int i = lane; i < Cp / 4; i += 64) d[i] = src[i];
    }
}

// results of an exact scan on a compacted work list -> the anchors' rows; the rows are marked RESOLVED (argmin / min_dist exact)
__global__ __launch_bounds__(256) void match_scatter_exact_kernel(int cap_a, int cap_c, const int32_t *__restrict__ count,
                                                                   const int32_t *__restrict__ idx, int idx_stride,
                                                                   const float *__restrict__ md_c, const int32_t *__restrict__ am_c,
                                                                   const uint8_t *__restrict__ va_c, float *__restrict__ min_dist,
                                                                   int32_t *__restrict__ argmin, uint8_t *__restrict__ valid,
                                                                   uint8_t *__restrict__ state)
{
    const int p = blockIdx.y, sl = blockIdx.x * 256 + threadIdx.x;
    const int n = count[p] < cap_c ? count[p] : cap_c;
    if (sl >= n) return;
    const size_t src = (size_t)p * cap_c + sl, dst = (size_t)p * cap_a + idx[(size_t)p * idx_stride + sl];
    min_dist[dst] = md_c[src];
    argmin[dst] = am_c[src];
    valid[dst] = va_c[src];
    state[dst] = LZ_RESOLVED;
}

// the sampled slots whose anchor row is AMB_VALID (argmin still unknown) -> work list for the second level (K1x3 / exact scan).  One
// workgroup per pair.  The list comes out in ASCENDING anchor-row order, i.e. in image order: neighbouring anchors share a wave of
// match_x3_scan_kernel, their matches are neighbours in the query map, and the scan can skip the query tiles none of a wave's anchors
// can match (it is also deterministic; the first version appended in atomic order).
__global__ __launch_bounds__(256) void match_list_sampled_amb_kernel(int cap_a, const uint8_t *__restrict__ state,
                                                                      const int32_t *__restrict__ pair_eager, const int32_t *__restrict__ n_sel,
                                                                      const int32_t *__restrict__ sel_rows, int corr_rows,
                                                                      int32_t *__restrict__ mark, int32_t *__restrict__ n_list,
                                                                      int32_t *__restrict__ list)
{
    __shared__ int wave_cnt[4];
    const int p = blockIdx.x;
    if (pair_eager[p]) return;
    const int n = n_sel[p], t = threadIdx.x, lane = t & 63, wave = t >> 6;
    // a row drawn several times (sampling with replacement) is marked once
    for (int s = t; s < n; s += 256) {
        const int a = sel_rows[(size_t)p * corr_rows + s];
        if (state[(size_t)p * cap_a + a] == LZ_AMB_VALID) mark[(size_t)p * cap_a + a] = 1;
    }
    __threadfence_block();
    __syncthreads();
    // ordered compaction in ONE scan: thread t owns the consecutive rows [t R, (t + 1) R), counts its marks, the block scans the 256 counts
    const int R = (cap_a + 255) / 256;
    const int32_t *mk = mark + (size_t)p * cap_a;
    int mine = 0;
    for (int i = 0; i < R; ++i) {
        const int a = t * R + i;
        mine += (a < cap_a && mk[a] != 0) ? 1 : 0;
    }
    int incl = mine;                                            // inclusive scan inside the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off);
        if (lane >= off) incl += v;
    }
    if (lane == 63) wave_cnt[wave] = incl;
    __syncthreads();
    int off0 = incl - mine;
    for (int w = 0; w < wave; ++w) off0 += wave_cnt[w];
    for (int i = 0; i < R; ++i) {
        const int a = t * R + i;
        if (a < cap_a && mk[a] != 0) list[(size_t)p * corr_rows + off0++] = a;
    }
    if (t == 255) n_list[p] = off0;
}

// ---- validity cascade, second pass (round 6; oryon_match_corrs_mx6_x3).  The first screening pass stops a panel once its anchors are all
// valid for sure and leaves their runner-up open (match_mx6_screen_w4_kernel<.., EXIT>); the sampled ones among them (the list of
// match_list_sampled_amb_kernel, <= corr_rows per pair) get the COMPLETE screen here: their operand rows compacted into one panel per pair,
// the same kernel over all query tiles, and the (m1, slice, m2) triples merged back into the per-anchor arrays - a row whose winner turns
// out unambiguous becomes LZ_VALID (resolved from its winning slice like any other), the others keep LZ_AMB_VALID with their true winning
// slice as the second level's seed.
// after the windowed first launch: is every live anchor of a 1024-row panel valid for sure already (its best score over the splits' windows
// above match_decide_lite_kernel's line)?  Then gate = 0 and its runner-ups read +inf (no margin: a partial scan rules nothing out);
// otherwise gate = 1: the gated launch scans everything for this panel and overwrites the triples.
__global__ __launch_bounds__(256) void match_panel_settle_kernel(int cap_a, int T8, int S, const int32_t *__restrict__ n_a,
                                                                  const float *__restrict__ ws_m1, float *__restrict__ ws_m2,
                                                                  const float *__restrict__ a_err, const float *__restrict__ q_err,
                                                                  float cut0, int32_t *__restrict__ gate)
{
    __shared__ int bad;
    const int p = blockIdx.y, panel = blockIdx.x, t = threadIdx.x;
    const int na = n_a[p], a0 = panel * 1024;
    if (a0 >= na) return;
    if (t == 0) bad = 0;
    __syncthreads();
    const float ea = a_err[p], eq = q_err[p];
    const float delta = ea + eq + ea * eq + 1.2e-4f;
    const float thr = delta < 0.2f ? cut0 + delta + 2e-5f : INFINITY;      // (match_decide_lite_kernel: m1 > cut0 + delta + 1e-5)
    int mine = 0;
    for (int a = a0 + t; a < a0 + 1024 && a < na; a += 256) {
        float m1 = -INFINITY;
        for (int s_ = 0; s_ < S; ++s_) m1 = fmaxf(m1, ws_m1[((size_t)p * S + s_) * cap_a + a]);
        mine |= !(m1 > thr);
    }
    if (mine) atomicOr(&bad, 1);
    __syncthreads();
    const int open = bad;
    if (t == 0) gate[p * T8 + panel] = open;
    if (!open)
        for (int a = a0 + t; a < a0 + 1024 && a < na; a += 256)
            for (int s_ = 0; s_ < S; ++s_) ws_m2[((size_t)p * S + s_) * cap_a + a] = INFINITY;
}

__global__ __launch_bounds__(256) void match_compact_rows_kernel(const uint8_t *__restrict__ rows, int row_bytes, int cap_a, int cap_c,
                                                                  const int32_t *__restrict__ count, const int32_t *__restrict__ idx,
                                                                  int idx_stride, uint8_t *__restrict__ out)
{
    // 16 lanes per 256-byte row (uint4 each); rows [count, cap_c) are zero rows (exponent byte 0: finite scores nobody reads)
    const int p = blockIdx.y, sl = blockIdx.x * 16 + (threadIdx.x >> 4), l = threadIdx.x & 15;
    if (sl >= cap_c) return;
    const int n = count[p] < cap_c ? count[p] : cap_c;
    uint4 *d = reinterpret_cast<uint4 *>(out + ((size_t)p * cap_c + sl) * row_bytes);
    if (sl >= n) {
        for (int i = l; i < row_bytes / 16; i += 16) d[i] = make_uint4(0, 0, 0, 0);
        return;
    }
    const uint4 *src = reinterpret_cast<const uint4 *>(rows + ((size_t)p * cap_a + idx[(size_t)p * idx_stride + sl]) * row_bytes);
    for (int i = l; i < row_bytes / 16; i += 16) d[i] = src[i];
}

__global__ __launch_bounds__(256) void match_decide_sampled_kernel(int cap_a, int cap_c, int S, const int32_t *__restrict__ count,
                                                                    const int32_t *__restrict__ idx, int idx_stride,
                                                                    const float *__restrict__ ws_m1, const int32_t *__restrict__ ws_i1,
                                                                    const float *__restrict__ ws_m2, const float *__restrict__ a_err,
                                                                    const float *__restrict__ q_err, float *__restrict__ m_final,
                                                                    int32_t *__restrict__ sid_final, float *__restrict__ margin_out,
                                                                    uint8_t *__restrict__ state, int32_t *__restrict__ mark,
                                                                    int32_t *__restrict__ count_before)
{
    const int p = blockIdx.y, sl = blockIdx.x * 256 + threadIdx.x;
    const int n = count[p] < cap_c ? count[p] : cap_c;
    if (sl == 0) count_before[p] = n;
    if (sl >= n) return;
    const int a = idx[(size_t)p * idx_stride + sl];
    float m1 = -INFINITY, m2 = -INFINITY;
    int sid = 0;
    for (int s = 0; s < S; ++s) {
        const size_t o = ((size_t)p * S + s) * cap_c + sl;
        const float x1 = ws_m1[o], x2 = ws_m2[o];
        m2 = fmaxf(fminf(m1, x1), fmaxf(m2, x2));
        if (x1 > m1) { m1 = x1; sid = ws_i1[o]; }
    }
    const float ea = a_err[p], eq = q_err[p];
    const float delta = ea + eq + ea * eq + 1.2e-4f;                  // match_decide_lite_kernel, fmt 1
    const float margin = delta < 0.2f ? 2.0f * delta + 2e-7f : INFINITY;
    const size_t arow = (size_t)p * cap_a + a;
    m_final[arow] = m1;                                                // the complete scan's maximum (>= the partial one that settled validity)
    sid_final[arow] = sid;
    margin_out[arow] = margin;
    if (m1 - m2 > margin) {                                            // unambiguous after all: no second level for this row
        state[arow] = LZ_VALID;
        mark[arow] = 0;
    }
}

// Exact resolution of ONE unambiguous anchor by one wave: candidates = rows of the winning 16-row slice within the int8 margin of its
// maximum (re-scored from the int8 rows, as match_decide_kernel does), then the canonical fp32 chain per candidate on x_k / d read
// from the raw map (as match_rescore_raw_kernel does).  Returns (distance, first index of the minimum) in lane 0.
typedef float f32x32r __attribute__((ext_vector_type(32)));
typedef unsigned u32x6r __attribute__((ext_vector_type(6)));

// dequantised dot product of two mx6 slots (32 channels): codes decoded by the conversion instruction (exact: multiples of 1/8 up to
// 7.5), 32 exact products summed in fp32 (every partial sum is a multiple of 1/64 below 2^11: exact), times both block exponents
__device__ __forceinline__ float mx6_block_dot(const uint4 a_lo, const uint4 a_up, const uint4 q_lo, const uint4 q_up)
{
    const u32x6r ca = {a_lo.x, a_lo.y, a_lo.z, a_lo.w, a_up.x, a_up.y}, cq = {q_lo.x, q_lo.y, q_lo.z, q_lo.w, q_up.x, q_up.y};
    const f32x32r fa = __builtin_amdgcn_cvt_scalef32_pk32_f32_fp6(ca, 1.0f), fq = __builtin_amdgcn_cvt_scalef32_pk32_f32_fp6(cq, 1.0f);
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < 32; ++i) sum = __fmaf_rn(fa[i], fq[i], sum);
    return sum * ldexpf(1.0f, (int)(a_up.z & 255u) + (int)(q_up.z & 255u) - 254);
}

template <bool NHWC, bool NEED_DIST = true, int FMT = 0>
__device__ __forceinline__ void resolve_anchor(int p, int a, const float *__restrict__ a_hat, const int8_t *__restrict__ a8,
                                               const int8_t *__restrict__ q8, const float *__restrict__ q_scale8,
                                               const float *__restrict__ a_scale8, const float *__restrict__ feat_q, int C_true, int HW,
                                               const int32_t *__restrict__ roi_q, int roi_stride, const float *__restrict__ norm_q,
                                               int Cp, int cap_a, int cap_q, int nq, float m1, int sid, float margin, float *lds /*[2*Cp]*/,
                                               int round_f16, float &d_out, int &j_out)
{
    const int lane = threadIdx.x & 63;
    const size_t arow = (size_t)p * cap_a + a;
    const int half = sid & 1, blk = sid >> 1;
    const int r = lane >> 2, seg = lane & 3;
    const int q = blk * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
    bool hit;
    if constexpr (FMT == 1) {
        // mx6 rows: the 16 rows of the winning slice re-scored from the very operands the screen multiplied (software sum: the order of
        // the additions differs from the MFMA's, inside the bound's slack); candidates = rows within the margin of the SLICE maximum
        // (every exact minimiser lies in this slice and scores at least max - 2 delta)
        float s6 = 0.0f;
        if (q < nq) {
            const uint4 *ar = reinterpret_cast<const uint4 *>(a8 + arow * Cp) + seg * (Cp / 64);
            const uint4 *qr = reinterpret_cast<const uint4 *>(q8 + ((size_t)p * cap_q + q) * Cp) + seg * (Cp / 64);
            for (int b = 0; b < Cp / 128; ++b) s6 += mx6_block_dot(ar[2 * b], ar[2 * b + 1], qr[2 * b], qr[2 * b + 1]);
        }
        s6 += __shfl_xor(s6, 1);
        s6 += __shfl_xor(s6, 2);
        float mx = (q < nq) ? s6 : -INFINITY;
        mx = fmaxf(mx, __shfl_xor(mx, 4));
        mx = fmaxf(mx, __shfl_xor(mx, 8));
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        hit = (seg == 0) && (q < nq) && (s6 >= mx - margin - 4e-5f);
        (void)m1;
    } else {
    const float sa = a_scale8[(size_t)p * (cap_a / 16) + (a >> 5) * 2 + ((a >> 2) & 1)];
    int idot = 0;
    if (q < nq) {
        const uint4 *ar = reinterpret_cast<const uint4 *>(a8 + arow * Cp) + seg * (Cp / 64);
        const uint4 *qr = reinterpret_cast<const uint4 *>(q8 + ((size_t)p * cap_q + q) * Cp) + seg * (Cp / 64);
        for (int i0 = 0; i0 < Cp / 64; ++i0) {
            const uint4 av = ar[i0], qv = qr[i0];
            idot = __builtin_amdgcn_sdot4((int)av.x, (int)qv.x, idot, false);
            idot = __builtin_amdgcn_sdot4((int)av.y, (int)qv.y, idot, false);
            idot = __builtin_amdgcn_sdot4((int)av.z, (int)qv.z, idot, false);
            idot = __builtin_amdgcn_sdot4((int)av.w, (int)qv.w, idot, false);
        }
    }
    idot += __shfl_xor(idot, 1);
    idot += __shfl_xor(idot, 2);
    const float s8 = (float)idot * q_scale8[(size_t)p * (cap_q / 16) + sid] * sa;
    hit = (seg == 0) && (q < nq) && (s8 >= m1 - margin);
    }
    unsigned long long hits = __ballot(hit);
    if (!NEED_DIST && __popcll(hits) == 1) {
        // a single row inside the int8 margin IS the argmin (every other row is provably farther): no fp32 work, and none of the 256
        // scattered 4-byte reads of its raw descriptor (the NCHW gather is what bounds this kernel: one 64-byte sector per channel)
        j_out = __shfl(q, __ffsll((long long)hits) - 1);
        d_out = __builtin_nanf("");
        return;
    }
    // anchor row (k-permuted: position 8g + 4h + j holds k = 8g + 2j + h) -> natural order in LDS
    float *A = lds, *Q = lds + Cp;
    for (int pos = lane; pos < Cp; pos += 64) {
        const int g = pos >> 3, hh = (pos >> 2) & 1, jj = pos & 3;
        A[8 * g + 2 * jj + hh] = a_hat[arow * Cp + pos];
    }
    float d = INFINITY;
    int j = 0x7fffffff;
    const float *fq = feat_q + (size_t)p * C_true * HW;
    while (hits) {
        const int src = __ffsll((long long)hits) - 1;
        hits &= hits - 1;
        const int jj = __shfl(q, src);
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
        float dot = 0.0f;                               // every lane runs the same chain on broadcast LDS reads
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
    d_out = d;
    j_out = j;
}

// anchors whose VALIDITY the int8 bound could not settle (pairs on the lazy route only): exact distance now, before the sampling
template <bool NHWC, int FMT = 0>
__global__ __launch_bounds__(256) void match_resolve_uncertain_kernel(
    const float *__restrict__ a_hat, const int8_t *__restrict__ a8, const int8_t *__restrict__ q8, const float *__restrict__ q_scale8,
    const float *__restrict__ a_scale8, const float *__restrict__ feat_q, int C_true, int HW, const int32_t *__restrict__ roi_q,
    int roi_stride, const float *__restrict__ norm_q, int Cp, int cap_a, int cap_q, const int32_t *__restrict__ n_q, float thr,
    const float *__restrict__ m_final, const int32_t *__restrict__ sid_final, const float *__restrict__ margin_in,
    const int32_t *__restrict__ n_unc, const int32_t *__restrict__ unc_idx, const int32_t *__restrict__ pair_eager,
    uint8_t *__restrict__ state, uint8_t *__restrict__ valid, float *__restrict__ min_dist, int32_t *__restrict__ argmin, int round_f16)
{
    extern __shared__ float lds_res[];
    const int p = blockIdx.y;
    if (pair_eager[p]) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = n_unc[p];
    for (int i = blockIdx.x * 4 + wave; i < n; i += gridDim.x * 4) {
        const int a = unc_idx[(size_t)p * cap_a + i];
        const size_t arow = (size_t)p * cap_a + a;
        float d;
        int j;
        resolve_anchor<NHWC, true, FMT>(p, a, a_hat, a8, q8, q_scale8, a_scale8, feat_q, C_true, HW, roi_q, roi_stride, norm_q, Cp, cap_a, cap_q, n_q[p],
                             m_final[arow], sid_final[arow], margin_in[arow], lds_res + wave * 2 * Cp, round_f16, d, j);
        if (lane == 0) {
            min_dist[arow] = d;
            argmin[arow] = j;
            valid[arow] = (d < thr) ? 1 : 0;
            state[arow] = LZ_RESOLVED;
        }
    }
}

// the sampled rows of the lazy pairs: exact argmin -> query half of the correspondence
template <bool NHWC, int FMT = 0>
__global__ __launch_bounds__(256) void match_resolve_selected_kernel(
    const float *__restrict__ a_hat, const int8_t *__restrict__ a8, const int8_t *__restrict__ q8, const float *__restrict__ q_scale8,
    const float *__restrict__ a_scale8, const float *__restrict__ feat_q, int C_true, int HW, const int32_t *__restrict__ roi_q,
    int roi_stride, const float *__restrict__ norm_q, int Cp, int cap_a, int cap_q, const int32_t *__restrict__ n_q, int W,
    const float *__restrict__ m_final, const int32_t *__restrict__ sid_final, const float *__restrict__ margin_in,
    const uint8_t *__restrict__ state, const int32_t *__restrict__ pair_eager, const int32_t *__restrict__ n_sel,
    const int32_t *__restrict__ sel_rows, int corr_rows, float *__restrict__ min_dist, int32_t *__restrict__ argmin,
    int32_t *__restrict__ corrs, int round_f16)
{
    extern __shared__ float lds_res[];
    const int p = blockIdx.y;
    if (pair_eager[p]) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int slot = blockIdx.x * 4 + wave;
    if (slot >= n_sel[p]) return;
    const int a = sel_rows[(size_t)p * corr_rows + slot];
    const size_t arow = (size_t)p * cap_a + a;
    float d = 0.0f;
    int j;
    if (state[arow] == LZ_RESOLVED) {
        j = argmin[arow];
    } else {
        resolve_anchor<NHWC, false, FMT>(p, a, a_hat, a8, q8, q_scale8, a_scale8, feat_q, C_true, HW, roi_q, roi_stride, norm_q, Cp, cap_a, cap_q,
                                    n_q[p], m_final[arow], sid_final[arow], margin_in[arow], lds_res + wave * 2 * Cp, round_f16, d, j);
        if (lane == 0) {                                                // same values from every slot that drew this row
            argmin[arow] = j;
            if (d == d) min_dist[arow] = d;                             // NaN: single candidate, the distance was never needed
        }
    }
    if (lane == 0) {
        j = (j < 0 || j >= n_q[p]) ? 0 : j;                            // cannot happen for a row that passed the validity cut
        const int pq = roi_q[(size_t)p * roi_stride + j];
        corrs[((size_t)p * corr_rows + slot) * 4 + 2] = pq / W;
        corrs[((size_t)p * corr_rows + slot) * 4 + 3] = pq % W;
    }
}
}  // namespace oryon

namespace {
struct LazyWs {
    Screen8RawWs raw;
    float *margin;
    int32_t *sid_final, *pair_eager, *n_unc, *unc_idx, *n_a_eager, *n_a_lazy, *sel_rows, *scratch;
    int32_t *n_ambu, *ambu_idx, *n_ambv, *ambv_idx, *need_f32_lazy, *n_amb_total, *mark;
    void *exact_ws;
    size_t exact_ws_bytes;
    // K1x3 (C_pad 256): hi / lo anchor rows, the overflow fall-back's compact rows and outputs, candidate lists
    __half *x3_ah, *x3_al;
    float *x3_a_ovf, *x3_md_o;
    int32_t *x3_am_o;
    uint8_t *x3_va_o;
    void *x3_scratch;
    uint8_t *state;
    // validity cascade (hard route, C_pad 256): the sampled anchors' mx6 rows as one 512-row panel per pair + the second pass's triples
    uint8_t *cs_panel;
    float *cs_max, *cs_m2;
    int32_t *cs_i1;
    int32_t *n_ambv0, *cs_gate;
    size_t bytes, zero_off, zero_bytes;
};

LazyWs carve_lazy(void *base, int B, int C, int cap_a, int cap_q, int S, int corr_rows)
{
    LazyWs w;
    w.raw = carve_screen8_raw(base, B, C, cap_a, cap_q, S);
    char *p = static_cast<char *>(base);
    size_t off = (w.raw.bytes + 255) / 256 * 256;
    auto take = [&](size_t n) { size_t o = off; off = (off + n + 255) / 256 * 256; return o; };
    const size_t o_mg = take((size_t)B * cap_a * sizeof(float));
    const size_t o_sf = take((size_t)B * cap_a * sizeof(int32_t));
    const size_t o_ui = take((size_t)B * cap_a * sizeof(int32_t));
    const size_t o_st = take((size_t)B * cap_a);
    const size_t o_sr = take((size_t)B * corr_rows * sizeof(int32_t));
    const size_t o_sc = take((size_t)B * cap_a * sizeof(int32_t));
    const size_t o_ne = take((size_t)B * sizeof(int32_t));
    const size_t o_nl = take((size_t)B * sizeof(int32_t));
    const size_t o_au = take((size_t)B * cap_a * sizeof(int32_t));
    const size_t o_av = take((size_t)B * corr_rows * sizeof(int32_t));
    const int cap_s0 = (corr_rows + 127) / 128 * 128;
    const int cap_s = cap_s0 < cap_a ? cap_s0 : cap_a;
    const size_t e1 = oryon_match_workspace_bytes(B, cap_a), e2 = oryon_match_workspace_bytes(B, cap_s);
    w.exact_ws_bytes = e1 > e2 ? e1 : e2;                                  // split-merge scratch of the exact scan (either list capacity)
    const size_t o_ew = take(w.exact_ws_bytes > 16 ? w.exact_ws_bytes : 16);
    const size_t o_xah = take((size_t)B * cap_s * C * sizeof(__half));
    const size_t o_xal = take((size_t)B * cap_s * C * sizeof(__half));
    const size_t o_xao = take((size_t)B * cap_s * C * sizeof(float));
    const size_t o_xmd = take((size_t)B * cap_s * sizeof(float));
    const size_t o_xam = take((size_t)B * cap_s * sizeof(int32_t));
    const size_t o_xva = take((size_t)B * cap_s);
    const size_t o_xsc = take(match_x3_scratch_bytes(B, cap_s, 8, cap_q));
    const bool cascade = C == 256 && corr_rows <= MX6_SAMPLED_PANEL;
    const int csS = mx6_sampled_splits(B);
    const size_t o_csp = take(cascade ? (size_t)B * MX6_SAMPLED_PANEL * C : 16);
    const size_t o_csm = take(cascade ? (size_t)B * csS * MX6_SAMPLED_PANEL * sizeof(float) : 16);
    const size_t o_cs2 = take(cascade ? (size_t)B * csS * MX6_SAMPLED_PANEL * sizeof(float) : 16);
    const size_t o_csi = take(cascade ? (size_t)B * csS * MX6_SAMPLED_PANEL * sizeof(int32_t) : 16);
    const size_t o_na0 = take((size_t)B * sizeof(int32_t));
    const size_t o_gat = take((size_t)B * mx6_panels_per_pair(cap_a) * sizeof(int32_t));
    w.zero_off = off;
    const size_t o_pe = take((size_t)B * sizeof(int32_t));
    const size_t o_nu = take((size_t)B * sizeof(int32_t));
    const size_t o_nau = take((size_t)B * sizeof(int32_t));
    const size_t o_nav = take((size_t)B * sizeof(int32_t));
    const size_t o_nfl = take((size_t)B * sizeof(int32_t));
    const size_t o_nat = take((size_t)B * sizeof(int32_t));
    const size_t o_mk = take((size_t)B * cap_a * sizeof(int32_t));
    w.zero_bytes = off - w.zero_off;
    w.bytes = off;
    auto at = [&](size_t o) { return base ? p + o : nullptr; };
    w.margin = reinterpret_cast<float *>(at(o_mg));
    w.sid_final = reinterpret_cast<int32_t *>(at(o_sf));
    w.unc_idx = reinterpret_cast<int32_t *>(at(o_ui));
    w.state = reinterpret_cast<uint8_t *>(at(o_st));
    w.sel_rows = reinterpret_cast<int32_t *>(at(o_sr));
    w.scratch = reinterpret_cast<int32_t *>(at(o_sc));
    w.n_a_eager = reinterpret_cast<int32_t *>(at(o_ne));
    w.n_a_lazy = reinterpret_cast<int32_t *>(at(o_nl));
    w.pair_eager = reinterpret_cast<int32_t *>(at(o_pe));
    w.n_unc = reinterpret_cast<int32_t *>(at(o_nu));
    w.ambu_idx = reinterpret_cast<int32_t *>(at(o_au));
    w.ambv_idx = reinterpret_cast<int32_t *>(at(o_av));
    w.exact_ws = at(o_ew);
    w.x3_ah = reinterpret_cast<__half *>(at(o_xah));
    w.x3_al = reinterpret_cast<__half *>(at(o_xal));
    w.x3_a_ovf = reinterpret_cast<float *>(at(o_xao));
    w.x3_md_o = reinterpret_cast<float *>(at(o_xmd));
    w.x3_am_o = reinterpret_cast<int32_t *>(at(o_xam));
    w.x3_va_o = reinterpret_cast<uint8_t *>(at(o_xva));
    w.x3_scratch = at(o_xsc);
    w.cs_panel = cascade ? reinterpret_cast<uint8_t *>(at(o_csp)) : nullptr;
    w.cs_max = reinterpret_cast<float *>(at(o_csm));
    w.cs_m2 = reinterpret_cast<float *>(at(o_cs2));
    w.cs_i1 = reinterpret_cast<int32_t *>(at(o_csi));
    w.n_ambv0 = reinterpret_cast<int32_t *>(at(o_na0));
    w.cs_gate = reinterpret_cast<int32_t *>(at(o_gat));
    w.n_ambu = reinterpret_cast<int32_t *>(at(o_nau));
    w.n_ambv = reinterpret_cast<int32_t *>(at(o_nav));
    w.need_f32_lazy = reinterpret_cast<int32_t *>(at(o_nfl));
    w.n_amb_total = reinterpret_cast<int32_t *>(at(o_nat));
    w.mark = reinterpret_cast<int32_t *>(at(o_mk));
    return w;
}
}  // namespace

extern "C" size_t oryon_match_corrs_i8_workspace_bytes(int B, int C, int cap_a, int cap_q, int corr_rows)
{
    if (B <= 0 || C <= 0 || cap_a <= 0 || cap_a % MT16 || cap_q <= 0 || corr_rows <= 0) return 0;
    return carve_lazy(nullptr, B, C, cap_a, cap_q, pick_split16(B, cap_a / MT16), corr_rows).bytes;
}

// fmt 0: int8 rows (a_scale / q_scale per 16-row slice, q_eps_max per pair).  fmt 1: mx6 rows in a_i8 / q_i8, a_scale = the largest
// anchor-row error norm per pair [B], q_eps_max = the largest query-row error norm per pair [B], q_scale unused; lazy route only.
static int match_corrs_lazy_impl(const float *a_hat, const int8_t *a_i8, const float *a_scale, const float *feat_q, int C_true, int HW,
                                 int layout, const int32_t *roi_a, int roi_stride_a, const int32_t *roi_q, int roi_stride_q,
                                 const float *q_norm, const int8_t *q_i8, const float *q_scale, const float *q_eps_max, int B, int C,
                                 int cap_a, int cap_q, const int32_t *n_a, const int32_t *n_q, float threshold, int W, int max_corrs,
                                 int corr_rows, uint64_t seed, const int64_t *pair_key, int force_eager, float *min_dist,
                                 int32_t *argmin, uint8_t *valid, int32_t *corrs, int32_t *n_valid, int32_t *n_sel, int32_t *status,
                                 int32_t *n_undecided, int round_f16, void *workspace, size_t workspace_bytes, void *stream, int fmt,
                                 const void *q_hi_lo = nullptr, const float *q_lo_sq_max = nullptr);

extern "C" int oryon_match_corrs_i8(const float *a_hat, const int8_t *a_i8, const float *a_scale, const float *feat_q, int C_true, int HW,
                                    int layout, const int32_t *roi_a, int roi_stride_a, const int32_t *roi_q, int roi_stride_q,
                                    const float *q_norm, const int8_t *q_i8, const float *q_scale, const float *q_eps_max, int B, int C,
                                    int cap_a, int cap_q, const int32_t *n_a, const int32_t *n_q, float threshold, int W, int max_corrs,
                                    int corr_rows, uint64_t seed, const int64_t *pair_key, int force_eager, float *min_dist,
                                    int32_t *argmin, uint8_t *valid, int32_t *corrs, int32_t *n_valid, int32_t *n_sel, int32_t *status,
                                    int32_t *n_undecided, int round_f16, void *workspace, size_t workspace_bytes, void *stream)
{
    ORYON_CHECK_ARG(a_scale && q_scale);
    return match_corrs_lazy_impl(a_hat, a_i8, a_scale, feat_q, C_true, HW, layout, roi_a, roi_stride_a, roi_q, roi_stride_q, q_norm, q_i8, q_scale,
                                 q_eps_max, B, C, cap_a, cap_q, n_a, n_q, threshold, W, max_corrs, corr_rows, seed, pair_key, force_eager, min_dist,
                                 argmin, valid, corrs, n_valid, n_sel, status, n_undecided, round_f16, workspace, workspace_bytes, stream, 0);
}

extern "C" int oryon_match_corrs_mx6(const float *a_hat, const uint8_t *a_mx6, const float *a_err_max, const float *feat_q, int C_true, int HW,
                                     int layout, const int32_t *roi_a, int roi_stride_a, const int32_t *roi_q, int roi_stride_q,
                                     const float *q_norm, const uint8_t *q_mx6, const float *q_err_max, int B, int C, int cap_a, int cap_q,
                                     const int32_t *n_a, const int32_t *n_q, float threshold, int W, int max_corrs, int corr_rows,
                                     uint64_t seed, const int64_t *pair_key, float *min_dist, int32_t *argmin, uint8_t *valid,
                                     int32_t *corrs, int32_t *n_valid, int32_t *n_sel, int32_t *status, int32_t *n_undecided, int round_f16,
                                     void *workspace, size_t workspace_bytes, void *stream)
{
    ORYON_CHECK_ARG(a_err_max && q_err_max);
    return match_corrs_lazy_impl(a_hat, reinterpret_cast<const int8_t *>(a_mx6), a_err_max, feat_q, C_true, HW, layout, roi_a, roi_stride_a, roi_q,
                                 roi_stride_q, q_norm, reinterpret_cast<const int8_t *>(q_mx6), nullptr, q_err_max, B, C, cap_a, cap_q, n_a, n_q,
                                 threshold, W, max_corrs, corr_rows, seed, pair_key, 0, min_dist, argmin, valid, corrs, n_valid, n_sel, status,
                                 n_undecided, round_f16, workspace, workspace_bytes, stream, 1);
}

extern "C" int oryon_match_corrs_mx6_x3(const float *a_hat, const uint8_t *a_mx6, const float *a_err_max, const float *feat_q, int C_true, int HW,
                                        int layout, const int32_t *roi_a, int roi_stride_a, const int32_t *roi_q, int roi_stride_q,
                                        const float *q_norm, const uint8_t *q_mx6, const float *q_err_max, const void *q_hi_lo_f16,
                                        const float *q_lo_sq_max, int B, int C, int cap_a, int cap_q, const int32_t *n_a, const int32_t *n_q,
                                        float threshold, int W, int max_corrs, int corr_rows, uint64_t seed, const int64_t *pair_key,
                                        float *min_dist, int32_t *argmin, uint8_t *valid, int32_t *corrs, int32_t *n_valid, int32_t *n_sel,
                                        int32_t *status, int32_t *n_undecided, int round_f16, void *workspace, size_t workspace_bytes,
                                        void *stream)
{
    ORYON_CHECK_ARG(a_err_max && q_err_max && q_hi_lo_f16 && q_lo_sq_max && C == 256);
    return match_corrs_lazy_impl(a_hat, reinterpret_cast<const int8_t *>(a_mx6), a_err_max, feat_q, C_true, HW, layout, roi_a, roi_stride_a, roi_q,
                                 roi_stride_q, q_norm, reinterpret_cast<const int8_t *>(q_mx6), nullptr, q_err_max, B, C, cap_a, cap_q, n_a, n_q,
                                 threshold, W, max_corrs, corr_rows, seed, pair_key, 0, min_dist, argmin, valid, corrs, n_valid, n_sel, status,
                                 n_undecided, round_f16, workspace, workspace_bytes, stream, 1, q_hi_lo_f16, q_lo_sq_max);
}

static int match_corrs_lazy_impl(const float *a_hat, const int8_t *a_i8, const float *a_scale, const float *feat_q, int C_true, int HW,
                                 int layout, const int32_t *roi_a, int roi_stride_a, const int32_t *roi_q, int roi_stride_q,
                                 const float *q_norm, const int8_t *q_i8, const float *q_scale, const float *q_eps_max, int B, int C,
                                 int cap_a, int cap_q, const int32_t *n_a, const int32_t *n_q, float threshold, int W, int max_corrs,
                                 int corr_rows, uint64_t seed, const int64_t *pair_key, int force_eager, float *min_dist,
                                 int32_t *argmin, uint8_t *valid, int32_t *corrs, int32_t *n_valid, int32_t *n_sel, int32_t *status,
                                 int32_t *n_undecided, int round_f16, void *workspace, size_t workspace_bytes, void *stream, int fmt,
                                 const void *q_hi_lo, const float *q_lo_sq_max)
{
    ORYON_CHECK_ARG(a_hat && a_i8 && a_scale && feat_q && roi_a && roi_q && q_norm && q_i8 && q_eps_max && n_a && n_q);
    ORYON_CHECK_ARG(min_dist && argmin && valid && corrs && n_valid && n_sel && status && !(fmt == 1 && force_eager));
    ORYON_CHECK_ARG(B >= 0 && (C == 256 || C == 512) && C_true > 0 && C_true <= C && HW > 0 && W > 0 && max_corrs > 0 && corr_rows >= max_corrs);
    ORYON_CHECK_ARG(layout == ORYON_LAYOUT_NCHW || layout == ORYON_LAYOUT_NHWC);
    ORYON_CHECK_ARG(cap_a > 0 && cap_a % MT16 == 0 && cap_q > 0 && cap_q % 256 == 0 && threshold > 0.0f && threshold <= 0.5f);
    if (B == 0) return ORYON_OK;
    const int T = cap_a / MT16;
    const int S = pick_split16(B, T);
    LazyWs lw = carve_lazy(workspace, B, C, cap_a, cap_q, S, corr_rows);
    if (!workspace || workspace_bytes < lw.bytes) {
        set_error("oryon_match_corrs_i8: workspace too small (%zu < %zu)", workspace_bytes, lw.bytes);
        return ORYON_ERR_WORKSPACE;
    }
    Screen8RawWs &wr = lw.raw;
    Screen8Ws &w8 = wr.base;
    ScreenWs &w = w8.top;
    hipStream_t st = as_stream(stream);
    ORYON_CHECK_HIP(hipMemsetAsync(static_cast<char *>(workspace) + w.zero_off, 0, w.zero_bytes, st));
    if (force_eager) ORYON_CHECK_HIP(hipMemsetAsync(wr.need_f32, 0, (size_t)B * sizeof(int32_t), st));      // only the eager route reads it
    ORYON_CHECK_HIP(hipMemsetAsync(static_cast<char *>(workspace) + lw.zero_off, 0, lw.zero_bytes, st));
    const float cut0 = 1.0f - 2.0f * threshold;
    const float valid_cut16 = cut0 - SCREEN_DELTA - 1e-6f;
    const int groups = ((B * S + 7) / 8) * 8 * T;
    // second level for the sampled anchors no screen can separate: fp16x3 two-sweep scan (K1x3, match_x3.hip; C_pad 256) instead of the
    // exact fp32 scan - same results, hard-descriptor step 9.8 -> 8.5 ms; ~30 us of empty launches per step when no anchor needs it.
    // ORYON_AMB_X3=0 keeps the exact scan (the tests run both settings).
    static const bool x3_env = dev_env_int("ORYON_AMB_X3", 1) != 0;
    const int use_x3 = (x3_env && C == 256 && !force_eager) ? 1 : 0;
    // validity cascade (round 6): on the route the engine takes once its feedback says "hard" (q_hi_lo given: K0 wrote the hi / lo rows),
    // the screen stops a panel whose anchors are all valid for sure, and a second, complete pass serves the sampled anchors only
    static const bool cascade_env = dev_env_int("ORYON_CASCADE", 1) != 0;
    const bool cascade = cascade_env && fmt == 1 && use_x3 && q_hi_lo != nullptr && lw.cs_panel != nullptr;
    if (fmt == 1) {
        const uint8_t *a6 = reinterpret_cast<const uint8_t *>(a_i8), *q6 = reinterpret_cast<const uint8_t *>(q_i8);
        profile_begin(st, screen_mx6_name(C));
        if (cascade) {
            // first pass of the cascade: two tiles per (panel, split) near the panel's own place in the map, then the complete scan for the
            // panels that still hold an anchor whose validity is open (device-gated: the others return at once)
            const int T8 = mx6_panels_per_pair(cap_a);
            launch_screen_mx6(C, groups, T, st, a6, q6, B, cap_a, cap_q, n_a, n_q, S, w.ws_max, w.ws_i1, w.ws_m2, C_true, 1, nullptr, 2);
            hipLaunchKernelGGL(match_panel_settle_kernel, dim3(T8, B), dim3(256), 0, st, cap_a, T8, S, n_a, w.ws_max, w.ws_m2, a_scale, q_eps_max,
                               cut0, lw.cs_gate);
            launch_screen_mx6(C, groups, T, st, a6, q6, B, cap_a, cap_q, n_a, n_q, S, w.ws_max, w.ws_i1, w.ws_m2, C_true, 1, lw.cs_gate, 0);
        } else
        launch_screen_mx6(C, groups, T, st, a6, q6, B, cap_a, cap_q, n_a, n_q, S, w.ws_max, w.ws_i1, w.ws_m2, C_true);
        profile_end(st);
    } else {
    profile_begin(st, C == 256 ? screen8_name<256>() : screen8_name<512>());
    if (C == 256) launch_screen8<256>(groups, st, a_i8, q_i8, q_scale, B, cap_a, cap_q, n_a, n_q, T, S, w.ws_max, w.ws_i1, w.ws_m2);
    else launch_screen8<512>(groups, st, a_i8, q_i8, q_scale, B, cap_a, cap_q, n_a, n_q, T, S, w.ws_max, w.ws_i1, w.ws_m2);
    profile_end(st);
    }
    ORYON_CHECK_LAUNCH();
    const float sqrt_c = sqrtf((float)C_true);
    hipLaunchKernelGGL(match_decide_lite_kernel, dim3(cap_a / 256, B), dim3(256), 0, st, cap_a, n_a, S, w.ws_max, w.ws_i1, w.ws_m2, a_scale,
                       q_eps_max, cut0, sqrt_c, (float)C_true, force_eager, fmt, use_x3, w.m_final, lw.sid_final, lw.margin, lw.state, valid, min_dist,
                       argmin, lw.pair_eager, lw.n_unc, lw.unc_idx, lw.n_ambu, lw.ambu_idx, lw.need_f32_lazy, lw.n_amb_total);
    hipLaunchKernelGGL(match_mask_counts_kernel, dim3((B + 255) / 256), dim3(256), 0, st, B, n_a, lw.pair_eager, lw.n_a_eager, lw.n_a_lazy);
    ORYON_CHECK_LAUNCH();
    if (force_eager) {
    // ---- eager route: the complete tail of oryon_match_screened8_raw for every pair (pair_eager is set only by force_eager since the
    // ambiguous anchors of lazy pairs are resolved by compacted exact scans), then the sampler on the complete outputs
    const int32_t *nae = lw.n_a_eager;
    hipLaunchKernelGGL((match_decide_kernel<128>), dim3(cap_a / 64, B), dim3(256), 0, st, static_cast<const __half *>(nullptr),
                       static_cast<const __half *>(nullptr), C, cap_a, cap_q, nae, n_q, S, valid_cut16, w.ws_max, w.ws_i1, w.ws_m2, w.m_final,
                       w.cnt, w.cand, w.n_amb, w.amb_idx, a_scale, nullptr, q_eps_max, cut0, sqrt_c, (float)C_true, a_i8, q_i8, q_scale);
#define RESCORE_RAW_E(NHWCV)                                                                                                   \
    hipLaunchKernelGGL((match_rescore_raw_kernel<2, NHWCV>), dim3(cap_a / 128, B), dim3(256), 0, st, a_hat, feat_q, C_true, HW, roi_q,  \
                       roi_stride_q, q_norm, C, cap_a, cap_q, nae, n_q, threshold, w.m_final, w.cnt, w.cand, min_dist, argmin, valid,    \
                       w.row_flag, w.panel_flag, wr.need_f32, round_f16)
    if (layout == ORYON_LAYOUT_NHWC) RESCORE_RAW_E(true); else RESCORE_RAW_E(false);
#undef RESCORE_RAW_E
    ORYON_CHECK_LAUNCH();
    hipLaunchKernelGGL(match_need_f32_kernel, dim3((B + 255) / 256), dim3(256), 0, st, B, w.n_amb, wr.need_f32);
    int rc = gather_q8_launch(feat_q, B, C_true, HW, layout, roi_q, roi_stride_q, n_q, wr.need_f32, cap_q, C, wr.q8_scratch, wr.scale_scratch,
                              wr.eps_scratch, nullptr, wr.q_hat, 1, round_f16, st);
    if (rc) { set_error("oryon_match_corrs_i8: fall-back gather launch failed"); return rc; }
    rc = match_f32_flagged(a_hat, wr.q_hat, B, C, cap_a, cap_q, nae, n_q, threshold, min_dist, argmin, valid, w.panel_flag, w.row_flag, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(match_compact8_kernel, dim3(cap_a / 64, B), dim3(256), 0, st, a_hat, static_cast<const __half *>(nullptr), C, cap_a,
                       w.n_amb, w.amb_idx, w8.a_hat_c, w.a16c);
    hipLaunchKernelGGL(match_make_q16_kernel, dim3(64, B), dim3(256), 0, st, wr.q_hat, C, cap_q, n_q, w.n_amb, w8.q16);
    ORYON_CHECK_LAUNCH();
    rc = oryon_match_screened(w8.a_hat_c, wr.q_hat, w.a16c, w8.q16, B, C, cap_a, cap_q, w.n_amb, n_q, threshold, w8.md_c, w8.am_c, w8.va_c,
                              w8.nested, w8.nested_bytes, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(match_scatter8_kernel, dim3(cap_a / 256, B), dim3(256), 0, st, cap_a, w.n_amb, w.amb_idx, w8.md_c, w8.am_c, w8.va_c,
                       min_dist, argmin, valid);
    ORYON_CHECK_LAUNCH();
    rc = select_corrs_launch(roi_a, roi_q, roi_stride_a, roi_stride_q, n_a, n_q, argmin, valid, cap_a, B, W, max_corrs, corr_rows, seed,
                             pair_key, lw.scratch, corrs, n_valid, n_sel, status, lw.sel_rows, lw.pair_eager, st);
    if (rc) { set_error("oryon_match_corrs_i8: select launch failed"); return rc; }
    if (n_undecided) ORYON_CHECK_HIP(hipMemcpyAsync(n_undecided, w.n_amb, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    return ORYON_OK;
    }
    // ---- lazy route.  (1) pairs with ambiguous possibly-valid anchors get their fp32 query rows (device-gated, as on the eager route)
    int rc = gather_q8_launch(feat_q, B, C_true, HW, layout, roi_q, roi_stride_q, n_q, lw.need_f32_lazy, cap_q, C, wr.q8_scratch, wr.scale_scratch,
                          wr.eps_scratch, nullptr, wr.q_hat, 1, round_f16, st);
    if (rc) { set_error("oryon_match_corrs_i8: lazy fp32 gather launch failed"); return rc; }
    // (2) ambiguous anchors whose VALIDITY is open: exact fp32 scan (K1) of exactly those rows, before the sampling
    hipLaunchKernelGGL(match_compact_f32_kernel, dim3(cap_a / 64, B), dim3(256), 0, st, a_hat, C, cap_a, cap_a, lw.n_ambu, lw.ambu_idx, cap_a,
                       w8.a_hat_c);
    ORYON_CHECK_LAUNCH();
    rc = oryon_match_f32(w8.a_hat_c, wr.q_hat, B, C, cap_a, cap_q, lw.n_ambu, n_q, threshold, w8.md_c, w8.am_c, w8.va_c, lw.exact_ws,
                         lw.exact_ws_bytes, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(match_scatter_exact_kernel, dim3(cap_a / 256, B), dim3(256), 0, st, cap_a, cap_a, lw.n_ambu, lw.ambu_idx, cap_a, w8.md_c,
                       w8.am_c, w8.va_c, min_dist, argmin, valid, lw.state);
    ORYON_CHECK_LAUNCH();
    // (3) unambiguous anchors whose validity is open: exact distance from the winning slice's candidates
    const size_t lds_res = (size_t)4 * 2 * C * sizeof(float);
#define RESOLVE_U(NHWCV) RESOLVE_U2(NHWCV, 0)
#define RESOLVE_U2(NHWCV, FMTV)                                                                                                \
    hipLaunchKernelGGL((match_resolve_uncertain_kernel<NHWCV, FMTV>), dim3(64, B), dim3(256), lds_res, st, a_hat, a_i8, q_i8, q_scale, a_scale,     \
                       feat_q, C_true, HW, roi_q, roi_stride_q, q_norm, C, cap_a, cap_q, n_q, threshold, w.m_final, lw.sid_final, lw.margin,   \
                       lw.n_unc, lw.unc_idx, lw.pair_eager, lw.state, valid, min_dist, argmin, round_f16)
    if (fmt == 1) { if (layout == ORYON_LAYOUT_NHWC) RESOLVE_U2(true, 1); else RESOLVE_U2(false, 1); }
    else if (layout == ORYON_LAYOUT_NHWC) RESOLVE_U(true); else RESOLVE_U(false);
#undef RESOLVE_U
#undef RESOLVE_U2
    ORYON_CHECK_LAUNCH();
    // (4) the sampling, on the exact valid set
    rc = select_corrs_launch(roi_a, roi_q, roi_stride_a, roi_stride_q, n_a, n_q, argmin, valid, cap_a, B, W, max_corrs, corr_rows, seed,
                             pair_key, lw.scratch, corrs, n_valid, n_sel, status, lw.sel_rows, lw.pair_eager, st);
    if (rc) { set_error("oryon_match_corrs_i8: select launch failed"); return rc; }
    // (5) sampled rows that are ambiguous (valid for sure, argmin open): exact fp32 scan of just those <= max_corrs rows per pair
    const int cap_s0 = (corr_rows + 127) / 128 * 128;
    const int cap_s = cap_s0 < cap_a ? cap_s0 : cap_a;           // distinct sampled rows <= min(max_corrs, n_a)
    hipLaunchKernelGGL(match_list_sampled_amb_kernel, dim3(B), dim3(256), 0, st, cap_a, lw.state, lw.pair_eager, n_sel, lw.sel_rows, corr_rows,
                       lw.mark, lw.n_ambv, lw.ambv_idx);
    if (cascade) {
        // second pass: the complete screen for the listed rows (one 512-row panel per pair), triples merged back, list rebuilt
        const int csS = mx6_sampled_splits(B);
        hipLaunchKernelGGL(match_compact_rows_kernel, dim3(MX6_SAMPLED_PANEL / 16, B), dim3(256), 0, st, reinterpret_cast<const uint8_t *>(a_i8), C,
                           cap_a, MX6_SAMPLED_PANEL, lw.n_ambv, lw.ambv_idx, corr_rows, lw.cs_panel);
        launch_screen_mx6_sampled(st, lw.cs_panel, reinterpret_cast<const uint8_t *>(q_i8), B, cap_q, lw.n_ambv, n_q, csS, lw.cs_max, lw.cs_i1,
                                  lw.cs_m2, C_true);
        hipLaunchKernelGGL(match_decide_sampled_kernel, dim3((MX6_SAMPLED_PANEL + 255) / 256, B), dim3(256), 0, st, cap_a, MX6_SAMPLED_PANEL, csS,
                           lw.n_ambv, lw.ambv_idx, corr_rows, lw.cs_max, lw.cs_i1, lw.cs_m2, a_scale, q_eps_max, w.m_final, lw.sid_final,
                           lw.margin, lw.state, lw.mark, lw.n_ambv0);
        hipLaunchKernelGGL(match_list_sampled_amb_kernel, dim3(B), dim3(256), 0, st, cap_a, lw.state, lw.pair_eager, n_sel, lw.sel_rows, corr_rows,
                           lw.mark, lw.n_ambv, lw.ambv_idx);
        ORYON_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(match_compact_f32_kernel, dim3((cap_s + 63) / 64, B), dim3(256), 0, st, a_hat, C, cap_a, cap_s, lw.n_ambv, lw.ambv_idx,
                       corr_rows, w8.a_hat_c);
    ORYON_CHECK_LAUNCH();
    if (use_x3) {
        // K1x3: hi / lo half query rows into the (now free) fp32-row area, fp16x3 scan with candidate lists, exact chain on the few
        // candidates; anchors whose lists overflowed (duplicate crowds) fall back to the exact scan on fp32 rows materialised for their pair
        __half *qh = reinterpret_cast<__half *>(wr.q_hat), *ql = qh + (size_t)B * cap_q * C;
        int32_t *n_ovf = nullptr, *ovf_idx = nullptr;
        rc = match_x3_resolve(w8.a_hat_c, lw.n_ambv, cap_s, feat_q, C_true, HW, layout, roi_q, roi_stride_q, q_norm, n_q, B, cap_q, threshold,
                              round_f16, qh, ql, lw.x3_ah, lw.x3_al, lw.x3_scratch, w8.md_c, w8.am_c, w8.va_c, &n_ovf, &ovf_idx, lw.ambv_idx, corr_rows,
                              lw.sid_final, cap_a, static_cast<const __half *>(q_hi_lo), q_lo_sq_max, st);
        if (rc) { set_error("oryon_match_corrs: fp16x3 second-level launch failed"); return rc; }
        // fp32 query rows for the pairs with overflowed anchors only: the gather's per-map gate reads n_ovf itself
        rc = gather_q8_launch(feat_q, B, C_true, HW, layout, roi_q, roi_stride_q, n_q, n_ovf, cap_q, C, wr.q8_scratch, wr.scale_scratch,
                              wr.eps_scratch, nullptr, wr.q_hat, 1, round_f16, st);
        if (rc) { set_error("oryon_match_corrs: overflow fp32 gather launch failed"); return rc; }
        hipLaunchKernelGGL(match_compact_f32_kernel, dim3((cap_s + 63) / 64, B), dim3(256), 0, st, w8.a_hat_c, C, cap_s, cap_s, n_ovf, ovf_idx,
                           cap_s, lw.x3_a_ovf);
        ORYON_CHECK_LAUNCH();
        rc = oryon_match_f32(lw.x3_a_ovf, wr.q_hat, B, C, cap_s, cap_q, n_ovf, n_q, threshold, lw.x3_md_o, lw.x3_am_o, lw.x3_va_o, lw.exact_ws,
                             lw.exact_ws_bytes, stream);
        if (rc) return rc;
        match_x3_scatter_ovf(B, cap_s, n_ovf, ovf_idx, lw.x3_md_o, lw.x3_am_o, lw.x3_va_o, w8.md_c, w8.am_c, w8.va_c, st);
    } else {
    rc = oryon_match_f32(w8.a_hat_c, wr.q_hat, B, C, cap_s, cap_q, lw.n_ambv, n_q, threshold, w8.md_c, w8.am_c, w8.va_c, lw.exact_ws,
                         lw.exact_ws_bytes, stream);
    if (rc) return rc;
    }
    hipLaunchKernelGGL(match_scatter_exact_kernel, dim3((cap_s + 255) / 256, B), dim3(256), 0, st, cap_a, cap_s, lw.n_ambv, lw.ambv_idx, corr_rows,
                       w8.md_c, w8.am_c, w8.va_c, min_dist, argmin, valid, lw.state);
    ORYON_CHECK_LAUNCH();
    // (6) query half of every sampled correspondence: resolved rows read their argmin, the others get it from their winning slice
#define RESOLVE_S(NHWCV) RESOLVE_S2(NHWCV, 0)
#define RESOLVE_S2(NHWCV, FMTV)                                                                                                \
    hipLaunchKernelGGL((match_resolve_selected_kernel<NHWCV, FMTV>), dim3((max_corrs + 3) / 4, B), dim3(256), lds_res, st, a_hat, a_i8, q_i8,       \
                       q_scale, a_scale, feat_q, C_true, HW, roi_q, roi_stride_q, q_norm, C, cap_a, cap_q, n_q, W, w.m_final, lw.sid_final,    \
                       lw.margin, lw.state, lw.pair_eager, n_sel, lw.sel_rows, corr_rows, min_dist, argmin, corrs, round_f16)
    if (fmt == 1) { if (layout == ORYON_LAYOUT_NHWC) RESOLVE_S2(true, 1); else RESOLVE_S2(false, 1); }
    else if (layout == ORYON_LAYOUT_NHWC) RESOLVE_S(true); else RESOLVE_S(false);
#undef RESOLVE_S
#undef RESOLVE_S2
    ORYON_CHECK_LAUNCH();
    if (n_undecided && cascade) {
        // cascade: the first pass calls every anchor of a panel that stopped early "ambiguous"; the feedback the engine steers by is that
        // count scaled by the share of the SAMPLED ambiguous rows which the complete second pass left ambiguous
        hipLaunchKernelGGL(match_cascade_counts_kernel, dim3((B + 255) / 256), dim3(256), 0, st, B, lw.n_amb_total, lw.n_ambv, lw.n_ambv0, n_undecided);
        ORYON_CHECK_LAUNCH();
    } else
    if (n_undecided) {       // anchors the int8 stage could not fully decide: the fp16-stage anchors of eager pairs + the ambiguous anchors of lazy pairs
        hipLaunchKernelGGL(match_sum_counts_kernel, dim3((B + 255) / 256), dim3(256), 0, st, B, w.n_amb, lw.n_amb_total, n_undecided);
        ORYON_CHECK_LAUNCH();
    }
    return ORYON_OK;
}
