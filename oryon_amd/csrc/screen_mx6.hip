// K1s6: the MX-fp6 screen of the lazy matcher (round 3) - the dominant kernel of the cfg2 step.  Its own translation unit because it is
// compiled with -fno-honor-nans (Makefile): the running-maximum reductions are chains of fmaxf, and with NaNs honoured the compiler
// quiets every operand it cannot prove canonical (v_max_f32 x, x, x: three extra VALU instructions per 32x32 block) - extra live values
// that push match_mx6_screen_w4_kernel<256, 8> from 246 VGPRs to 52 spilled registers (266 MB of scratch writes per launch in the
// WRITE_SIZE counter, +12 % time).  No NaN can occur here: fp6 e2m3 has no NaN / Inf codes, the E8M0 exponents K0 writes are finite
// (dead rows of a tile are zero rows with exponent byte 0), and a garbage anchor row beyond the pair's count only feeds its own,
// ignored, output column.  match16.hip, which uses a NaN as a marker (resolve_anchor), keeps the default semantics.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "match_common.h"

namespace oryon {

typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr int screen8_tile_bytes(int CP) { return CP * 128; }     // 128 query rows per tile (as the int8 screen, match16.hip)

// ------------------------------------------------------------------------------------------------ K1s6: MX-fp6 screen (round 3)
// The same single-pass (m1, slice, m2) screening on v_mfma_scale_f32_32x32x64_f8f6f4 with fp6 (e2m3) operands: 64 channels per
// instruction at the int8 instruction's issue rate, i.e. twice the multiply-accumulates per matrix-pipe cycle (guide: 8.9 PFLOP/s
// measured for MX-fp6 32x32x64 against 4.4 POP/s for i8 32x32x32).  Operands are K0's mx6 rows (gather8.hip, FMT = 1): per row and
// 32-channel block one 32-byte slot = 24 bytes of codes + the block's E8M0 exponent - the share of one lane (row l & 31, block
// 2 S + (l >> 5) of k-step S), so a lane's A operand is two ds_read_b128 and the hardware applies both exponents: accumulators are
// the dequantised dot products themselves (no per-slice integer scale).  Row bytes, tile geometry, LDS image, DMA, swizzle, block
// map, outputs and the meaning of a slice are those of match_i8_screen_v2_kernel.
//
// Bound.  With e = x^ - dequant(x^) the MEASURED quantisation error of a row (K0 accumulates |e|_2 per row and keeps the largest per
// map), s6_ij - a^_i.q^_j = a^_i.eq_j + ea_i.q^_j + ea_i.eq_j, hence by Cauchy-Schwarz on unit rows
//     |s6_ij - a^_i.q^_j| <= |eq| + |ea| + |ea||eq| (+ 1.2e-4: fp32 accumulation of 256-512 exact products here and in the canonical chain)
// - a 2-norm bound on errors that really occur (~0.02 per row for e2m3 with a per-block exponent) instead of a worst-case 1-norm one.
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16s __attribute__((ext_vector_type(16)));

// VAR: 0 = the product kernel; 1 = DIAGNOSTIC (epilogue reduced to one max per block: wrong results, prices the VALU share)
template <int CP, int WAVES = 8, int VAR = 0>
__global__ __launch_bounds__(64 * WAVES, CP == 512 ? 1 : 2) void match_mx6_screen_kernel(
    const uint8_t *__restrict__ a6, const uint8_t *__restrict__ q6, int B, int cap_a, int cap_q, const int32_t *__restrict__ n_a,
    const int32_t *__restrict__ n_q, int T, int S, float *__restrict__ ws_max, int32_t *__restrict__ ws_i1, float *__restrict__ ws_m2)
{
    constexpr int RB = CP;
    constexpr int TILE_BYTES = screen8_tile_bytes(CP);
    constexpr int ROWS = 128, NQB = 4, NAB = 2;
    constexpr int NKS = CP / 64;                          // k-steps of 64 channels (two 32-channel blocks, one per lane half)
    constexpr int NI = TILE_BYTES / (1024 * WAVES);
    constexpr int LPR = RB / 256;
    char *smem;
    if constexpr (2 * TILE_BYTES > 65536) {
        extern __shared__ __attribute__((aligned(256))) char smem_dyn6[];
        smem = smem_dyn6;
    } else {
        __shared__ __attribute__((aligned(256))) char smem_st6[2 * TILE_BYTES];
        smem = smem_st6;
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
    const char *qp = reinterpret_cast<const char *>(q6) + (size_t)p * cap_q * RB;

    // stationary B operand: this wave's 64 anchors, slot (2 S + hi) of every k-step (dwords 0-5 codes, dword 6 = exponent byte)
    i32x8 breg[NAB][NKS];
#pragma unroll
    for (int ab = 0; ab < NAB; ++ab) {
        const int arow_i = a0 + wave * 64 + ab * 32 + l31;
        const char *arow = reinterpret_cast<const char *>(a6) + ((size_t)p * cap_a + (arow_i < cap_a ? arow_i : cap_a - 1)) * RB + 32 * hi;
#pragma unroll
        for (int s = 0; s < NKS; ++s) {
            const i32x4 lo = *reinterpret_cast<const i32x4 *>(arow + 64 * s), up = *reinterpret_cast<const i32x4 *>(arow + 64 * s + 16);
            breg[ab][s] = __builtin_shufflevector(lo, up, 0, 1, 2, 3, 4, 5, 6, 7);      // dword 7 (padding) is not read by the fp6 format
        }
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
    // LDS offsets of the two 16-byte chunks of slot (2 (S & 3) + hi) in the lane's row: chunk index XOR (row & 15) inside a 256-byte line
    unsigned koff[4][2];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e = 0; e < 2; ++e) koff[c][e] = (unsigned)(l31 * RB) + ((((unsigned)(4 * c + 2 * hi + e)) ^ (unsigned)(l31 & 15)) << 4);
    // the constant part of an address (tile buffer, 32-row block, 256-byte line) goes into the instruction's 16-bit offset field where it fits
    auto rd = [&](int s, int qb, unsigned tile) -> i32x8 {
        const unsigned base = tile + (unsigned)(qb * 32 * RB + (s >> 2) * 256);
        const i32x4 lo = *reinterpret_cast<const i32x4 *>(smem + koff[s & 3][0] + base);
        const i32x4 up = *reinterpret_cast<const i32x4 *>(smem + koff[s & 3][1] + base);
        return __builtin_shufflevector(lo, up, 0, 1, 2, 3, 4, 5, 6, 7);
    };

    float runmax[NAB], run2[NAB];
    int runidx[NAB];
#pragma unroll
    for (int ab = 0; ab < NAB; ++ab) { runmax[ab] = -INFINITY; run2[ab] = -INFINITY; runidx[ab] = 0; }
    auto reduce_block = [&](const f32x16s &c, int sid, int ab) {
        if constexpr (VAR == 1) { runmax[ab] = fmaxf(runmax[ab], c[0] + c[15]); runidx[ab] = sid; return; }
        const float m0 = fmaxf(fmaxf(c[0], c[1]), c[2]), m1 = fmaxf(fmaxf(c[3], c[4]), c[5]), m2 = fmaxf(fmaxf(c[6], c[7]), c[8]);
        const float m3 = fmaxf(fmaxf(c[9], c[10]), c[11]), m4 = fmaxf(fmaxf(c[12], c[13]), c[14]);
        const float x = fmaxf(fmaxf(fmaxf(m0, m1), m2), fmaxf(fmaxf(m3, m4), c[15]));
        const bool improved = x > runmax[ab];
        run2[ab] = __builtin_amdgcn_fmed3f(runmax[ab], run2[ab], x);              // run2 <= runmax: the median is the second largest
        runmax[ab] = fmaxf(runmax[ab], x);
        runidx[ab] = improved ? sid : runidx[ab];
    };
    const f32x16s zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    if (qt_end > qt_begin) {
#pragma unroll
        for (int j = 0; j < NI; ++j) issue_one(qt_begin, 0, j);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    i32x8 areg[NKS];
#pragma unroll
    for (int s = 0; s < NKS; ++s) areg[s] = rd(s, 0, 0u);
    f32x16s prev[NAB];
#pragma unroll
    for (int ab = 0; ab < NAB; ++ab)
#pragma unroll
        for (int r = 0; r < 16; ++r) prev[ab][r] = -3.0e38f;          // the dummy "previous block" before the first one never wins
    int prev_sid = 0;
    // one 128-row tile out of buffer BUF (a compile-time constant: the loop below alternates the two instances)
    auto do_tile = [&](int qt, auto BUFC) {
        constexpr int BUF = decltype(BUFC)::value;
        constexpr unsigned tile = BUF * TILE_BYTES;
        const int qt_next = qt + 1 < qt_end ? qt + 1 : qt;
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) {
            f32x16s acc[NAB];
            if (qb == 0) {
#pragma unroll
                for (int j = 0; j < NI; ++j) issue_one(qt_next, BUF ^ 1, j);
            }
#pragma unroll
            for (int s = 0; s < NKS; ++s) {
#pragma unroll
                for (int ab = 0; ab < NAB; ++ab)
                    acc[ab] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(areg[s], breg[ab][s], s == 0 ? zero16 : acc[ab], 2, 2, 0,
                                                                               areg[s][6], 0, breg[ab][s][6]);
                if (qb + 1 < NQB) areg[s] = rd(s, qb + 1, tile);
            }
#pragma unroll
            for (int ab = 0; ab < NAB; ++ab) reduce_block(prev[ab], prev_sid, ab);
            // pin the interleave: the previous block's epilogue (~22 VALU) and the next A operand's reads under this block's 2 NKS MFMAs
#pragma unroll
            for (int i = 0; i < NKS; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 24 / (2 * NKS), 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 24 / (2 * NKS), 0);
                if (qb + 1 < NQB) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
#pragma unroll
            for (int ab = 0; ab < NAB; ++ab) prev[ab] = acc[ab];
            prev_sid = (qt * NQB + qb) * 2 + hi;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int s = 0; s < NKS; ++s) areg[s] = rd(s, 0, (BUF ^ 1) * TILE_BYTES);
    };
    for (int qt = qt_begin; qt < qt_end; qt += 2) {
        do_tile(qt, std::integral_constant<int, 0>{});
        if (qt + 1 < qt_end) do_tile(qt + 1, std::integral_constant<int, 1>{});
    }
#pragma unroll
    for (int ab = 0; ab < NAB; ++ab) reduce_block(prev[ab], prev_sid, ab);
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

// K1s6 at C_pad = 256, second cut (round 3): 128 anchors per wave.  The kernel above feeds every 32x32x64 MFMA one kilobyte of LDS reads
// (each wave reads the whole 32-row block of the query tile for its TWO anchor blocks): at four MFMAs per 32 cycles and CU that is the
// LDS's entire 128 B/clk - an epilogue-free diagnostic build of it ran no faster than the product kernel: the LDS and the matrix pipe
// saturate together.  Here a wave keeps FOUR anchor blocks stationary (96 code + 4 packed exponent registers; the instruction picks the
// exponent byte by op_sel), so one A-operand read feeds four MFMAs: half the LDS bytes per MFMA.  No accumulator copies: the accumulators
// of anchor blocks 0/1 and 2/3 alternate - while the MFMAs of one pair run, the VALU reduces the other pair's previous block.  246 VGPRs,
// two waves per SIMD.  WAVES = 8 (default): 1024-anchor panels, ONE 64 KB workgroup per CU - half the L2 -> LDS bytes as well, and K0 /
// the registration still find LDS beside it (1.46 ms alone against 1.58-1.61, pipelined step 3.84 against 3.87 ms on the same box).
// WAVES = 4: 512-anchor panels, two workgroups per CU: 1.47 ms alone, but its 128 KB of LDS keep K0 off the CU (pipelined 3.94 ms).
// 256-row tiles (half the barriers, 128 KB of LDS): the eight unrolled 32-row blocks spill 22 registers - 1.58 ms alone, step 4.07 ms: not kept.
// C_pad = 512 (cfg4): the same loop with 8 k-steps, 4 waves and one wave per SIMD (192 code + 8 exponent registers stationary, 482 VGPRs incl. AGPRs,
// 2 x 64 KB of LDS): 19.7 ms per cfg4 launch against 21.7 ms for the two-block kernel, cfg4 shard 3.57 k -> 3.84 k pairs/s.
// Outputs and slice meaning unchanged.
typedef int i32x3 __attribute__((ext_vector_type(3)));
typedef int i32x6 __attribute__((ext_vector_type(6)));

// compile-time loop over the k-steps (the step index is an instruction operand: the op_sel byte of the packed exponents)
template <int S0, int N, class F>
__device__ __forceinline__ void mx6_static_for(F &&f)
{
    if constexpr (S0 < N) {
        f(std::integral_constant<int, S0>{});
        mx6_static_for<S0 + 1, N>(f);
    }
}

// C_pad = 512: WAVES = 4, one wave per SIMD (192 code + 8 exponent registers of stationary operands), 2 x 64 KB of dynamic LDS
// KL = live k-steps of 64 channels (round 4: narrow maps - the reference's own C = 32 - are zero-padded to the 256-channel rows; their dead
// k-steps hold zero codes and contribute exactly 0, so they are simply not multiplied: same results, a quarter of the MFMAs at C <= 64)
template <int CP, int WAVES, int KL = CP / 64>
__global__ __launch_bounds__(64 * WAVES, CP == 512 ? 1 : 2) void match_mx6_screen_w4_kernel(
    const uint8_t *__restrict__ a6, const uint8_t *__restrict__ q6, int B, int cap_a, int cap_q, const int32_t *__restrict__ n_a,
    const int32_t *__restrict__ n_q, int T, int S, float *__restrict__ ws_max, int32_t *__restrict__ ws_i1, float *__restrict__ ws_m2,
    long long *__restrict__ dbg_wg /* ORYON_MX6_DEBUG: per-workgroup (start, end, hw_id, xcc_id), else NULL */)
{
    static_assert(CP == 256 || (CP == 512 && WAVES == 4), "geometries: C_pad 256 with 4 / 8 waves, C_pad 512 with 4 waves");
    constexpr int RB = CP, NAB = 4;
    constexpr int TILE_BYTES = screen8_tile_bytes(CP);
    constexpr int ROWS = 128, NQB = 4;
    constexpr int NKS = CP / 64;
    constexpr int NI = TILE_BYTES / (1024 * WAVES);
    constexpr int LPR = RB / 256;
    char *smem;
    if constexpr (2 * TILE_BYTES > 65536) {
        extern __shared__ __attribute__((aligned(256))) char smem_dyn_w4[];
        smem = smem_dyn_w4;
    } else {
        __shared__ __attribute__((aligned(256))) char smem_st_w4[2 * TILE_BYTES];
        smem = smem_st_w4;
    }
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int unit = (slot / T) * 8 + xcd;
    if (unit >= B * S) return;
    const int panel = slot % T;
    const int p = unit / S, split = unit % S;
    const int na = n_a[p], nq = n_q[p];
    const int a0 = panel * (32 * NAB * WAVES);
    if (a0 >= na) return;
    const int nqt = (nq + ROWS - 1) / ROWS;
    const int qt_per = (nqt + S - 1) / S;
    const int qt_begin = split * qt_per;
    const int qt_end = (qt_begin + qt_per < nqt) ? qt_begin + qt_per : nqt;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, hi = lane >> 5;
    const char *qp = reinterpret_cast<const char *>(q6) + (size_t)p * cap_q * RB;
    const long long tk0 = dbg_wg ? wall_clock64() : 0;

    // stationary B operands: slot (2 s + hi) of k-step s of the lane's anchor row in each of the four blocks; the four exponent bytes of a
    // block packed into one register (the instruction picks the byte by op_sel)
    i32x8 breg[NAB][NKS];
    int bsc[NAB][NKS / 4];
#pragma unroll
    for (int ab = 0; ab < NAB; ++ab) {
        const int arow_i = a0 + wave * (32 * NAB) + ab * 32 + l31;
        const char *arow = reinterpret_cast<const char *>(a6) + ((size_t)p * cap_a + (arow_i < cap_a ? arow_i : cap_a - 1)) * RB + 32 * hi;
        unsigned sc[NKS / 4];
#pragma unroll
        for (int w = 0; w < NKS / 4; ++w) sc[w] = 0;
#pragma unroll
        for (int s = 0; s < KL; ++s) {
            const i32x4 lo = *reinterpret_cast<const i32x4 *>(arow + 64 * s), up = *reinterpret_cast<const i32x4 *>(arow + 64 * s + 16);
            breg[ab][s] = __builtin_shufflevector(lo, up, 0, 1, 2, 3, 4, 5, -1, -1);   // the fp6 format reads six dwords
            sc[s >> 2] |= ((unsigned)up[2] & 0xffu) << (8 * (s & 3));
        }
#pragma unroll
        for (int w = 0; w < NKS / 4; ++w) bsc[ab][w] = (int)sc[w];
    }
    // DMA instruction j of a wave moves the four 256-byte lines (wave NI + j) 4 .. + 3 of the tile (4 / LPR rows); rows 16 apart share the
    // swizzle, so instructions j and j + DJ (DJ = 4 LPR) differ by 16 rows - in the scalar base, not in another pair of address registers
    constexpr int DJ = 4 * LPR;
    static_assert(NI % DJ == 0 || NI == DJ, "DMA instructions per wave and tile");
    unsigned dma_off[DJ];
#pragma unroll
    for (int j = 0; j < DJ; ++j) {
        const int line = (wave * NI + j) * 4 + (lane >> 4), sl = lane & 15;
        const int row = line / LPR;
        dma_off[j] = (unsigned)(row * RB + (((line % LPR) * 16 + (sl ^ (row & 15))) << 4));
    }
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto issue_one = [&](int qt, int buf, int j) {
        const char *qb = qp + (size_t)qt * TILE_BYTES + (j / DJ) * (16 * RB);
        char *dst = smem + buf * TILE_BYTES + (wave_u * NI + j) * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(qb + dma_off[j % DJ]),
                                         (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
    };
    unsigned koff[4][2];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e = 0; e < 2; ++e) koff[c][e] = (unsigned)(l31 * RB) + ((((unsigned)(4 * c + 2 * hi + e)) ^ (unsigned)(l31 & 15)) << 4);
    // A operand of one k-step: 16 + 12 bytes of the lane's slot (dword 6 = exponent byte, the scale operand)
    auto rd = [&](int s, int qb, unsigned tile) -> i32x8 {
        const unsigned base = tile + (unsigned)(qb * 32 * RB + (s >> 2) * 256);
        const i32x4 lo = *reinterpret_cast<const i32x4 *>(smem + koff[s & 3][0] + base);
        const i32x4 up = *reinterpret_cast<const i32x4 *>(smem + koff[s & 3][1] + base);
        return __builtin_shufflevector(lo, up, 0, 1, 2, 3, 4, 5, 6, -1);
    };

    float runmax[NAB], run2[NAB];
    int runidx[NAB];
#pragma unroll
    for (int ab = 0; ab < NAB; ++ab) { runmax[ab] = -INFINITY; run2[ab] = -INFINITY; runidx[ab] = 0; }
    auto reduce_block = [&](const f32x16s &c, int sid, int ab) {
        const float m0 = fmaxf(fmaxf(c[0], c[1]), c[2]), m1 = fmaxf(fmaxf(c[3], c[4]), c[5]), m2 = fmaxf(fmaxf(c[6], c[7]), c[8]);
        const float m3 = fmaxf(fmaxf(c[9], c[10]), c[11]), m4 = fmaxf(fmaxf(c[12], c[13]), c[14]);
        const float x = fmaxf(fmaxf(fmaxf(m0, m1), m2), fmaxf(fmaxf(m3, m4), c[15]));
        const bool improved = x > runmax[ab];
        run2[ab] = __builtin_amdgcn_fmed3f(runmax[ab], run2[ab], x);              // run2 <= runmax: the median is the second largest
        runmax[ab] = __builtin_amdgcn_fmed3f(runmax[ab], x, INFINITY);           // = max (no NaNs here); as the intrinsic it needs no operand
        runidx[ab] = improved ? sid : runidx[ab];                                // canonicalisation (v_max x, x) of the loop-carried state
    };
    const f32x16s zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    if (qt_end > qt_begin) {
#pragma unroll
        for (int j = 0; j < NI; ++j) issue_one(qt_begin, 0, j);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    i32x8 areg[NKS];
#pragma unroll
    for (int s = 0; s < KL; ++s) areg[s] = rd(s, 0, 0u);
    f32x16s acc[NAB];
#pragma unroll
    for (int ab = 0; ab < NAB; ++ab)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ab][r] = -3.0e38f;           // "previous block" of the pairs before the first one: never wins
    int sid01 = 0, sid23 = 0;                                         // slice ids of the blocks acc[0..1] / acc[2..3] currently hold

    // one MFMA: k-step SC (a compile-time constant: it is also the op_sel byte of the packed B exponents) of anchor block ab
    auto mfma = [&](f32x16s &d, int ab, auto SC) {
        constexpr int s_ = decltype(SC)::value;
        d = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(areg[s_], breg[ab][s_], s_ == 0 ? zero16 : d, 2, 2, 0, areg[s_][6], s_ & 3,
                                                            bsc[ab][s_ >> 2]);
    };
    int buf = 0;
    for (int qt = qt_begin; qt < qt_end; ++qt) {
        const unsigned tile = buf * TILE_BYTES;
        const int qt_next = qt + 1 < qt_end ? qt + 1 : qt;
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) {
            const int sid = (qt * NQB + qb) * 2 + hi;
            if (qb == 0) {
#pragma unroll
                for (int j = 0; j < NI; ++j) issue_one(qt_next, buf ^ 1, j);
            }
            // first half: the VALU reduces the previous block of anchor blocks 2 / 3, the matrix pipe starts this block for 0 / 1
            reduce_block(acc[2], sid23, 2);
            reduce_block(acc[3], sid23, 3);
            // (the reductions read acc[2..3] before the second half overwrites them: program order)
            mx6_static_for<0, KL>([&](auto SC) { mfma(acc[0], 0, SC); mfma(acc[1], 1, SC); });
#pragma unroll
            for (int i = 0; i < 2 * KL; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 16 / KL, 0);
            }
            // second half: this block for 2 / 3 while the VALU reduces what 0 / 1 just finished; next A operand behind its last use
            mx6_static_for<0, KL>([&](auto SC) {
                constexpr int s_ = decltype(SC)::value;
                mfma(acc[2], 2, SC); mfma(acc[3], 3, SC);
                if (qb + 1 < NQB) areg[s_] = rd(s_, qb + 1, tile);
            });
            reduce_block(acc[0], sid, 0);
            reduce_block(acc[1], sid, 1);
#pragma unroll
            for (int i = 0; i < 2 * KL; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 16 / KL, 0);
                if (qb + 1 < NQB && (i & 1)) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
            sid01 = sid;
            sid23 = sid;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        buf ^= 1;
#pragma unroll
        for (int s = 0; s < KL; ++s) areg[s] = rd(s, 0, buf * TILE_BYTES);
    }
    (void)sid01;
    reduce_block(acc[2], sid23, 2);
    reduce_block(acc[3], sid23, 3);
#pragma unroll
    for (int ab = 0; ab < NAB; ++ab) {
        const float om1 = __shfl_xor(runmax[ab], 32), om2 = __shfl_xor(run2[ab], 32);
        const int oi1 = __shfl_xor(runidx[ab], 32);
        const float m1 = fmaxf(runmax[ab], om1);
        const float m2 = fmaxf(fminf(runmax[ab], om1), fmaxf(run2[ab], om2));
        const int i1 = (om1 > runmax[ab]) ? oi1 : runidx[ab];
        const int a = a0 + wave * (32 * NAB) + ab * 32 + l31;
        if (hi == 0 && a < cap_a) {
            const size_t o = ((size_t)p * S + split) * cap_a + a;
            ws_max[o] = m1;
            ws_i1[o] = i1;
            ws_m2[o] = m2;
        }
    }
    if (dbg_wg && t == 0) {
        dbg_wg[(size_t)blockIdx.x * 4 + 0] = tk0;
        dbg_wg[(size_t)blockIdx.x * 4 + 1] = wall_clock64();
        dbg_wg[(size_t)blockIdx.x * 4 + 2] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_REG_HW_ID
        dbg_wg[(size_t)blockIdx.x * 4 + 3] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 20);       // HW_REG_XCC_ID
    }
}

// The validity cascade of the hard route (round 6; oryon_match_corrs_mx6_x3) runs the screen in two launches of this copy of the kernel
// above (same loop, two more scalar arguments; the headline's kernel itself is untouched):
//   win > 0  : every (panel, split) multiplies only `win` query tiles, placed in its split where the panel sits in its own map - on smooth
//              maps (every anchor matches a near-by query and its neighbours almost as well) that settles the VALIDITY of every anchor
//              of the panel ("some query within the threshold" needs one witness) at a few percent of the multiply-accumulates;
//   gate     : the complete scan, for the panels match_panel_settle_kernel found an unsettled anchor in; panels that were settled keep
//              their partial triples with m2 = +inf (no margin: argmin open, which is what every anchor of such maps is anyway), and
//              the SAMPLED anchors get the complete screen in a second pass over one 512-row panel per pair (match_corrs_lazy_impl).
// (An in-loop early exit was built first: the break cost the loop its register allocation - 161 spilled registers, six times slower per tile.)
template <int CP, int WAVES, int KL = CP / 64>
__global__ __launch_bounds__(64 * WAVES, CP == 512 ? 1 : 2) void match_mx6_screen_w4_win_kernel(
    const uint8_t *__restrict__ a6, const uint8_t *__restrict__ q6, int B, int cap_a, int cap_q, const int32_t *__restrict__ n_a,
    const int32_t *__restrict__ n_q, int T, int S, float *__restrict__ ws_max, int32_t *__restrict__ ws_i1, float *__restrict__ ws_m2,
    const int32_t *__restrict__ gate /* [B, T] or NULL: only panels with a non-zero entry run */, int win /* > 0: that many tiles of the split */)
{
    static_assert(CP == 256 || (CP == 512 && WAVES == 4), "geometries: C_pad 256 with 4 / 8 waves, C_pad 512 with 4 waves");
    constexpr int RB = CP, NAB = 4;
    constexpr int TILE_BYTES = screen8_tile_bytes(CP);
    constexpr int ROWS = 128, NQB = 4;
    constexpr int NKS = CP / 64;
    constexpr int NI = TILE_BYTES / (1024 * WAVES);
    constexpr int LPR = RB / 256;
    char *smem;
    if constexpr (2 * TILE_BYTES > 65536) {
        extern __shared__ __attribute__((aligned(256))) char smem_dyn_w4w[];
        smem = smem_dyn_w4w;
    } else {
        __shared__ __attribute__((aligned(256))) char smem_st_w4w[2 * TILE_BYTES];
        smem = smem_st_w4w;
    }
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int unit = (slot / T) * 8 + xcd;
    if (unit >= B * S) return;
    const int panel = slot % T;
    const int p = unit / S, split = unit % S;
    const int na = n_a[p], nq = n_q[p];
    const int a0 = panel * (32 * NAB * WAVES);
    if (a0 >= na) return;
    if (gate && gate[p * T + panel] == 0) return;
    const int nqt = (nq + ROWS - 1) / ROWS;
    const int qt_per = (nqt + S - 1) / S;
    int qt_begin = split * qt_per;
    int qt_end = (qt_begin + qt_per < nqt) ? qt_begin + qt_per : nqt;
    if (win > 0 && qt_end - qt_begin > win) {
        // the window sits in the split where the panel sits among the pair's panels (smooth maps match near-by pixels)
        const int cnt = qt_end - qt_begin, panels = (na + 32 * NAB * WAVES - 1) / (32 * NAB * WAVES);
        int k0 = (int)(((long long)(2 * panel + 1) * cnt) / (2 * panels)) - win / 2;
        k0 = k0 < 0 ? 0 : (k0 > cnt - win ? cnt - win : k0);
        qt_begin += k0;
        qt_end = qt_begin + win;
    }
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, hi = lane >> 5;
    const char *qp = reinterpret_cast<const char *>(q6) + (size_t)p * cap_q * RB;

    // stationary B operands: slot (2 s + hi) of k-step s of the lane's anchor row in each of the four blocks; the four exponent bytes of a
    // block packed into one register (the instruction picks the byte by op_sel)
    i32x8 breg[NAB][NKS];
    int bsc[NAB][NKS / 4];
#pragma unroll
    for (int ab = 0; ab < NAB; ++ab) {
        const int arow_i = a0 + wave * (32 * NAB) + ab * 32 + l31;
        const char *arow = reinterpret_cast<const char *>(a6) + ((size_t)p * cap_a + (arow_i < cap_a ? arow_i : cap_a - 1)) * RB + 32 * hi;
        unsigned sc[NKS / 4];
#pragma unroll
        for (int w = 0; w < NKS / 4; ++w) sc[w] = 0;
#pragma unroll
        for (int s = 0; s < KL; ++s) {
            const i32x4 lo = *reinterpret_cast<const i32x4 *>(arow + 64 * s), up = *reinterpret_cast<const i32x4 *>(arow + 64 * s + 16);
            breg[ab][s] = __builtin_shufflevector(lo, up, 0, 1, 2, 3, 4, 5, -1, -1);   // the fp6 format reads six dwords
            sc[s >> 2] |= ((unsigned)up[2] & 0xffu) << (8 * (s & 3));
        }
#pragma unroll
        for (int w = 0; w < NKS / 4; ++w) bsc[ab][w] = (int)sc[w];
    }
    // DMA instruction j of a wave moves the four 256-byte lines (wave NI + j) 4 .. + 3 of the tile (4 / LPR rows); rows 16 apart share the
    // swizzle, so instructions j and j + DJ (DJ = 4 LPR) differ by 16 rows - in the scalar base, not in another pair of address registers
    constexpr int DJ = 4 * LPR;
    static_assert(NI % DJ == 0 || NI == DJ, "DMA instructions per wave and tile");
    unsigned dma_off[DJ];
#pragma unroll
    for (int j = 0; j < DJ; ++j) {
        const int line = (wave * NI + j) * 4 + (lane >> 4), sl = lane & 15;
        const int row = line / LPR;
        dma_off[j] = (unsigned)(row * RB + (((line % LPR) * 16 + (sl ^ (row & 15))) << 4));
    }
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto issue_one = [&](int qt, int buf, int j) {
        const char *qb = qp + (size_t)qt * TILE_BYTES + (j / DJ) * (16 * RB);
        char *dst = smem + buf * TILE_BYTES + (wave_u * NI + j) * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(qb + dma_off[j % DJ]),
                                         (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
    };
    unsigned koff[4][2];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e = 0; e < 2; ++e) koff[c][e] = (unsigned)(l31 * RB) + ((((unsigned)(4 * c + 2 * hi + e)) ^ (unsigned)(l31 & 15)) << 4);
    // A operand of one k-step: 16 + 12 bytes of the lane's slot (dword 6 = exponent byte, the scale operand)
    auto rd = [&](int s, int qb, unsigned tile) -> i32x8 {
        const unsigned base = tile + (unsigned)(qb * 32 * RB + (s >> 2) * 256);
        const i32x4 lo = *reinterpret_cast<const i32x4 *>(smem + koff[s & 3][0] + base);
        const i32x4 up = *reinterpret_cast<const i32x4 *>(smem + koff[s & 3][1] + base);
        return __builtin_shufflevector(lo, up, 0, 1, 2, 3, 4, 5, 6, -1);
    };

    float runmax[NAB], run2[NAB];
    int runidx[NAB];
#pragma unroll
    for (int ab = 0; ab < NAB; ++ab) { runmax[ab] = -INFINITY; run2[ab] = -INFINITY; runidx[ab] = 0; }
    auto reduce_block = [&](const f32x16s &c, int sid, int ab) {
        const float m0 = fmaxf(fmaxf(c[0], c[1]), c[2]), m1 = fmaxf(fmaxf(c[3], c[4]), c[5]), m2 = fmaxf(fmaxf(c[6], c[7]), c[8]);
        const float m3 = fmaxf(fmaxf(c[9], c[10]), c[11]), m4 = fmaxf(fmaxf(c[12], c[13]), c[14]);
        const float x = fmaxf(fmaxf(fmaxf(m0, m1), m2), fmaxf(fmaxf(m3, m4), c[15]));
        const bool improved = x > runmax[ab];
        run2[ab] = __builtin_amdgcn_fmed3f(runmax[ab], run2[ab], x);              // run2 <= runmax: the median is the second largest
        runmax[ab] = __builtin_amdgcn_fmed3f(runmax[ab], x, INFINITY);           // = max (no NaNs here); as the intrinsic it needs no operand
        runidx[ab] = improved ? sid : runidx[ab];                                // canonicalisation (v_max x, x) of the loop-carried state
    };
    const f32x16s zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    if (qt_end > qt_begin) {
#pragma unroll
        for (int j = 0; j < NI; ++j) issue_one(qt_begin, 0, j);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    i32x8 areg[NKS];
#pragma unroll
    for (int s = 0; s < KL; ++s) areg[s] = rd(s, 0, 0u);
    f32x16s acc[NAB];
#pragma unroll
    for (int ab = 0; ab < NAB; ++ab)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ab][r] = -3.0e38f;           // "previous block" of the pairs before the first one: never wins
    int sid01 = 0, sid23 = 0;                                         // slice ids of the blocks acc[0..1] / acc[2..3] currently hold

    // one MFMA: k-step SC (a compile-time constant: it is also the op_sel byte of the packed B exponents) of anchor block ab
    auto mfma = [&](f32x16s &d, int ab, auto SC) {
        constexpr int s_ = decltype(SC)::value;
        d = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(areg[s_], breg[ab][s_], s_ == 0 ? zero16 : d, 2, 2, 0, areg[s_][6], s_ & 3,
                                                            bsc[ab][s_ >> 2]);
    };
    int buf = 0;
    for (int qt = qt_begin; qt < qt_end; ++qt) {
        const unsigned tile = buf * TILE_BYTES;
        const int qt_next = qt + 1 < qt_end ? qt + 1 : qt;
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) {
            const int sid = (qt * NQB + qb) * 2 + hi;
            if (qb == 0) {
#pragma unroll
                for (int j = 0; j < NI; ++j) issue_one(qt_next, buf ^ 1, j);
            }
            // first half: the VALU reduces the previous block of anchor blocks 2 / 3, the matrix pipe starts this block for 0 / 1
            reduce_block(acc[2], sid23, 2);
            reduce_block(acc[3], sid23, 3);
            // (the reductions read acc[2..3] before the second half overwrites them: program order)
            mx6_static_for<0, KL>([&](auto SC) { mfma(acc[0], 0, SC); mfma(acc[1], 1, SC); });
#pragma unroll
            for (int i = 0; i < 2 * KL; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 16 / KL, 0);
            }
            // second half: this block for 2 / 3 while the VALU reduces what 0 / 1 just finished; next A operand behind its last use
            mx6_static_for<0, KL>([&](auto SC) {
                constexpr int s_ = decltype(SC)::value;
                mfma(acc[2], 2, SC); mfma(acc[3], 3, SC);
                if (qb + 1 < NQB) areg[s_] = rd(s_, qb + 1, tile);
            });
            reduce_block(acc[0], sid, 0);
            reduce_block(acc[1], sid, 1);
#pragma unroll
            for (int i = 0; i < 2 * KL; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 16 / KL, 0);
                if (qb + 1 < NQB && (i & 1)) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
            sid01 = sid;
            sid23 = sid;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        buf ^= 1;
#pragma unroll
        for (int s = 0; s < KL; ++s) areg[s] = rd(s, 0, buf * TILE_BYTES);
    }
    (void)sid01;
    reduce_block(acc[2], sid23, 2);
    reduce_block(acc[3], sid23, 3);
#pragma unroll
    for (int ab = 0; ab < NAB; ++ab) {
        const float om1 = __shfl_xor(runmax[ab], 32), om2 = __shfl_xor(run2[ab], 32);
        const int oi1 = __shfl_xor(runidx[ab], 32);
        const float m1 = fmaxf(runmax[ab], om1);
        const float m2 = fmaxf(fminf(runmax[ab], om1), fmaxf(run2[ab], om2));
        const int i1 = (om1 > runmax[ab]) ? oi1 : runidx[ab];
        const int a = a0 + wave * (32 * NAB) + ab * 32 + l31;
        if (hi == 0 && a < cap_a) {
            const size_t o = ((size_t)p * S + split) * cap_a + a;
            ws_max[o] = m1;
            ws_i1[o] = i1;
            ws_m2[o] = m2;
        }
    }
}

// ORYON_MX6_DEBUG=1 (development aid; synchronises): per-workgroup wall-clock records of the screen launch -> how many workgroups were
// resident over the launch, i.e. whether the dispatcher keeps the CUs full (the K1x3 scan's were not: match_x3.hip)
static void mx6_debug_report(const long long *dbg_dev, int groups, hipStream_t st)
{
    (void)hipStreamSynchronize(st);
    long long *w = static_cast<long long *>(malloc((size_t)groups * 4 * sizeof(long long)));
    (void)hipMemcpy(w, dbg_dev, (size_t)groups * 4 * sizeof(long long), hipMemcpyDeviceToHost);
    long long t0 = -1, t1 = 0;
    double sum = 0;
    int ran = 0;
    long long longest = 0;
    for (int g = 0; g < groups; ++g)
        if (w[4 * g + 1]) {
            if (t0 < 0 || w[4 * g] < t0) t0 = w[4 * g];
            if (w[4 * g + 1] > t1) t1 = w[4 * g + 1];
            sum += (double)(w[4 * g + 1] - w[4 * g]);
            if (w[4 * g + 1] - w[4 * g] > longest) longest = w[4 * g + 1] - w[4 * g];
            ++ran;
        }
    fprintf(stderr, "[mx6] %d of %d workgroups ran: span %.1f us, mean %.1f us, longest %.1f us, mean residency %.1f workgroups\n", ran, groups,
            (t1 - t0) * 0.01, ran ? sum * 0.01 / ran : 0.0, longest * 0.01, t1 > t0 ? sum / (double)(t1 - t0) : 0.0);
    fprintf(stderr, "[mx6] resident at 5%%..95%% of the span:");
    for (int k = 0; k < 10; ++k) {
        const long long ts = t0 + (t1 - t0) * (2 * k + 1) / 20;
        int r = 0;
        for (int g = 0; g < groups; ++g) r += w[4 * g + 1] && w[4 * g] <= ts && ts < w[4 * g + 1];
        fprintf(stderr, " %d", r);
    }
    fprintf(stderr, "\n");
    free(w);
}

static int mx6_var()
{
    // 0: default; 1: diagnostic build of the 8-wave loop (wrong results); 2: the first 8-wave kernel at C_pad 256; 3: 4-wave workgroups
    static const int var = dev_env_int("ORYON_MX6_VAR", 0);
    return var;
}

namespace {
template <int CP>
void launch_screen_mx6_t(int groups, int T, hipStream_t st, const uint8_t *a6, const uint8_t *q6, int B, int cap_a, int cap_q, const int32_t *n_a,
                       const int32_t *n_q, int S, float *ws_max, int32_t *ws_i1, float *ws_m2, int kl = 0, bool gate_or_win = false,
                       const int32_t *gate = nullptr, int win = 0)
{
    // C_pad 256: 512-anchor panels (8 waves; `groups` was sized for 256-anchor panels, T of them per unit).  C_pad 512: the stationary
    // operand is 128 registers, so 4 waves per workgroup and one workgroup per CU (512 registers per wave), as the int8 kernel
    constexpr size_t dyn = 2 * screen8_tile_bytes(CP) > 65536 ? 2 * screen8_tile_bytes(CP) : 0;
    constexpr int W_ = CP == 512 ? 4 : 8;
    const int Tw = (cap_a + 64 * W_ - 1) / (64 * W_);
    const int var = mx6_var();
    if (var == 1) {
        if (dyn) allow_dynamic_lds(reinterpret_cast<const void *>(&match_mx6_screen_kernel<CP, W_, 1>), (int)dyn);
        hipLaunchKernelGGL((match_mx6_screen_kernel<CP, W_, 1>), dim3(groups / T * Tw), dim3(64 * W_), dyn, st, a6, q6, B, cap_a, cap_q, n_a, n_q, Tw,
                           S, ws_max, ws_i1, ws_m2);
        return;
    }
    if constexpr (CP == 256) {
        if (var == 0) {                                          // default: 8 waves x 128 anchors (1024-anchor panels, one workgroup per CU)
            const int T8 = (cap_a + 1023) / 1024;
            static const bool dbg = dev_env_set("ORYON_MX6_DEBUG");
            static long long *dbg_wg = nullptr;
            const int g8 = groups / T * T8;
            if (dbg && !dbg_wg) (void)hipMalloc(&dbg_wg, (size_t)65536 * 4 * sizeof(long long));
            if (dbg && dbg_wg && g8 <= 65536) (void)hipMemsetAsync(dbg_wg, 0, (size_t)g8 * 4 * sizeof(long long), st);
            if (gate_or_win) {
                // the cascade's launches: windowed (win > 0) or gated (gate != NULL) copy of the kernel
#define ORYON_LAUNCH_WIN(KLV)                                                                                                         \
    hipLaunchKernelGGL((match_mx6_screen_w4_win_kernel<CP, 8, KLV>), dim3(g8), dim3(512), 0, st, a6, q6, B, cap_a, cap_q, n_a, n_q, T8, S,   \
                       ws_max, ws_i1, ws_m2, gate, win)
                if (kl == 1) ORYON_LAUNCH_WIN(1);
                else if (kl == 2) ORYON_LAUNCH_WIN(2);
                else ORYON_LAUNCH_WIN(4);
#undef ORYON_LAUNCH_WIN
                return;
            }
            if (kl == 1)
                hipLaunchKernelGGL((match_mx6_screen_w4_kernel<CP, 8, 1>), dim3(g8), dim3(512), 0, st, a6, q6, B, cap_a, cap_q, n_a, n_q, T8, S,
                                   ws_max, ws_i1, ws_m2, (dbg && g8 <= 65536) ? dbg_wg : nullptr);
            else if (kl == 2)
                hipLaunchKernelGGL((match_mx6_screen_w4_kernel<CP, 8, 2>), dim3(g8), dim3(512), 0, st, a6, q6, B, cap_a, cap_q, n_a, n_q, T8, S,
                                   ws_max, ws_i1, ws_m2, (dbg && g8 <= 65536) ? dbg_wg : nullptr);
            else
            hipLaunchKernelGGL((match_mx6_screen_w4_kernel<CP, 8>), dim3(g8), dim3(512), 0, st, a6, q6, B, cap_a, cap_q, n_a, n_q, T8, S,
                               ws_max, ws_i1, ws_m2, (dbg && g8 <= 65536) ? dbg_wg : nullptr);
            if (dbg && dbg_wg && g8 <= 65536) mx6_debug_report(dbg_wg, g8, st);
            return;
        }
        if (var == 3) {                                          // 4 waves x 128 anchors (512-anchor panels, two workgroups per CU)
            hipLaunchKernelGGL((match_mx6_screen_w4_kernel<CP, 4>), dim3(groups / T * Tw), dim3(256), 0, st, a6, q6, B, cap_a, cap_q, n_a, n_q, Tw, S,
                               ws_max, ws_i1, ws_m2, nullptr);
            return;
        }
    }
    if constexpr (CP == 512) {
        if (var == 0) {                                          // default at C_pad 512: 4 waves x 128 anchors, one wave per SIMD
            allow_dynamic_lds(reinterpret_cast<const void *>(&match_mx6_screen_w4_kernel<CP, 4>), (int)dyn);
            hipLaunchKernelGGL((match_mx6_screen_w4_kernel<CP, 4>), dim3(groups / T * Tw), dim3(256), dyn, st, a6, q6, B, cap_a, cap_q, n_a, n_q, Tw,
                               S, ws_max, ws_i1, ws_m2, nullptr);
            return;
        }
    }
    if (dyn) allow_dynamic_lds(reinterpret_cast<const void *>(&match_mx6_screen_kernel<CP, W_>), (int)dyn);
    hipLaunchKernelGGL((match_mx6_screen_kernel<CP, W_>), dim3(groups / T * Tw), dim3(64 * W_), dyn, st, a6, q6, B, cap_a, cap_q, n_a, n_q, Tw, S,
                       ws_max, ws_i1, ws_m2);
}
}  // namespace

// the kernel launch_screen_mx6 dispatches (for oryon_dominant_kernel / profile markers)
const char *screen_mx6_name(int C)
{
    if (C != 256) return mx6_var() == 0 ? "match_mx6_screen_w4_kernel<512, 4>" : "match_mx6_screen_kernel<512, 4, 0>";
    return mx6_var() == 0 ? "match_mx6_screen_w4_kernel<256, 8>" : mx6_var() == 3 ? "match_mx6_screen_w4_kernel<256, 4>" : mx6_var() == 1 ? "match_mx6_screen_kernel<256, 8, 1>" : "match_mx6_screen_kernel<256, 8, 0>";
}

void launch_screen_mx6(int C, int groups, int T, hipStream_t st, const uint8_t *a6, const uint8_t *q6, int B, int cap_a, int cap_q,
                       const int32_t *n_a, const int32_t *n_q, int S, float *ws_max, int32_t *ws_i1, float *ws_m2, int C_true,
                       int cascade, const int32_t *gate, int win)
{
    // live k-steps of narrow maps (1 for C <= 64, 2 for C <= 128; otherwise all four): see match_mx6_screen_w4_kernel
    const int kl = (C == 256 && C_true > 0 && C_true <= 64) ? 1 : (C == 256 && C_true > 0 && C_true <= 128) ? 2 : 0;
    // cascade != 0 (C_pad 256 only): the windowed / gated launches of the validity cascade (match_mx6_screen_w4_win_kernel)
    if (C == 256) launch_screen_mx6_t<256>(groups, T, st, a6, q6, B, cap_a, cap_q, n_a, n_q, S, ws_max, ws_i1, ws_m2, kl, cascade != 0, gate, win);
    else launch_screen_mx6_t<512>(groups, T, st, a6, q6, B, cap_a, cap_q, n_a, n_q, S, ws_max, ws_i1, ws_m2);
}

// Second pass of the validity cascade: the complete screen over ONE 512-row panel per pair (the sampled anchors whose argmin is open,
// compacted by match_compact_rows_kernel), the query tiles of a pair dealt to S workgroups of four waves (two per CU): a tenth of the
// first pass's multiply-accumulates, spread over the whole chip.  Same kernel, same scores, same slice ids as the full screen.
void launch_screen_mx6_sampled(hipStream_t st, const uint8_t *a6_panel, const uint8_t *q6, int B, int cap_q, const int32_t *n_rows,
                               const int32_t *n_q, int S, float *ws_max, int32_t *ws_i1, float *ws_m2, int C_true)
{
    const int groups = ((B * S + 7) / 8) * 8;                      // T = 1 panel per (pair, split) unit
    const int kl = (C_true > 0 && C_true <= 64) ? 1 : (C_true > 0 && C_true <= 128) ? 2 : 0;
    if (kl == 1)
        hipLaunchKernelGGL((match_mx6_screen_w4_kernel<256, 4, 1>), dim3(groups), dim3(256), 0, st, a6_panel, q6, B, 512, cap_q, n_rows, n_q, 1, S,
                           ws_max, ws_i1, ws_m2, static_cast<long long *>(nullptr));
    else if (kl == 2)
        hipLaunchKernelGGL((match_mx6_screen_w4_kernel<256, 4, 2>), dim3(groups), dim3(256), 0, st, a6_panel, q6, B, 512, cap_q, n_rows, n_q, 1, S,
                           ws_max, ws_i1, ws_m2, static_cast<long long *>(nullptr));
    else
        hipLaunchKernelGGL((match_mx6_screen_w4_kernel<256, 4>), dim3(groups), dim3(256), 0, st, a6_panel, q6, B, 512, cap_q, n_rows, n_q, 1, S,
                           ws_max, ws_i1, ws_m2, static_cast<long long *>(nullptr));
}

}  // namespace oryon
