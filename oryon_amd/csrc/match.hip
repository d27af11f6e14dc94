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
#include "common.h"

namespace oryon {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int MT = ORYON_MATCH_TILE;  // 128 queries x 128 anchors per workgroup step
constexpr int BK = 32;                // k-tile
constexpr int LD = BK + 1;            // padded LDS row: ds_read_b32 of a 32-row column is conflict-free
constexpr int MATCH_THREADS = 256;
constexpr int MAX_SPLIT = 16;
constexpr int TILE_FLOATS = MT * LD;

__device__ __forceinline__ void lex_min(float &d, int &i, float od, int oi)
{
    const bool take = (od < d) || (od == d && oi < i);
    d = take ? od : d;
    i = take ? oi : i;
}

__global__ __launch_bounds__(MATCH_THREADS, 2) void match_f32_kernel(
    const float *__restrict__ a_hat, const float *__restrict__ q_hat, int B, int Cp, int cap_a, int cap_q,
    const int32_t *__restrict__ n_a, const int32_t *__restrict__ n_q, float thr, int T, int S,
    float *__restrict__ min_dist, int32_t *__restrict__ argmin, uint8_t *__restrict__ valid,
    float *__restrict__ ws_dist, int32_t *__restrict__ ws_idx)
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

        const float *Qs = smem + cur * 2 * TILE_FLOATS + (wm * 64 + l31) * LD + hi;
        const float *As = smem + cur * 2 * TILE_FLOATS + TILE_FLOATS + (wn * 64 + l31) * LD + hi;
#pragma unroll
        for (int ks = 0; ks < BK / 2; ++ks) {
            const float q0 = Qs[2 * ks], q1 = Qs[32 * LD + 2 * ks];
            const float b0 = As[2 * ks], b1 = As[32 * LD + 2 * ks];
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
            min_dist[o] = d;
            argmin[o] = (i == 0x7fffffff) ? 0 : i;
            valid[o] = (d < thr) ? 1 : 0;
        } else {
            const size_t o = ((size_t)p * S + split) * cap_a + a0 + t;
            ws_dist[o] = d;
            ws_idx[o] = i;
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
    // aim for >= ~1024 workgroups (256 CUs x 2 resident x 2 rounds) when the batch alone cannot fill the chip
    int S = (1024 + B * T - 1) / (B * T);
    if (S < 1) S = 1;
    if (S > MAX_SPLIT) S = MAX_SPLIT;
    return S;
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
    const int groups = ((B + 7) / 8) * 8 * T * S;
    hipLaunchKernelGGL(match_f32_kernel, dim3(groups), dim3(MATCH_THREADS), 0, as_stream(stream), a_hat, q_hat, B, C, cap_a,
                       cap_q, n_a, n_q, threshold, T, S, min_dist, argmin, valid, ws_dist, ws_idx);
    ORYON_CHECK_LAUNCH();
    if (S > 1) {
        hipLaunchKernelGGL(match_merge_kernel, dim3(cap_a / 256 + 1, B), dim3(256), 0, as_stream(stream), ws_dist, ws_idx, B,
                           S, cap_a, n_a, threshold, min_dist, argmin, valid);
        ORYON_CHECK_LAUNCH();
    }
    return ORYON_OK;
}
