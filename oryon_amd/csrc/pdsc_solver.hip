// K5-K10: PointDSC seed selection, per-seed consensus sets, power iteration, weighted Kabsch, hypothesis
// scoring and the iterative refinement - all device-side, no host synchronisation.
//
// Replaces models/pointdsc/PointDSC.py:199-217 (pick_seeds), :234-336 (cal_seed_trans), :338-358
// (cal_leading_eigenvector, incl. the joint allclose early exit over all seeds of a pair), :403-438
// (post_refinement; the reference syncs to the host every iteration through int(inlier_num - previous) and
// ships every 3x3 covariance to the CPU for LAPACK), and models/pointdsc/common.py:48-69 (knn, evaluated
// only for the seed rows that are consumed).
//
// Ordering semantics: torch.argsort / topk tie order is implementation-defined in the reference
// (SURVEY.md §7); here every ranking is "by value, ties by ascending index" (rank counting), which is what
// a stable sort gives.
#include <stdlib.h>
#include "common.h"
#include "kabsch.h"
#include "pdsc.h"

namespace oryon {

// ------------------------------------------------------------------------------------------------ K5
// is_local_max_i = AND_j (score_i >= score_j  OR  |s_i - s_j| >= R);  seeds = first S of the descending
// order of score * is_local_max (ties by ascending index).
//   pdsc_seed_keys_kernel  (n_cap/64 x B workgroups)  the O(n^2) NMS test: 4 lanes per row, each a quarter of the j range
//   pdsc_seed_rank_kernel  (B workgroups)             bitonic sort of (key, index), first S indices
__global__ __launch_bounds__(256) void pdsc_seed_keys_kernel(const float *__restrict__ src, const float *__restrict__ conf,
                                                              const int32_t *__restrict__ n_rows, int n_cap, float radius,
                                                              float *__restrict__ key)
{
    extern __shared__ float sm[];
    float *px = sm, *py = sm + n_cap, *pz = sm + 2 * n_cap, *sc = sm + 3 * n_cap;
    const int b = blockIdx.y, t = threadIdx.x;
    const int n = n_rows[b];
    if ((int)blockIdx.x * 64 >= n) return;
    for (int i = t; i < n; i += 256) {
        px[i] = src[((size_t)b * n_cap + i) * 3 + 0];
        py[i] = src[((size_t)b * n_cap + i) * 3 + 1];
        pz[i] = src[((size_t)b * n_cap + i) * 3 + 2];
        sc[i] = conf[(size_t)b * n_cap + i];
    }
    __syncthreads();
    const int i = blockIdx.x * 64 + (t >> 2), part = t & 3;
    bool lm = true;
    float s = 0.0f;
    if (i < n) {
        const float x = px[i], y = py[i], z = pz[i];
        s = sc[i];
        const int per = (n + 3) / 4, j0 = part * per, j1 = (j0 + per < n) ? j0 + per : n;
        for (int j = j0; j < j1; ++j) {
            const float dx = x - px[j], dy = y - py[j], dz = z - pz[j];
            const float d = sqrt_rn(dx * dx + dy * dy + dz * dz);
            lm = lm && ((s >= sc[j]) || (d >= radius));
        }
    }
    int ok = lm ? 1 : 0;
    ok &= __shfl_xor(ok, 1);
    ok &= __shfl_xor(ok, 2);
    if (i < n && part == 0) key[(size_t)b * n_cap + i] = s * (ok ? 1.0f : 0.0f);
}

__global__ __launch_bounds__(256) void pdsc_seed_rank_kernel(const float *__restrict__ key_in, const int32_t *__restrict__ n_rows,
                                                              int n_cap, int S_cap, float ratio, int32_t *__restrict__ seeds,
                                                              int32_t *__restrict__ n_seeds)
{
    extern __shared__ float sm[];
    const int b = blockIdx.x, t = threadIdx.x;
    const int n = n_rows[b];
    int S = (int)((double)n * (double)ratio);
    S = S < S_cap ? S : S_cap;
    int P = 256;
    while (P < n) P <<= 1;
    float *key = sm;                                   // [P]
    int *order = reinterpret_cast<int *>(sm + P);       // [P]
    for (int j = t; j < P; j += 256) {
        key[j] = j < n ? key_in[(size_t)b * n_cap + j] : -INFINITY;
        order[j] = j;
    }
    __syncthreads();
    // descending by key, ties by ascending index (what a stable descending sort gives)
    for (int size = 2; size <= P; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int e = t; e < P / 2; e += 256) {
                const int lo = 2 * e - (e & (stride - 1)), hi = lo + stride;
                const bool up = (lo & size) == 0;
                const float kl = key[lo], kh = key[hi];
                const int il = order[lo], ih = order[hi];
                const bool hi_first = (kh > kl) || (kh == kl && ih < il);      // hi belongs before lo
                if (hi_first == up) {
                    key[lo] = kh; key[hi] = kl;
                    order[lo] = ih; order[hi] = il;
                }
            }
            __syncthreads();
        }
    }
    for (int r = t; r < S; r += 256) seeds[(size_t)b * S_cap + r] = order[r];
    if (t == 0) n_seeds[b] = S;
}

// Round 5: K5 as ONE launch - one 16-wave workgroup per pair (n_cap <= 2048).  Two threads per row walk half of the j range each
// with the pair's (x, y, z, score) quadruples in LDS (one broadcast ds_read_b128 per test); `|s_i - s_j| >= R` is tested on the SQUARED
// distance against x0 = the smallest float whose correctly rounded root reaches R (found once per workgroup; sqrt is monotone, so the
// outcome is the one sqrt_rn(d2) >= R gives) - no square root in the O(n^2) loop.  The ranking is RANK COUNTING:
// rank_i = #{j : key_j > key_i, or key_j == key_i and j < i} is the position of row i in "descending by key, ties by ascending index"
// - the order of the 45-barrier bitonic sort it replaces - and rows with rank < S are the seeds.  22.4 + 12.4 us -> one launch.
// Reference: PointDSC.py:199-217 (pick_seeds).
__global__ __launch_bounds__(1024) void pdsc_seeds_fused_kernel(const float *__restrict__ src, const float *__restrict__ conf,
                                                                 const int32_t *__restrict__ n_rows, int n_cap, float radius, int S_cap,
                                                                 float ratio, float *__restrict__ key_out, int32_t *__restrict__ seeds,
                                                                 int32_t *__restrict__ n_seeds)
{
    extern __shared__ float sm[];
    float4 *pt = reinterpret_cast<float4 *>(sm);           // [n_cap] x, y, z, score
    float *key = sm + 4 * n_cap;                           // [n_cap]
    __shared__ float s_x0;
    const int b = blockIdx.x, t = threadIdx.x;
    const int n = n_rows[b];
    int S = (int)((double)n * (double)ratio);
    S = S < S_cap ? S : S_cap;
    for (int i = t; i < n; i += 1024) {
        float4 q;
        q.x = src[((size_t)b * n_cap + i) * 3 + 0];
        q.y = src[((size_t)b * n_cap + i) * 3 + 1];
        q.z = src[((size_t)b * n_cap + i) * 3 + 2];
        q.w = conf[(size_t)b * n_cap + i];
        pt[i] = q;
    }
    if (t == 0) {
        // smallest float x0 with sqrt_rn(x0) >= radius: within a few ulps of radius^2
        float x = radius * radius;
        for (int u = 0; u < 4; ++u) x = __uint_as_float(__float_as_uint(x) - 1u);
        while (!(sqrt_rn(x) >= radius)) x = __uint_as_float(__float_as_uint(x) + 1u);
        s_x0 = x;
    }
    __syncthreads();
    const float x0 = s_x0;
    // thread -> (row i = t & 511 (+ 512, ..), half of the j range = t >> 9, wave-uniform).  The j side of the NMS test comes through the
    // SCALAR cache (src / conf of row j: uniform addresses -> s_load, SGPR operands), the j side of the ranking out of a register by
    // v_readlane: no LDS instruction in either inner loop.  (Measured: a same-address ds_read_b128 / b64 does not broadcast - ~64 LDS
    // cycles per wave instruction; with the (x, y, z, score) quadruples read that way this kernel took 105-118 us.)
    int *part_ok = reinterpret_cast<int *>(key + n_cap);           // [2][n_cap] per-half verdicts, then per-half rank counts
    const int part = __builtin_amdgcn_readfirstlane(t >> 9);
    const int lane = t & 63;
    const int per = (n + 1) / 2, j0 = part * per, j1 = (j0 + per < n) ? j0 + per : n;
    const float *sp = src + (size_t)b * n_cap * 3, *cp = conf + (size_t)b * n_cap;
    for (int base = 0; base < n; base += 512) {
        const int i = base + (t & 511);
        const float4 me = i < n ? pt[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        bool lm = true;
        int j = j0;
        for (; j + 8 <= j1; j += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float ox = sp[3 * (j + u)], oy = sp[3 * (j + u) + 1], oz = sp[3 * (j + u) + 2], ow = cp[j + u];
                const float dx = me.x - ox, dy = me.y - oy, dz = me.z - oz;
                const float d2 = dx * dx + dy * dy + dz * dz;
                lm = lm & ((me.w >= ow) | (d2 >= x0));
            }
        }
        for (; j < j1; ++j) {
            const float ox = sp[3 * j], oy = sp[3 * j + 1], oz = sp[3 * j + 2], ow = cp[j];
            const float dx = me.x - ox, dy = me.y - oy, dz = me.z - oz;
            const float d2 = dx * dx + dy * dy + dz * dz;
            lm = lm & ((me.w >= ow) | (d2 >= x0));
        }
        if (i < n) part_ok[part * n_cap + i] = lm ? 1 : 0;
    }
    __syncthreads();
    for (int i = t; i < n; i += 1024) {
        const float kv = pt[i].w * ((part_ok[i] & part_ok[n_cap + i]) ? 1.0f : 0.0f);
        // rank counting needs a TOTAL order: a NaN confidence (upstream features overflowed) is neither greater than, equal to nor less
        // than anything, two rows would share a rank and a seed slot would keep an index of an earlier step.  NaN ranks last (the most
        // negative finite float; the padding value -inf below stays "neither greater than nor equal to any key").
        key[i] = (kv != kv) ? -3.402823466e38f : kv;
        key_out[(size_t)b * n_cap + i] = kv;
    }
    __syncthreads();
    for (int base = 0; base < n; base += 512) {
        const int i = base + (t & 511);
        const float ki = i < n ? key[i] : 0.0f;
        int rank = 0;
        for (int jb = j0; jb < j1; jb += 64) {
            const float kreg = (jb + lane < j1) ? key[jb + lane] : -INFINITY;      // -inf: neither greater than nor equal to any key
#pragma unroll
            for (int u = 0; u < 64; ++u) {
                const float kj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, kreg), u));
                rank += ((kj > ki) | ((kj == ki) & (jb + u < i))) ? 1 : 0;
            }
        }
        if (i < n) part_ok[part * n_cap + i] = rank;
    }
    __syncthreads();
    for (int i = t; i < n; i += 1024) {
        const int rank = part_ok[i] + part_ok[n_cap + i];
        if (rank < S) seeds[(size_t)b * S_cap + rank] = i;
    }
    if (t == 0) n_seeds[b] = S;
}

// ------------------------------------------------------------------------------------------------ K6 + K7a
// One workgroup per (seed, pair): feature-space kNN of the seed row, then the k x k compatibility matrix
//   M_ab = clamp(1 - (1 - f_a.f_b)/sigma^2, 0) * clamp(1 - (|s_a-s_b| - |t_a-t_b|)^2/sigma_d^2, 0),  M_aa = 0.
constexpr int KNN_MAX_K = 64;
// 2 - 2 f_seed.f_j for every (seed, row) of a pair, with the feature matrix read once per 64-row block instead of once per seed:
// grid (n_cap/64, B), 4 lanes per row (each a quarter of the seeds), the row's C values in registers, seed features in LDS.
// The dot product is the same k-ordered fmaf chain the per-seed kernel used.
template <int C>
__global__ __launch_bounds__(256) void pdsc_seed_dist_kernel(const float *__restrict__ feat_n, const int32_t *__restrict__ n_rows, int n_cap,
                                                              const int32_t *__restrict__ seeds, const int32_t *__restrict__ n_seeds,
                                                              int S_cap, float *__restrict__ dist /*[B,S_cap,n_cap]*/)
{
    extern __shared__ float fs[];           // [S][C]
    const int b = blockIdx.y, t = threadIdx.x;
    const int n = n_rows[b], S = n_seeds[b];
    if ((int)blockIdx.x * 64 >= n || S <= 0) return;
    const float *F = feat_n + (size_t)b * n_cap * C;
    for (int e = t; e < S * C; e += 256) fs[e] = F[(size_t)seeds[(size_t)b * S_cap + e / C] * C + e % C];
    __syncthreads();
    const int j = blockIdx.x * 64 + (t >> 2), part = t & 3;
    if (j >= n) return;
    float row[C];
    const float4 *rp = reinterpret_cast<const float4 *>(F + (size_t)j * C);
#pragma unroll
    for (int c4 = 0; c4 < C / 4; ++c4) {
        const float4 v = rp[c4];
        row[4 * c4] = v.x; row[4 * c4 + 1] = v.y; row[4 * c4 + 2] = v.z; row[4 * c4 + 3] = v.w;
    }
    for (int s = part; s < S; s += 4) {
        const float4 *f4 = reinterpret_cast<const float4 *>(fs + s * C);
        float acc = 0.0f;
#pragma unroll
        for (int c4 = 0; c4 < C / 4; ++c4) {
            const float4 f = f4[c4];
            acc = fmaf(f.x, row[4 * c4], acc);
            acc = fmaf(f.y, row[4 * c4 + 1], acc);
            acc = fmaf(f.z, row[4 * c4 + 2], acc);
            acc = fmaf(f.w, row[4 * c4 + 3], acc);
        }
        dist[((size_t)b * S_cap + s) * n_cap + j] = 2.0f - 2.0f * acc;
    }
}

// Round 5: the same distances on the fp32 matrix pipe.  v_mfma_f32_32x32x2_f32 accumulates its two k values one after the other
// into each output element, i.e. it IS the k-ordered fmaf chain (the property the exact matcher K1 rests on, match.hip): D[row, seed]
// comes out bit for bit as pdsc_seed_dist_kernel computes it, at 64 MFMAs per 32 x 32 tile instead of 128 dependent VALU fmas per
// element.  Workgroup = 64 rows of a pair; wave w = (row half w & 1, seed tiles w >> 1, + 2, ..); seed features in LDS, rows padded
// to 129 floats (a lane's seed = its column: conflict-free).  39.5 -> ~10 us per 64 registrations.
typedef float f32x16s __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void pdsc_seed_dist_mfma_kernel(const float *__restrict__ feat_n, const int32_t *__restrict__ n_rows, int n_cap,
                                                                   const int32_t *__restrict__ seeds, const int32_t *__restrict__ n_seeds,
                                                                   int S_cap, float *__restrict__ dist /*[B,S_cap,n_cap]*/)
{
    constexpr int C = 128, FS = C + 1;
    extern __shared__ float fs[];           // [S_pad][129]
    const int b = blockIdx.y, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int n = n_rows[b], S = n_seeds[b];
    if ((int)blockIdx.x * 64 >= n || S <= 0) return;
    const int tiles = (S + 31) / 32;
    const float *F = feat_n + (size_t)b * n_cap * C;
    for (int e = t; e < tiles * 32 * C; e += 256) {
        const int sd = e / C, c = e % C;
        fs[sd * FS + c] = sd < S ? F[(size_t)seeds[(size_t)b * S_cap + sd] * C + c] : 0.0f;
    }
    const int half = lane >> 5, col = lane & 31;
    const int j = blockIdx.x * 64 + (wave & 1) * 32 + col;          // this lane's A row (rows >= n are zero rows of feat_n)
    float a[C / 2];                                                  // the row's values k = 2 step + half
    {
        const float4 *rp = reinterpret_cast<const float4 *>(F + (size_t)j * C);
#pragma unroll
        for (int q = 0; q < C / 4; ++q) {
            const float4 v = rp[q];
            a[2 * q] = half ? v.y : v.x;
            a[2 * q + 1] = half ? v.w : v.z;
        }
    }
    __syncthreads();
    for (int st = wave >> 1; st < tiles; st += 2) {
        f32x16s acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const float *bp = fs + (st * 32 + col) * FS + half;
#pragma unroll
        for (int step = 0; step < C / 2; ++step) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[step], bp[2 * step], acc, 0, 0, 0);
        const int sd = st * 32 + col;                                // the lane's column = seed
        if (sd < S) {
            float *dp = dist + ((size_t)b * S_cap + sd) * n_cap + blockIdx.x * 64 + (wave & 1) * 32;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int r0 = 8 * g + 4 * half;                     // rows r0 .. r0 + 3 of the tile
                float4 o;
                o.x = 2.0f - 2.0f * acc[4 * g + 0];
                o.y = 2.0f - 2.0f * acc[4 * g + 1];
                o.z = 2.0f - 2.0f * acc[4 * g + 2];
                o.w = 2.0f - 2.0f * acc[4 * g + 3];
                *reinterpret_cast<float4 *>(dp + r0) = o;
            }
        }
    }
}

__global__ __launch_bounds__(256) void pdsc_knn_matrix_kernel(const float *__restrict__ feat_n, const float *__restrict__ src,
                                                               const float *__restrict__ tgt,
                                                               const int32_t *__restrict__ n_rows, int n_cap, int C,
                                                               const int32_t *__restrict__ seeds,
                                                               const int32_t *__restrict__ n_seeds, int S_cap, int k_cfg,
                                                               float inv_sigma2, float inv_sigma_d2,
                                                               int32_t *__restrict__ knn_out, float *__restrict__ M_out,
                                                               const float *__restrict__ dist_pre /* [B,S_cap,n_cap] or NULL */, int n_batch)
{
    extern __shared__ float sm[];
    // XCD-aware block map: the seeds of one pair gather rows of the same feature matrix (256 KB at n = 500, C = 128), so they run on one XCD
    // (linear block id mod 8) and share it through that L2.  gridDim.y is the pair count rounded up to a multiple of 8.
    const int lin = blockIdx.x + gridDim.x * blockIdx.y;
    const int b = (lin / 8 / (int)gridDim.x) * 8 + (lin & 7), s = (lin / 8) % (int)gridDim.x;
    if (b >= n_batch || s >= n_seeds[b]) return;
    const int n = n_rows[b];
    const int k = k_cfg < n - 1 ? k_cfg : n - 1;
    int P = 256;                            // sort size: power of two >= n
    while (P < n) P <<= 1;
    float *dist = sm;                       // [P] sort keys
    int *order = reinterpret_cast<int *>(dist + P);           // [P] row indices, sorted along with the keys
    float *fs = reinterpret_cast<float *>(order + P);          // [C] seed feature
    float *kf = fs + C;                     // [KNN_MAX_K][C+4]: 16-byte aligned rows for the b128 reads of the matrix phase
    float *kc = kf + KNN_MAX_K * (C + 4);   // [KNN_MAX_K][6]
    int *kidx = reinterpret_cast<int *>(kc + KNN_MAX_K * 6);  // [KNN_MAX_K]
    const int t = threadIdx.x;
    const int seed_row = seeds[(size_t)b * S_cap + s];
    const float *F = feat_n + (size_t)b * n_cap * C;
    for (int c = t; c < C; c += 256) fs[c] = F[(size_t)seed_row * C + c];
    __syncthreads();
    // 2 - 2 f_seed.f_j : one thread per row (16-byte loads, all of a row's loads independent), k-ordered fmaf chain
    for (int j = t; j < P; j += 256) {
        float d = INFINITY;
        if (j < n && dist_pre) {
            d = dist_pre[((size_t)b * S_cap + s) * n_cap + j];
        } else if (j < n) {
            const float4 *row = reinterpret_cast<const float4 *>(F + (size_t)j * C);
            float acc = 0.0f;
            for (int c4 = 0; c4 < C / 4; c4 += 8) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = (c4 + u < C / 4) ? row[c4 + u] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (c4 + u < C / 4) {
                        const float *f4 = fs + 4 * (c4 + u);
                        acc = fmaf(f4[0], v[u].x, acc);
                        acc = fmaf(f4[1], v[u].y, acc);
                        acc = fmaf(f4[2], v[u].z, acc);
                        acc = fmaf(f4[3], v[u].w, acc);
                    }
                }
            }
            d = 2.0f - 2.0f * acc;
        }
        dist[j] = d;
        order[j] = j;
    }
    __syncthreads();
    // ascending bitonic sort of (distance, row index): ties resolve to the smaller index; rank 0 is dropped, ranks 1..k are the
    // neighbours (common.py:68)
    for (int size = 2; size <= P; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int e = t; e < P / 2; e += 256) {
                const int lo = 2 * e - (e & (stride - 1));          // index with bit `stride` clear
                const int hi = lo + stride;
                const bool up = (lo & size) == 0;
                const float dl = dist[lo], dh = dist[hi];
                const int il = order[lo], ih = order[hi];
                const bool lo_gt = (dl > dh) || (dl == dh && il > ih);
                if (lo_gt == up) {
                    dist[lo] = dh; dist[hi] = dl;
                    order[lo] = ih; order[hi] = il;
                }
            }
            __syncthreads();
        }
    }
    if (t < k) kidx[t] = order[t + 1];
    __syncthreads();
    for (int e = t; e < k * C; e += 256) {
        const int a = e / C, c = e % C;
        kf[a * (C + 4) + c] = F[(size_t)kidx[a] * C + c];
    }
    for (int e = t; e < k * 3; e += 256) {
        const int a = e / 3, d = e % 3;
        kc[a * 6 + d] = src[((size_t)b * n_cap + kidx[a]) * 3 + d];
        kc[a * 6 + 3 + d] = tgt[((size_t)b * n_cap + kidx[a]) * 3 + d];
    }
    if (t < k) knn_out[((size_t)b * S_cap + s) * k_cfg + t] = kidx[t];
    __syncthreads();
    float *Mo = M_out + ((size_t)b * S_cap + s) * k_cfg * k_cfg;
    // the matrix is symmetric: one thread per unordered pair (a < c2), both entries written; the diagonal is zero
    for (int e = t; e < k; e += 256) Mo[e * k_cfg + e] = 0.0f;
    const int n_pairs = k * (k - 1) / 2;
    for (int e = t; e < n_pairs; e += 256) {
        // e -> (a, c2), a < c2, row-major over the strict upper triangle
        int a = (int)((2.0f * k - 1.0f - sqrt_rn((2.0f * k - 1.0f) * (2.0f * k - 1.0f) - 8.0f * (float)e)) * 0.5f);
        while (a > 0 && a * (2 * k - a - 1) / 2 > e) --a;
        while ((a + 1) * (2 * k - a - 2) / 2 <= e) ++a;
        const int c2 = a + 1 + (e - a * (2 * k - a - 1) / 2);
        float dot = 0.0f;
        const float4 *fa = reinterpret_cast<const float4 *>(kf + a * (C + 4)), *fb = reinterpret_cast<const float4 *>(kf + c2 * (C + 4));
        for (int c4 = 0; c4 < C / 4; ++c4) {
            const float4 x = fa[c4], y = fb[c4];
            dot = fmaf(x.x, y.x, dot);
            dot = fmaf(x.y, y.y, dot);
            dot = fmaf(x.z, y.z, dot);
            dot = fmaf(x.w, y.w, dot);
        }
        float fm = 1.0f - (1.0f - dot) * inv_sigma2;
        fm = fm > 0.0f ? fm : 0.0f;
        const float *pa = kc + a * 6, *pb = kc + c2 * 6;
        const float dx = pa[0] - pb[0], dy = pa[1] - pb[1], dz = pa[2] - pb[2];
        const float ex = pa[3] - pb[3], ey = pa[4] - pb[4], ez = pa[5] - pb[5];
        const float df = sqrt_rn(dx * dx + dy * dy + dz * dz) - sqrt_rn(ex * ex + ey * ey + ez * ez);
        float smv = 1.0f - df * df * inv_sigma_d2;
        smv = smv > 0.0f ? smv : 0.0f;
        const float v = fm * smv;
        Mo[a * k_cfg + c2] = v;
        Mo[c2 * k_cfg + a] = v;
    }
}

// The same kNN + compatibility matrix with ONE WAVE per (seed, pair) and no workgroup barrier (round 3; n_cap <= 64 R, distances from
// pdsc_seed_dist_kernel): the 256-thread kernel above spends its time in the 45 barrier-separated stages of a 512-element bitonic sort of
// which only ranks 1..k (k = 40) are used (100 us per 64 registrations, 219 us outliers).  Here a lane holds R distances as order-preserving
// integer keys, sorts its own R (stable odd-even transposition in registers: ties keep the smaller row index first), and the wave then
// extracts the k + 1 smallest (key, row) pairs one by one: minimum over the lanes' heads (two 6-step xor reductions: key, then the smallest
// row among the lanes that hold it), the winning lane pops its head.  Same order as the full sort - ascending distance, ties by row index -
// hence the same neighbours, the same matrix, bit for bit.  The neighbour rows go to LDS (22 KB per wave: 7 waves per CU) for the matrix phase.
// MEASURED: 91.5 us against 100 us - the kernel is bound by its instruction count (3200 waves x ~7.5 k VALU instructions over 1024 SIMDs:
// the 780 k-ordered 128-term dot products of the matrix phase and the 41 extraction rounds), not by the barriers; ORYON_PDSC_KNN_WAVE=1
// selects it, the tests run both.
// wave-wide minimum on the DPP data path (row shifts inside the 16-lane rows, then the two row broadcasts gfx9 has), result read from lane 63:
// ~10 dependent VALU instructions; the same reduction written with __shfl_xor is six dependent ds_bpermute round trips (the first version of
// this kernel, latency-bound on them, was no faster than the sort it replaces)
__device__ __forceinline__ unsigned wave_min_u32(unsigned v)
{
#define ORYON_DPP_MIN(ctrl, rmask)                                                                                              \
    {                                                                                                                           \
        const unsigned o = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, ctrl, rmask, 0xf, false);                      \
        v = o < v ? o : v;                                                                                                      \
    }
    ORYON_DPP_MIN(0x111, 0xf)      // row_shr:1
    ORYON_DPP_MIN(0x112, 0xf)      // row_shr:2
    ORYON_DPP_MIN(0x114, 0xf)      // row_shr:4
    ORYON_DPP_MIN(0x118, 0xf)      // row_shr:8   -> lane 15 of every row holds the row's minimum
    ORYON_DPP_MIN(0x142, 0xa)      // row_bcast:15 into rows 1 and 3
    ORYON_DPP_MIN(0x143, 0xc)      // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's minimum
#undef ORYON_DPP_MIN
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

template <int C, int R>
__global__ __launch_bounds__(64) void pdsc_knn_wave_kernel(const float *__restrict__ feat_n, const float *__restrict__ src,
                                                            const float *__restrict__ tgt, const int32_t *__restrict__ n_rows, int n_cap,
                                                            const int32_t *__restrict__ n_seeds, int S_cap, int k_cfg, float inv_sigma2,
                                                            float inv_sigma_d2, int32_t *__restrict__ knn_out, float *__restrict__ M_out,
                                                            const float *__restrict__ dist_pre /* [B,S_cap,n_cap] */, int n_batch)
{
    extern __shared__ float sm[];
    const int lin = blockIdx.x + gridDim.x * blockIdx.y;
    const int b = (lin / 8 / (int)gridDim.x) * 8 + (lin & 7), s = (lin / 8) % (int)gridDim.x;
    if (b >= n_batch || s >= n_seeds[b]) return;
    const int n = n_rows[b];
    const int k = k_cfg < n - 1 ? k_cfg : n - 1;
    float *kf = sm;                               // [k_cfg][C + 4]
    float *kc = kf + k_cfg * (C + 4);             // [k_cfg][6]
    const int lane = threadIdx.x;
    const float *F = feat_n + (size_t)b * n_cap * C;
    const float *dp = dist_pre + ((size_t)b * S_cap + s) * n_cap;
    // order-preserving keys (no -0.0 / NaN among 2 - 2 dot); absent rows: the largest key
    unsigned key[R];
    int rr[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int j = lane + 64 * r;
        unsigned kx = 0xffffffffu;
        if (j < n) {
            const unsigned bits = __float_as_uint(dp[j]);
            kx = (bits >> 31) ? ~bits : (bits | 0x80000000u);
        }
        key[r] = kx;
        rr[r] = r;
    }
    // the lane's own R entries, ascending, stable (strict >): equal keys keep ascending row order
#pragma unroll
    for (int pass = 0; pass < R; ++pass) {
#pragma unroll
        for (int r = pass & 1; r + 1 < R; r += 2) {
            const bool sw = key[r] > key[r + 1];
            const unsigned ka = key[r], kb = key[r + 1];
            const int ra = rr[r], rb = rr[r + 1];
            key[r] = sw ? kb : ka; key[r + 1] = sw ? ka : kb;
            rr[r] = sw ? rb : ra;  rr[r + 1] = sw ? ra : rb;
        }
    }
    int mine = 0;                                 // lane t holds neighbour t (rank t + 1 of the order)
    for (int t = 0; t <= k; ++t) {
        const unsigned mk = wave_min_u32(key[0]);
        const unsigned cand = key[0] == mk ? (unsigned)(lane + 64 * rr[0]) : 0xffffffffu;
        const unsigned mi = wave_min_u32(cand);
        if (lane == (int)(mi & 63u)) {
#pragma unroll
            for (int r = 0; r + 1 < R; ++r) { key[r] = key[r + 1]; rr[r] = rr[r + 1]; }
            key[R - 1] = 0xffffffffu;
        }
        if (t >= 1 && lane == t - 1) mine = (int)mi;
    }
    if (lane < k) knn_out[((size_t)b * S_cap + s) * k_cfg + lane] = mine;
    // neighbour features / coordinates -> LDS
    static_assert(C == 128, "two channels per lane");
    for (int a = 0; a < k; ++a) {
        const int row = __shfl(mine, a);
        const float2 v = reinterpret_cast<const float2 *>(F + (size_t)row * C)[lane];
        *reinterpret_cast<float2 *>(kf + a * (C + 4) + 2 * lane) = v;
    }
    if (lane < k) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            kc[lane * 6 + d] = src[((size_t)b * n_cap + mine) * 3 + d];
            kc[lane * 6 + 3 + d] = tgt[((size_t)b * n_cap + mine) * 3 + d];
        }
    }
    __syncthreads();
    float *Mo = M_out + ((size_t)b * S_cap + s) * k_cfg * k_cfg;
    for (int e = lane; e < k; e += 64) Mo[e * k_cfg + e] = 0.0f;
    const int n_pairs = k * (k - 1) / 2;
    for (int e = lane; e < n_pairs; e += 64) {
        int a = (int)((2.0f * k - 1.0f - sqrt_rn((2.0f * k - 1.0f) * (2.0f * k - 1.0f) - 8.0f * (float)e)) * 0.5f);
        while (a > 0 && a * (2 * k - a - 1) / 2 > e) --a;
        while ((a + 1) * (2 * k - a - 2) / 2 <= e) ++a;
        const int c2 = a + 1 + (e - a * (2 * k - a - 1) / 2);
        float dot = 0.0f;
        const float4 *fa = reinterpret_cast<const float4 *>(kf + a * (C + 4)), *fb = reinterpret_cast<const float4 *>(kf + c2 * (C + 4));
        for (int c4 = 0; c4 < C / 4; ++c4) {
            const float4 x = fa[c4], y = fb[c4];
            dot = fmaf(x.x, y.x, dot);
            dot = fmaf(x.y, y.y, dot);
            dot = fmaf(x.z, y.z, dot);
            dot = fmaf(x.w, y.w, dot);
        }
        float fm = 1.0f - (1.0f - dot) * inv_sigma2;
        fm = fm > 0.0f ? fm : 0.0f;
        const float *pa = kc + a * 6, *pb = kc + c2 * 6;
        const float dx = pa[0] - pb[0], dy = pa[1] - pb[1], dz = pa[2] - pb[2];
        const float ex = pa[3] - pb[3], ey = pa[4] - pb[4], ez = pa[5] - pb[5];
        const float df = sqrt_rn(dx * dx + dy * dy + dz * dz) - sqrt_rn(ex * ex + ey * ey + ez * ez);
        float smv = 1.0f - df * df * inv_sigma_d2;
        smv = smv > 0.0f ? smv : 0.0f;
        const float v = fm * smv;
        Mo[a * k_cfg + c2] = v;
        Mo[c2 * k_cfg + a] = v;
    }
}

// ------------------------------------------------------------------------------------------------ K7b + K8 + K9
// Seed hypotheses in three small launches over (seed, pair) instead of one workgroup per pair:
//   power    one wave per (seed, pair): the leading eigenvector iteration of the k x k compatibility matrix with the row of M
//            in registers; ALL num_iterations iterates and their closeness flags are stored, because the reference stops every
//            seed of a pair at the first iteration at which all of them are close (PointDSC.py:347-357, joint allclose)
//   solve    one wave per (seed, pair): picks that joint iteration, normalises the weights, weighted Kabsch on the k neighbours,
//            fitness over all n correspondences
//   select   one workgroup per pair: first-argmax over the seeds, best transform, labels
// Arithmetic per seed is unchanged from the single-workgroup version (seeds only interact through the stopping iteration).
constexpr int HYP_MAX_IT = 16;
__global__ __launch_bounds__(64) void pdsc_power_kernel(const int32_t *__restrict__ n_rows, const int32_t *__restrict__ n_seeds,
                                                         int S_cap, int k_cfg, int num_iterations, const float *__restrict__ Mmat,
                                                         float *__restrict__ v_hist /*[B,S_cap,it,64]*/,
                                                         int32_t *__restrict__ close_hist /*[B,S_cap,it]*/)
{
    const int b = blockIdx.y, s = blockIdx.x, lane = threadIdx.x;
    if (s >= n_seeds[b]) return;
    const int n = n_rows[b];
    const int k = k_cfg < n - 1 ? k_cfg : n - 1;
    __shared__ float v_sh[KNN_MAX_K];
    float mrow[KNN_MAX_K];
    const float *Ms = Mmat + ((size_t)b * S_cap + s) * k_cfg * k_cfg;
#pragma unroll
    for (int c = 0; c < KNN_MAX_K; ++c) mrow[c] = (lane < k && c < k) ? Ms[lane * k_cfg + c] : 0.0f;
    float vl = 1.0f;
    for (int it = 0; it < num_iterations; ++it) {
        v_sh[lane] = vl;
        __syncthreads();
        float u = 0.0f;
#pragma unroll
        for (int c = 0; c < KNN_MAX_K; ++c)
            if (c < k) u = fmaf(mrow[c], v_sh[c], u);
        __syncthreads();
        if (lane >= k) u = 0.0f;
        float sq = u * u;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off);
        const float vn = u / (sqrt_rn(sq) + 1e-6f);
        const bool close = (lane >= k) || (fabsf(vn - vl) <= 1e-8f + 1e-5f * fabsf(vl));
        const bool all = __all(close);
        const size_t h = ((size_t)b * S_cap + s) * num_iterations + it;
        v_hist[h * KNN_MAX_K + lane] = vn;
        if (lane == 0) close_hist[h] = all ? 1 : 0;
        vl = (lane < k) ? vn : 1.0f;
    }
}

// Round 5: K6b + K7a as ONE launch per registration (C = 128, n_cap <= 1024): one workgroup per (seed, pair) takes the seed's
// feature distances from pdsc_seed_dist_kernel, selects the k + 1 nearest rows with a WAVE-LEVEL bitonic top-64 - every lane holds EPT
// (key, row) pairs as 64-bit composites; the 64-element lists are sorted in registers (shuffles, no barrier), merged pairwise by
// min(a_i, b_63-i) + a 6-stage bitonic merge, across the four waves through LDS with two barriers - i.e. ascending distance, ties by
// row index: the order of the 45-barrier bitonic sort of all n_cap rows it replaces, hence the same neighbours; then builds the k x k
// compatibility matrix and writes it to M_out for pdsc_power_kernel (the power iteration stays its own launch: inside this kernel one
// wave iterated while the workgroup's 31 KB of LDS kept all but five workgroups off the CU, +77 us - DESIGN.md round 5).  Replaces
// pdsc_knn_matrix_kernel.  Reference: common.py:48-69 (knn), PointDSC.py:257-281 (compatibility); :338-358 is pdsc_power_kernel.
__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int m)
{
    const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)v, m), hi = (unsigned)__shfl_xor((int)(unsigned)(v >> 32), m);
    return ((unsigned long long)hi << 32) | lo;
}
// one compare-exchange stage of a 64-lane bitonic network: lanes whose `keep_min` is set keep the smaller of (own, partner)
__device__ __forceinline__ unsigned long long bitonic_cx(unsigned long long x, int stride, bool keep_min)
{
    const unsigned long long p = shfl_xor_u64(x, stride);
    const bool lt = x < p;
    return (lt == keep_min) ? x : p;
}
// full sort of one value per lane (21 stages); desc: descending order
__device__ __forceinline__ unsigned long long wave_sort64(unsigned long long x, int lane, bool desc)
{
#pragma unroll
    for (int size = 2; size <= 64; size <<= 1)
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            const bool up = (((lane & size) == 0) || size == 64) != desc;
            x = bitonic_cx(x, stride, ((lane & stride) == 0) == up);
        }
    return x;
}
// a bitonic sequence (one value per lane) -> sorted (6 stages)
__device__ __forceinline__ unsigned long long wave_merge64(unsigned long long x, int lane, bool desc)
{
#pragma unroll
    for (int stride = 32; stride > 0; stride >>= 1) x = bitonic_cx(x, stride, ((lane & stride) == 0) != desc);
    return x;
}

template <int EPT>
__global__ __launch_bounds__(256) void pdsc_hyp_fused_kernel(const float *__restrict__ feat_n, const float *__restrict__ src,
                                                              const float *__restrict__ tgt, const int32_t *__restrict__ n_rows, int n_cap,
                                                              const float *__restrict__ dist_pre /* [B,S_cap,n_cap] */,
                                                              const int32_t *__restrict__ n_seeds,
                                                              int S_cap, int k_cfg, float inv_sigma2, float inv_sigma_d2,
                                                              int32_t *__restrict__ knn_out, float *__restrict__ M_out, int n_batch)
{
    constexpr int C = 128;
    static_assert(EPT == 2 || EPT == 4, "lists per wave");
    extern __shared__ float sm[];
    const int lin = blockIdx.x + gridDim.x * blockIdx.y;
    const int b = (lin / 8 / (int)gridDim.x) * 8 + (lin & 7), s = (lin / 8) % (int)gridDim.x;
    if (b >= n_batch || s >= n_seeds[b]) return;
    const int n = n_rows[b];
    const int k = k_cfg < n - 1 ? k_cfg : n - 1;
    unsigned long long *lists = reinterpret_cast<unsigned long long *>(sm);     // [4][64] the waves' sorted 64-lists
    float *kf = sm + 512;                                          // [k_cfg][C + 4]
    float *kc = kf + k_cfg * (C + 4);                              // [k_cfg][6]
    int *kidx = reinterpret_cast<int *>(kc + k_cfg * 6);           // [KNN_MAX_K]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float *F = feat_n + (size_t)b * n_cap * C;
    const float *dp = dist_pre + ((size_t)b * S_cap + s) * n_cap;
    {
        // list e of wave w = rows [(w EPT + e) 64, + 64); order-preserving keys (no NaN / -0.0 among 2 - 2 dot), absent rows last
        unsigned long long x[EPT];
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int j = (wave * EPT + e) * 64 + lane;
            unsigned kx = 0xffffffffu;
            if (j < n) {
                const unsigned bits = __float_as_uint(dp[j]);
                kx = (bits >> 31) ? ~bits : (bits | 0x80000000u);
                kx = kx == 0xffffffffu ? 0xfffffffeu : kx;
            }
            x[e] = ((unsigned long long)kx << 32) | (unsigned)j;
        }
#pragma unroll
        for (int e = 0; e < EPT; ++e) x[e] = wave_sort64(x[e], lane, (e & 1) != 0);          // even lists ascending, odd descending
        // pairwise: the 64 smallest of (asc, desc) = lane-wise minimum, a bitonic sequence
        unsigned long long y;
        if constexpr (EPT == 4) {
            unsigned long long y0 = x[0] < x[1] ? x[0] : x[1], y1 = x[2] < x[3] ? x[2] : x[3];
            y0 = wave_merge64(y0, lane, false);
            y1 = wave_merge64(y1, lane, true);
            y = y0 < y1 ? y0 : y1;
        } else {
            y = x[0] < x[1] ? x[0] : x[1];
        }
        y = wave_merge64(y, lane, false);                              // this wave's 64 smallest, ascending
        lists[wave * 64 + lane] = y;
        __syncthreads();
        if ((wave & 1) == 0) {
            const unsigned long long o = lists[(wave + 1) * 64 + 63 - lane];
            y = wave_merge64(y < o ? y : o, lane, false);
        }
        __syncthreads();
        if (wave == 2) lists[2 * 64 + lane] = y;
        __syncthreads();
        if (wave == 0) {
            const unsigned long long o = lists[2 * 64 + 63 - lane];
            y = wave_merge64(y < o ? y : o, lane, false);
            // rank 0 (the row itself, or its duplicate with the smallest index) is dropped; ranks 1..k are the neighbours (common.py:68)
            if (lane >= 1 && lane <= k) kidx[lane - 1] = (int)(unsigned)y;
        }
    }
    __syncthreads();
    for (int e = t; e < k * (C / 4); e += 256) {
        const int a = e / (C / 4), c4 = e % (C / 4);
        *reinterpret_cast<float4 *>(kf + a * (C + 4) + 4 * c4) = reinterpret_cast<const float4 *>(F + (size_t)kidx[a] * C)[c4];
    }
    for (int e = t; e < k * 3; e += 256) {
        const int a = e / 3, d = e % 3;
        kc[a * 6 + d] = src[((size_t)b * n_cap + kidx[a]) * 3 + d];
        kc[a * 6 + 3 + d] = tgt[((size_t)b * n_cap + kidx[a]) * 3 + d];
    }
    if (t < k) knn_out[((size_t)b * S_cap + s) * k_cfg + t] = kidx[t];
    __syncthreads();
    float *Mo = M_out + ((size_t)b * S_cap + s) * k_cfg * k_cfg;
    for (int e = t; e < k; e += 256) Mo[e * k_cfg + e] = 0.0f;
    const int n_pairs = k * (k - 1) / 2;
    for (int e = t; e < n_pairs; e += 256) {
        int a = (int)((2.0f * k - 1.0f - sqrt_rn((2.0f * k - 1.0f) * (2.0f * k - 1.0f) - 8.0f * (float)e)) * 0.5f);
        while (a > 0 && a * (2 * k - a - 1) / 2 > e) --a;
        while ((a + 1) * (2 * k - a - 2) / 2 <= e) ++a;
        const int c2 = a + 1 + (e - a * (2 * k - a - 1) / 2);
        float dot = 0.0f;
        const float4 *fa = reinterpret_cast<const float4 *>(kf + a * (C + 4)), *fb = reinterpret_cast<const float4 *>(kf + c2 * (C + 4));
#pragma unroll 8
        for (int c4 = 0; c4 < C / 4; ++c4) {
            const float4 x = fa[c4], y = fb[c4];
            dot = fmaf(x.x, y.x, dot);
            dot = fmaf(x.y, y.y, dot);
            dot = fmaf(x.z, y.z, dot);
            dot = fmaf(x.w, y.w, dot);
        }
        float fm = 1.0f - (1.0f - dot) * inv_sigma2;
        fm = fm > 0.0f ? fm : 0.0f;
        const float *pa = kc + a * 6, *pb = kc + c2 * 6;
        const float dx = pa[0] - pb[0], dy = pa[1] - pb[1], dz = pa[2] - pb[2];
        const float ex = pa[3] - pb[3], ey = pa[4] - pb[4], ez = pa[5] - pb[5];
        const float df = sqrt_rn(dx * dx + dy * dy + dz * dz) - sqrt_rn(ex * ex + ey * ey + ez * ez);
        float smv = 1.0f - df * df * inv_sigma_d2;
        smv = smv > 0.0f ? smv : 0.0f;
        const float v = fm * smv;
        Mo[a * k_cfg + c2] = v;
        Mo[c2 * k_cfg + a] = v;
    }
}

__global__ __launch_bounds__(64) void pdsc_seed_solve_kernel(const float *__restrict__ src, const float *__restrict__ tgt,
                                                              const int32_t *__restrict__ n_rows, int n_cap,
                                                              const int32_t *__restrict__ n_seeds, int S_cap, int k_cfg,
                                                              int num_iterations, float inlier_thr, const int32_t *__restrict__ knn,
                                                              const float *__restrict__ v_hist, const int32_t *__restrict__ close_hist,
                                                              float *__restrict__ seed_T, float *__restrict__ fitness)
{
    const int b = blockIdx.y, s = blockIdx.x, lane = threadIdx.x;
    const int S = n_seeds[b];
    if (s >= S) return;
    const int n = n_rows[b];
    const int k = k_cfg < n - 1 ? k_cfg : n - 1;
    // joint stopping iteration: the first one at which every seed of the pair is close (else the last)
    int stop = num_iterations - 1;
    for (int it = 0; it < num_iterations; ++it) {
        int ok = 1;
        for (int q = lane; q < S; q += 64) ok &= close_hist[((size_t)b * S_cap + q) * num_iterations + it];
        if (__all(ok)) { stop = it; break; }
    }
    const float *sp = src + (size_t)b * n_cap * 3, *tp = tgt + (size_t)b * n_cap * 3;
    const float vv = lane < k ? v_hist[(((size_t)b * S_cap + s) * num_iterations + stop) * KNN_MAX_K + lane] : 0.0f;
    float sum = vv;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    float w = vv / (sum + 1e-6f);
    w = w < 0.0f ? 0.0f : w;
    KabschAcc acc;
    acc.clear();
    if (lane < k) {
        const int j = knn[((size_t)b * S_cap + s) * k_cfg + lane];
        acc.add(sp[3 * j], sp[3 * j + 1], sp[3 * j + 2], tp[3 * j], tp[3 * j + 1], tp[3 * j + 2], w);
    }
    acc.wave_reduce();
    float T[16];
    acc.solve(T);
    int cnt = 0;
    for (int j = lane; j < n; j += 64) {
        const float x = sp[3 * j], y = sp[3 * j + 1], z = sp[3 * j + 2];
        const float dx = (T[0] * x + T[1] * y + T[2] * z + T[3]) - tp[3 * j];
        const float dy = (T[4] * x + T[5] * y + T[6] * z + T[7]) - tp[3 * j + 1];
        const float dz = (T[8] * x + T[9] * y + T[10] * z + T[11]) - tp[3 * j + 2];
        cnt += (sqrt_rn(dx * dx + dy * dy + dz * dz) < inlier_thr) ? 1 : 0;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
    if (lane == 0) {
        fitness[(size_t)b * S_cap + s] = (float)cnt / (float)n;
#pragma unroll
        for (int i = 0; i < 16; ++i) seed_T[((size_t)b * S_cap + s) * 16 + i] = T[i];
    }
}

__global__ __launch_bounds__(256) void pdsc_seed_select_kernel(const float *__restrict__ src, const float *__restrict__ tgt,
                                                                const int32_t *__restrict__ n_rows, int n_cap,
                                                                const int32_t *__restrict__ n_seeds, int S_cap, float inlier_thr,
                                                                const float *__restrict__ seed_T, const float *__restrict__ fitness,
                                                                int32_t *__restrict__ best, float *__restrict__ T_best,
                                                                uint8_t *__restrict__ labels)
{
    __shared__ int s_best;
    const int b = blockIdx.x, t = threadIdx.x;
    const int n = n_rows[b], S = n_seeds[b];
    if (S <= 0) {
        if (t == 0) best[b] = -1;
        return;
    }
    if (t == 0) {
        int bi = 0;
        float bf = fitness[(size_t)b * S_cap];
        for (int s = 1; s < S; ++s) {
            const float f = fitness[(size_t)b * S_cap + s];
            if (f > bf) { bf = f; bi = s; }                 // first maximum, as torch.argmax
        }
        s_best = bi;
        best[b] = bi;
    }
    __syncthreads();
    float T[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) T[i] = seed_T[((size_t)b * S_cap + s_best) * 16 + i];
    if (t < 16) T_best[(size_t)b * 16 + t] = T[t];
    const float *sp = src + (size_t)b * n_cap * 3, *tp = tgt + (size_t)b * n_cap * 3;
    if (labels)
        for (int j = t; j < n_cap; j += 256) {
            uint8_t lab = 0;
            if (j < n) {
                const float x = sp[3 * j], y = sp[3 * j + 1], z = sp[3 * j + 2];
                const float dx = (T[0] * x + T[1] * y + T[2] * z + T[3]) - tp[3 * j];
                const float dy = (T[4] * x + T[5] * y + T[6] * z + T[7]) - tp[3 * j + 1];
                const float dz = (T[8] * x + T[9] * y + T[10] * z + T[11]) - tp[3 * j + 2];
                lab = (sqrt_rn(dx * dx + dy * dy + dz * dz) < inlier_thr) ? 1 : 0;
            }
            labels[(size_t)b * n_cap + j] = lab;
        }
}

// ------------------------------------------------------------------------------------------------ K10
// post_refinement: <= 20 rounds of {warp, inliers, stop if the count repeats, weighted Kabsch on inliers}.
__global__ __launch_bounds__(256) void pdsc_refine_kernel(const float *__restrict__ src, const float *__restrict__ tgt,
                                                           const int32_t *__restrict__ n_rows, int n_cap, float tau,
                                                           const float *__restrict__ T_in, const int32_t *__restrict__ status_in,
                                                           const int32_t *__restrict__ n_seeds, float *__restrict__ T_out,
                                                           int32_t *__restrict__ status_out)
{
    __shared__ float sT[16];
    __shared__ double s_red[4][17];
    __shared__ int s_cnt[4];
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int n = n_rows[b];
    int st = status_in ? status_in[b] : ORYON_PAIR_OK;
    if (st == ORYON_PAIR_OK && ((n_seeds && n_seeds[b] <= 0) || n < 2)) st = ORYON_PAIR_NO_CORR;
    if (st != ORYON_PAIR_OK) {   // failure path of the reference: identity pose (pipeline.py:341,350)
        if (t < 16) T_out[(size_t)b * 16 + t] = (t % 5 == 0) ? 1.0f : 0.0f;
        if (t == 0 && status_out) status_out[b] = st;
        return;
    }
    if (t < 16) sT[t] = T_in[(size_t)b * 16 + t];
    __syncthreads();
    const float *sp = src + (size_t)b * n_cap * 3, *tp = tgt + (size_t)b * n_cap * 3;
    int prev = 0;
    for (int it = 0; it < 20; ++it) {
        float T[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) T[i] = sT[i];
        KabschAcc acc;
        acc.clear();
        int cnt = 0;
        for (int j = t; j < n; j += 256) {
            const float x = sp[3 * j], y = sp[3 * j + 1], z = sp[3 * j + 2];
            const float bx = tp[3 * j], by = tp[3 * j + 1], bz = tp[3 * j + 2];
            const float dx = (T[0] * x + T[1] * y + T[2] * z + T[3]) - bx;
            const float dy = (T[4] * x + T[5] * y + T[6] * z + T[7]) - by;
            const float dz = (T[8] * x + T[9] * y + T[10] * z + T[11]) - bz;
            const float d = sqrt_rn(dx * dx + dy * dy + dz * dz);
            if (d < tau) {
                ++cnt;
                const float q = d / tau;
                acc.add(x, y, z, bx, by, bz, 1.0f / (1.0f + q * q));
            }
        }
        acc.wave_reduce();
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
        __syncthreads();
        if (lane == 0) {
            s_cnt[wave] = cnt;
            s_red[wave][0] = acc.sw;
#pragma unroll
            for (int i = 0; i < 3; ++i) { s_red[wave][1 + i] = acc.sa[i]; s_red[wave][4 + i] = acc.sb[i]; }
#pragma unroll
            for (int i = 0; i < 9; ++i) s_red[wave][7 + i] = acc.sab[i];
        }
        __syncthreads();
        const int total = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        if (total == prev) break;      // abs(int(inlier_num - previous)) < 1  (PointDSC.py:426)
        prev = total;
        if (t == 0) {
            KabschAcc a;
            a.sw = s_red[0][0] + s_red[1][0] + s_red[2][0] + s_red[3][0];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                a.sa[i] = s_red[0][1 + i] + s_red[1][1 + i] + s_red[2][1 + i] + s_red[3][1 + i];
                a.sb[i] = s_red[0][4 + i] + s_red[1][4 + i] + s_red[2][4 + i] + s_red[3][4 + i];
            }
#pragma unroll
            for (int i = 0; i < 9; ++i) a.sab[i] = s_red[0][7 + i] + s_red[1][7 + i] + s_red[2][7 + i] + s_red[3][7 + i];
            float Tn[16];
            a.solve(Tn);
#pragma unroll
            for (int i = 0; i < 16; ++i) sT[i] = Tn[i];
        }
        __syncthreads();
    }
    if (t < 16) T_out[(size_t)b * 16 + t] = sT[t];
    if (t == 0 && status_out) status_out[b] = ORYON_PAIR_OK;
}

// ------------------------------------------------------------------------------------------------ host side
int pdsc_run_seeds(const PdscModel &M, const float *src, const float *conf, const int32_t *n_rows, int B, int n_cap, int S_cap,
                   int32_t *seeds, int32_t *n_seeds, float *key_scratch /* [B,n_cap] */, hipStream_t st)
{
    static const bool fused_seeds = dev_env_int("ORYON_PDSC_FUSED_SEEDS", 1) != 0;      // dev build: 0 = keys + bitonic rank kernels
    if (fused_seeds && n_cap <= 2048) {
        hipLaunchKernelGGL(pdsc_seeds_fused_kernel, dim3(B), dim3(1024), (size_t)7 * n_cap * sizeof(float), st, src, conf, n_rows, n_cap,
                           M.cfg.nms_radius, S_cap, M.cfg.ratio, key_scratch, seeds, n_seeds);
        return hipGetLastError() == hipSuccess ? ORYON_OK : ORYON_ERR_HIP;
    }
    size_t P = 256;
    while (P < (size_t)n_cap) P <<= 1;
    hipLaunchKernelGGL(pdsc_seed_keys_kernel, dim3(n_cap / 64, B), dim3(256), (size_t)4 * n_cap * sizeof(float), st, src, conf, n_rows, n_cap,
                       M.cfg.nms_radius, key_scratch);
    hipLaunchKernelGGL(pdsc_seed_rank_kernel, dim3(B), dim3(256), 2 * P * sizeof(float), st, key_scratch, n_rows, n_cap, S_cap, M.cfg.ratio,
                       seeds, n_seeds);
    return hipGetLastError() == hipSuccess ? ORYON_OK : ORYON_ERR_HIP;
}

int pdsc_run_hypotheses(const PdscModel &M, const PdscWorkspace &ws, const float *src, const float *tgt, const float *feat_n,
                        const int32_t *n_rows, const int32_t *seeds, const int32_t *n_seeds, int B, int n_cap, float *seed_T,
                        float *fitness, int32_t *best, float *T_best, uint8_t *labels, hipStream_t st)
{
    const int C = M.cfg.num_channels, k = M.cfg.k, S_cap = ws.S_cap;
    size_t P = 256;
    while (P < (size_t)n_cap) P <<= 1;
    const size_t sh1 = (2 * P + C + (size_t)KNN_MAX_K * (C + 4) + KNN_MAX_K * 6 + KNN_MAX_K) * sizeof(float);
    const int nit = M.cfg.num_iterations < HYP_MAX_IT ? M.cfg.num_iterations : HYP_MAX_IT;
    // round 5: distances + kNN (rank counting) + compatibility matrix + power iteration in ONE launch (0 = the separate kernels, dev build)
    static const bool fused_hyp = dev_env_int("ORYON_PDSC_FUSED_HYP", 1) != 0;
    if (fused_hyp && C == 128 && n_cap <= 1024 && k <= KNN_MAX_K - 1 && (size_t)S_cap * 128 * sizeof(float) <= 64 * 1024) {
        static const bool dist_mfma = dev_env_int("ORYON_PDSC_DIST_MFMA", 1) != 0;          // dev build: 0 = the VALU kernel
        if (dist_mfma) {
            const size_t shd = (size_t)((S_cap + 31) / 32 * 32) * 129 * sizeof(float);
            allow_dynamic_lds(reinterpret_cast<const void *>(pdsc_seed_dist_mfma_kernel), (int)shd);
            hipLaunchKernelGGL(pdsc_seed_dist_mfma_kernel, dim3(n_cap / 64, B), dim3(256), shd, st, feat_n, n_rows, n_cap, seeds, n_seeds, S_cap,
                               ws.seed_dist);
        } else
        hipLaunchKernelGGL((pdsc_seed_dist_kernel<128>), dim3(n_cap / 64, B), dim3(256), (size_t)S_cap * 128 * sizeof(float), st, feat_n, n_rows,
                           n_cap, seeds, n_seeds, S_cap, ws.seed_dist);
        const int ept = n_cap <= 512 ? 2 : 4;
        const size_t shf = ((size_t)512 + (size_t)k * (C + 4) + k * 6 + KNN_MAX_K) * sizeof(float);
        const dim3 grid(S_cap, (B + 7) / 8 * 8);
        if (ept == 2)
            hipLaunchKernelGGL(pdsc_hyp_fused_kernel<2>, grid, dim3(256), shf, st, feat_n, src, tgt, n_rows, n_cap, ws.seed_dist, n_seeds, S_cap, k,
                               1.0f / (M.sigma * M.sigma), 1.0f / (M.sigma_d * M.sigma_d), ws.knn, ws.Mmat, B);
        else
            hipLaunchKernelGGL(pdsc_hyp_fused_kernel<4>, grid, dim3(256), shf, st, feat_n, src, tgt, n_rows, n_cap, ws.seed_dist, n_seeds, S_cap, k,
                               1.0f / (M.sigma * M.sigma), 1.0f / (M.sigma_d * M.sigma_d), ws.knn, ws.Mmat, B);
        if (hipGetLastError() != hipSuccess) return ORYON_ERR_HIP;
        hipLaunchKernelGGL(pdsc_power_kernel, dim3(S_cap, B), dim3(64), 0, st, n_rows, n_seeds, S_cap, k, nit, ws.Mmat, ws.v_hist, ws.close_hist);
    } else {
    const float *dist_pre = nullptr;
    if (C == 128 && (size_t)S_cap * 128 * sizeof(float) <= 64 * 1024) {
        hipLaunchKernelGGL((pdsc_seed_dist_kernel<128>), dim3(n_cap / 64, B), dim3(256), (size_t)S_cap * 128 * sizeof(float), st, feat_n, n_rows,
                           n_cap, seeds, n_seeds, S_cap, ws.seed_dist);
        dist_pre = ws.seed_dist;
    }
    // 1: the one-wave-per-seed kernel (same results; 91 vs 100 us per 64 registrations in the kernel trace, no difference in the step: not the default)
    static const bool knn_wave = dev_env_int("ORYON_PDSC_KNN_WAVE", 0) != 0;
    if (knn_wave && dist_pre && n_cap <= 1024 && k <= 63) {
        const size_t shw = ((size_t)k * (128 + 4) + (size_t)k * 6) * sizeof(float);
        if (n_cap <= 512)
            hipLaunchKernelGGL((pdsc_knn_wave_kernel<128, 8>), dim3(S_cap, (B + 7) / 8 * 8), dim3(64), shw, st, feat_n, src, tgt, n_rows, n_cap,
                               n_seeds, S_cap, k, 1.0f / (M.sigma * M.sigma), 1.0f / (M.sigma_d * M.sigma_d), ws.knn, ws.Mmat, dist_pre, B);
        else
            hipLaunchKernelGGL((pdsc_knn_wave_kernel<128, 16>), dim3(S_cap, (B + 7) / 8 * 8), dim3(64), shw, st, feat_n, src, tgt, n_rows, n_cap,
                               n_seeds, S_cap, k, 1.0f / (M.sigma * M.sigma), 1.0f / (M.sigma_d * M.sigma_d), ws.knn, ws.Mmat, dist_pre, B);
    } else
    hipLaunchKernelGGL(pdsc_knn_matrix_kernel, dim3(S_cap, (B + 7) / 8 * 8), dim3(256), sh1, st, feat_n, src, tgt, n_rows, n_cap, C, seeds,
                       n_seeds, S_cap, k, 1.0f / (M.sigma * M.sigma), 1.0f / (M.sigma_d * M.sigma_d), ws.knn, ws.Mmat, dist_pre, B);
    if (hipGetLastError() != hipSuccess) return ORYON_ERR_HIP;
    hipLaunchKernelGGL(pdsc_power_kernel, dim3(S_cap, B), dim3(64), 0, st, n_rows, n_seeds, S_cap, k, nit, ws.Mmat, ws.v_hist, ws.close_hist);
    }
    hipLaunchKernelGGL(pdsc_seed_solve_kernel, dim3(S_cap, B), dim3(64), 0, st, src, tgt, n_rows, n_cap, n_seeds, S_cap, k, nit,
                       M.cfg.inlier_threshold, ws.knn, ws.v_hist, ws.close_hist, seed_T, fitness);
    hipLaunchKernelGGL(pdsc_seed_select_kernel, dim3(B), dim3(256), 0, st, src, tgt, n_rows, n_cap, n_seeds, S_cap, M.cfg.inlier_threshold,
                       seed_T, fitness, best, T_best, labels);
    return hipGetLastError() == hipSuccess ? ORYON_OK : ORYON_ERR_HIP;
}

int pdsc_run_refine(const PdscModel &M, const float *src, const float *tgt, const int32_t *n_rows, int B, int n_cap,
                    const float *T_in, const int32_t *status_in, const int32_t *n_seeds, float *T_out, int32_t *status_out,
                    hipStream_t st)
{
    const float tau = (M.cfg.inlier_threshold == 0.10f) ? 0.10f : 1.2f;   // PointDSC.py:415-418
    hipLaunchKernelGGL(pdsc_refine_kernel, dim3(B), dim3(256), 0, st, src, tgt, n_rows, n_cap, tau, T_in, status_in, n_seeds,
                       T_out, status_out);
    return hipGetLastError() == hipSuccess ? ORYON_OK : ORYON_ERR_HIP;
}

}  // namespace oryon
