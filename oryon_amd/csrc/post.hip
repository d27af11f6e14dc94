// K1b (correspondence selection), K2 (scale / validate / lift) and K8 (batched weighted Kabsch + 3x3 SVD).
// Replaces utils/pcd.py:205-214, pipeline.py:447-460 + utils/coordinates.py:5-48 + utils/pcd.py:44-74 and
// models/pointdsc/common.py:7-45 of the reference.  Small, latency-bound kernels: one workgroup (or one
// wave) per problem, ordered compaction by wave ballots, no host round trips (the reference does an
// H.cpu() -> LAPACK -> .to(device) per Kabsch call).
#include "common.h"
#include "kabsch.h"

namespace oryon {

constexpr int SEL_THREADS = 1024;
constexpr int SEL_WAVES = SEL_THREADS / 64;

__device__ __forceinline__ int block_rank_sel(bool flag, int *s_wave, int &total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long b = __ballot(flag);
    const int before = __popcll(b & ((1ull << lane) - 1ull));
    __syncthreads();
    if (lane == 0) s_wave[wave] = __popcll(b);
    __syncthreads();
    int base = 0, tot = 0;
    const int nw = blockDim.x >> 6;
    for (int w = 0; w < nw; ++w) {
        const int c = s_wave[w];
        base += (w < wave) ? c : 0;
        tot += c;
    }
    total = tot;
    return base + before;
}

// Ordered block scan of per-thread COUNTS (exclusive offset of this thread, chunk total through `total`): lets a thread own several
// consecutive entries per scan step - these one-workgroup-per-pair kernels are bound by the latency of their scan steps.
__device__ __forceinline__ int block_scan_counts_sel(int cnt, int *s_wave, int &total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        incl += (lane >= o) ? v : 0;
    }
    __syncthreads();
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int base = 0, tot = 0;
    const int nw = blockDim.x >> 6;
    for (int w = 0; w < nw; ++w) {
        const int c = s_wave[w];
        base += (w < wave) ? c : 0;
        tot += c;
    }
    total = tot;
    return base + incl - cnt;
}

// One workgroup per pair.  scratch[p, :] receives the ordered list of valid anchor rows.
__global__ __launch_bounds__(SEL_THREADS) void select_corrs_kernel(
    const int32_t *__restrict__ roi_a, const int32_t *__restrict__ roi_q, int stride_a, int stride_q,
    const int32_t *__restrict__ n_a, const int32_t *__restrict__ n_q, const int32_t *__restrict__ argmin,
    const uint8_t *__restrict__ valid, int cap_a, int W, int max_corrs, int corr_rows, uint64_t seed,
    const int64_t *__restrict__ pair_key, int32_t *__restrict__ scratch, int32_t *__restrict__ corrs,
    int32_t *__restrict__ n_valid, int32_t *__restrict__ n_sel, int32_t *__restrict__ status,
    int32_t *__restrict__ sel_rows, const int32_t *__restrict__ pair_eager)
{
    // sel_rows != NULL (lazy matcher, match16.hip): also record WHICH anchor row fills every slot; for pairs whose argmin column is
    // not materialised yet (pair_eager[p] == 0) the query half of the row is left to match_resolve_selected_kernel
    __shared__ int s_wave[SEL_WAVES];
    __shared__ unsigned s_hist[256];
    __shared__ unsigned s_prefix, s_remaining;
    const int p = blockIdx.x;
    const int na = n_a[p], nq = n_q[p];
    int32_t *out = corrs + (size_t)p * corr_rows * 4;
    if (na <= 0 || nq <= 0) {
        if (threadIdx.x == 0) { n_valid[p] = 0; status[p] = ORYON_PAIR_NO_MASK; if (n_sel) n_sel[p] = 0; }
        return;
    }
    const uint8_t *v = valid + (size_t)p * cap_a;
    const int32_t *am = argmin + (size_t)p * cap_a;
    int32_t *list = scratch + (size_t)p * cap_a;
    int nv = 0;
    for (int i0 = 0; i0 < na; i0 += SEL_THREADS * 8) {              // eight consecutive rows per thread and scan step
        const int i = i0 + (int)threadIdx.x * 8;
        unsigned bits = 0u;
#pragma unroll
        for (int e = 0; e < 8; ++e) bits |= (i + e < na && v[i + e] != 0 ? 1u : 0u) << e;
        int tot;
        int o = nv + block_scan_counts_sel(__popc(bits), s_wave, tot);
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (bits & (1u << e)) list[o++] = i + e;
        nv += tot;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        n_valid[p] = nv;
        status[p] = nv > 1 ? ORYON_PAIR_OK : ORYON_PAIR_NO_CORR;
        if (n_sel) n_sel[p] = nv > 1 ? max_corrs : 0;
    }
    if (nv <= 1) return;
    const uint64_t key = pair_key ? (uint64_t)pair_key[p] : (uint64_t)p;
    const int32_t *ra = roi_a + (size_t)p * stride_a, *rq = roi_q + (size_t)p * stride_q;
    const bool query_known = !sel_rows || !pair_eager || pair_eager[p] != 0;
    auto emit = [&](int slot, int row) {
        const int pa = ra[row];
        int4 c;
        c.x = pa / W; c.y = pa % W; c.z = 0; c.w = 0;
        if (query_known) {
            const int pq = rq[am[row]];
            c.z = pq / W; c.w = pq % W;
        }
        *reinterpret_cast<int4 *>(out + (size_t)slot * 4) = c;
        if (sel_rows) sel_rows[(size_t)p * corr_rows + slot] = row;
    };
    if (nv < max_corrs) {
        // with replacement (utils/misc.py:251-252): max_corrs independent uniform draws
        for (int j = threadIdx.x; j < max_corrs; j += SEL_THREADS) {
            const uint32_t u = rng_u32(seed, key, 2u, (uint32_t)j);
            const int pick = (int)(((uint64_t)u * (uint64_t)nv) >> 32);
            emit(j, list[pick]);
        }
        return;
    }
    // without replacement: the max_corrs smallest keys among the nv valid rows, emitted in row order
    if (threadIdx.x == 0) { s_prefix = 0u; s_remaining = (unsigned)max_corrs; }
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        if (threadIdx.x < 256) s_hist[threadIdx.x] = 0u;
        __syncthreads();
        const unsigned prefix = s_prefix;
        const unsigned hi_mask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
        for (int i = threadIdx.x; i < nv; i += SEL_THREADS) {
            const unsigned k = rng_u32(seed, key, 1u, (uint32_t)i);
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
    const unsigned Tk = s_prefix;
    const int ties_to_take = (int)s_remaining;
    int kept = 0, ties_seen = 0;
    for (int i0 = 0; i0 < nv; i0 += SEL_THREADS * 8) {
        const int i = i0 + (int)threadIdx.x * 8;
        unsigned below = 0u, tie = 0u;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const bool in = i + e < nv;
            const unsigned k = in ? rng_u32(seed, key, 1u, (uint32_t)(i + e)) : 0xFFFFFFFFu;
            below |= (in && k < Tk ? 1u : 0u) << e;
            tie |= (in && k == Tk ? 1u : 0u) << e;
        }
        int tie_total, keep_total;
        int tie_rank = ties_seen + block_scan_counts_sel(__popc(tie), s_wave, tie_total);
        unsigned keep = below;
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (tie & (1u << e)) { if (tie_rank < ties_to_take) keep |= 1u << e; ++tie_rank; }
        int pos = kept + block_scan_counts_sel(__popc(keep), s_wave, keep_total);
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (keep & (1u << e)) emit(pos++, list[i + e]);
        kept += keep_total;
        ties_seen += tie_total;
    }
}

// K2.  One workgroup per pair; fp32 operation order is the reference's, with explicit _rn intrinsics so the
// compiler cannot contract mul+sub into fma:  y' = float(y) * (float(HA)/float(FH));  X = ((x - cx) * z) / fx.
__global__ __launch_bounds__(512) void lift_pairs_kernel(
    const int32_t *__restrict__ corrs, const int32_t *__restrict__ n_corr, int n_cap, float sya, float sxa, float syq,
    float sxq, const float *__restrict__ depth_a, int HA, int WA, const float *__restrict__ depth_q, int HQ, int WQ,
    const float *__restrict__ cam_a, const float *__restrict__ cam_q, const int32_t *__restrict__ status,
    float *__restrict__ pcd_a, float *__restrict__ pcd_q, int32_t *__restrict__ n_out)
{
    __shared__ int s_wave[8];
    const int p = blockIdx.x;
    float *oa = pcd_a + (size_t)p * n_cap * 3, *oq = pcd_q + (size_t)p * n_cap * 3;
    // rows past the lifted count are zero (the callers hand over uninitialised buffers: no separate fill launches)
    auto zero_from = [&](int first) {
        for (int e = first * 3 + (int)threadIdx.x; e < n_cap * 3; e += (int)blockDim.x) { oa[e] = 0.0f; oq[e] = 0.0f; }
    };
    if (status && status[p] != ORYON_PAIR_OK) {
        if (threadIdx.x == 0) n_out[p] = 0;
        zero_from(0);
        return;
    }
    const int n = n_corr ? n_corr[p] : n_cap;
    const int32_t *c = corrs + (size_t)p * n_cap * 4;
    const float *da = depth_a + (size_t)p * HA * WA, *dq = depth_q + (size_t)p * HQ * WQ;
    const float fxa = cam_a[p * 9 + 0], cxa = cam_a[p * 9 + 2], fya = cam_a[p * 9 + 4], cya = cam_a[p * 9 + 5];
    const float fxq = cam_q[p * 9 + 0], cxq = cam_q[p * 9 + 2], fyq = cam_q[p * 9 + 4], cyq = cam_q[p * 9 + 5];
    int base = 0;
    for (int i0 = 0; i0 < n; i0 += blockDim.x) {
        const int i = i0 + threadIdx.x;
        bool ok = false;
        float ya = 0, xa = 0, yq = 0, xq = 0;
        if (i < n) {
            const int4 cc = *reinterpret_cast<const int4 *>(c + (size_t)i * 4);
            ya = __fmul_rn((float)cc.x, sya); xa = __fmul_rn((float)cc.y, sxa);
            yq = __fmul_rn((float)cc.z, syq); xq = __fmul_rn((float)cc.w, sxq);
            ok = ya >= 0.0f && ya < (float)HA && xa >= 0.0f && xa < (float)WA && yq >= 0.0f && yq < (float)HQ &&
                 xq >= 0.0f && xq < (float)WQ;
        }
        int tot;
        const int r = block_rank_sel(ok, s_wave, tot);
        if (ok) {
            const int iya = (int)ya, ixa = (int)xa, iyq = (int)yq, ixq = (int)xq;
            const float za = da[(size_t)iya * WA + ixa], zq = dq[(size_t)iyq * WQ + ixq];
            const int o = (base + r) * 3;
            oa[o + 0] = __fdiv_rn(__fdiv_rn(__fmul_rn(__fsub_rn((float)ixa, cxa), za), fxa), 1000.0f);
            oa[o + 1] = __fdiv_rn(__fdiv_rn(__fmul_rn(__fsub_rn((float)iya, cya), za), fya), 1000.0f);
            oa[o + 2] = __fdiv_rn(za, 1000.0f);
            oq[o + 0] = __fdiv_rn(__fdiv_rn(__fmul_rn(__fsub_rn((float)ixq, cxq), zq), fxq), 1000.0f);
            oq[o + 1] = __fdiv_rn(__fdiv_rn(__fmul_rn(__fsub_rn((float)iyq, cyq), zq), fyq), 1000.0f);
            oq[o + 2] = __fdiv_rn(zq, 1000.0f);
        }
        base += tot;
    }
    if (threadIdx.x == 0) n_out[p] = base;
    zero_from(base);
}

// lift_pcd proper (utils/pcd.py:44-74): selected pixels of one depth map -> [n,3] millimetres.
__global__ void lift_points_kernel(const float *__restrict__ depth, int H, int W, const float *__restrict__ cam,
                                   const int32_t *__restrict__ x_idx, const int32_t *__restrict__ y_idx, int n,
                                   float *__restrict__ out)
{
    const float fx = cam[0], cx = cam[2], fy = cam[4], cy = cam[5];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int x = x_idx[i], y = y_idx[i];
        const float z = depth[(size_t)y * W + x];
        out[3 * i + 0] = __fdiv_rn(__fmul_rn(__fsub_rn((float)x, cx), z), fx);
        out[3 * i + 1] = __fdiv_rn(__fmul_rn(__fsub_rn((float)y, cy), z), fy);
        out[3 * i + 2] = z;
    }
}

// K8.  One wave per problem; lanes stride over the m points, fp64 accumulation, wave reduction, lane 0 solves.
__global__ __launch_bounds__(256) void kabsch_batched_kernel(const float *__restrict__ A, const float *__restrict__ Bp,
                                                              const float *__restrict__ w, int nb, int m,
                                                              float *__restrict__ T)
{
    const int prob = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (prob >= nb) return;
    const int lane = threadIdx.x & 63;
    const float *a = A + (size_t)prob * m * 3, *b = Bp + (size_t)prob * m * 3;
    const float *ww = w ? w + (size_t)prob * m : nullptr;
    KabschAcc acc;
    acc.clear();
    for (int i = lane; i < m; i += 64) {
        float wi = ww ? ww[i] : 1.0f;
        wi = wi < 0.0f ? 0.0f : wi;
        acc.add(a[3 * i], a[3 * i + 1], a[3 * i + 2], b[3 * i], b[3 * i + 1], b[3 * i + 2], wi);
    }
    acc.wave_reduce();
    if (lane == 0) {
        float Tm[16];
        acc.solve(Tm);
#pragma unroll
        for (int i = 0; i < 16; ++i) T[(size_t)prob * 16 + i] = Tm[i];
    }
}

}  // namespace oryon

using namespace oryon;

namespace oryon {
// "sample first" (engine option): the matcher runs on a random first-stage subset of the anchors; a pair whose first stage found fewer
// than max_corrs valid rows - and had more anchors than that subset - is redone on all of its anchors.  These two kernels gate the
// second stage on the device and merge its results over the first stage's.
__global__ void sample_first_gate_kernel(int B, const int32_t *__restrict__ n_valid1, const int32_t *__restrict__ n_a1,
                                         const int32_t *__restrict__ n_a, int max_corrs, int32_t *__restrict__ n_a2)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < B) n_a2[p] = (n_valid1[p] < max_corrs && n_a[p] > n_a1[p]) ? n_a[p] : 0;
}

__global__ __launch_bounds__(256) void sample_first_merge_kernel(int corr_rows, const int32_t *__restrict__ n_a2,
                                                                  const int32_t *__restrict__ corrs2, const int32_t *__restrict__ n_valid2,
                                                                  const int32_t *__restrict__ n_sel2, const int32_t *__restrict__ status2,
                                                                  int32_t *__restrict__ corrs1, int32_t *__restrict__ n_valid1,
                                                                  int32_t *__restrict__ n_sel1, int32_t *__restrict__ status1)
{
    const int p = blockIdx.x;
    if (n_a2[p] <= 0) return;                                // first stage stands
    const int4 *src = reinterpret_cast<const int4 *>(corrs2 + (size_t)p * corr_rows * 4);
    int4 *dst = reinterpret_cast<int4 *>(corrs1 + (size_t)p * corr_rows * 4);
    for (int i = threadIdx.x; i < corr_rows; i += 256) dst[i] = src[i];
    if (threadIdx.x == 0) { n_valid1[p] = n_valid2[p]; n_sel1[p] = n_sel2[p]; status1[p] = status2[p]; }
}

// internal entry shared with the lazy matcher (match16.hip)
int select_corrs_launch(const int32_t *roi_a, const int32_t *roi_q, int roi_stride_a, int roi_stride_q, const int32_t *n_a,
                        const int32_t *n_q, const int32_t *argmin, const uint8_t *valid, int cap_a, int B, int W, int max_corrs,
                        int corr_rows, uint64_t seed, const int64_t *pair_key, int32_t *scratch, int32_t *corrs, int32_t *n_valid,
                        int32_t *n_sel, int32_t *status, int32_t *sel_rows, const int32_t *pair_eager, hipStream_t st)
{
    hipLaunchKernelGGL(select_corrs_kernel, dim3(B), dim3(SEL_THREADS), 0, st, roi_a, roi_q, roi_stride_a, roi_stride_q, n_a, n_q, argmin,
                       valid, cap_a, W, max_corrs, corr_rows, seed, pair_key, scratch, corrs, n_valid, n_sel, status, sel_rows, pair_eager);
    return hipGetLastError() == hipSuccess ? ORYON_OK : ORYON_ERR_HIP;
}
}  // namespace oryon

extern "C" int oryon_select_corrs(const int32_t *roi_a, const int32_t *roi_q, int roi_stride_a, int roi_stride_q,
                                  const int32_t *n_a, const int32_t *n_q, const int32_t *argmin, const uint8_t *valid,
                                  int cap_a, int B, int W, int max_corrs, int corr_rows, uint64_t seed,
                                  const int64_t *pair_key, int32_t *scratch, int32_t *corrs, int32_t *n_valid,
                                  int32_t *n_sel, int32_t *status, void *stream)
{
    ORYON_CHECK_ARG(roi_a && roi_q && n_a && n_q && argmin && valid && scratch && corrs && n_valid && status);
    ORYON_CHECK_ARG(B >= 0 && cap_a > 0 && W > 0 && max_corrs > 0 && corr_rows >= max_corrs && roi_stride_a > 0 && roi_stride_q > 0);
    if (B == 0) return ORYON_OK;
    hipLaunchKernelGGL(select_corrs_kernel, dim3(B), dim3(SEL_THREADS), 0, as_stream(stream), roi_a, roi_q, roi_stride_a,
                       roi_stride_q, n_a, n_q, argmin, valid, cap_a, W, max_corrs, corr_rows, seed, pair_key, scratch, corrs,
                       n_valid, n_sel, status, static_cast<int32_t *>(nullptr), static_cast<const int32_t *>(nullptr));
    ORYON_CHECK_LAUNCH();
    return ORYON_OK;
}

extern "C" int oryon_sample_first_gate(const int32_t *n_valid1, const int32_t *n_a1, const int32_t *n_a, int B, int max_corrs, int32_t *n_a2,
                                       void *stream)
{
    ORYON_CHECK_ARG(n_valid1 && n_a1 && n_a && n_a2 && B >= 0 && max_corrs > 0);
    if (B == 0) return ORYON_OK;
    hipLaunchKernelGGL(sample_first_gate_kernel, dim3((B + 255) / 256), dim3(256), 0, as_stream(stream), B, n_valid1, n_a1, n_a, max_corrs, n_a2);
    ORYON_CHECK_LAUNCH();
    return ORYON_OK;
}

extern "C" int oryon_sample_first_merge(const int32_t *n_a2, const int32_t *corrs2, const int32_t *n_valid2, const int32_t *n_sel2,
                                        const int32_t *status2, int B, int corr_rows, int32_t *corrs1, int32_t *n_valid1, int32_t *n_sel1,
                                        int32_t *status1, void *stream)
{
    ORYON_CHECK_ARG(n_a2 && corrs2 && n_valid2 && n_sel2 && status2 && corrs1 && n_valid1 && n_sel1 && status1 && B >= 0 && corr_rows > 0);
    if (B == 0) return ORYON_OK;
    hipLaunchKernelGGL(sample_first_merge_kernel, dim3(B), dim3(256), 0, as_stream(stream), corr_rows, n_a2, corrs2, n_valid2, n_sel2, status2,
                       corrs1, n_valid1, n_sel1, status1);
    ORYON_CHECK_LAUNCH();
    return ORYON_OK;
}

extern "C" int oryon_lift_pairs(const int32_t *corrs, const int32_t *n_corr, int B, int n_cap, int FH, int FW,
                                const float *depth_a, int HA, int WA, const float *depth_q, int HQ, int WQ,
                                const float *cam_a, const float *cam_q, const int32_t *status, float *pcd_a, float *pcd_q,
                                int32_t *n_out, void *stream)
{
    ORYON_CHECK_ARG(corrs && depth_a && depth_q && cam_a && cam_q && pcd_a && pcd_q && n_out);
    ORYON_CHECK_ARG(B >= 0 && n_cap > 0 && FH > 0 && FW > 0 && HA > 0 && WA > 0 && HQ > 0 && WQ > 0);
    if (B == 0) return ORYON_OK;
    const float sya = (float)HA / (float)FH, sxa = (float)WA / (float)FW;
    const float syq = (float)HQ / (float)FH, sxq = (float)WQ / (float)FW;
    hipLaunchKernelGGL(lift_pairs_kernel, dim3(B), dim3(512), 0, as_stream(stream), corrs, n_corr, n_cap, sya, sxa, syq, sxq,
                       depth_a, HA, WA, depth_q, HQ, WQ, cam_a, cam_q, status, pcd_a, pcd_q, n_out);
    ORYON_CHECK_LAUNCH();
    return ORYON_OK;
}

extern "C" int oryon_lift_points(const float *depth, int H, int W, const float *cam9, const int32_t *x_idx,
                                 const int32_t *y_idx, int n, float *out, void *stream)
{
    ORYON_CHECK_ARG(depth && cam9 && x_idx && y_idx && out && H > 0 && W > 0 && n >= 0);
    if (n == 0) return ORYON_OK;
    const int blocks = ceil_div(n, 256) < 1024 ? ceil_div(n, 256) : 1024;
    hipLaunchKernelGGL(lift_points_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), depth, H, W, cam9, x_idx, y_idx, n,
                       out);
    ORYON_CHECK_LAUNCH();
    return ORYON_OK;
}

extern "C" int oryon_kabsch_batched(const float *A, const float *B, const float *w, int nb, int m, float *T, void *stream)
{
    ORYON_CHECK_ARG(A && B && T && nb >= 0 && m > 0);
    if (nb == 0) return ORYON_OK;
    hipLaunchKernelGGL(kabsch_batched_kernel, dim3(ceil_div(nb, 4)), dim3(256), 0, as_stream(stream), A, B, w, nb, m, T);
    ORYON_CHECK_LAUNCH();
    return ORYON_OK;
}
