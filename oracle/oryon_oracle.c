/* C restatement of the integer/bit-exact parts of the Oryon hot path (TEST INFRASTRUCTURE).
 *
 * This file is the bit-exact oracle for the HIP kernels whose outputs are integers or must be
 * reproduced to the last bit (ROI compaction, cosine nearest-neighbour argmin, coordinate
 * scale/validate/truncate, pin-hole lift).  It follows the reference algorithm
 *   utils/pcd.py:184-205      (ROI = row-major nonzero(mask==1); dist = 0.5*(1-cos); amin/argmin; < thr)
 *   utils/coordinates.py:5-48 + pipeline.py:447-460 + utils/pcd.py:44-74   (scale, validate, trunc, lift)
 * and fixes ONE canonical fp32 evaluation order, which is also the order the gfx950 kernels use:
 *   |x|^2  = k-ordered fmaf chain starting at 0          (k = 0..C-1)
 *   x^_k   = x_k / max(sqrtf(|x|^2), 1e-8f)              (IEEE division; eps of torch.cosine_similarity)
 *   dot    = k-ordered fmaf chain of a^_k * b^_k from 0  (== v_mfma_f32_32x32x2_f32 accumulation)
 *   dist   = fmaf(-0.5f, dot, 0.5f)                      (== 0.5f * (1.0f - dot) bit for bit)
 *   argmin = first column attaining the row minimum
 * Pinned against the real reference by tests/test_oracle_goldens.py (form "c").
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * Build: make -C oracle      (gcc -O3 -mfma -mavx2 -ffp-contract=off; no fast-math: fmaf stays one correctly rounded operation)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* row-major compaction of pixels == 1; returns the count; roi_lin[i] = y*W + x */
int orc_roi_from_mask(const int32_t *mask, int H, int W, int32_t *roi_lin)
{
    int n = 0;
    for (int p = 0; p < H * W; ++p)
        if (mask[p] == 1) roi_lin[n++] = p;
    return n;
}

/* gather one pixel's descriptor from a channel-planar [C,HW] map and normalise it */
static void gather_normalise(const float *feat, int C, int HW, int pix, float *out)
{
    float n2 = 0.0f;
    for (int k = 0; k < C; ++k) {
        float v = feat[(size_t)k * HW + pix];
        n2 = fmaf(v, v, n2);
    }
    float d = sqrtf(n2);
    if (d < 1e-8f) d = 1e-8f;
    for (int k = 0; k < C; ++k) out[k] = feat[(size_t)k * HW + pix] / d;
}

/* normalised, gathered descriptors [n,C] (what the HIP gather kernel writes) */
void orc_gather_normalise(const float *feat, int C, int HW, const int32_t *roi_lin, int n, float *out)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) gather_normalise(feat, C, HW, roi_lin[i], out + (size_t)i * C);
}

/* cosine nearest neighbour of every anchor ROI pixel among the query ROI pixels.
 * Every (anchor, query) dot product is its own k-ordered fmaf chain; the loop nest only interleaves QB independent
 * chains (query rows stored block-transposed, [n2/QB][C][QB]) so the compiler can keep them in vector lanes - the
 * value of each chain is what the scalar loop gives, bit for bit.  anchor_rows (optional) restricts the scan to a
 * subset of the anchor rows (full-size parity checks score a sample of rows against ALL query rows). */
#define QB 16
void orc_match_f32_rows(const float *feat_a, const float *feat_q, int C, int HW, const int32_t *roi_a, int n1,
                        const int32_t *roi_q, int n2, float thr, const int32_t *anchor_rows, int n_rows,
                        float *min_dist, int32_t *argmin, uint8_t *valid)
{
    const int nb = (n2 + QB - 1) / QB;
    float *an = (float *)malloc((size_t)(n1 > 0 ? n1 : 1) * C * sizeof(float));
    float *qn = (float *)malloc((size_t)(n2 > 0 ? n2 : 1) * C * sizeof(float));
    float *qt = (float *)calloc((size_t)(nb > 0 ? nb : 1) * C * QB, sizeof(float));
    orc_gather_normalise(feat_a, C, HW, roi_a, n1, an);
    orc_gather_normalise(feat_q, C, HW, roi_q, n2, qn);
#pragma omp parallel for schedule(static)
    for (int j = 0; j < n2; ++j)
        for (int k = 0; k < C; ++k) qt[((size_t)(j / QB) * C + k) * QB + (j % QB)] = qn[(size_t)j * C + k];
    const int rows = anchor_rows ? n_rows : n1;
#pragma omp parallel for schedule(dynamic, 16)
    for (int r = 0; r < rows; ++r) {
        const int i = anchor_rows ? anchor_rows[r] : r;
        const float *a = an + (size_t)i * C;
        float best = INFINITY;
        int bj = 0;
        for (int jb = 0; jb < nb; ++jb) {
            const float *b = qt + (size_t)jb * C * QB;
            float dot[QB];
            for (int u = 0; u < QB; ++u) dot[u] = 0.0f;
            for (int k = 0; k < C; ++k) {
                const float ak = a[k];
                for (int u = 0; u < QB; ++u) dot[u] = fmaf(ak, b[(size_t)k * QB + u], dot[u]);
            }
            const int lim = (n2 - jb * QB) < QB ? (n2 - jb * QB) : QB;
            for (int u = 0; u < lim; ++u) {
                float dist = fmaf(-0.5f, dot[u], 0.5f);
                if (dist < best) { best = dist; bj = jb * QB + u; }
            }
        }
        min_dist[r] = best;
        argmin[r] = bj;
        valid[r] = (uint8_t)(best < thr);
    }
    free(an);
    free(qn);
    free(qt);
}

void orc_match_f32(const float *feat_a, const float *feat_q, int C, int HW, const int32_t *roi_a, int n1,
                   const int32_t *roi_q, int n2, float thr, float *min_dist, int32_t *argmin, uint8_t *valid)
{
    orc_match_f32_rows(feat_a, feat_q, C, HW, roi_a, n1, roi_q, n2, thr, NULL, 0, min_dist, argmin, valid);
}

/* the plain scalar loop nest (one chain at a time): kept to pin the interleaved version above, bit for bit */
void orc_match_f32_scalar(const float *feat_a, const float *feat_q, int C, int HW, const int32_t *roi_a, int n1,
                          const int32_t *roi_q, int n2, float thr, float *min_dist, int32_t *argmin, uint8_t *valid)
{
    float *an = (float *)malloc((size_t)(n1 > 0 ? n1 : 1) * C * sizeof(float));
    float *qn = (float *)malloc((size_t)(n2 > 0 ? n2 : 1) * C * sizeof(float));
    orc_gather_normalise(feat_a, C, HW, roi_a, n1, an);
    orc_gather_normalise(feat_q, C, HW, roi_q, n2, qn);
    for (int i = 0; i < n1; ++i) {
        const float *a = an + (size_t)i * C;
        float best = INFINITY;
        int bj = 0;
        for (int j = 0; j < n2; ++j) {
            const float *b = qn + (size_t)j * C;
            float dot = 0.0f;
            for (int k = 0; k < C; ++k) dot = fmaf(a[k], b[k], dot);
            float dist = fmaf(-0.5f, dot, 0.5f);
            if (dist < best) { best = dist; bj = j; }
        }
        min_dist[i] = best;
        argmin[i] = bj;
        valid[i] = (uint8_t)(best < thr);
    }
    free(an);
    free(qn);
}

/* (y,x) featmap coords -> original pixel coords: fp32 scale (target/source), bounds check on both
 * images, truncation; then pin-hole lift in millimetres and /1000 (pipeline.py:447-460).
 * corrs: [n,4] int64 (y1,x1,y2,x2).  Outputs are compacted over the valid rows; returns their count. */
int orc_scale_validate_lift(const int64_t *corrs, int n, int FH, int FW, const float *depth_a, int HA, int WA,
                            const float *depth_q, int HQ, int WQ, const double *cam_a, const double *cam_q,
                            uint8_t *row_valid, float *pcd_a, float *pcd_q)
{
    const float sya = (float)HA / (float)FH, sxa = (float)WA / (float)FW;
    const float syq = (float)HQ / (float)FH, sxq = (float)WQ / (float)FW;
    const float fxa = (float)cam_a[0], cxa = (float)cam_a[2], fya = (float)cam_a[4], cya = (float)cam_a[5];
    const float fxq = (float)cam_q[0], cxq = (float)cam_q[2], fyq = (float)cam_q[4], cyq = (float)cam_q[5];
    int m = 0;
    for (int i = 0; i < n; ++i) {
        float ya = (float)corrs[4 * i + 0] * sya, xa = (float)corrs[4 * i + 1] * sxa;
        float yq = (float)corrs[4 * i + 2] * syq, xq = (float)corrs[4 * i + 3] * sxq;
        int ok = ya >= 0.0f && ya < (float)HA && xa >= 0.0f && xa < (float)WA &&
                 yq >= 0.0f && yq < (float)HQ && xq >= 0.0f && xq < (float)WQ;
        row_valid[i] = (uint8_t)ok;
        if (!ok) continue;
        int iya = (int)ya, ixa = (int)xa, iyq = (int)yq, ixq = (int)xq;
        float za = depth_a[(size_t)iya * WA + ixa], zq = depth_q[(size_t)iyq * WQ + ixq];
        float X, Y;
        X = ((float)ixa - cxa) * za; X = X / fxa;
        Y = ((float)iya - cya) * za; Y = Y / fya;
        pcd_a[3 * m + 0] = X / 1000.0f; pcd_a[3 * m + 1] = Y / 1000.0f; pcd_a[3 * m + 2] = za / 1000.0f;
        X = ((float)ixq - cxq) * zq; X = X / fxq;
        Y = ((float)iyq - cyq) * zq; Y = Y / fyq;
        pcd_q[3 * m + 0] = X / 1000.0f; pcd_q[3 * m + 1] = Y / 1000.0f; pcd_q[3 * m + 2] = zq / 1000.0f;
        ++m;
    }
    return m;
}
