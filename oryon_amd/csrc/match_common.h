// Shared bits of the matcher translation units (match.hip, match16.hip).
#pragma once
#include <hip/hip_fp16.h>
#include "common.h"

namespace oryon {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// lexicographic (distance, index) minimum: smaller distance, then smaller index (first-index tie rule of torch.argmin)
__device__ __forceinline__ void lex_min(float &d, int &i, float od, int oi)
{
    const bool take = (od < d) || (od == d && oi < i);
    d = take ? od : d;
    i = take ? oi : i;
}

// Exact fp32 recomputation (K1) of the anchor panels flagged in panel_flag [B, cap_a/128]; only rows with row_flag set
// are written.  Used by the screened matcher when a candidate list overflows.
int match_f32_flagged(const float *a_hat, const float *q_hat, int B, int C, int cap_a, int cap_q, const int32_t *n_a,
                      const int32_t *n_q, float threshold, float *min_dist, int32_t *argmin, uint8_t *valid,
                      const int32_t *panel_flag, const uint8_t *row_flag, void *stream);

// K1x3 (match_x3.hip): fp32-grade scan of a compacted anchor list on the fp16 matrix pipe; see the file's header
size_t match_x3_scratch_bytes(int B, int cap_s, int S, int cap_q);
int match_x3_resolve(const float *a_c, const int32_t *n_c, int cap_s, const float *feat_q, int C_true, int HW, int layout,
                     const int32_t *roi_q, int roi_stride_q, const float *q_norm, const int32_t *n_q, int B, int cap_q, float threshold,
                     int round_f16, __half *qh, __half *ql, __half *ah, __half *al, void *scratch, float *md_c, int32_t *am_c, uint8_t *va_c,
                     int32_t **n_ovf_out, int32_t **ovf_idx_out, const int32_t *orig_idx, int orig_stride, const int32_t *sid_final, int cap_a,
                     const __half *q_hi_lo_pre, const float *q_lo_sq_max_pre, hipStream_t st);
void match_x3_scatter_ovf(int B, int cap_s, const int32_t *n_ovf, const int32_t *ovf_idx, const float *md_o, const int32_t *am_o,
                          const uint8_t *va_o, float *md_c, int32_t *am_c, uint8_t *va_c, hipStream_t st);

const char *screen_mx6_name(int C);
// screen_mx6.hip: K1s6 launch (C = 256 / 512); groups / T as sized for 256-anchor panels by the caller
void launch_screen_mx6(int C, int groups, int T, hipStream_t st, const uint8_t *a6, const uint8_t *q6, int B, int cap_a, int cap_q,
                       const int32_t *n_a, const int32_t *n_q, int S, float *ws_max, int32_t *ws_i1, float *ws_m2, int C_true,
                       int cascade = 0, const int32_t *gate = nullptr, int win = 0);
inline int mx6_panels_per_pair(int cap_a) { return (cap_a + 1023) / 1024; }     // panels of the 8-wave C_pad 256 screen (the cascade's gate is [B, that])
// second pass of the validity cascade: one 512-row panel of compacted anchor rows per pair, S query splits (C_pad 256)
void launch_screen_mx6_sampled(hipStream_t st, const uint8_t *a6_panel, const uint8_t *q6, int B, int cap_q, const int32_t *n_rows,
                               const int32_t *n_q, int S, float *ws_max, int32_t *ws_i1, float *ws_m2, int C_true);
constexpr int MX6_SAMPLED_PANEL = 512;          // rows of that panel (>= the sampled rows of a pair: corr_rows <= 512 on this route)
inline int mx6_sampled_splits(int B) { int s = (512 + B - 1) / B; return s < 1 ? 1 : s > 16 ? 16 : s; }      // ~512 four-wave workgroups

}  // namespace oryon
