// Shared bits of the matcher translation units (match.hip, match16.hip).
#pragma once
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

}  // namespace oryon
