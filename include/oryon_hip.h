/* liboryon_hip.so — C ABI of the MI355X-native Oryon hot path (gfx950 only).
 *
 * The reference (jcorsetti/oryon) is pure Python/PyTorch and has no FFI; its boundary for this path is a
 * set of Python callables.  Each entry point below replaces the PyTorch expression(s) cited next to it
 * (file:line into the reference tree) and is what a ctypes binding on the reference side would load
 * (INTEGRATION.md shows that binding).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host; the caller owns all buffers;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all calls are asynchronous on
 *     it, never synchronise, never allocate (except the handle constructors: oryon_pointdsc_create/_finalize,
 *     oryon_engine_create, oryon_decoder_create - the last one also waits for its packing launches once);
 *   - return value: ORYON_OK or a negative ORYON_ERR_*; nothing throws; oryon_last_error() gives text;
 *   - per-pair outcomes (no mask / no correspondences) are DATA, reported in `status` arrays with the
 *     reference's own failure semantics (pipeline.py:335-350), not error codes;
 *   - thread-safe for distinct streams.  Global state of the library, all of it per device or per thread: the last-error string
 *     (thread local); the step engine's stream pool (eight HIP streams per device, created under a mutex on first use or by
 *     oryon_engine_warm_streams, shared by every engine of the process and never destroyed); the fp16x3 range-flag word per device
 *     (oryon_x3_range_flag); the profile-event pair armed by oryon_profile_events (thread local).
 */
#ifndef ORYON_HIP_H
#define ORYON_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORYON_OK 0
#define ORYON_ERR_INVALID_ARG (-1)
#define ORYON_ERR_HIP (-2)
#define ORYON_ERR_WORKSPACE (-3)
#define ORYON_ERR_NO_DEVICE (-4)
#define ORYON_ERR_STATE (-5)

/* per-pair status values (pipeline.py:316-350: invalid detection / matcher returned None / pose ok) */
#define ORYON_PAIR_OK 0
#define ORYON_PAIR_NO_MASK 1
#define ORYON_PAIR_NO_CORR 2

/* matcher tile geometry the padded capacities must respect */
#define ORYON_MATCH_TILE 128

const char *oryon_version(void);
const char *oryon_last_error(void);
/* Measurement hook used by bench.py: the two hipEvent_t handles (created by the caller) are recorded on the launch
 * stream immediately before / after the DOMINANT kernel launch of the next oryon_match_f32 / oryon_match_screened call
 * made by this thread (match_f32_regb_kernel, resp. the match_f16_screen_kernel screening pass), then forgotten. */
int oryon_profile_events(void *start_event, void *stop_event);
/* Name (template arguments included) of the kernel the events above bracketed most recently; "" before the first armed call. */
const char *oryon_dominant_kernel(void);
/* ORYON_OK iff device `device` exists and is gfx950. */
int oryon_device_check(int device);

/* K-1 batched image pre-processing in front of the network: what the reference does per sample on dataloader workers
 *     (utils/data/common.py:49 `rgb.transpose(2,0,1)/255.`, utils/augmentations.py:137-139 resize to dataset.img_size,
 *     datasets.py:204 `.to(float32)`).  Resampling = torch upsample_bilinear2d, align_corners=False.
 * rgb_hwc [n,HI,WI,3] uint8 -> out [n,3,HO,WO] fp32 (fp64 arithmetic, one rounding, like the reference's float64 tensor).
 * in [n,HI,WI] fp32 -> out [n,HO,WO] fp32 (fp32 arithmetic; round_output != 0 rounds half-to-even as torchvision does
 * for integer images such as the depth map). */
int oryon_rgb_resize_bilinear(const uint8_t *rgb_hwc, int n, int HI, int WI, int HO, int WO, float *out, void *stream);
int oryon_resize_bilinear_f32(const float *in, int n, int HI, int WI, int HO, int WO, int round_output, float *out, void *stream);

/* K1' the reference's fp16 matcher branch (corrs_device='cuda', utils/pcd.py:195-197) casts the descriptors to float16 first:
 *     out[i] = float(half(in[i])) (round to nearest even; may alias).  The exact matcher then runs on the rounded values. */
int oryon_round_to_f16_f32(const float *in, float *out, int64_t n, void *stream);

/* B1  QuickGELU of the CLIP residual blocks, y = x * sigmoid(1.702 x)  (third-party clip model.py, loaded at
 *     models/vlm.py:19), fused into one pass for bf16 activations: x, y [n] bf16 (16-byte aligned, may alias).
 *     Arithmetic in fp32, one rounding.  The fp32 backbone path keeps torch's own ops. */
int oryon_quick_gelu_bf16(const void *x, void *y, int64_t n, void *stream);

/* B2  residual add + LayerNorm of the CLIP residual stream (clip model.py ResidualAttentionBlock.forward:
 *     x = x + attention(ln_1(x)); x = x + mlp(ln_2(x)); torch.nn.LayerNorm semantics, biased variance), bf16 inference only:
 *         s = bf16(x + delta)   -> x_out   (delta == NULL: s = x, x_out not written)
 *         h = bf16((s - mean(s)) * rsqrt(var(s) + eps) * gamma + beta) -> h_out          statistics in fp32 over the rounded s
 *     x, delta, x_out, h_out [rows, D] bf16, gamma / beta [D] bf16; D % 8 == 0, D <= 4096; 16-byte aligned; x_out may alias x
 *     or delta, h_out must not alias an input. */
int oryon_add_layernorm_bf16(const void *x, const void *delta, const void *gamma, const void *beta, int64_t rows, int D, float eps,
                             void *x_out, void *h_out, void *stream);
/*     fp32 twin for the fp32 / fp16x3 inference path (same formula without the bf16 roundings; D % 4 == 0, D <= 2048). */
int oryon_add_layernorm_f32(const float *x, const float *delta, const float *gamma, const float *beta, int64_t rows, int D, float eps,
                            float *x_out, float *h_out, void *stream);

/* B3  shifted-window attention of the Swin guidance backbone (torchvision swin_transformer.shifted_window_attention, the
 *     swin_b feature extractor of net.py:60-75), window 7, head dim 32, bf16 inference: everything between the q|k|v Linear and
 *     the output projection in one kernel (pad, cyclic shift, window partition, q scale, QK^T + relative position bias + shift
 *     mask, softmax, PV, window merge, reverse shift, crop).
 *     qkv     [B, H, W, 3C] bf16: the q|k|v Linear applied to the un-windowed tokens (it commutes with the partition)
 *     pad_qkv [3C] bf16: q|k|v of a zero-padded token = the Linear's bias (zeros if it has none)
 *     bias_t  [heads, 49 (key), 49 (query)] fp32: relative_position_bias_table[relative_position_index], key-major
 *     out     [B, H, W, C] bf16;  C == heads * 32, heads <= 8, 0 <= shift < 7 (torchvision passes 0 or 3) */
int oryon_swin_window_attention_bf16(const void *qkv, const void *pad_qkv, const float *bias_t, int B, int H, int W, int C, int heads,
                                     int shift, void *out, void *stream);
/* the same operation on fp32 tensors (fp32 evaluation of the guidance tower): both products on the fp16 matrix pipe with error-compensated
 * operands (fp32-grade, ~1e-6 relative; round 4 - the fp32-VALU form of the bf16 kernel took twice as long), fp32 bias / mask / softmax */
int oryon_swin_window_attention_f32(const float *qkv, const float *pad_qkv, const float *bias_t, int B, int H, int W, int C, int heads,
                                    int shift, float *out, void *stream);

/* ---------------------------------------------------------------------------------------------------
 * K0  mask -> ROI.   Replaces torch.nonzero(mask == 1) (utils/pcd.py:184-185) and the validity test
 *     count_nonzero(mask == 1) > 0 (pipeline.py:391-393).
 * mask  [n_maps, HW] int32        roi [n_maps, HW] int32 (linear pixel index y*W+x, row-major order)
 * count [n_maps] int32
 */
int oryon_roi_compact(const int32_t *mask, int n_maps, int HW, int32_t *roi, int32_t *count, void *stream);

/* K0  sigmoid(logit) > threshold -> {0,1} int32 mask.  Replaces losses.py:58-59 (test.mask == predicted). */
int oryon_mask_from_logits(const float *logits, int64_t n, float threshold, int32_t *mask, void *stream);

/* K0  legacy-nearest resize of an integer mask to the feature-map grid (src = floor(dst * in/out) computed
 *     with the fp32 scale, as torch F.interpolate(mode='nearest') does).  Replaces pipeline.py:408-411. */
int oryon_mask_resize_nearest(const uint8_t *mask_in, int n_maps, int HI, int WI, int HO, int WO,
                              int32_t *mask_out, void *stream);

/* K0  device-side subsample WITHOUT replacement to at most max_keep entries per map, keeping row-major
 *     order.  Counter-based RNG keyed by (seed, map_key[m], element) so the result is independent of how
 *     maps are sharded over GPUs.  Stands in for torch_sample_select (utils/misc.py:242-254, called at
 *     utils/pcd.py:187-190) in the batched path; the drop-in Python facade keeps the host torch RNG.
 * map_key [n_maps] int64 (e.g. global pair index), may be NULL (then the map index is used). */
int oryon_roi_subsample(int32_t *roi, int32_t *count, int n_maps, int roi_stride, int max_keep, uint64_t seed,
                        const int64_t *map_key, void *stream);

/* K0  gather the ROI descriptors of channel-planar maps and L2-normalise them.
 *     Replaces feats[:, roi[:,0], roi[:,1]].T (utils/pcd.py:192-193) and the x / max(|x|, 1e-8) half of
 *     cosine_similarity (utils/pcd.py:28).
 * feat [n_maps, C, HW] fp32; roi [n_maps, roi_stride]; out [n_maps, rows_cap, C_pad] fp32, row-major; rows
 * >= count[m] up to the next multiple of 256 and columns C..C_pad-1 are zero-filled (zero columns do not
 * change the fmaf chain).  C_pad is a multiple of 32 (>= C), rows_cap a multiple of 256.
 * Layout note: inside every group of 8 columns the fp32 rows are stored k-permuted (position 8g+4h+j holds
 * k = 8g+2j+h, the order the MFMA operands consume 16-byte chunks in); consumers are K1 and K1s only.
 * out_f16 (may be NULL): [n_maps, rows_cap, C_pad] IEEE half copy of the same unit rows (round-to-nearest), natural k
 * order - the operand of the screening pass of oryon_match_screened. */
int oryon_gather_normalise_f32(const float *feat, int n_maps, int C, int HW, const int32_t *roi, int roi_stride,
                               const int32_t *count, int rows_cap, int C_pad, float *out, void *out_f16, void *stream);

/* K1  cosine nearest neighbour: for every anchor row the query row minimising 0.5*(1 - a^.q^).
 *     Replaces pdist(...,'inv_norm_cosine') + amin + argmin + (min_dist < th) (utils/pcd.py:202-205)
 *     without materialising [N1,N2] (the reference materialises [N1,N2,C]).
 * a_hat [B, cap_a, C], q_hat [B, cap_q, C]  (outputs of oryon_gather_normalise_f32; caps multiples of
 * ORYON_MATCH_TILE), n_a/n_q [B] int32 on device.
 * min_dist [B,cap_a] fp32, argmin [B,cap_a] int32 (first index on ties), valid [B,cap_a] uint8.
 * Arithmetic: exact fp32 — dot = k-ordered fmaf chain (v_mfma_f32_32x32x2_f32), dist = fma(-0.5,dot,0.5).
 * workspace: oryon_match_workspace_bytes(B, cap_a) bytes (used only when the launch splits the query range). */
size_t oryon_match_workspace_bytes(int B, int cap_a);
int oryon_match_f32(const float *a_hat, const float *q_hat, int B, int C, int cap_a, int cap_q,
                    const int32_t *n_a, const int32_t *n_q, float threshold, float *min_dist, int32_t *argmin,
                    uint8_t *valid, void *workspace, size_t workspace_bytes, void *stream);

/* K1s same contract as oryon_match_f32 for every anchor row that can reach the threshold, computed as an fp16-MFMA
 *     screening pass (v_mfma_f32_32x32x16_f16 on the IEEE-half copies written by oryon_gather_normalise_f32) followed by
 *     an exact fp32 re-scoring of the surviving candidates with K1's canonical fmaf chain.  A proven error bound on the
 *     screening scores (csrc/match16.hip) guarantees that every exact minimiser is a candidate, so on rows with
 *     valid == 1 `min_dist` / `argmin` are bit-identical to oryon_match_f32, and `valid` is identical on every row.
 *     Rows that provably cannot reach the threshold get valid = 0, argmin = 0 and the screening estimate as min_dist.
 * C (padded) must be 128, 256 or 512, cap_a a multiple of 256; a_f16 / q_f16 are the half copies [B, cap, C]. */
size_t oryon_match_screened_workspace_bytes(int B, int C, int cap_a);
int oryon_match_screened(const float *a_hat, const float *q_hat, const void *a_f16, const void *q_f16, int B, int C,
                         int cap_a, int cap_q, const int32_t *n_a, const int32_t *n_q, float threshold, float *min_dist,
                         int32_t *argmin, uint8_t *valid, void *workspace, size_t workspace_bytes, void *stream);

/* K1s8 the same contract again with an INT8 pre-screen (v_mfma_i32_32x32x32_i8: twice the fp16 matrix rate) in front of K1s:
 *     oryon_gather_normalise_q8 additionally writes int8 rows q = rint(x^ * 2^E) with one exponent per 16-row slice of the
 *     screening kernel's accumulator layout (slice_scale = 2^-E, eps_max = largest 2^-(E+1) of the map).  Anchors whose int8
 *     maximum is decided within the proven int8 bound go straight to fp16 slice re-scoring + exact fp32 re-scoring; every other
 *     anchor runs through the complete K1s pipeline on a compacted set.  Outputs are therefore identical to oryon_match_screened
 *     (and to oryon_match_f32 on valid rows) for every input; only the run time depends on the data.  C_pad 256 or 512.
 *     The fp16 operands of that second stage are derived from the fp32 rows inside the call, and only for pairs that need
 *     them, so out_f16 of oryon_gather_normalise_q8 may be NULL on this path.
 *     C_true = number of real channels (<= C_pad), used in the bound. */
int oryon_gather_normalise_q8(const float *feat, int n_maps, int C, int HW, const int32_t *roi, int roi_stride, const int32_t *count,
                              int rows_cap, int C_pad, float *out, void *out_f16, int8_t *out_i8, float *slice_scale, float *eps_max,
                              void *stream);
size_t oryon_match_screened8_workspace_bytes(int B, int C, int cap_a, int cap_q);
int oryon_match_screened8(const float *a_hat, const float *q_hat, const int8_t *a_i8, const int8_t *q_i8, const float *a_scale, const float *q_scale, const float *q_eps_max, int B, int C_true,
                          int C, int cap_a, int cap_q, const int32_t *n_a, const int32_t *n_q, float threshold, float *min_dist,
                          int32_t *argmin, uint8_t *valid, int32_t *n_undecided /* [B] or NULL: anchors handed to the fp16 stage */,
                          void *workspace, size_t workspace_bytes, void *stream);

/* K0v3 + K1s8 without an fp32 copy of the query rows (round 2; csrc/gather8.hip).
 *     Replaces the same reference lines as oryon_gather_normalise_q8 / oryon_match_screened8 (utils/pcd.py:192-193, :28-29, :202-205).
 *     oryon_gather_q8 reads the raw descriptors of the ROI rows once - from a channel-planar [n_maps,C,H,W] map (ORYON_LAYOUT_NCHW, what
 *     net.py:162-167 returns) or from a channels_last [n_maps,H,W,C] map (ORYON_LAYOUT_NHWC, zero-copy view of a torch channels_last
 *     tensor) - and writes the int8 rows, the per-slice scales, eps_max, the canonical row norm d (row_norm [n_maps, rows_cap], may be
 *     NULL) and, only when out_f32 != NULL, the canonical fp32 unit rows (k-permuted, as oryon_gather_normalise_f32 writes them).
 *     oryon_match_screened8_raw is oryon_match_screened8 for such operands: anchors come as materialised fp32 + int8 rows, queries as
 *     int8 rows + row norms + the raw map itself; the exact re-scoring pass recovers a candidate's canonical unit values x_k / d from
 *     the raw map, so its outputs are bit for bit those of oryon_match_screened8.  Pairs that need the fp32 query rows after all
 *     (anchors the int8 stage could not decide, overflowed candidate lists) get them materialised inside the call, gated on the device.
 *     Rows [n, round_up(n,256)) of every output are zero rows; rows beyond are not written.
 *     round_f16 != 0 selects the reference's half-descriptor branch (utils/pcd.py:195-197, `corrs_device='cuda'`: feats.half() before
 *     pdist): every raw descriptor value is rounded to the nearest float16 on the way in, in all three calls alike; the arithmetic on
 *     the rounded values is unchanged (fp32), which is what BASELINE configs[4] ("fp16 descriptors") runs. */
#define ORYON_LAYOUT_NCHW 0
#define ORYON_LAYOUT_NHWC 1
int oryon_gather_q8(const float *feat, int n_maps, int C, int HW, int layout, const int32_t *roi, int roi_stride, const int32_t *count,
                    int rows_cap, int C_pad, int8_t *out_i8, float *slice_scale, float *eps_max, float *row_norm, float *out_f32,
                    int round_f16, void *stream);
size_t oryon_match_screened8_raw_workspace_bytes(int B, int C, int cap_a, int cap_q);
int oryon_match_screened8_raw(const float *a_hat, const int8_t *a_i8, const float *a_scale, const float *feat_q, int C_true, int HW,
                              int layout, const int32_t *roi_q, int roi_stride, const float *q_norm, const int8_t *q_i8,
                              const float *q_scale, const float *q_eps_max, int B, int C, int cap_a, int cap_q, const int32_t *n_a,
                              const int32_t *n_q, float threshold, float *min_dist, int32_t *argmin, uint8_t *valid,
                              int32_t *n_undecided, int round_f16, void *workspace, size_t workspace_bytes, void *stream);

/* K1s8 + K1b fused and LAZY (round 2): from the K0v3 operands straight to the sampled correspondences of every pair
 * (utils/pcd.py:202-214).  The int8 bound settles the validity flag of almost every anchor without its argmin; candidate generation
 * and exact re-scoring then run for the <= max_corrs sampled anchors only.  Anchors whose validity the bound cannot settle are resolved
 * exactly before the sampling.  AMBIGUOUS anchors (runner-up slice within the int8 margin of the best one) are resolved by the exact
 * fp32 scan K1 on a compacted list of just those rows - before the sampling if their validity is open, after it (sampled rows only)
 * if the bound already proves them valid - against fp32 query rows materialised for that pair inside the call.  force_eager != 0
 * sends every pair through the complete tail of oryon_match_screened8_raw instead (all of min_dist / argmin exact).  corrs / n_valid / n_sel / status are exactly what oryon_select_corrs returns on the outputs of
 * oryon_match_screened8_raw.  valid [B,cap_a] is exact on every row; argmin is exact on sampled rows and on every row of an eager
 * pair; min_dist is exact on every row of an eager pair and on the sampled / resolved rows that needed an fp32 comparison (more than one
 * candidate inside the int8 margin, or validity open); elsewhere both hold the screening estimate / 0. */
size_t oryon_match_corrs_i8_workspace_bytes(int B, int C, int cap_a, int cap_q, int corr_rows);
int oryon_match_corrs_i8(const float *a_hat, const int8_t *a_i8, const float *a_scale, const float *feat_q, int C_true, int HW, int layout,
                         const int32_t *roi_a, int roi_stride_a, const int32_t *roi_q, int roi_stride_q, const float *q_norm,
                         const int8_t *q_i8, const float *q_scale, const float *q_eps_max, int B, int C, int cap_a, int cap_q,
                         const int32_t *n_a, const int32_t *n_q, float threshold, int W, int max_corrs, int corr_rows, uint64_t seed,
                         const int64_t *pair_key, int force_eager, float *min_dist, int32_t *argmin, uint8_t *valid, int32_t *corrs,
                         int32_t *n_valid, int32_t *n_sel, int32_t *status, int32_t *n_undecided, int round_f16, void *workspace,
                         size_t workspace_bytes, void *stream);

/* K0 + K1s6 (round 3): the same lazy matcher with an MX-fp6 screen in place of the int8 one.  Replaces the same reference lines
 * (utils/pcd.py:192-193, :28-29, :202-214); the results are those of oryon_match_corrs_i8 bit for bit (every decision the screen takes is
 * backed by a proven bound, everything else is resolved exactly), only the dominant kernel changes: v_mfma_scale_f32_32x32x64_f8f6f4
 * multiplies 64 channels per instruction at the int8 instruction's rate.
 *     oryon_gather_mx6 writes, per ROI row and 32-channel block, one 32-byte slot: bytes 0-23 = 32 fp6 (e2m3) codes of x^ / 2^e (element t
 *     at bits [6t, 6t+6)), byte 24 = e + 127 (E8M0), bytes 25-31 = 0 - rows of C_pad bytes, like the int8 rows; the block exponent puts the
 *     block's largest magnitude into (3.75, 7.5].  err_max [n_maps] receives the largest MEASURED quantisation error |x^ - dequant|_2 over
 *     the map's live rows; row_norm / out_f32 as oryon_gather_q8.
 *     oryon_match_corrs_mx6 = oryon_match_corrs_i8 on such operands (lazy route only): |s6 - a^.q^| <= |ea| + |eq| + |ea||eq| + 1.2e-4 with
 *     ea / eq the pair's a_err_max / q_err_max decides validity and unambiguity; the winning slice's 16 rows are re-scored from the mx6
 *     slots, candidates inside the margin of the slice maximum go to the exact fp32 chain on the raw map.  Workspace:
 *     oryon_match_corrs_i8_workspace_bytes. */
int oryon_gather_mx6(const float *feat, int n_maps, int C, int HW, int layout, const int32_t *roi, int roi_stride, const int32_t *count,
                     int rows_cap, int C_pad, uint8_t *out_mx6, float *err_max, float *row_norm, float *out_f32, int round_f16, void *stream);
int oryon_match_corrs_mx6(const float *a_hat, const uint8_t *a_mx6, const float *a_err_max, const float *feat_q, int C_true, int HW, int layout,
                          const int32_t *roi_a, int roi_stride_a, const int32_t *roi_q, int roi_stride_q, const float *q_norm,
                          const uint8_t *q_mx6, const float *q_err_max, int B, int C, int cap_a, int cap_q, const int32_t *n_a,
                          const int32_t *n_q, float threshold, int W, int max_corrs, int corr_rows, uint64_t seed, const int64_t *pair_key,
                          float *min_dist, int32_t *argmin, uint8_t *valid, int32_t *corrs, int32_t *n_valid, int32_t *n_sel, int32_t *status,
                          int32_t *n_undecided, int round_f16, void *workspace, size_t workspace_bytes, void *stream);

/* K0 + K1x3 operands in ONE pass (round 4).  On descriptor fields where the screen cannot separate a sampled anchor's near-ties (smooth decoder
 * outputs) the matcher needs the query rows once more as error-compensated half rows (hi = half(u), lo = half(u - hi) of the canonical unit
 * row u) for its fp16x3 second level - a second read of the maps inside oryon_match_corrs_mx6.  oryon_gather_mx6_x3 writes them in the same
 * pass as the mx6 slots (replaces the same reference lines as oryon_gather_mx6, utils/pcd.py:192-193, :28-29): hi_lo_f16 = [2][n_maps,
 * rows_cap, 256] halves (all hi rows, then all lo rows), lo_sq_max [n_maps] = largest |u - hi|^2 of the map's rows.  C_pad = 256 only.
 * oryon_match_corrs_mx6_x3 = oryon_match_corrs_mx6 that takes those rows instead of making them: same corrs / n_valid / n_sel / status /
 * valid, bit for bit.  Since round 6 its screen runs as a VALIDITY CASCADE (this is the route of maps on which every anchor is valid and
 * ambiguous, where the full screen bought nothing but validity): (1) a windowed launch - two query tiles per (1024-anchor panel, query
 * split), placed where the panel sits in its own map; a panel all of whose anchors are thereby valid for sure (one witness above
 * 1 - 2 thr + the proven bound suffices: utils/pcd.py:204 is "min over queries < threshold") is settled, its runner-ups read "open";
 * (2) the complete scan for the other panels only (device-gated); (3) after the sampling, the complete screen for the <= max_corrs
 * sampled rows whose argmin is open (one 512-row panel per pair), which gives them their true winning slice - rows that turn out
 * unambiguous are resolved from it, the others go to the fp16x3 second level with it as their seed.  corr_rows <= 512.  min_dist of rows
 * that were neither sampled nor resolved holds the estimate of the scan that settled them (here possibly a partial one); n_undecided =
 * the first pass's count scaled by the share of sampled rows that stayed ambiguous after (3) (the engine's route feedback). */
int oryon_gather_mx6_x3(const float *feat, int n_maps, int C, int HW, int layout, const int32_t *roi, int roi_stride, const int32_t *count,
                        int rows_cap, int C_pad, uint8_t *out_mx6, float *err_max, float *row_norm, void *hi_lo_f16, float *lo_sq_max,
                        int round_f16, void *stream);
int oryon_match_corrs_mx6_x3(const float *a_hat, const uint8_t *a_mx6, const float *a_err_max, const float *feat_q, int C_true, int HW, int layout,
                             const int32_t *roi_a, int roi_stride_a, const int32_t *roi_q, int roi_stride_q, const float *q_norm,
                             const uint8_t *q_mx6, const float *q_err_max, const void *q_hi_lo_f16, const float *q_lo_sq_max, int B, int C,
                             int cap_a, int cap_q, const int32_t *n_a, const int32_t *n_q, float threshold, int W, int max_corrs, int corr_rows,
                             uint64_t seed, const int64_t *pair_key, float *min_dist, int32_t *argmin, uint8_t *valid, int32_t *corrs,
                             int32_t *n_valid, int32_t *n_sel, int32_t *status, int32_t *n_undecided, int round_f16, void *workspace,
                             size_t workspace_bytes, void *stream);

/* "Sample first" (optional engine schedule, off by default).  Only max_corrs correspondences per pair leave the matcher
 * (utils/pcd.py:205-214), drawn uniformly from the valid anchor rows - so a uniformly random first-stage subset of the anchors that
 * already holds >= max_corrs valid rows yields an identically distributed sample.  The engine runs the matcher on such a subset;
 * oryon_sample_first_gate computes, on the device, which pairs must be redone on all of their anchors (n_a2 = n_a where the first stage
 * found fewer than max_corrs valid rows although anchors were left out, else 0 - every second-stage launch then sees no anchors for the
 * other pairs) and oryon_sample_first_merge copies the second stage's rows / counts / status over the first stage's for those pairs. */
int oryon_sample_first_gate(const int32_t *n_valid1, const int32_t *n_a1, const int32_t *n_a, int B, int max_corrs, int32_t *n_a2,
                            void *stream);
int oryon_sample_first_merge(const int32_t *n_a2, const int32_t *corrs2, const int32_t *n_valid2, const int32_t *n_sel2,
                             const int32_t *status2, int B, int corr_rows, int32_t *corrs1, int32_t *n_valid1, int32_t *n_sel1,
                             int32_t *status1, void *stream);

/* K1b turn matcher outputs into sampled correspondences (device RNG; batched path only).
 *     Replaces utils/pcd.py:205-214: keep rows with valid, need more than one, sample exactly max_corrs
 *     (with replacement iff fewer are available).
 * corrs [B, corr_rows, 4] int32 (y1,x1,y2,x2 in feature-map coordinates; corr_rows >= max_corrs is the row
 * stride per pair, rows >= max_corrs are left untouched), n_valid [B] (rows under the threshold), n_sel [B]
 * (max_corrs or 0; may be NULL), status [B] (ORYON_PAIR_OK / ORYON_PAIR_NO_MASK when n_a or n_q is 0 /
 * ORYON_PAIR_NO_CORR when <= 1 valid row).  scratch [B, cap_a] int32 (ordered list of valid anchor rows). */
int oryon_select_corrs(const int32_t *roi_a, const int32_t *roi_q, int roi_stride_a, int roi_stride_q,
                       const int32_t *n_a, const int32_t *n_q, const int32_t *argmin, const uint8_t *valid, int cap_a,
                       int B, int W, int max_corrs, int corr_rows, uint64_t seed, const int64_t *pair_key,
                       int32_t *scratch, int32_t *corrs, int32_t *n_valid, int32_t *n_sel, int32_t *status, void *stream);

/* K2  scale (y,x) to the original image, keep rows inside both images, truncate, gather depth, pin-hole
 *     lift, /1000.  Replaces pipeline.py:447-460 + utils/coordinates.py:5-48 + utils/pcd.py:44-74.
 * corrs [B, n_cap, 4] int32; n_corr [B] or NULL (then n_cap rows each); depth_* [B,H*,W*] fp32 millimetres;
 * cam_* [B,9] fp32 (row-major K, already rounded from the reference's fp64); status [B] may be NULL
 * (pairs whose status != ORYON_PAIR_OK are skipped and get n_out = 0).
 * pcd_a/pcd_q [B, n_cap, 3] fp32 metres, compacted over valid rows in order, rows >= n_out zeroed; n_out [B]. */
int oryon_lift_pairs(const int32_t *corrs, const int32_t *n_corr, int B, int n_cap, int FH, int FW,
                     const float *depth_a, int HA, int WA, const float *depth_q, int HQ, int WQ,
                     const float *cam_a, const float *cam_q, const int32_t *status, float *pcd_a, float *pcd_q,
                     int32_t *n_out, void *stream);

/* K2' lift_pcd itself (utils/pcd.py:35-81 with xy_idxs): selected pixels of ONE depth map [H,W] fp32 ->
 *     [n,3] fp32 in the depth's unit (millimetres), X = ((x - cx) * z) / fx etc., no fma contraction.
 *     x_idx / y_idx [n] int32 must be inside the image (the reference would raise an IndexError). */
int oryon_lift_points(const float *depth, int H, int W, const float *cam9, const int32_t *x_idx, const int32_t *y_idx,
                      int n, float *out, void *stream);

/* K8  batched weighted Kabsch with an in-kernel 3x3 SVD (one-sided Jacobi, fp64 internal).
 *     Replaces rigid_transform_3d (models/pointdsc/common.py:7-45) incl. its H.cpu() -> LAPACK round trip.
 * A,B [nb, m, 3] fp32; w [nb, m] fp32 or NULL (all ones); negative weights count as 0 (common.py:20).
 * T [nb, 16] fp32 row-major 4x4. */
int oryon_kabsch_batched(const float *A, const float *B, const float *w, int nb, int m, float *T, void *stream);

/* ---------------------------------------------------------------------------------------------------
 * K3-K10  PointDSC registration (models/pointdsc/PointDSC.py:128-197, inference branch, wrapped as
 *         utils/pointdsc/init.py:10-29 does: corr_pos = cat(src,tgt) - mean).
 */
typedef struct oryon_pointdsc oryon_pointdsc_t;

typedef struct {
    int in_dim;             /* 6 */
    int num_layers;         /* release config: 12 */
    int num_channels;       /* 128 (32, 64 or 128 supported) */
    int num_iterations;     /* power-iteration cap, 10 */
    float ratio;            /* seeds = int(n * ratio), 0.1 */
    float inlier_threshold; /* 0.10 */
    float sigma_d;          /* 0.10 (sigma_spat) */
    int k;                  /* 40 */
    float nms_radius;       /* 0.10 */
} oryon_pointdsc_config_t;

int oryon_pointdsc_create(oryon_pointdsc_t **handle, const oryon_pointdsc_config_t *cfg);
void oryon_pointdsc_destroy(oryon_pointdsc_t *handle);
/* Load one tensor by its reference state-dict name (models/pointdsc/PointDSC.py:9-25,49-63,97-113), fp32
 * host data.  Unknown names -> ORYON_ERR_INVALID_ARG; num_batches_tracked is accepted and ignored. */
int oryon_pointdsc_load_param(oryon_pointdsc_t *handle, const char *name, const float *data_host, int64_t numel);
/* Fold eval-mode BatchNorm into the preceding conv, upload to the current device. */
int oryon_pointdsc_finalize(oryon_pointdsc_t *handle, void *stream);
size_t oryon_pointdsc_workspace_bytes(const oryon_pointdsc_t *handle, int B, int n_cap);
/* src,tgt [B, n_cap, 3] fp32 metres; n [B] int32 (rows actually used, <= n_cap); status_in [B] or NULL.
 * T [B,16] fp32 (identity for pairs that fail, as pipeline.py:341/350), labels [B,n_cap] uint8 or NULL,
 * status_out [B] int32 or NULL. */
int oryon_pointdsc_register(oryon_pointdsc_t *handle, const float *src, const float *tgt, const int32_t *n, int B,
                            int n_cap, const int32_t *status_in, void *workspace, size_t workspace_bytes, float *T,
                            uint8_t *labels, int32_t *status_out, void *stream);
/* Stage-level views for parity tests (same kernels as oryon_pointdsc_register):
 *   encode : corr features [B,n_cap,C] + confidence [B,n_cap]
 *   seeds  : NMS seeds [B, S_cap] int32 + n_seeds [B]
 *   hypotheses : given seeds, per-seed transforms [B,S_cap,16], fitness [B,S_cap], best [B]
 *   refine : post_refinement of given transforms */
int oryon_pointdsc_encode(oryon_pointdsc_t *handle, const float *src, const float *tgt, const int32_t *n, int B,
                          int n_cap, void *workspace, size_t workspace_bytes, float *feat, float *confidence,
                          void *stream);
int oryon_pointdsc_seeds(oryon_pointdsc_t *handle, const float *src, const float *confidence, const int32_t *n, int B,
                         int n_cap, int S_cap, int32_t *seeds, int32_t *n_seeds, void *stream);
int oryon_pointdsc_hypotheses(oryon_pointdsc_t *handle, const float *src, const float *tgt, const float *feat,
                              const int32_t *n, const int32_t *seeds, const int32_t *n_seeds, int B, int n_cap,
                              int S_cap, void *workspace, size_t workspace_bytes, float *seed_T, float *fitness,
                              int32_t *best, void *stream);
int oryon_pointdsc_refine(oryon_pointdsc_t *handle, const float *src, const float *tgt, const int32_t *n, int B,
                          int n_cap, const float *T_in, float *T_out, uint8_t *labels, void *stream);

/* ---------------------------------------------------------------------------------------------------
 * The whole batched step as ONE call (round 3): what the per-sample loop of FPM_Pipeline.test_step does for every pair of a batch
 * (pipeline.py:313-355: is_detection_valid -> get_featmap_corrs [utils/pcd.py:177-216] -> get_pose [pipeline.py:429-472:
 * scale / validate / lift, get_pointdsc_pose]), for B pairs, enqueued from C++ on streams and events the engine owns, over a
 * persistent arena the caller hands over once (the streams come from a per-device pool that lives as long as the process: every engine of
 * a process runs on the same four streams, so a re-created engine keeps the hardware queues of the first).  oryon_engine_submit issues,
 * without allocating or synchronising,
 *     gather stream : oryon_roi_compact x2, oryon_roi_subsample, oryon_gather_q8 x2                       (K0)
 *     match stream  : oryon_match_corrs_i8, oryon_lift_pairs                                               (K1s8 + K1b, K2)
 *     reg stream    : oryon_pointdsc_register                                                              (K3-K10)
 * with exactly the arguments oryon_amd/engine.py passes to those entry points, so results are identical bit for bit.  The K0
 * outputs alternate between two buffer sets, everything else a step produces between n_slots result slots, registrations between
 * two streams: K0 of step k+1 and the registrations of steps k-1, k-2 overlap the matching of step k.
 * Route: the screened lazy matcher (C <= 512, 0 < dist_th <= 0.5; descriptors narrower than 256 channels - the reference's own
 *          C = 32 @ 192^2, configs/config.yaml:34-35 - are zero-padded to the 256-channel operand rows by K0, which changes no result);
 *          wider descriptors / other thresholds stay with the per-call entry points.
 *
 * overlap: 0 = everything on the caller's stream (no engine streams), 1 = match on an engine stream, registration on one stream
 *          per slot, 2 = additionally K0 on its own stream (n_slots >= 2 for overlap >= 1).
 * Per-pair results live in the arena: oryon_engine_buffer gives offset / size of a slot's named buffer ("pose" [B,16] fp32,
 *          "status_out", "n_valid_out", "n_lift_out" [B] i32, "n_valid", "n_lift", "n_a", "n_q", "n_und" [B] i32, "corrs" [B,n_cap,4] i32, "pcd_a", "pcd_q"
 *          [B,n_cap,3] fp32, "roi_a", "roi_q" [B,FH*FW] i32, "min_dist", "argmin", "valid" [B,cap_a], ...); a slot's buffers are valid
 *          from oryon_engine_wait(slot) until the n_slots-th next submit.
 *          Slot lifetime, precisely: that submit orders the overwrite of "pose" / "status_out" / "n_valid_out" / "n_lift_out" (the
 *          protected block: written on the registration stream; the last two are the step's "n_valid" / "n_lift" copied there) after
 *          everything queued on caller_stream before it, whatever inputs_resident says.  K0 and the matcher overwrite the slot's OTHER buffers (ROI lists,
 *          counts, matcher outputs, correspondences, lifted points) without waiting for caller_stream when inputs_resident != 0: a
 *          caller that may still have reads of those queued when the slot comes round again passes inputs_resident = 0 for that submit.
 *          "corrs" rows >= the pair's n_sel are undefined (a re-used slot keeps an earlier step's rows there; K2 reads n_sel rows).
 * oryon_engine_submit returns the slot index (>= 0) or a negative error.  inputs_resident != 0: the inputs were complete before
 *          the call (nothing pending on caller_stream produces them), so K0 need not wait for the caller's stream.  The caller
 *          must not overwrite the inputs before oryon_engine_wait(slot, stream) has been passed on the stream that overwrites them.
 * Measurement: oryon_engine_set_timing(1) brackets the three sections and the screening kernel of every step with HIP events;
 *          oryon_engine_timing(step) (step = 0-based index of the submit, one of the last 64) -> {gather, match+lift, screening
 *          kernel, registration} durations in ms and the start / end of the match and registration sections relative to the start
 *          of the gather (valid once the step has completed).
 *          oryon_engine_host_stats: host time spent inside oryon_engine_submit.
 * Hardware queues: with overlap == 2 the engine owns four streams; the HIP runtime maps all streams of the process onto
 *          GPU_MAX_HW_QUEUES (default 4) hardware queues, and streams that share one execute in submission order.  Run the host
 *          process with GPU_MAX_HW_QUEUES >= 6 (the Python package sets 8 at import) or the two registration streams serialise. */
typedef struct oryon_engine oryon_engine_t;
typedef struct {
    int B, C, FH, FW;        /* pairs per step; descriptor maps [B,C,FH,FW] */
    int HA, WA, HQ, WQ;      /* depth maps [B,HA,WA] / [B,HQ,WQ] fp32 millimetres */
    int layout;              /* ORYON_LAYOUT_NCHW | ORYON_LAYOUT_NHWC */
    float dist_th;           /* test.dist_th (0.25) */
    int n_corrs;             /* test.n_corrs (500) */
    int src_sampling;        /* test.src_sampling (5000); 0 = keep every ROI pixel */
    uint64_t seed;
    int round_f16;           /* the reference's half-descriptor branch (utils/pcd.py:195-197) */
    int n_slots;             /* result slots: a step's results stay readable until the n_slots-th next submit (6; <= 8) */
    int overlap;             /* see above (2) */
    int gather_sets;         /* K0 output sets: K0 may run this many steps ahead of the matcher minus one (<= 4, <= n_slots).  2 is the Python binding's
                                default since round 5: a third set adds 3.4 ms of queueing to a step's latency (14.1 -> 10.7 ms at cfg2) and no throughput */
    int reg_streams;         /* registration streams the steps alternate over (2; <= 4) */
    int reg_lag;             /* > 0: the matcher of step k waits for the registration of step k - reg_lag (< n_slots); 0 = never (default) */
    int screen;              /* 0 = int8 screen (oryon_gather_q8 + oryon_match_corrs_i8), 1 = MX-fp6 screen (oryon_gather_mx6 +
                                oryon_match_corrs_mx6; steps submitted with force_eager take the int8 route) */
    int sample_first;        /* 0 = off (default).  N > 0: the "sample first" schedule (see oryon_sample_first_gate): the matcher first
                                sees a uniformly random N-anchor subset per pair; pairs whose subset holds fewer than n_corrs valid rows are
                                redone on all anchors, gated on the device.  Same distribution of the sampled correspondences, not the same
                                sample as the default schedule; steps submitted with force_eager ignore it */
    int x3_prefetch;         /* 1 (default in the Python binding): when the steps that completed most recently left more than a quarter of their
                                anchors to the second level (n_und, read back through pinned memory without a synchronisation), K0 writes the
                                query rows' hi / lo halves in its own pass (oryon_gather_mx6_x3) and the matcher skips its second read of the maps
                                (oryon_match_corrs_mx6_x3).  Results are unchanged; C <= 256 and the MX-fp6 screen only.  0 = never */
    int stream_roles;        /* which of the device's eight pooled streams (numbered in creation order) serve as match / gather / registration 0 /
                                registration 1: four decimal digits, each 0..7.  0 = the library's default (2345).  The HIP runtime multiplexes a
                                process's streams onto a few hardware queues (the arbitration between them is undocumented; found by measurement):
                                the placement alone moves the pipelined step by up to 30 % at 64 pairs and 70 % at 16, and which placements
                                are good depends on what else the process created first (DESIGN.md "stream placement": the default is
                                the one that stayed within 2 % of the best both in a plain process and behind an RCCL communicator).  A
                                host calls oryon_engine_warm_streams() before it creates other streams, and measures the candidates on its
                                own process with oryon_engine_set_stream_roles (oryon_amd.engine.MatchPoseEngine.tune_stream_roles does).
                                Results never
                                depend on it */
} oryon_engine_config_t;
size_t oryon_engine_config_bytes(void);      /* sizeof(oryon_engine_config_t) in the built library: a binding's mirror of the struct must match */
size_t oryon_engine_arena_bytes(const oryon_engine_config_t *cfg, const oryon_pointdsc_t *solver);
int oryon_engine_create(oryon_engine_t **handle, const oryon_engine_config_t *cfg, oryon_pointdsc_t *solver, void *arena,
                        size_t arena_bytes);
void oryon_engine_destroy(oryon_engine_t *handle);
/* Creates the current device's stream pool now (idempotent, thread-safe; every new stream runs one empty kernel, which binds its hardware
 * queue): call it once per process BEFORE anything else creates HIP streams on the device - torch.distributed.init_process_group("nccl",
 * device_id=...) creates RCCL's - so that the engine's streams are the process's first (run_test.py:31 launches one process per GPU).
 * That alone does not make every placement equally good behind a communicator (measured, see cfg.stream_roles): the default one is. */
int oryon_engine_warm_streams(void);
/* Re-assigns a live engine's streams (same digits as cfg.stream_roles; 0 = default).  Drains the engine's streams first (synchronises
 * the host with the steps in flight): a tuning aid for warm-up, not for the steady state.  oryon_engine_stream_roles reads the placement. */
int oryon_engine_set_stream_roles(oryon_engine_t *handle, int roles);
int oryon_engine_stream_roles(const oryon_engine_t *handle, int *roles);
int oryon_engine_buffer(const oryon_engine_t *handle, int slot, const char *name, size_t *offset, size_t *bytes);
int oryon_engine_geometry(const oryon_engine_t *handle, int *cap_a, int *cap_q, int *c_pad, int *n_cap);
/* feat_* [B,C,FH,FW] fp32 in cfg.layout; mask_* [B,FH*FW] int32 (== 1 selects); depth_* fp32; cam_* [B,9] fp32; pair_key [B] int64 or NULL */
int oryon_engine_submit(oryon_engine_t *handle, const float *feat_a, const float *feat_q, const int32_t *mask_a, const int32_t *mask_q,
                        const float *depth_a, const float *depth_q, const float *cam_a, const float *cam_q, const int64_t *pair_key,
                        int force_eager, int inputs_resident, void *caller_stream);
int oryon_engine_wait(oryon_engine_t *handle, int slot, void *caller_stream);
int oryon_engine_set_timing(oryon_engine_t *handle, int enable);
int oryon_engine_timing(oryon_engine_t *handle, int64_t step, float *out8);
/* ms from timing event `event_a` of step `step_a` to event `event_b` of step `step_b` (events per step: 0/1 gather begin / end,
 * 2/3 match + lift begin / end, 4/5 screening kernel begin / end, 6/7 registration begin / end): timelines across steps. */
int oryon_engine_elapsed(oryon_engine_t *handle, int64_t step_a, int event_a, int64_t step_b, int event_b, float *ms);
/* ms the two K0 gather launches (queries, anchors: the HBM-bound kernels of the matcher) of step `step` took, i.e. the gather section
 * without the ROI kernels in front of it (timing must have been on; the step must have completed) */
int oryon_engine_gather_ms(oryon_engine_t *handle, int64_t step, float *ms);
int oryon_engine_host_stats(const oryon_engine_t *handle, int64_t *n_submit, double *submit_ms_total, double *submit_ms_last);
/* number of submits so far whose K0 pass wrote the hi / lo rows (cfg.x3_prefetch) */
int oryon_engine_x3_steps(const oryon_engine_t *handle, int64_t *n_steps);
/* the engine's own feedback, for callers that want the statistic without queueing reads of slot buffers (round 5): sums of the newest
 * COMPLETED step's per-pair counts of anchors the screen left to the second level ("n_und") and of anchors ("n_a") - the pinned-memory
 * copies the x3_prefetch decision is made from (MX-fp6 screen only).  *step = that step's submit index, -1 when none has completed yet.
 * Never waits: an event query per slot. */
int oryon_engine_feedback(oryon_engine_t *handle, int64_t *step, int64_t *n_undecided, int64_t *n_anchors);

/* B4  error-compensated fp16x3 linear layer for the frozen fp32 towers (CLIP ViT-L/14@336, Swin) of Oryon.forward
 *     (net.py:142-167, models/vlm.py:43-61; the reference evaluates them with fp32 torch linears):
 *         C[M,N] = act(A[M,K] * W[N,K]^T + bias[N]),   W = W_hi + W_lo (two fp16 matrices made once by oryon_split_f16x3)
 *     with every product accumulated as Ahi*Whi + Ahi*Wlo + Alo*Whi on the fp16 matrix pipe (fp32 accumulate): ~2^-22 relative, i.e.
 *     fp32-grade results at ~3x the fp32-MFMA rate.  act: 0 = none, 1 = QuickGELU x*sigmoid(1.702x) (CLIP), 2 = GELU x*Phi(x) with erf (Swin's nn.GELU), fused
 *     into the epilogue.
 *     K % 32 == 0, N % 256 == 0 (N % 128 == 0 when K >= 64), N * K < 2^30, |values| < 65504.
 *     W_lo == NULL (K >= 64 only) declares that every weight IS an fp16 value (its low half would be all zeros: what `clip.load` leaves in
 *     the reference's CLIPEncoder, models/vlm.py:19-22 - an fp16 checkpoint widened to fp32): the Ahi*Wlo products are left out, two
 *     instead of three MFMAs per product, results bit-identical to passing a zero W_lo. */
int oryon_split_f16x3(const float *x, int64_t n, void *hi_f16, void *lo_f16, void *stream);
/* Range check of the fp16x3 path (round 5; per stream since round 6).  The PRODUCERS of every tensor the fp16x3 kernels split - the B4
 * linear (both kernels; the in-place residual form checks its own finished sum, and the residual value its atomics leave in C is
 * checked by the fused residual-add + LayerNorm pass that reads it next, oryon_add_layernorm_f32), the 24 x 24 and decoder
 * convolutions / up-convolutions - OR 1 into a flag word when one of their finished values is not a finite value below
 * 60000 in magnitude: that is what an operand beyond float16's range (|x| >= 65520: hi = inf) produces in every product it enters, and
 * what an output that the NEXT kernel could not split looks like.  The attention kernels (B5, the Swin / fusion window attentions) and
 * the single-slab persistent decoder convolutions carry no check of their own: their q | k | v operands are outputs of checked linears
 * (an overflowing operand has raised the flag already, and their own outputs are convex combinations of checked v rows), the
 * convolutions' inputs are checked up-convolution outputs and GroupNorm-normalised maps, and whatever they produce goes through a
 * checked linear / convolution next (tests/test_backbone_pins.py::test_fp16x3_overflow_behind_an_unchecked_kernel_is_flagged_by_the_next executes
 * that argument).
 * The word belongs to (device, stream): a launch raises the word of the stream it runs on, oryon_x3_range_flag(stream) reads that word
 * after everything queued on `stream` (synchronises that stream: once per forward, not per layer) and clears it when reset != 0;
 * value_out == NULL only queues the clear (and allocates the device's flag table at the first call - make that call at initialisation so
 * that no launch allocates, e.g. under graph capture).  Forwards on different streams, threads or model instances do not see each
 * other's flags.  oryon_amd.net.Oryon.forward clears its stream's word before the forward, reads it after, and re-evaluates a flagged
 * forward with the fp32 torch modules. */
int oryon_x3_range_flag(int *value_out, int reset, void *stream);
int oryon_linear_f16x3(const float *A, int M, int K, const void *W_hi, const void *W_lo, const float *bias, int N, int act, float *C,
                       void *stream);
/* C[M,N] += A W^T + bias (no activation; K >= 64, N % 128 == 0, N * K < 2^30): the towers' residual update x = x + linear(h)
 * (clip's ResidualAttentionBlock: x + attn(ln_1(x)), x + mlp(ln_2(x)); torchvision's SwinTransformerBlock the same) done by the linear
 * itself - every element of C receives ONE fire-and-forget fp32 atomic add of its finished sum, i.e. exactly the rounding of the separate
 * add, in a fixed order (no two lanes touch the same element) - so that the LayerNorm pass which follows reads one tensor, not two.
 * C must not alias A. */
int oryon_linear_f16x3_acc(const float *A, int M, int K, const void *W_hi, const void *W_lo, const float *bias, int N, float *C,
                           void *stream);

/* B5  multi-head self-attention of the frozen CLIP image tower (clip's ResidualAttentionBlock.attention reached through
 *     models/vlm.py:46-56; nn.MultiheadAttention without mask) in fp32-grade arithmetic on the fp16 matrix pipe (both products
 *     error-compensated like B4, flash-style, fp32 softmax): qkv [N, L, 3*heads*64] fp32 = the in_proj output, out [N, L, heads*64]. */
int oryon_mha_f16x3(const float *qkv, int N, int L, int heads, int head_dim, float *out, void *stream);

/* f3  pose-accuracy metrics on the device for a batch of pairs.
 *     Replaces utils/metrics.py:194-220 (compute_add / compute_adds, with the FLOAT16 model transform of utils/pcd.py:127-133) and
 *     utils/metrics.py:222-259 (compute_RT_distances) of the reference's evaluator (utils/evaluator.py:206-256).
 * pred_pose / gt_pose [B,16] row-major 4x4 (metres); model_pts [sum M, 3] fp32 (metres) with pts_offset [n_models+1] (prefix sums) and
 * model_of_pair [B] (NULL: every pair uses model 0); max_pts = largest model; workspace [B,2] fp32.
 * out [B,4] = (ADD, ADD-S, rotation error in degrees, translation error in centimetres). */
int oryon_pose_metrics(const float *pred_pose, const float *gt_pose, int B, const float *model_pts, const int32_t *pts_offset, int n_models,
                       int max_pts, const int32_t *model_of_pair, float *workspace, float *out, void *stream);

/* f3  the BOP errors the reference's evaluator reports next to ADD(-S): MSSD (maximum symmetry-aware surface distance, millimetres)
 *     and MSPD (maximum symmetry-aware projection distance, pixels) for a batch of pairs.
 *     Replaces utils/evaluator.py:258-275 (both poses rounded to FLOAT16, translation = half(t) * 1000 in half arithmetic) +
 *     bop_toolkit_lib/pose_error.py:370-427 (my_mssd, my_mspd: float64 arithmetic on the rounded poses).
 * pred_pose / gt_pose [B,16] float64 row-major 4x4 (metres; float64 so that the half rounding sees the caller's values, whatever their
 * type); K [B,9] float64 (the query camera); model_pts_mm [sum M, 3] float64 MILLIMETRES (what the evaluator holds) with pts_offset
 * [n_models+1]; syms [sum S, 12] float64 = the models' symmetry sets as row-major 3x4 [R|t] (bop_toolkit_lib/misc.py:43-90,
 * format_sym_set :402-411; the identity included) with sym_offset [n_models+1]; max_syms = largest set; model_of_pair [B] or NULL.
 * max_points: 3 = what the reference computes (its np_transform, pose_error.py:339-352, slices `pts[:, :3]` on the POINT axis of a
 * [1,N,3] array, so my_mssd / my_mspd see the first three model points only); 0 = every point (the BOP definition).
 * workspace: oryon_pose_bop_workspace_bytes(B, max_syms).  out [B,2] float64 = (MSSD error, MSPD error). */
size_t oryon_pose_bop_workspace_bytes(int B, int max_syms);
int oryon_pose_bop_errors(const double *pred_pose, const double *gt_pose, const double *K, int B, const double *model_pts_mm,
                          const int32_t *pts_offset, const double *syms, const int32_t *sym_offset, int n_models, int max_syms,
                          const int32_t *model_of_pair, int max_points, double *workspace, double *out, void *stream);

/* a5  StandardDecoder.forward (models/decoder.py:82-108) on the device, fp32 in / fp32 out, for the decoder the reference builds
 *     (get_decoder, models/decoder.py:119-125: input_dim 128, decoder_dims [64, 32], extra_upsampling, guidance projections 256->32 and
 *     128->16): three Up blocks (ConvTranspose2d 2x2 s2 -> cat guidance -> (conv3x3 - GroupNorm(C/16) - ReLU) x 2, :9-42), the two
 *     guidance projections (conv3x3 + bias + ReLU, :66-72) and the 3x3 head (:80).  Replaces the cuDNN / MIOpen convolutions and the
 *     separate GroupNorm / ReLU / cat / clone passes of the reference graph: every convolution is an implicit GEMM on the fp16 matrix
 *     pipe with error-compensated (hi + lo) operands - fp32-grade, see csrc/decoder.hip - GroupNorm + ReLU are applied by the next
 *     layer's loader, the statistics are reduced in a fixed order (bit-reproducible).
 * Weights: DEVICE pointers to the module's fp32 parameters in torch layout (state-dict names in the comments); oryon_decoder_create packs
 * them (a few small launches on `stream`, then one synchronise: the caller's tensors are not referenced afterwards). */
typedef struct oryon_decoder oryon_decoder_t;
typedef struct {
    const float *gp_w[2], *gp_b[2];   /* decoder_guidance_projection.{0,1}.0.{weight,bias}: [32,256,3,3] [32]; [16,128,3,3] [16] */
    const float *up_w[3], *up_b[3];   /* decoder{1,2,3}.up.{weight,bias}: [128,96,2,2] [96]; [64,48,2,2] [48]; [32,32,2,2] [32] */
    const float *c1_w[3];             /* decoder{1,2,3}.conv.double_conv.0.weight: [64,128,3,3]; [32,64,3,3]; [32,32,3,3] */
    const float *n1_g[3], *n1_b[3];   /* ....double_conv.1.{weight,bias}: [64]; [32]; [32] */
    const float *c2_w[3];             /* ....double_conv.3.weight: [64,64,3,3]; [32,32,3,3]; [32,32,3,3] */
    const float *n2_g[3], *n2_b[3];   /* ....double_conv.4.{weight,bias} */
    const float *head_w, *head_b;     /* head.{weight,bias}: [1,32,3,3] [1] */
} oryon_decoder_weights_t;
int oryon_decoder_create(const oryon_decoder_weights_t *weights, oryon_decoder_t **handle, void *stream);
void oryon_decoder_destroy(oryon_decoder_t *handle);
/* x [n_img, 128, h, w] NCHW (= rearrange(fusion output, 'B C T H W -> (B T) C H W'), :93), g2 [n_img, 256, 2h, 2w], g3 [n_img, 128, 4h, 4w]
 * (guidance[1:], :86; guidance_layout = ORYON_LAYOUT_NCHW, or ORYON_LAYOUT_NHWC for [n_img, 2h, 2w, 256] / [n_img, 4h, 4w, 128] - the Swin
 * tower's own token layout, of which net.py:72-75 returns permuted views), h % 8 == 0, w % 8 == 0 (the reference: 24 x 24) -> featmap [n_img, 32, 8h, 8w] NCHW, logits [n_img, 8h, 8w].
 * workspace: oryon_decoder_workspace_bytes(n_img, h, w) bytes (0 = unsupported shape), 256-byte aligned, no other requirements; three
 * activation buffers at the offsets oryon_decoder_workspace_layout reports (tests read intermediates there).
 * stop_after: 0 = the whole module; k = 3 i + j (debug / tests): return after block i's cat buffer (j = 1), first (j = 2) or second
 * (j = 3) convolution - raw NHWC outputs in buffers 0, 1, 2. */
int64_t oryon_decoder_workspace_bytes(int n_img, int h, int w);
int oryon_decoder_workspace_layout(int n_img, int h, int w, int64_t *offsets3);
int oryon_decoder_forward(const oryon_decoder_t *handle, const float *x, const float *g2, const float *g3, int n_img, int h, int w,
                          void *workspace, int64_t workspace_bytes, float *featmap, float *logits, int guidance_layout, int stop_after, void *stream);

/* a4  window attention of ImageTextFusion's guided Swin blocks (models/fusion.py:75-103 inside :173-213) on un-windowed tokens:
 *     qk [B, H, W, 2C] fp32 (the q projection, then the k projection of [x | guidance]), v [B, H, W, C] fp32 -> out [B, H, W, C] fp32 =
 *     roll(-shift) -> 12 x 12 windows -> softmax(q k^T / sqrt(32) + shift mask) v per head -> windows back -> roll(+shift), one kernel;
 *     both products on the fp16 matrix pipe with error-compensated operands (fp32-grade, ~1e-6), fp32 softmax.  head_dim 32 (C == heads * 32), window == 12, H % 12 == 0, W % 12 == 0 (the reference: 24 x 24, 4 heads). */
int oryon_fusion_window_attention_f32(const float *qk, const float *v, int B, int H, int W, int C, int heads, int window, int shift,
                                      float *out, void *stream);

/* a4  the two convolutions of ImageTextFusion on its 24 x 24 maps (models/fusion.py:562 + :595-600: conv1 7x7 pad 3 on the cost volume;
 *     :566-570 + :614-615: guidance_projection 3x3 pad 1 + ReLU) as implicit GEMMs on the fp16 matrix pipe with error-compensated operands
 *     (fp32-grade).  x [n, 24, 24, cin] fp32 NHWC -> y [n, 24, 24, cout] fp32 NHWC = act(conv(x, w) + bias); ksize 3 or 7 (stride 1, padding
 *     ksize / 2), cout % 64 == 0, cin % 4 == 0.  `image`: the weights packed once by oryon_conv24_pack_f16x3 from torch's [cout, cin, k, k]
 *     layout into oryon_conv24_image_bytes(cout, cin, ksize) bytes (0 = unsupported shape). */
int64_t oryon_conv24_image_bytes(int cout, int cin, int ksize);
int oryon_conv24_pack_f16x3(const float *w, int cout, int cin, int ksize, void *image, void *stream);
int oryon_conv24_f16x3(const float *x, int n, int cin, const void *image, const float *bias, int cout, int ksize, int relu, float *y, void *stream);

/* a4  the class-aggregation layer of the fusion module's aggregator (models/fusion.py:300-332 around the LinearAttention of :240-266) for the
 *     reference's single prompt axis (T = 1) on 24 x 24 maps with 128 channels, 4 heads, 6 x 6 pooling, 128-wide text guidance: AvgPool ->
 *     LayerNorm -> q | k from [tokens | text guidance], v -> elu + 1 linear attention -> residual -> LayerNorm -> MLP 128 -> 512 -> 128 (ReLU)
 *     -> residual -> bilinear upsampling (align_corners) -> residual on the map, one kernel, fp32 arithmetic.
 *     x [B, 24, 24, 128] fp32 NHWC, text_guidance [B, 128] fp32 -> out [B, 24, 24, 128] fp32 NHWC.  Weights: device pointers, torch layouts
 *     (norm1 / norm2: [128]; attention.q / .k: [128, 256] + [128]; attention.v: [128, 128] + [128]; MLP.0: [512, 128] + [512]; MLP.2: [128, 512] + [128]). */
typedef struct {
    const float *ln1_w, *ln1_b, *wq, *bq, *wk, *bk, *wv, *bv, *ln2_w, *ln2_b, *w1, *b1, *w2, *b2;
} oryon_fusion_class_weights_t;
int oryon_fusion_class_layer_f32(const float *x, const float *text_guidance, const oryon_fusion_class_weights_t *weights, int B, float *out,
                                 void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ORYON_HIP_H */
