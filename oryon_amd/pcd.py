"""Drop-in counterparts of the reference's utils/pcd.py entry points on the hot path.

    nn_correspondences(feats1, feats2, mask1, mask2, threshold, max_corrs, subsample_source, corrs_device)
    lift_pcd(depth, camera, xy_idxs)
    torch_sample_select(t, n)

Same names, argument meaning, return types and failure behaviour as utils/pcd.py:177-216, :35-81 and
utils/misc.py:242-254; the arithmetic runs in liboryon_hip.so (K0 + K1 + K2').  The two random draws keep
the reference's host-side torch.multinomial on the global CPU generator, in the same order, so a seeded
run consumes the RNG stream exactly as the reference does with corrs_device='cpu'.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import Tensor

from . import ops
from ._lib import require_gpu


def torch_sample_select(t: Tensor, n: int) -> Tensor:
    """Exactly n indices into t's first dimension; replacement only when n > N (utils/misc.py:242-254).
    Always drawn from the global CPU generator (the reference's default corrs_device)."""
    N = t.shape[0]
    w = torch.ones(N, dtype=torch.float64)
    return torch.multinomial(w, n, replacement=(n > N))


def match_presample(feats1: Tensor, feats2: Tensor, mask1: Tensor, mask2: Tensor, threshold: float,
                    subsample_source: Optional[int] = None, half_descriptors: bool = False, mode: str = "exact"):
    """Deterministic half of the matcher (utils/pcd.py:184-205) on the GPU.

    Returns dict(roi1 [N1,2] i64 (y,x), roi2 [N2,2] i64, min_dist [N1] f32, argmin [N1] i64, valid [N1] bool).
    When subsample_source is given and N1 exceeds it, the first RNG draw is made exactly as the reference does.
    mode: "exact" = full fp32-MFMA scan (K1); "screened16" = fp16-MFMA screening + exact fp32 re-scoring (K1s);
    "screened8" = int8-MFMA pre-screen in front of K1s (K1s8, the batched engine's default).  The screened modes return the same
    `valid` set and, on valid rows, the same argmin / min_dist bits; rows that provably cannot reach the threshold report
    valid = 0, argmin = 0 and the screening estimate of the distance.  Channels are zero-padded to the kernels' widths.
    "screened6" = the step engine's default route (K0 oryon_gather_mx6 -> oryon_match_corrs_mx6: MX-fp6 screen, lazy tail, K1x3 second
    level, device-RNG sampler): `valid` is exact on every row; `argmin` is exact on the SAMPLED rows only and `min_dist` only where the
    route needed the fp32 comparison (include/oryon_hip.h) - the dict additionally carries `corrs` [n_sel,4] i64 (y1,x1,y2,x2), the
    sampled correspondences (seed 1, max_corrs 500), and `status`."""
    if mode not in ("exact", "screened16", "screened8", "screened6"):
        raise ValueError(f"match_presample: unknown mode {mode!r}")
    if mode != "exact" and not (0.0 < threshold <= 0.5):
        mode = "exact"                       # the screens' validity cut needs 1 - 2*threshold >= 0
    dev = require_gpu(feats1.device)
    W = feats1.shape[2]
    W2 = feats2.shape[2]
    if half_descriptors:
        # K1': the reference's corrs_device='cuda' branch rounds the descriptors to float16 first (utils/pcd.py:195-197); the
        # cosine itself is then evaluated exactly in fp32 on the rounded values (the reference's half arithmetic agrees with
        # this to ~4e-4 in distance, see tests/golden/g1_matcher_half.npz)
        feats1, feats2 = ops.round_to_f16(feats1.to(dev)), ops.round_to_f16(feats2.to(dev))
    f1 = feats1.to(torch.float32).contiguous()[None]
    f2 = feats2.to(torch.float32).contiguous()[None]
    roi1_lin, c1 = ops.roi_compact(mask1.to(dev))
    roi2_lin, c2 = ops.roi_compact(mask2.to(dev))
    n1, n2 = int(c1.item()), int(c2.item())          # the reference syncs here as well (shape of nonzero)
    roi1_lin = roi1_lin[:, :n1]
    if subsample_source is not None and n1 > subsample_source:
        idxs = torch_sample_select(roi1_lin[0], subsample_source).to(dev)
        roi1_lin = roi1_lin[:, idxs]
        n1 = subsample_source
        c1 = torch.full((1,), n1, dtype=torch.int32, device=dev)
    roi1_lin = roi1_lin.contiguous()
    out = dict(
        roi1=torch.stack((roi1_lin[0] // W, roi1_lin[0] % W), dim=1).to(torch.int64),
        roi2=torch.stack((roi2_lin[0, :n2] // W2, roi2_lin[0, :n2] % W2), dim=1).to(torch.int64))
    if n1 == 0 or n2 == 0:
        out.update(min_dist=torch.zeros(n1, device=dev), argmin=torch.zeros(n1, dtype=torch.int64, device=dev),
                   valid=torch.zeros(n1, dtype=torch.bool, device=dev))
        if mode == "screened6":
            out.update(corrs=torch.zeros((0, 4), dtype=torch.int64, device=dev), status=torch.tensor(1, device=dev))
        return out
    C = f1.shape[1]
    cap1, cap2 = ops.round_up(n1, ops.ROW_PAD), ops.round_up(n2, ops.ROW_PAD)
    if mode != "exact" and C > 512:
        mode = "exact"                       # the screening kernels are built for C_pad 128 / 256 / 512
    if mode == "screened6":
        c_pad = 256 if C <= 256 else 512
        a6, a_err, _, a_hat = ops.gather_mx6(f1, roi1_lin, c1, cap1, c_pad, want_f32=True)
        q6, q_err, q_norm, _ = ops.gather_mx6(f2, roi2_lin, c2, cap2, c_pad)
        corrs, n_valid, n_sel, status, min_dist, argmin, valid = ops.match_corrs_mx6(
            a_hat, a6, a_err, f2, roi1_lin, roi2_lin, q_norm, q6, q_err, c1, c2, threshold, W2, 500, 1, corr_rows=512)
        out.update(corrs=corrs[0, : int(n_sel.item())].to(torch.int64), status=status[0])
    elif mode == "screened8":
        c_pad = 256 if C <= 256 else 512
        a_hat, _, a8, a_sc, _ = ops.gather_normalise_q8(f1, roi1_lin, c1, cap1, c_pad)
        q_hat, _, q8, q_sc, q_eps = ops.gather_normalise_q8(f2, roi2_lin, c2, cap2, c_pad)
        min_dist, argmin, valid = ops.match_screened8(a_hat, q_hat, a8, q8, a_sc, q_sc, q_eps, c1, c2, threshold, C)
    elif mode == "screened16":
        c_pad = 128 if C <= 128 else (256 if C <= 256 else 512)
        a_hat, a16 = ops.gather_normalise(f1, roi1_lin, c1, cap1, c_pad=c_pad, want_f16=True)
        q_hat, q16 = ops.gather_normalise(f2, roi2_lin, c2, cap2, c_pad=c_pad, want_f16=True)
        min_dist, argmin, valid = ops.match_screened(a_hat, q_hat, a16, q16, c1, c2, threshold)
    else:
        a_hat = ops.gather_normalise(f1, roi1_lin, c1, cap1)
        q_hat = ops.gather_normalise(f2, roi2_lin, c2, cap2)
        min_dist, argmin, valid = ops.match(a_hat, q_hat, c1, c2, threshold)
    out.update(min_dist=min_dist[0, :n1], argmin=argmin[0, :n1].to(torch.int64), valid=valid[0, :n1].bool())
    return out


def nn_correspondences(feats1: Tensor, feats2: Tensor, mask1: Tensor, mask2: Tensor, threshold: float, max_corrs: int,
                       subsample_source: Optional[int], corrs_device: str = "cuda") -> Optional[Tensor]:
    """Matches between two [D,H,W] feature maps restricted to mask==1; int64 [max_corrs,4] rows
    (y1,x1,y2,x2) on feats1.device, or None when at most one anchor pixel finds a match under the
    threshold (utils/pcd.py:177-216).  The contraction always runs on the GPU and the two
    host RNG draws are always made on the CPU generator.  corrs_device='cuda' selects the reference's fp16-descriptor
    branch (descriptors rounded to float16, utils/pcd.py:195-197); anything else its fp32 branch."""
    orig_device = feats1.device
    pre = match_presample(feats1, feats2, mask1, mask2, threshold, subsample_source, half_descriptors=(corrs_device == "cuda"))
    valid_corr = torch.nonzero(pre["valid"]).squeeze(1)
    if valid_corr.shape[0] > 1:
        roi2 = pre["roi2"][pre["argmin"]]
        final_corrs = torch.cat((pre["roi1"][valid_corr], roi2[valid_corr]), dim=1)
        idxs = torch_sample_select(final_corrs, max_corrs).to(final_corrs.device)
        return final_corrs[idxs].to(orig_device)
    return None


def lift_pcd(depth: Tensor, camera: Tensor, xy_idxs: Optional[Tuple[Tensor, Tensor]] = None) -> Tensor:
    """Depth [H,W,1] (millimetres) + flattened K [9] -> [N,3] points (millimetres), fp32
    (utils/pcd.py:35-81).  With xy_idxs=(x_idx, y_idx) only those pixels are lifted; the caller divides
    by 1000 as pipeline.py:459-460 does."""
    dev = require_gpu(depth.device)
    H, W, D = depth.shape
    if D != 1:
        raise NotImplementedError("RGB-D lifting (D > 1) is off the registration path (utils/pcd.py:76-80)")
    d = depth[:, :, 0].to(torch.float32).contiguous()
    if xy_idxs is None:
        ys, xs = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing="ij")
        x_idx, y_idx = xs.reshape(-1), ys.reshape(-1)
    else:
        x_idx, y_idx = xy_idxs[0].to(dev), xy_idxs[1].to(dev)
        if x_idx.numel() and (int(x_idx.min()) < 0 or int(x_idx.max()) >= W or int(y_idx.min()) < 0 or int(y_idx.max()) >= H):
            raise IndexError("lift_pcd: pixel index out of the depth image")
    cam = camera.reshape(9).to(torch.float32).to(dev)
    return ops.lift_points(d, cam, x_idx, y_idx)
