"""Batched match -> lift -> registration engine: the whole per-sample loop of FPM_Pipeline.test_step
(pipeline.py:313-355) for B pairs at once, with no host synchronisation between stages.

    masks -> ROI (K0) -> subsample <=5000 (K0, device RNG) -> gather+normalise (K0) -> cosine NN (K1)
          -> sample 500 correspondences (K1b, device RNG) -> scale/validate/lift (K2) -> PointDSC (K3-K10)

Differences from the drop-in per-sample facade (oryon_amd.pcd / oryon_amd.pointdsc), by design:
  * the two random draws use a counter-based device RNG keyed by (seed, global pair index), so results do
    not depend on how pairs are sharded over GPUs; the reference's host torch.multinomial stream is kept
    only in the per-sample facade;
  * failures are reported as per-pair status codes (ORYON_PAIR_*) with an identity pose, exactly the values
    pipeline.py:335-350 writes.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional

import torch
from torch import Tensor

from . import _lib, ops
from .pointdsc import PointDSC

PAIR_OK, PAIR_NO_MASK, PAIR_NO_CORR = 0, 1, 2


@dataclass
class MatchPoseConfig:
    """Flag names follow configs/config.yaml of the reference (test.* section)."""
    dist_th: float = 0.25          # test.dist_th
    n_corrs: int = 500             # test.n_corrs (= dataset.max_corrs)
    src_sampling: Optional[int] = 5000   # test.src_sampling
    seed: int = 1                  # seed (pipeline.py:296-299)
    # "screened":   int8 pre-screen (K1s8, C_pad 256 / 512) -> fp16 screening (K1s) -> exact fp32 re-scoring; identical valid
    #               set / argmin / min_dist on valid rows for every input, used when 64 < C <= 512
    # "screened16": the same without the int8 stage
    # "exact":      full fp32-MFMA scan (K1) for every row
    match_mode: str = "screened"
    # True = the reference's half-descriptor branch (utils/pcd.py:195-197, corrs_device='cuda'; BASELINE configs[4] "fp16 descriptors"):
    # every descriptor value is rounded to float16 before anything else; the contraction itself is unchanged
    half_descriptors: bool = False
    # > 0: "sample first" schedule of the int8 route - the matcher runs on a uniformly random subset of this many anchors per pair first;
    # since only n_corrs correspondences leave it, drawn uniformly from the valid rows (utils/pcd.py:205-214), a subset that already holds
    # >= n_corrs valid rows gives an identically distributed sample at a fraction of the contraction.  Pairs whose subset came up short
    # are redone on all of their anchors (device-gated, no host round trip).  Off (0) by default: the default route computes the
    # validity of every one of the <= src_sampling anchors, as the reference does.
    sample_first: int = 0


class MatchPoseEngine:
    def __init__(self, solver: PointDSC, cfg: Optional[MatchPoseConfig] = None, overlap_registration: bool = False,
                 overlap_gather: bool = False):
        """overlap_registration: run the registration stage (K3-K10: many small, latency-bound launches) on a second HIP
        stream so that it overlaps with the matching stage of the NEXT batch submitted by the caller; `run` then returns
        immediately after queueing and `finish(out)` makes the caller's stream wait for the poses.
        overlap_gather: additionally run K0 (ROI + gather/normalise: HBM-bound) on its own stream, so that the gather of
        the next batch overlaps the MFMA-bound screening of the current one."""
        self.solver = solver
        self.cfg = cfg or MatchPoseConfig()
        self.n_cap = ops.round_up(self.cfg.n_corrs, 128)
        self.overlap = overlap_registration
        self.overlap_gather = overlap_gather
        self._reg_stream = None
        self.reg_streams = 2
        self._gather_stream = None
        # adaptive use of the int8 pre-screen: the fraction of anchors it had to hand to the fp16 stage is read back
        # asynchronously (pinned buffer + event, never a sync); above `i8_max_undecided` the next batches skip the int8 stage
        # and it is tried again every `i8_retry_every` batches.  Exactness never depends on this - only the run time does.
        # (round 2: the lazy tail resolves ambiguous anchors with an exact scan of the sampled rows only, which is cheaper than the fp16
        # route on every distribution tried, so the back-off is off by default - thresholds >= 1 never trigger)
        self.i8_max_undecided = 2.0
        self.collect_i8_stats = False       # measurement aid: keep `_i8_frac` up to date even though the back-off is off
        self.i8_retry_every = 16
        self._i8_pending = None        # (pinned [2] int64 tensor, event)
        self._i8_host = None           # pinned buffer, allocated once
        self._i8_frac = 0.0
        self._i8_skipped = 0

    def finish(self, out: Dict[str, Tensor]) -> Dict[str, Tensor]:
        """Order the caller's current stream after the registration of `out` (no-op without overlap)."""
        ev = out.pop("_done", None)
        if ev is not None:
            cur = torch.cuda.current_stream(out["pose"].device)
            cur.wait_event(ev)
            for k in ("pose", "status"):
                out[k].record_stream(cur)
        return out

    @torch.no_grad()
    def run(self, feat_a: Tensor, feat_q: Tensor, mask_a: Tensor, mask_q: Tensor, depth_a: Tensor, depth_q: Tensor,
            cam_a: Tensor, cam_q: Tensor, pair_key: Optional[Tensor] = None, keep: bool = False,
            inputs_event: Optional["torch.cuda.Event"] = None, inputs_resident: bool = False) -> Dict[str, Tensor]:
        """feat_* [B,C,FH,FW] fp32, mask_* [B,FH,FW] int (==1 selects), depth_* [B,H,W] fp32 mm, cam_* [B,3,3] or [B,9].
        Returns pose [B,4,4] fp32 (identity on failure), status [B] int32, n_valid [B], n_lifted [B].
        With overlap_gather the gather stream starts after `inputs_event` (the producer's event), or immediately when
        `inputs_resident` says the inputs were complete before this call; by default it waits for everything queued on the
        caller's stream so far (always correct, but then it cannot overlap the previous batch's matching)."""
        dev = _lib.require_gpu(feat_a.device)
        cfg = self.cfg
        B, C, FH, FW = feat_a.shape
        if pair_key is None:
            pair_key = torch.arange(B, dtype=torch.int64, device=dev)
        main = torch.cuda.current_stream(dev)
        # the screens' validity cut is 1 - 2*dist_th: thresholds above 0.5 (or non-positive) take the exact scan
        screened = cfg.match_mode in ("screened", "screened16") and 64 < C <= 512 and 0.0 < cfg.dist_th <= 0.5
        use_i8 = screened and cfg.match_mode == "screened" and C > 128
        if use_i8:
            if self._i8_pending is not None and self._i8_pending[1].query():
                h = self._i8_pending[0]
                self._i8_frac = int(h[0].sum()) / max(1, int(h[1].sum()))
                self._i8_pending = None
            if self._i8_frac > self.i8_max_undecided:
                self._i8_skipped += 1
                if self._i8_skipped < self.i8_retry_every:
                    use_i8 = False
                else:
                    self._i8_skipped, self._i8_frac = 0, 0.0
        if cfg.half_descriptors and not use_i8:
            # every route but the int8 one takes pre-rounded maps (one extra pass; K0v3 rounds on the way in)
            feat_a, feat_q = ops.round_to_f16(feat_a), ops.round_to_f16(feat_q)
        if self.overlap_gather:
            if self._gather_stream is None:
                self._gather_stream = torch.cuda.Stream(device=dev)
            if inputs_event is None and not inputs_resident:
                inputs_event = torch.cuda.Event()
                inputs_event.record(main)
            gctx = torch.cuda.stream(self._gather_stream)
            gctx.__enter__()
            if inputs_event is not None:
                self._gather_stream.wait_event(inputs_event)
        # two launches on the callers' tensors instead of a torch.cat + one launch: no torch arithmetic inside the step
        roi_a, n_a = ops.roi_compact(mask_a.reshape(B, FH, FW))
        roi_q, n_q = ops.roi_compact(mask_q.reshape(B, FH, FW))
        if cfg.src_sampling is not None:
            ops.roi_subsample_(roi_a, n_a, cfg.src_sampling, cfg.seed, pair_key)
            cap_a = ops.round_up(min(cfg.src_sampling, FH * FW), ops.ROW_PAD)
        else:
            cap_a = ops.round_up(FH * FW, ops.ROW_PAD)
        cap_q = ops.round_up(FH * FW, ops.ROW_PAD)
        a16 = q16 = a8 = q8 = a_sc = q_sc = q_eps = q_norm = q_hat = roi_a1 = n_a1 = None
        stage1 = use_i8 and cfg.sample_first > 0 and not keep
        if use_i8:
            # K0v3: anchors -> fp32 + int8 rows, queries -> int8 rows + row norms only (the re-scoring pass reads its few candidates
            # from the raw map); contiguous and channels_last maps are both read in place
            c_pad = 256 if C <= 256 else 512
            q8, q_sc, q_eps, q_norm, _ = ops.gather_q8(feat_q, roi_q, n_q, cap_q, c_pad, round_f16=cfg.half_descriptors)
            if stage1:
                # first-stage anchors: a random subset (second-level device-RNG subsample, row-major order kept) of the pair's anchors
                roi_a1, n_a1 = roi_a.clone(), n_a.clone()
                ops.roi_subsample_(roi_a1, n_a1, cfg.sample_first, cfg.seed ^ 0x5A17F125, pair_key)
                cap_a1 = ops.round_up(min(cfg.sample_first, FH * FW), ops.ROW_PAD)
                a8, a_sc, _, _, a_hat = ops.gather_q8(feat_a, roi_a1, n_a1, cap_a1, c_pad, want_f32=True, round_f16=cfg.half_descriptors)
            else:
                a8, a_sc, _, _, a_hat = ops.gather_q8(feat_a, roi_a, n_a, cap_a, c_pad, want_f32=True, round_f16=cfg.half_descriptors)
        elif screened:
            c_pad = 128 if C <= 128 else (256 if C <= 256 else 512)
            a_hat, a16 = ops.gather_normalise(feat_a.contiguous(), roi_a, n_a, cap_a, c_pad=c_pad, want_f16=True)
            q_hat, q16 = ops.gather_normalise(feat_q.contiguous(), roi_q, n_q, cap_q, c_pad=c_pad, want_f16=True)
        else:
            a_hat = ops.gather_normalise(feat_a.contiguous(), roi_a, n_a, cap_a)
            q_hat = ops.gather_normalise(feat_q.contiguous(), roi_q, n_q, cap_q)
        if self.overlap_gather:
            gathered = torch.cuda.Event()
            gathered.record(self._gather_stream)
            gctx.__exit__(None, None, None)
            main.wait_event(gathered)
            # EVERY tensor allocated under the gather stream and read on the main stream: without the record the allocator would hand
            # its memory to the next batch's gather (which runs ahead) while this batch's matcher still reads it
            for t_ in (roi_a, roi_q, n_a, n_q, a_hat, q_hat, a16, q16, a8, q8, a_sc, q_sc, q_eps, q_norm, roi_a1, n_a1):
                if t_ is not None:
                    t_.record_stream(main)
        if use_i8:
            # lazy K1s8 + K1b: validity of every anchor from the int8 bound, exact argmin for the sampled anchors only; `keep` (the
            # caller wants the complete min_dist / argmin arrays) forces the eager route
            n_und = torch.empty((B,), dtype=torch.int32, device=dev)
            if stage1:
                corrs, n_valid, n_sel, status, min_dist, argmin, valid = ops.match_corrs_i8(
                    a_hat, a8, a_sc, feat_q, roi_a1, roi_q, q_norm, q8, q_sc, q_eps, n_a1, n_q, cfg.dist_th, FW, cfg.n_corrs, cfg.seed,
                    pair_key, corr_rows=self.n_cap, n_undecided=n_und, round_f16=cfg.half_descriptors)
                # second stage, gated on the device: all anchors of the pairs whose subset held fewer than n_corrs valid rows
                n_a2 = ops.sample_first_gate(n_valid, n_a1, n_a, cfg.n_corrs)
                a8b, a_scb, _, _, a_hatb = ops.gather_q8(feat_a, roi_a, n_a2, cap_a, c_pad, want_f32=True, round_f16=cfg.half_descriptors)
                c2, nv2, ns2, st2, _, _, _ = ops.match_corrs_i8(
                    a_hatb, a8b, a_scb, feat_q, roi_a, roi_q, q_norm, q8, q_sc, q_eps, n_a2, n_q, cfg.dist_th, FW, cfg.n_corrs, cfg.seed,
                    pair_key, corr_rows=self.n_cap, round_f16=cfg.half_descriptors)
                ops.sample_first_merge_(n_a2, c2, nv2, ns2, st2, corrs, n_valid, n_sel, status)
            else:
                corrs, n_valid, n_sel, status, min_dist, argmin, valid = ops.match_corrs_i8(
                    a_hat, a8, a_sc, feat_q, roi_a, roi_q, q_norm, q8, q_sc, q_eps, n_a, n_q, cfg.dist_th, FW, cfg.n_corrs, cfg.seed,
                    pair_key, corr_rows=self.n_cap, force_eager=keep, n_undecided=n_und, round_f16=cfg.half_descriptors)
            # statistics for the back-off (skip the int8 stage while most anchors come back undecided): only collected when the
            # back-off can trigger at all (i8_max_undecided < 1; off by default - the lazy tail handles such inputs on the device)
            if self._i8_pending is None and (self.i8_max_undecided < 1.0 or self.collect_i8_stats):
                if self._i8_host is None:
                    self._i8_host = torch.empty((2, B), dtype=torch.int32, pin_memory=True)
                    self._i8_dev = torch.empty((2, B), dtype=torch.int32, device=dev)
                if self._i8_host.shape[1] == B:
                    # two device-to-device copies into one staging tensor + one async D2H: no torch arithmetic in the step
                    self._i8_dev[0].copy_(n_und)
                    self._i8_dev[1].copy_(n_a)
                    self._i8_host.copy_(self._i8_dev, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream(dev))
                    self._i8_pending = (self._i8_host, ev)
        else:
            if screened:
                min_dist, argmin, valid = ops.match_screened(a_hat, q_hat, a16, q16, n_a, n_q, cfg.dist_th)
            else:
                min_dist, argmin, valid = ops.match(a_hat, q_hat, n_a, n_q, cfg.dist_th)
            corrs, n_valid, n_sel, status = ops.select_corrs(roi_a, roi_q, n_a, n_q, argmin, valid, FW, cfg.n_corrs, cfg.seed,
                                                             pair_key, corr_rows=self.n_cap)
        cam_a = cam_a.reshape(B, 9).to(torch.float32).contiguous()
        cam_q = cam_q.reshape(B, 9).to(torch.float32).contiguous()
        pcd_a, pcd_q, n_lift = ops.lift_pairs(corrs, n_sel, (FH, FW), depth_a, depth_q, cam_a, cam_q, status)
        if self.overlap:
            # registrations of consecutive batches alternate between two streams (and two workspaces): their many small,
            # low-occupancy launches then overlap each other as well as the next batch's matching
            if self._reg_stream is None:
                self._reg_stream = [torch.cuda.Stream(device=dev) for _ in range(self.reg_streams)]
                self._reg_turn = 0
            slot = self._reg_turn % len(self._reg_stream)
            self._reg_turn += 1
            rs = self._reg_stream[slot]
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(dev))
            with torch.cuda.stream(rs):
                rs.wait_event(ready)
                for t_ in (pcd_a, pcd_q, n_lift, status):
                    t_.record_stream(rs)
                pose, _, status_out = self.solver.register(pcd_a, pcd_q, n_lift, status, ws_slot=slot)
                done = torch.cuda.Event()
                done.record(rs)
        else:
            pose, _, status_out = self.solver.register(pcd_a, pcd_q, n_lift, status)
            done = None
        out = dict(pose=pose, status=status_out, n_valid=n_valid, n_lifted=n_lift)
        if done is not None:
            out["_done"] = done
        if keep:
            out.update(roi_a=roi_a, roi_q=roi_q, n_a=n_a, n_q=n_q, min_dist=min_dist, argmin=argmin, valid=valid, corrs=corrs,
                       pcd_a=pcd_a, pcd_q=pcd_q)
        return out
