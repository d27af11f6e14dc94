"""Test-time counterpart of the reference's FPM_Pipeline (pipeline.py:306-355, 372-472, 490-497).

`Pipeline` offers the same four callables the reference's test loop uses, with the same argument dicts:

    is_detection_valid(results, batch, idx) -> bool
    get_featmap_corrs(batch, net_output, results, idx) -> (corrs | None, pos_a, pos_q)
    get_pose(batch, corrs, idx) -> Tensor[4,4] fp32
    test_step(batch, batch_idx)                       # per-sample loop, host RNG: reference semantics
    test_step_batched(batch, first_pair_index=0)      # same work for the whole batch with no host sync

Training, logging, the evaluator (ADD/VSD) and the dataloaders of the reference are out of scope
(SURVEY.md §2.1).  Config flags keep the names of configs/config.yaml.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
from torch import Tensor

from . import _lib, ops
from .engine import MatchPoseConfig, MatchPoseEngine, PAIR_NO_CORR, PAIR_NO_MASK, PAIR_OK
from .pcd import nn_correspondences
from .pointdsc import PointDSC, get_pointdsc_pose


def default_args(**overrides) -> SimpleNamespace:
    """The hot-path subset of configs/config.yaml (same names, same defaults)."""
    args = SimpleNamespace(
        device="cuda", corrs_device="cpu", seed=1,
        dataset=SimpleNamespace(img_size=[224, 224], max_corrs=500),
        model=SimpleNamespace(image_encoder=SimpleNamespace(img_size=[192, 192], out_channels=32)),
        test=SimpleNamespace(mask="predicted", src_sampling=5000, solver="pointdsc", n_corrs=500, dist_th=0.25,
                             mask_threshold=0.5),
        loss=SimpleNamespace(hard_negatives=True),
    )
    for k, v in overrides.items():
        node = args
        parts = k.split(".")
        for p in parts[:-1]:
            node = getattr(node, p)
        setattr(node, parts[-1], v)
    return args


class PrecomputedFeatures:
    """Stand-in for Oryon.forward when descriptors are given: returns the batch's own feature maps / mask logits
    under the output keys of net.py:162-167."""

    def forward(self, batch: Dict) -> Dict[str, Tensor]:
        return {k: batch[k] for k in ("featmap_a", "featmap_q", "mask_a", "mask_q") if k in batch}

    __call__ = forward


def mask_iou(gt: Tensor, pred: Tensor) -> Tensor:
    """Per-sample IoU of binary masks [B,H,W]: |and| / |or|; an empty union gives NaN, as in utils/metrics.py:18-40."""
    g, p = gt != 0, pred != 0
    inter = (g & p).flatten(1).sum(1).float()
    union = (g | p).flatten(1).sum(1).float()
    return inter / union


class Pipeline:
    def __init__(self, args: SimpleNamespace, model=None, pointdsc_solver: Optional[PointDSC] = None):
        self.args = args
        self.device = args.device
        self.corrs_device = args.corrs_device
        self.model = model if model is not None else PrecomputedFeatures()
        self.pointdsc_solver = pointdsc_solver
        self.pred_lines: List[str] = []
        self.failures: List[str] = []
        self._engine: Optional[MatchPoseEngine] = None

    # ------------------------------------------------------------------ mask post-processing (losses.py:56-60)
    def mask_results(self, batch: Dict, outputs: Dict) -> Dict[str, Tensor]:
        th = self.args.test.mask_threshold
        res = {"mask_a": ops.mask_from_logits(outputs["mask_a"].squeeze(1), th),
               "mask_q": ops.mask_from_logits(outputs["mask_q"].squeeze(1), th)}
        for side, key in (("anchor", "a"), ("query", "q")):
            gt = batch.get(side, {}).get("mask") if isinstance(batch.get(side), dict) else None
            pred = res["mask_" + key]
            if gt is not None:
                gt_r = ops.mask_resize_nearest(gt.to(pred.device), pred.shape[-2:]) if gt.shape[-2:] != pred.shape[-2:] else gt.to(pred.device)
                res["iou_" + key] = mask_iou(gt_r, pred)
            else:
                res["iou_" + key] = torch.zeros(pred.shape[0], device=pred.device)
        return res

    def _external_masks(self, batch: Dict, idx: int, size: Tuple[int, int]) -> Tuple[Tensor, Tensor]:
        dev = _lib.require_gpu(self.device)
        ma = ops.mask_resize_nearest(batch["anchor"]["mask"][idx].to(dev), size)[0]
        mq = ops.mask_resize_nearest(batch["query"]["mask"][idx].to(dev), size)[0]
        return ma, mq

    # ------------------------------------------------------------------ pipeline.py:372-395
    def is_detection_valid(self, results: Dict, batch: Dict, idx: int) -> bool:
        if self.args.test.mask != "predicted":
            mask_a, mask_q = self._external_masks(batch, idx, tuple(self.args.model.image_encoder.img_size))
        else:
            mask_a, mask_q = results["mask_a"][idx], results["mask_q"][idx]
        valid_a = torch.count_nonzero(mask_a == 1)
        valid_q = torch.count_nonzero(mask_q == 1)
        return (valid_a.item() > 0) and (valid_q.item() > 0)

    # ------------------------------------------------------------------ pipeline.py:397-427
    def get_featmap_corrs(self, batch: Dict, net_output: Dict, results: Dict, idx: int):
        NH, NW = net_output["featmap_a"].shape[2:]
        if self.args.test.mask != "predicted":
            mask_ai, mask_qi = self._external_masks(batch, idx, (NH, NW))
        else:
            mask_ai, mask_qi = results["mask_a"][idx], results["mask_q"][idx]
        featmap_ai, featmap_qi = net_output["featmap_a"][idx], net_output["featmap_q"][idx]
        pred_corrs = nn_correspondences(featmap_ai, featmap_qi, mask_ai, mask_qi, self.args.test.dist_th, self.args.test.n_corrs,
                                        self.args.test.src_sampling, self.corrs_device)
        if pred_corrs is not None:
            corr_ai, corr_qi = pred_corrs[:, :2], pred_corrs[:, 2:]
            pos_a = featmap_ai[:, corr_ai[:, 0], corr_ai[:, 1]].transpose(1, 0)
            pos_q = featmap_qi[:, corr_qi[:, 0], corr_qi[:, 1]].transpose(1, 0)
        else:
            pos_a, pos_q = None, None
        return pred_corrs, pos_a, pos_q

    # ------------------------------------------------------------------ pipeline.py:429-472
    def get_pose(self, batch: Dict, corrs: Tensor, idx: int) -> Tensor:
        dev = _lib.require_gpu(self.device)
        depth_a = batch["anchor"]["orig_depth"][idx].squeeze().to(dev, torch.float32)
        depth_q = batch["query"]["orig_depth"][idx].squeeze().to(dev, torch.float32)
        camera_a = batch["anchor"]["camera"][idx].reshape(1, 9).to(torch.float32).to(dev)
        camera_q = batch["query"]["camera"][idx].reshape(1, 9).to(torch.float32).to(dev)
        HO, WO = self.args.model.image_encoder.img_size
        if self.args.test.solver == "pointdsc":
            c = corrs.to(dev).to(torch.int32).contiguous()[None]
            pcd_a, pcd_q, n = ops.lift_pairs(c, None, (HO, WO), depth_a[None].contiguous(), depth_q[None].contiguous(),
                                             camera_a, camera_q)
            m = int(n.item())
            pose4 = get_pointdsc_pose(self.pointdsc_solver, pcd_a[0, :m], pcd_q[0, :m], self.device)
        else:
            raise RuntimeError(f"Solver {self.args.test.solver} not implemented")
        return pose4.to(torch.float32)

    # ------------------------------------------------------------------ pipeline.py:490-497
    def add_pred_pose(self, id_a: str, id_q: str, mask_a_iou, mask_q_iou, pred_pose: np.ndarray) -> str:
        pose_txt = " ".join([str(n) for n in pred_pose[:3, :].flatten()])
        line = ",".join([id_a, id_q, pose_txt, str(mask_a_iou), str(mask_q_iou)]) + "\n"
        self.pred_lines.append(line)
        return line

    # ------------------------------------------------------------------ pipeline.py:311 -> losses.py:64-88,196-199,250
    def feature_loss_rng_draws(self, batch: Dict, outputs: Dict) -> int:
        """The reference's test_step calls `self.feature_loss.forward(batch, outputs)` before the per-sample loop (pipeline.py:311) - for
        its mask post-processing, but the call also samples contrastive negatives and thereby CONSUMES random numbers ahead of the
        matcher's two draws.  The loss values are not part of the path; the generator state is: with `seed: 1` the matcher's samples
        depend on it.  This replays exactly those draws (same generator, same order, same arguments) and returns how many were made:
          loss.hard_negatives (config.yaml:43, default True): per side (anchor, then query) and per pair with batch['valid'] == 1,
              `torch_sample_select(featmap_i, 2000)` when FH*FW > 2000 (losses.py:196-199) = multinomial(ones(HW, float64), 2000, False)
              on the FEATURE MAP's device - the CUDA generator in a GPU run, i.e. it only interacts with the matcher's draws
              (CPU generator, `corrs_device: cpu`) when everything runs on one device;
          otherwise `torch.randint(0, HW, (n_gt_corrs,))` on the CPU generator (losses.py:250).
        Batches without the loader's 'valid' / 'corrs' entries (synthetic drivers) draw nothing."""
        if "valid" not in batch or "corrs" not in batch:
            return 0
        fm = outputs["featmap_a"]
        HW = fm.shape[2] * fm.shape[3]
        hard = bool(getattr(getattr(self.args, "loss", None), "hard_negatives", True))
        valid = batch["valid"]
        n = 0
        for _side in ("a", "q"):
            for i_b in range(fm.shape[0]):
                if valid[i_b] == 1:
                    if hard:
                        if HW > 2000:
                            torch.multinomial(torch.ones(HW, dtype=float).to(fm.device), 2000, replacement=False)
                            n += 1
                    else:
                        torch.randint(0, HW, (batch["corrs"].shape[1],))
                        n += 1
        return n

    # ------------------------------------------------------------------ pipeline.py:306-355
    def test_step(self, batch: Dict, batch_idx: int = 0) -> List[Dict]:
        outputs = self.model.forward(batch)
        BS = outputs["featmap_a"].shape[0]
        self.feature_loss_rng_draws(batch, outputs)          # generator state as after pipeline.py:311
        # the reference evaluates the predicted masks (and logs their IoU) whatever test.mask says (pipeline.py:311,352-354);
        # only the masks fed to the matcher switch to the external ones
        if "mask_a" in outputs and "mask_q" in outputs:
            results = self.mask_results(batch, outputs)
        elif self.args.test.mask == "predicted":
            raise KeyError("test.mask='predicted' needs the model's mask logits (outputs['mask_a'], outputs['mask_q'])")
        else:
            results = {"iou_a": torch.ones(BS), "iou_q": torch.ones(BS)}
        records = []
        for i_b in range(BS):
            id_a, id_q = batch["anchor"]["instance_id"][i_b], batch["query"]["instance_id"][i_b]
            status, pred_q = PAIR_OK, None
            if self.is_detection_valid(results, batch, i_b):
                pred_corrs, _, _ = self.get_featmap_corrs(batch, outputs, results, idx=i_b)
                if pred_corrs is not None:
                    pred_pose = self.get_pose(batch, pred_corrs, idx=i_b)
                    pred_q = pred_pose @ batch["anchor"]["pose"][i_b].cpu().detach().to(torch.float32)
                else:
                    status, pred_pose = PAIR_NO_CORR, torch.eye(4)
            else:
                status, pred_pose = PAIR_NO_MASK, torch.eye(4)
            if status != PAIR_OK:
                self.failures.append(batch["instance_id"][i_b] if "instance_id" in batch else id_q)
            iou_a = results["iou_a"][i_b].cpu().numpy()
            iou_q = results["iou_q"][i_b].cpu().numpy()
            self.add_pred_pose(id_a, id_q, iou_a, iou_q, pred_pose.cpu().numpy())
            records.append(dict(status=status, pred_pose_rel=pred_pose, pred_pose=pred_q))
        return records

    # ------------------------------------------------------------------ batched device path
    def test_step_batched(self, batch: Dict, first_pair_index: int = 0) -> Dict[str, Tensor]:
        """Same stages for the whole batch in ~25 launches and no host round trip.  orig_depth entries must share
        one size per side (NOCS / TOYL: 480x640).  Returns pose_rel [B,4,4], pred_q [B,4,4], status [B]."""
        dev = _lib.require_gpu(self.device)
        outputs = self.model.forward(batch)
        BS = outputs["featmap_a"].shape[0]
        FH, FW = outputs["featmap_a"].shape[2:]
        res = None
        if "mask_a" in outputs and "mask_q" in outputs:          # the predicted masks are evaluated (IoU) whatever test.mask says
            res = self.mask_results(batch, outputs)
        if self.args.test.mask == "predicted":
            if res is None:
                raise KeyError("test.mask='predicted' needs the model's mask logits (outputs['mask_a'], outputs['mask_q'])")
            mask_a, mask_q = res["mask_a"], res["mask_q"]
        else:
            mask_a = ops.mask_resize_nearest(batch["anchor"]["mask"].to(dev), (FH, FW))
            mask_q = ops.mask_resize_nearest(batch["query"]["mask"].to(dev), (FH, FW))
        if self._engine is None:
            self._engine = MatchPoseEngine(self.pointdsc_solver, MatchPoseConfig(
                dist_th=self.args.test.dist_th, n_corrs=self.args.test.n_corrs, src_sampling=self.args.test.src_sampling,
                seed=self.args.seed if self.args.seed is not None else 1))

        def stack(x):
            return (torch.stack([d.squeeze() for d in x]) if isinstance(x, (list, tuple)) else x).to(dev, torch.float32).contiguous()
        depth_a, depth_q = stack(batch["anchor"]["orig_depth"]), stack(batch["query"]["orig_depth"])
        key = torch.arange(first_pair_index, first_pair_index + BS, dtype=torch.int64, device=dev)
        out = self._engine.run(outputs["featmap_a"].float().contiguous(), outputs["featmap_q"].float().contiguous(), mask_a,
                               mask_q, depth_a, depth_q, batch["anchor"]["camera"].to(dev), batch["query"]["camera"].to(dev), key)
        anchor_pose = batch["anchor"]["pose"].to(dev, torch.float32)
        out["pred_q"] = torch.bmm(out["pose"], anchor_pose)
        if res is not None:
            out["iou_a"], out["iou_q"] = res["iou_a"], res["iou_q"]
        return out
