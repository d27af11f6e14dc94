"""The step in front of the hot path: raw sample dicts -> the batch dict `Pipeline.test_step` consumes.

Reference flow per sample, on dataloader workers (datasets.py:467-498):
    nocs/toyl.get_item_data (PIL decode)  ->  common.preprocess_item (utils/data/common.py:41-75)
    ->  augmentations.resize(dataset.img_size) (utils/augmentations.py:129-164)  ->  CollateWrapper (datasets.py:138-245)

Here `preprocess_item` keeps the host bookkeeping (mask-id selection, box, sizes, pose tensors) and `DeviceCollate` does the
pixel work for the whole batch on the GPU (K-1 kernels of liboryon_hip.so): uint8 HWC rgb -> /255 -> CHW -> bilinear 224x224,
nearest mask, bilinear depth, while `orig_depth` (what the 2D->3D lift reads) stays at sensor resolution.  The returned dict
has the keys / shapes / dtypes of the reference's collate (Appendix B of SURVEY.md); tensors that the network or the matcher
read live on the device, bookkeeping stays on the host.  No CPU pixel path exists in this module: without the HIP library
`DeviceCollate` raises.  PNG decoding itself (PIL) and the dataset index files are out of scope.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch
from torch import Tensor

from . import _lib, ops
from .coordinates import scale_coords


def box_from_mask(mask: np.ndarray, id: int) -> Tuple[int, int, int, int]:
    """[y1,x1,y2,x2] of the pixels equal to `id`; (0,0,2,2) when there are none (utils/misc.py:216-227)."""
    ys, xs = np.nonzero(mask == id)
    if ys.shape[0] > 0:
        return int(ys.min()), int(xs.min()), int(ys.max()), int(xs.max())
    return 0, 0, 2, 2


def get_mask_type(mask: str, eval: bool) -> str:
    """datasets.py:27-46: training always reads the oracle mask; at evaluation 'predicted' still loads the oracle mask (as GT)."""
    if eval:
        return "oracle" if mask == "predicted" else mask
    return "oracle"


def preprocess_item(item: dict) -> dict:
    """utils/data/common.py:41-75 without the pixel arithmetic: arrays become tensors, the instance mask becomes {0,1}, the box
    is taken from it, `orig_depth` / `eval_depth` keep the sensor-resolution depth.  `rgb` stays uint8 [H,W,3] - the
    `/255.`, CHW transpose and resize run on the device in `DeviceCollate` (the reference does them here, in float64)."""
    assert len(item["metadata"]["mask_ids"]) == 1, f" Problem with instance {item['instance_id']}: no objects found. Check cls_id!"
    item["hw_size"] = tuple(item["mask"].shape)
    for k, v in list(item.items()):
        if isinstance(v, np.ndarray):
            item[k] = torch.from_numpy(np.array(v, copy=True))          # PIL hands out read-only buffers
    item["orig_rgb"] = item["rgb"]
    item["orig_depth"] = item["depth"].clone()
    item["eval_depth"] = item["depth"].clone()
    if "poses" in item["metadata"]:
        item["metadata"]["poses"] = [torch.as_tensor(v) for v in item["metadata"]["poses"]]
    mask_id = item["metadata"]["mask_ids"][0]
    mask = torch.where(item["mask"] == mask_id, 1, 0)
    item["mask"] = mask
    y1, x1, y2, x2 = box_from_mask(mask.numpy(), id=1)
    item["metadata"]["boxes"] = torch.tensor([y1, x1, y2 - y1, x2 - x1])
    return item


def check_validity(item: dict) -> bool:
    """utils/data/common.py:104-110."""
    return int(torch.count_nonzero(item["mask"]).item()) > 0


def resize_annotations(item: dict, coords: Tensor, size: Sequence[int]) -> Tuple[Tensor, Tensor]:
    """The non-pixel half of augmentations.resize.resize_item (utils/augmentations.py:142-147): the box scaled by the
    size ratios and the GT correspondences' (y,x) scaled in fp32.  Returns (box [4] fp32-or-fp64 like the reference, coords)."""
    H, W = item["mask"].shape[-2:]
    y1, x1, h, w = item["metadata"]["boxes"]
    h_ratio, w_ratio = size[0] / float(H), size[1] / float(W)
    box = torch.tensor([y1 * h_ratio, x1 * w_ratio, h * h_ratio, w * w_ratio])
    return box, scale_coords(coords, (H, W), size)


class DeviceCollate:
    """`CollateWrapper(corr_n)` ∘ `resize(img_size)` for items that went through `preprocess_item` (datasets.py:138-245,
    utils/augmentations.py:129-164).  Call with the list of dataset tuples
        (item_a, item_q, prompt, sampled_corrs, all_corrs, pose, cls_id, instance_id, valid)
    and get the reference's batch dict.  Images of one side must share one sensor size (NOCS / TOYL: 480x640)."""

    def __init__(self, corr_n: int, img_size: Sequence[int] = (224, 224), device: str = "cuda"):
        self.max_corrs = int(corr_n)
        self.size = (int(img_size[0]), int(img_size[1]))
        self.device = device

    def _side(self, items: List[dict]) -> Dict:
        dev = _lib.require_gpu(self.device)
        rgb_u8 = torch.stack([it["rgb"] for it in items]).to(dev, non_blocking=True)                   # [B,H,W,3] uint8
        depth_raw = torch.stack([it["depth"] for it in items])
        integer_depth = not depth_raw.dtype.is_floating_point
        depth_dev = depth_raw.to(torch.float32).to(dev, non_blocking=True)                              # [B,H,W]
        mask_dev = torch.stack([it["mask"] for it in items]).to(torch.uint8).to(dev, non_blocking=True)
        boxes, sizes = [], []
        for it in items:
            box, _ = resize_annotations(it, torch.zeros((0, 2)), self.size)
            boxes.append(box.squeeze())
            s = it["hw_size"]
            sizes.append(s if isinstance(s, Tensor) else torch.tensor(s))
        orig_depth = [depth_dev[i] for i in range(len(items))]
        return {
            "rgb": ops.rgb_resize_bilinear(rgb_u8, self.size),
            "orig_rgb": [it["orig_rgb"] for it in items],
            "mask": ops.mask_resize_nearest(mask_dev, self.size).to(torch.uint8),
            "orig_depth": orig_depth,
            "eval_depth": orig_depth,
            "depth": ops.resize_bilinear(depth_dev, self.size, round_output=integer_depth),
            "camera": torch.stack([torch.as_tensor(it["camera"]).squeeze() for it in items]),
            "pose": torch.stack([it["metadata"]["poses"][0].squeeze() for it in items]),
            "box": torch.stack(boxes),
            "sizes": torch.stack(sizes),
            "instance_id": [it["instance_id"] for it in items],
        }

    def __call__(self, data: Sequence[Tuple]) -> dict:
        items_a, items_q = [d[0] for d in data], [d[1] for d in data]
        corr_list, all_corr_list, poses, valids = [], [], [], []
        for item_a, item_q, prompt, sampled_corrs, all_corrs, pose, cls_id, instance_id, valid in data:
            if valid and sampled_corrs.shape[0] > 0:
                valids.append(1.0)
                _, ca = resize_annotations(item_a, sampled_corrs[:, :2], self.size)
                _, cq = resize_annotations(item_q, sampled_corrs[:, 2:], self.size)
                sampled_corrs = torch.cat([ca, cq], dim=1)
            else:
                valids.append(0.0)
                sampled_corrs = torch.zeros((self.max_corrs, 4)).to(torch.long)
                all_corrs = torch.zeros((self.max_corrs, 4)).to(torch.long)
            corr_list.append(sampled_corrs)
            all_corr_list.append(all_corrs)
            if pose is not None:
                poses.append(pose)
        batch = {
            "anchor": self._side(items_a),
            "query": self._side(items_q),
            "corrs": torch.stack(corr_list, dim=0).to(torch.long) if len({c.shape for c in corr_list}) == 1 else corr_list,
            "all_corrs": all_corr_list,
            "prompt": [d[2] for d in data],
            "valid": torch.tensor(valids),
            "instance_id": [d[7] for d in data],
            "cls_id": [d[6] for d in data],
        }
        if len(poses) > 0:
            batch["pose"] = torch.tensor(np.stack(poses, axis=0))
        return batch


def make_raw_item(index: int, H: int = 480, W: int = 640, mask_id: int = 3) -> dict:
    """A synthetic sample shaped like nocs.get_item_data's return value (utils/data/nocs.py:228-278): uint8 rgb [H,W,3], the
    instance-id mask [H,W] uint8 (255 = background), an integer depth map in millimetres, metadata and the NOCS intrinsics."""
    rng = np.random.default_rng(1000 + index)
    rgb = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
    mask = np.full((H, W), 255, dtype=np.uint8)
    y0, x0 = int(rng.integers(20, H // 3)), int(rng.integers(20, W // 3))
    hh, ww = int(rng.integers(H // 6, H // 2)), int(rng.integers(W // 6, W // 2))
    mask[y0:y0 + hh, x0:x0 + ww] = mask_id
    mask[y0 + hh // 3:y0 + hh // 2, x0 + ww // 3:x0 + ww // 2] = 1            # another instance occluding part of it
    yy, xx = np.mgrid[0:H, 0:W]
    depth = (800 + 60 * np.sin(xx / 9.0) + 40 * np.cos(yy / 7.0)).astype(np.int32)
    depth[rng.random((H, W)) < 0.02] = 0
    pose = np.eye(4)
    pose[:3, 3] = rng.uniform(-0.2, 0.2, 3)
    return {
        "rgb": rgb, "mask": mask, "depth": depth, "instance_id": f"1 {index} synthetic_obj",
        "metadata": {"mask_ids": [mask_id], "cls_ids": [1], "cls_names": ["mug"], "cls_descs": [["white", "black"]],
                     "boxes": [[0, 0, 2, 2]], "poses": [pose]},
        "camera": np.asarray([[591.0125, 0, 322.525], [0, 590.16775, 244.11084], [0, 0, 1]]),
    }
