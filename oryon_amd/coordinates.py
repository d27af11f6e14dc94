"""utils/coordinates.py counterparts used between the matcher and the lift (scale_coords :5-13,
get_valid_coords :36-48).  On the product path these are fused into K2 (oryon_lift_pairs); the
standalone functions exist for callers that use them directly and are plain tensor plumbing."""
from typing import Tuple, Union

import torch
from torch import Tensor


def scale_coords(coords: Tensor, source_scale: Union[Tensor, Tuple], target_scale: Union[Tensor, Tuple]) -> Tensor:
    """(y,x) coordinates scaled by target/source in fp32; returns a copy."""
    new_coords = coords.clone().to(torch.float32)
    sy = torch.as_tensor(target_scale[0], dtype=torch.float32) / torch.as_tensor(source_scale[0], dtype=torch.float32)
    sx = torch.as_tensor(target_scale[1], dtype=torch.float32) / torch.as_tensor(source_scale[1], dtype=torch.float32)
    new_coords[:, 0] = new_coords[:, 0] * sy.to(new_coords.device)
    new_coords[:, 1] = new_coords[:, 1] * sx.to(new_coords.device)
    return new_coords


def get_valid_coords(coords: Tensor, bounds: Union[Tensor, Tuple]) -> Tensor:
    """Boolean mask of (y,x) rows inside [0,bounds)."""
    ys, xs = coords[:, 0], coords[:, 1]
    return (ys >= 0) & (ys < float(bounds[0])) & (xs >= 0) & (xs < float(bounds[1]))
