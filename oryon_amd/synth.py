"""Synthetic rigid RGB-D pair generator (descriptors given), used by bench.py, the tests and the
golden generator.  It follows the recipe the reference uses to build ground-truth correspondences
for its fixed test splits (scripts/data/make_nocs_test.py:223-234: lift anchor depth, transform,
re-project into the query view), but on closed-form inputs so nothing has to be shipped:

  anchor depth   D_a(y,x) = 800 + 60 sin(x/9) + 40 cos(y/7)            [mm]
  intrinsics     fx = fy = 1.25 W, cx = W/2, cy = H/2
  motion         rotation about the anchor cloud's centroid, axis uniform on S^2, angle U(0,20 deg),
                 translation U(-50,50)^3 mm
  descriptors    F_a ~ N(0,1);  F_q ~ N(0,1) then F_q[:, v, u] = F_a[:, y, x] + 0.05 N(0,1) at the
                 re-projected pixel (u,v) of every anchor pixel (last writer in row-major order wins)
  query depth    D_q[v,u] = z_q of the winning anchor pixel, 0 elsewhere
  masks          mask_a = centred square of side H/2 ; mask_q = D_q > 0

Everything is deterministic given (index, H, W, C) on a given device type.
"""
from __future__ import annotations

import math
from typing import Dict

import torch


def intrinsics(H: int, W: int) -> torch.Tensor:
    K = torch.zeros(3, 3, dtype=torch.float64)
    K[0, 0] = K[1, 1] = 1.25 * W
    K[0, 2] = W / 2.0
    K[1, 2] = H / 2.0
    K[2, 2] = 1.0
    return K


def _rotation(axis: torch.Tensor, angle: float) -> torch.Tensor:
    a = axis / axis.norm()
    Kx = torch.tensor([[0.0, -a[2], a[1]], [a[2], 0.0, -a[0]], [-a[1], a[0], 0.0]], dtype=torch.float64)
    return torch.eye(3, dtype=torch.float64) + math.sin(angle) * Kx + (1 - math.cos(angle)) * (Kx @ Kx)


def smooth_field(index: int, H: int, W: int, C: int, dev: torch.device, salt: int = 0) -> torch.Tensor:
    """[C, H*W] rank-8 descriptor field (constant, x, y, xy and four slow sinusoids with random per-channel coefficients): neighbouring
    pixels are nearly parallel, so every pixel has many near-ties in cosine - what a decoder's smooth output looks like and what no
    6- or 8-bit screen can separate (the `hard_descriptors` workload of bench.py uses the same basis)."""
    g = torch.Generator(device=dev)
    g.manual_seed(9000 + 2 * index + salt)
    yy, xx = torch.meshgrid(torch.linspace(0, 1, H, device=dev), torch.linspace(0, 1, W, device=dev), indexing="ij")
    coef = torch.stack([torch.ones_like(xx), xx, yy, xx * yy, torch.sin(3 * xx), torch.cos(3 * yy), torch.sin(7 * yy), torch.cos(5 * xx)])
    basis = torch.randn((C, coef.shape[0]), generator=g, device=dev)
    return (basis @ coef.reshape(coef.shape[0], H * W)).contiguous()


def make_pair(index: int, H: int, W: int, C: int, device: str = "cpu", noise: float = 0.05,
              feat_dtype: torch.dtype = torch.float32, smooth: float = 0.0) -> Dict[str, torch.Tensor]:
    """One synthetic pair.  Returns feat_a/feat_q [C,H,W], mask_a/mask_q [H,W] int32,
    depth_a/depth_q [H,W] fp32 (mm), camera [3,3] fp64, sizes (H,W), pose [4,4] fp64 (metres,
    maps anchor-camera points to query-camera points).
    smooth > 0: the Gaussian descriptors become `smooth` x N(0,1) noise on top of rank-8 smooth fields (one for the anchor map, another
    for the query background); the re-projected query pixels still carry their anchor pixel's descriptor + `noise` x N(0,1), so the
    geometry (and the ground-truth pose) is unchanged while every anchor now has a crowd of near-ties around its true match."""
    dev = torch.device(device)
    g = torch.Generator(device="cpu")
    g.manual_seed(1000 + index)
    axis = torch.randn(3, generator=g, dtype=torch.float64)
    angle = float(torch.rand(1, generator=g, dtype=torch.float64)) * math.radians(20.0)
    t = (torch.rand(3, generator=g, dtype=torch.float64) * 100.0 - 50.0)
    R = _rotation(axis, angle)

    ys = torch.arange(H, dtype=torch.float64)
    xs = torch.arange(W, dtype=torch.float64)
    D_a = 800.0 + 60.0 * torch.sin(xs / 9.0)[None, :] + 40.0 * torch.cos(ys / 7.0)[:, None]
    K = intrinsics(H, W)
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    X = (xs[None, :] - cx) * D_a / fx
    Y = (ys[:, None] - cy) * D_a / fy
    P_a = torch.stack((X, Y, D_a), dim=-1).reshape(-1, 3)
    c = P_a.mean(0)
    P_q = (P_a - c) @ R.T + c + t
    u = torch.round(P_q[:, 0] * fx / P_q[:, 2] + cx).long()
    v = torch.round(P_q[:, 1] * fy / P_q[:, 2] + cy).long()
    inside = (u >= 0) & (u < W) & (v >= 0) & (v < H) & (P_q[:, 2] > 0)
    src = torch.arange(H * W)[inside]
    tgt = (v * W + u)[inside]
    # last writer (largest row-major source index) wins, deterministically
    winner = torch.full((H * W,), -1, dtype=torch.long)
    winner.scatter_reduce_(0, tgt, src, reduce="amax", include_self=True)
    hit = winner >= 0
    D_q = torch.zeros(H * W, dtype=torch.float64)
    D_q[hit] = P_q[winner[hit], 2]

    gd = torch.Generator(device=dev)
    gd.manual_seed(7000 + index)
    feat_a = torch.randn(C, H * W, generator=gd, device=dev, dtype=torch.float32)
    feat_q = torch.randn(C, H * W, generator=gd, device=dev, dtype=torch.float32)
    pert = torch.randn(C, int(hit.sum()), generator=gd, device=dev, dtype=torch.float32) * noise
    if smooth > 0.0:
        feat_a = smooth_field(index, H, W, C, dev, 0) + smooth * feat_a
        feat_q = smooth_field(index, H, W, C, dev, 1) + smooth * feat_q
    hit_d = hit.to(dev)
    win_d = winner[hit].to(dev)
    feat_q[:, hit_d] = feat_a[:, win_d] + pert

    mask_a = torch.zeros(H, W, dtype=torch.int32)
    h0, w0 = H // 4, W // 4
    mask_a[h0:h0 + H // 2, w0:w0 + W // 2] = 1
    mask_q = (D_q > 0).reshape(H, W).to(torch.int32)

    pose = torch.eye(4, dtype=torch.float64)
    pose[:3, :3] = R
    pose[:3, 3] = (c - R @ c + t) / 1000.0
    return dict(
        feat_a=feat_a.reshape(C, H, W).to(feat_dtype), feat_q=feat_q.reshape(C, H, W).to(feat_dtype),
        mask_a=mask_a.to(dev), mask_q=mask_q.to(dev),
        depth_a=D_a.to(torch.float32).to(dev), depth_q=D_q.reshape(H, W).to(torch.float32).to(dev),
        camera=K, sizes=(H, W), pose=pose,
    )


def make_batch(first_index: int, B: int, H: int, W: int, C: int, device: str = "cpu") -> Dict[str, object]:
    """B pairs stacked: feat_* [B,C,H,W], mask_* [B,H,W], depth_* [B,H,W], camera [B,3,3], pose [B,4,4]."""
    items = [make_pair(first_index + i, H, W, C, device) for i in range(B)]
    out: Dict[str, object] = {}
    for k in ("feat_a", "feat_q", "mask_a", "mask_q", "depth_a", "depth_q", "camera", "pose"):
        out[k] = torch.stack([it[k] for it in items])
    out["sizes"] = (H, W)
    return out
