"""Truncated Swin-B guidance backbone of the reference (net.py:45-75), restated on plain PyTorch.

The reference takes torchvision's `swin_b(Swin_B_Weights.DEFAULT)` and cuts it with `create_feature_extractor` at
    features.1.1.add_1   -> guidance3 [B, 96, 96, 128]   (output of stage 1)
    features.2.reduction -> guidance2 [B, 48, 48, 256]   (output of the first patch merging)
    features.4.reduction -> guidance1 [B, 24, 24, 512]   (output of the second patch merging)
so only patch-embed, stage 1 (2 blocks, 4 heads), merge, stage 2 (2 blocks, 8 heads), merge are executed.
torchvision is not part of the reference tree nor installed here; this file restates the published Swin algorithm
(window 7, shift 3, relative position bias, -100 masks, zero padding to a multiple of the window, 2x2 patch merging
in the order (0,0),(1,0),(0,1),(1,1)) under torchvision's parameter names so `guidance_backbone.features.*` entries of
a reference checkpoint load unchanged.  PARITY UNPINNED (third-party arithmetic, no reference tests).
"""
from __future__ import annotations

from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor

from .clip import PatchEmbed

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


# fp32 inference switch for the whole guidance backbone: shifted-window attention as ONE HIP kernel
# (oryon_swin_window_attention_f32) instead of torch's dozen bandwidth-bound passes, residual add + LayerNorm in one pass
# (oryon_add_layernorm_f32), and the linears whose shapes allow it (N % 256 == 0: stages 2 and 3, the patch mergings) on the
# error-compensated fp16x3 kernel with the erf-GELU fused (B4).  Results stay fp32-grade: within ~2e-5 of the plain path
# (tests/test_backbone_pins.py).
FUSED_F32_ATTENTION = False


def _lin(m: nn.Linear, x: Tensor, gelu: bool = False) -> Tensor:
    """nn.Linear (+ erf-GELU) of the fp32 inference path: the fp16x3 kernel (B4) where its shape constraints hold, torch otherwise."""
    if FUSED_F32_ATTENTION and x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled():
        from .. import ops
        if ops.linear_f16x3_supported(x, m.weight):
            return ops.linear_f16x3(x, m.weight, m.bias, gelu=gelu)
    y = m(x)
    return F.gelu(y) if gelu else y


def _relative_index(w: int) -> Tensor:
    ys, xs = torch.meshgrid(torch.arange(w), torch.arange(w), indexing="ij")
    pos = torch.stack((ys.reshape(-1), xs.reshape(-1)))                 # [2, w*w]
    rel = pos[:, :, None] - pos[:, None, :] + (w - 1)                    # [2, N, N] in 0..2w-2
    return (rel[0] * (2 * w - 1) + rel[1]).reshape(-1)


class _WindowAttention(nn.Module):
    def __init__(self, dim: int, heads: int, window: int, shift: int):
        super().__init__()
        self.dim, self.heads, self.window, self.shift = dim, heads, window, shift
        self.qkv = nn.Linear(dim, 3 * dim)
        self.proj = nn.Linear(dim, dim)
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * window - 1) ** 2, heads))
        self.register_buffer("relative_position_index", _relative_index(window))
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)

    def _shift_mask(self, Hp: int, Wp: int, device, dtype) -> Tensor:
        w, s = self.window, self.shift
        region = torch.zeros((Hp, Wp), device=device, dtype=torch.float32)
        bands = ((0, Hp - w), (Hp - w, Hp - s), (Hp - s, Hp))
        bands_w = ((0, Wp - w), (Wp - w, Wp - s), (Wp - s, Wp))
        label = 0
        for h0, h1 in bands:
            for w0, w1 in bands_w:
                region[h0:h1, w0:w1] = label
                label += 1
        region = region.view(Hp // w, w, Wp // w, w).permute(0, 2, 1, 3).reshape(-1, w * w)   # [nW, N]
        diff = region[:, None, :] - region[:, :, None]
        return torch.where(diff != 0, torch.full_like(diff, -100.0), torch.zeros_like(diff)).to(dtype)   # [nW, N, N]

    def forward(self, x: Tensor) -> Tensor:                               # x: [B, H, W, C]
        B, H, W, C = x.shape
        w, s, nh = self.window, self.shift, self.heads
        if _fast_ln(x) and w == 7 and C == 32 * nh and nh <= 8 and self.qkv.weight.dtype == torch.bfloat16:
            from .. import ops                      # bf16 inference: one HIP kernel between the q|k|v Linear and the projection (B3)
            bias_t = self.relative_position_bias_table[self.relative_position_index].view(w * w, w * w, nh).permute(2, 1, 0)
            pad = self.qkv.bias if self.qkv.bias is not None else torch.zeros(3 * C, dtype=x.dtype, device=x.device)
            out = ops.swin_window_attention_bf16(self.qkv(x), pad, bias_t.float().contiguous(), nh, s)
            return self.proj(out)
        if FUSED_F32_ATTENTION and x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled() and w == 7 and C == 32 * nh and nh <= 8:
            from .. import ops                      # fp32 inference: the same single kernel on fp32 tensors
            bias_t = self.relative_position_bias_table[self.relative_position_index].view(w * w, w * w, nh).permute(2, 1, 0)
            pad = self.qkv.bias if self.qkv.bias is not None else torch.zeros(3 * C, dtype=x.dtype, device=x.device)
            out = ops.swin_window_attention_f32(_lin(self.qkv, x), pad.detach(), bias_t.detach().contiguous(), nh, s)
            return _lin(self.proj, out)
        pb, pr = (w - H % w) % w, (w - W % w) % w
        x = F.pad(x, (0, 0, 0, pr, 0, pb))
        Hp, Wp = H + pb, W + pr
        if s > 0:
            x = torch.roll(x, shifts=(-s, -s), dims=(1, 2))
        nW = (Hp // w) * (Wp // w)
        win = x.view(B, Hp // w, w, Wp // w, w, C).permute(0, 1, 3, 2, 4, 5).reshape(B * nW, w * w, C)
        qkv = self.qkv(win).view(B * nW, w * w, 3, nh, C // nh).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0] * (C // nh) ** -0.5, qkv[1], qkv[2]
        attn = q @ k.transpose(-2, -1)
        bias = self.relative_position_bias_table[self.relative_position_index].view(w * w, w * w, nh).permute(2, 0, 1)
        attn = attn + bias[None]
        if s > 0:
            m = self._shift_mask(Hp, Wp, x.device, attn.dtype)
            attn = (attn.view(B, nW, nh, w * w, w * w) + m[None, :, None]).view(B * nW, nh, w * w, w * w)
        out = (torch.softmax(attn, dim=-1) @ v).transpose(1, 2).reshape(B * nW, w * w, C)
        out = self.proj(out)
        out = out.view(B, Hp // w, Wp // w, w, w, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, C)
        if s > 0:
            out = torch.roll(out, shifts=(s, s), dims=(1, 2))
        return out[:, :H, :W, :].contiguous()


class _SwinBlock(nn.Module):
    def __init__(self, dim: int, heads: int, window: int, shift: int):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn = _WindowAttention(dim, heads, window, shift)
        self.norm2 = nn.LayerNorm(dim)
        # torchvision's MLP is Sequential(Linear, GELU, Dropout, Linear, Dropout): parameter slots 0 and 3
        self.mlp = nn.Sequential(nn.Linear(dim, 4 * dim), nn.GELU(), nn.Identity(), nn.Linear(4 * dim, dim), nn.Identity())

    def forward(self, x: Tensor) -> Tensor:
        if _fast_ln(x) and self.norm1.weight.dtype == x.dtype:
            from .. import ops                      # inference: LayerNorm / residual add + LayerNorm in one pass (B2)
            h = ops.add_layernorm(x, None, self.norm1.weight, self.norm1.bias, self.norm1.eps)[1]
            x, h = ops.add_layernorm(x, self.attn(h), self.norm2.weight, self.norm2.bias, self.norm2.eps)
            if x.dtype == torch.float32:
                return x + _lin(self.mlp[3], _lin(self.mlp[0], h, gelu=True))
            return x + self.mlp(h)
        x = x + self.attn(self.norm1(x))
        return x + self.mlp(self.norm2(x))


def _fast_ln(x: Tensor) -> bool:
    if not (x.is_cuda and not torch.is_grad_enabled() and x.shape[-1] % 8 == 0):
        return False
    return x.dtype == torch.bfloat16 or (FUSED_F32_ATTENTION and x.dtype == torch.float32 and x.shape[-1] <= 2048)


class _FastLayerNorm(nn.LayerNorm):
    """nn.LayerNorm whose bf16 CUDA inference path is the one-pass HIP kernel (torch's runs at ~1.3 TB/s on these row widths)."""

    def forward(self, x: Tensor) -> Tensor:
        if _fast_ln(x) and self.weight.dtype == x.dtype:
            from .. import ops
            return ops.add_layernorm(x, None, self.weight, self.bias, self.eps)[1]
        return super().forward(x)


class _PatchMerging(nn.Module):
    def __init__(self, dim: int):
        super().__init__()
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = _FastLayerNorm(4 * dim)

    def forward(self, x: Tensor) -> Tensor:                               # [B, H, W, C] -> [B, H/2, W/2, 2C]
        H, W = x.shape[1], x.shape[2]
        x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
        x = torch.cat((x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]), dim=-1)
        return _lin(self.reduction, self.norm(x))


class _PatchEmbedNHWC(PatchEmbed):
    def forward(self, x: Tensor) -> Tensor:                               # [B, 3, H, W] -> [B, H/4, W/4, C]
        return self.tokens(x)


class SwinGuidance(nn.Module):
    """features.0 .. features.4 of torchvision's swin_b, returning the three node outputs the reference extracts."""

    def __init__(self, embed: int = 128, window: int = 7, heads=(4, 8)):
        super().__init__()
        self.features = nn.Sequential(
            nn.Sequential(_PatchEmbedNHWC(3, embed, kernel_size=4, stride=4), nn.Identity(), _FastLayerNorm(embed)),   # .1 was the NHWC permute
            nn.Sequential(_SwinBlock(embed, heads[0], window, 0), _SwinBlock(embed, heads[0], window, window // 2)),
            _PatchMerging(embed),
            nn.Sequential(_SwinBlock(2 * embed, heads[1], window, 0), _SwinBlock(2 * embed, heads[1], window, window // 2)),
            _PatchMerging(2 * embed),
        )

    def forward(self, img: Tensor) -> dict:
        f = self.features
        s1 = f[1](f[0](img))             # features.1.1.add_1
        m1 = f[2](s1)                    # features.2.reduction
        m2 = f[4](f[3](m1))              # features.4.reduction
        return {"guidance3": s1, "guidance2": m1, "guidance1": m2}


def guidance_embeds(backbone: SwinGuidance, img: Tensor) -> List[Tensor]:
    """net.py:60-75: bicubic resize to 384 (align_corners=True), ImageNet normalisation, three NCHW guidance maps
    [B,512,24,24], [B,256,48,48], [B,128,96,96]."""
    x = F.interpolate(img, size=(384, 384), mode="bicubic", align_corners=True)
    mean = torch.tensor(IMAGENET_MEAN, device=x.device, dtype=x.dtype).view(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD, device=x.device, dtype=x.dtype).view(1, 3, 1, 1)
    outs = backbone((x - mean) / std)
    return [outs[k].permute(0, 3, 1, 2) for k in ("guidance1", "guidance2", "guidance3")]
