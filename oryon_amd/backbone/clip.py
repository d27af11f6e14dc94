"""CLIP ViT towers as used by the reference's CLIPEncoder (models/vlm.py:14-86), restated on plain PyTorch
(PyTorch-ROCm GEMMs; the north_star keeps the backbone on the framework's BLAS path).

The reference builds its towers with the third-party `clip` package (`clip.load("ViT-L/14@336px")`, vlm.py:19),
which is not part of the reference tree; this module restates the published architecture with the SAME parameter
names, so `clip_model.*` entries of a reference / CATSeg checkpoint load unchanged:

    visual.conv1.weight, visual.class_embedding, visual.positional_embedding, visual.ln_pre.*, visual.ln_post.*,
    visual.proj, visual.transformer.resblocks.{i}.{ln_1,ln_2}.*, .attn.{in_proj_weight,in_proj_bias,out_proj.*},
    .mlp.{c_fc,c_proj}.*, token_embedding.weight, positional_embedding, transformer.resblocks.*, ln_final.*,
    text_projection, logit_scale

What the reference touches (vlm.py:43-86): patch tokens after ln_post WITHOUT visual.proj, reshaped to
[B, width, 24, 24]; text: token + positional embedding -> causal transformer -> ln_final -> EOT token -> @ text_projection.
Residual block: x + attn(ln_1 x), x + c_proj(QuickGELU(c_fc(ln_2 x))), QuickGELU(x) = x * sigmoid(1.702 x).
Parity: no reference test or golden vector exists for these towers (third-party arithmetic): PARITY UNPINNED; the
structure is cross-checked against the independent `transformers` CLIP implementation in tests/test_backbone.py.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor


@dataclass
class CLIPConfig:
    image_size: int = 336
    patch: int = 14
    v_width: int = 1024
    v_layers: int = 24
    v_heads: int = 16
    embed_dim: int = 768
    ctx: int = 77
    vocab: int = 49408
    t_width: int = 768
    t_layers: int = 12
    t_heads: int = 12

    @staticmethod
    def vit_l14_336() -> "CLIPConfig":
        return CLIPConfig()


class PatchEmbed(nn.Conv2d):
    """Non-overlapping patch embedding (kernel == stride, no padding) with nn.Conv2d's parameters and state-dict names, computed as ONE
    GEMM over the unfolded patches.  MIOpen runs such a convolution as a per-image im2col + small GEMM (2 x 64 launches per CLIP
    forward, 7.8 ms per 64 images on MI355X); the unfold here is a single strided copy and the GEMM has N*g*g rows."""

    def tokens(self, x: Tensor) -> Tensor:                           # [N, Cin, H, W] -> [N, H/p, W/p, Cout]
        p = self.kernel_size[0]
        assert self.kernel_size == self.stride and self.kernel_size[0] == self.kernel_size[1] and self.padding == (0, 0)
        N, Cin, H, W = x.shape
        gh, gw = H // p, W // p
        cols = x[:, :, : gh * p, : gw * p].reshape(N, Cin, gh, p, gw, p).permute(0, 2, 4, 1, 3, 5).reshape(N * gh * gw, Cin * p * p)
        out = F.linear(cols, self.weight.view(self.out_channels, -1), self.bias)
        return out.view(N, gh, gw, self.out_channels)

    def forward(self, x: Tensor) -> Tensor:                          # nn.Conv2d's layout, for callers that want NCHW
        return self.tokens(x).permute(0, 3, 1, 2)


# Inference switch: evaluate the towers' linear layers with the error-compensated fp16x3 kernel (ops.linear_f16x3, B4) instead of
# torch's fp32 GEMMs.  Results stay fp32-grade (descriptor maps within ~1e-5 of the fp32 evaluation, tests/test_backbone_pins.py).
FP16X3_LINEAR = False
ACC_RESIDUAL = True           # fp16x3 path: the blocks' last linears add into the residual stream themselves (same values)


def _linear(x: Tensor, weight: Tensor, bias: Optional[Tensor], quick_gelu: bool = False) -> Tensor:
    if FP16X3_LINEAR:
        from .. import ops
        if ops.linear_f16x3_supported(x, weight):
            return ops.linear_f16x3(x, weight, bias, quick_gelu=quick_gelu)
    y = F.linear(x, weight, bias)
    return y * torch.sigmoid(1.702 * y) if quick_gelu else y


class _Attention(nn.Module):
    """Packed-qkv multi-head attention with nn.MultiheadAttention's parameter names."""

    def __init__(self, width: int, heads: int):
        super().__init__()
        self.heads = heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * width, width))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * width))
        self.out_proj = nn.Linear(width, width)
        nn.init.normal_(self.in_proj_weight, std=width ** -0.5)

    def forward(self, x: Tensor, causal: bool) -> Tensor:            # x: [N, L, D]
        N, L, D = x.shape
        if (FP16X3_LINEAR and not causal and x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled()
                and D == 64 * self.heads):
            from .. import ops                      # B5: the whole attention between the two linears as one fp16x3 kernel
            o = ops.mha_f16x3(_linear(x, self.in_proj_weight, self.in_proj_bias), self.heads)
            return _linear(o, self.out_proj.weight, self.out_proj.bias)
        qkv = _linear(x, self.in_proj_weight, self.in_proj_bias).view(N, L, 3, self.heads, D // self.heads)
        q, k, v = qkv.permute(2, 0, 3, 1, 4)                          # [N, H, L, d] each
        o = F.scaled_dot_product_attention(q, k, v, is_causal=causal)
        return _linear(o.transpose(1, 2).reshape(N, L, D), self.out_proj.weight, self.out_proj.bias)


class _MLP(nn.Module):
    def __init__(self, width: int):
        super().__init__()
        self.c_fc = nn.Linear(width, 4 * width)
        self.c_proj = nn.Linear(4 * width, width)

    def forward(self, x: Tensor) -> Tensor:
        if FP16X3_LINEAR and x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled():
            return self.forward_x3(x)
        h = self.c_fc(x)
        if h.is_cuda and h.dtype == torch.bfloat16 and not torch.is_grad_enabled():
            from .. import ops                      # fused QuickGELU (B1): one pass instead of three bandwidth-bound ones
            return self.c_proj(ops.quick_gelu_bf16(h))
        return self.c_proj(h * torch.sigmoid(1.702 * h))

    def forward_x3(self, x: Tensor) -> Tensor:
        return _linear(_linear(x, self.c_fc.weight, self.c_fc.bias, quick_gelu=True), self.c_proj.weight, self.c_proj.bias)


class _Block(nn.Module):
    def __init__(self, width: int, heads: int, causal: bool):
        super().__init__()
        self.causal = causal
        self.ln_1 = nn.LayerNorm(width)
        self.attn = _Attention(width, heads)
        self.ln_2 = nn.LayerNorm(width)
        self.mlp = _MLP(width)

    def forward(self, x: Tensor) -> Tensor:
        x = x + self.attn(self.ln_1(x), self.causal)
        return x + self.mlp(self.ln_2(x))

    def forward_fused(self, x: Tensor, h: Tensor, next_ln: Optional[nn.LayerNorm], owned: bool = False):
        """Same block with the residual adds fused into the following LayerNorm (ops.add_layernorm): takes the stream x and
        h = ln_1(x), returns the new stream and next_ln(new stream) (None for the last block)."""
        from .. import ops
        if (owned and FP16X3_LINEAR and not self.causal and x.dtype == torch.float32 and x.shape[-1] == 64 * self.attn.heads
                and h.is_cuda and h.dtype == torch.float32
                and ops.linear_f16x3_acc_supported(self.attn.out_proj.weight, x) and ops.linear_f16x3_acc_supported(self.mlp.c_proj.weight, x)
                and ops.linear_f16x3_supported(h, self.attn.in_proj_weight) and ops.linear_f16x3_supported(h, self.mlp.c_fc.weight)):
            # the residual stream is updated IN PLACE by the block's two last linears (oryon_linear_f16x3_acc: one fp32 add per element,
            # the value `x + linear(...)` has), so each LayerNorm pass reads one tensor instead of two.  `owned`: x is a buffer of this
            # forward (never the caller's tensor).
            o = ops.mha_f16x3(_linear(h, self.attn.in_proj_weight, self.attn.in_proj_bias), self.attn.heads)
            ops.linear_f16x3_acc(o, self.attn.out_proj.weight, self.attn.out_proj.bias, x)
            _, h = ops.add_layernorm(x, None, self.ln_2.weight, self.ln_2.bias, self.ln_2.eps)
            h = _linear(h, self.mlp.c_fc.weight, self.mlp.c_fc.bias, quick_gelu=True)
            ops.linear_f16x3_acc(h, self.mlp.c_proj.weight, self.mlp.c_proj.bias, x)
            if next_ln is None:
                return x, None
            return ops.add_layernorm(x, None, next_ln.weight, next_ln.bias, next_ln.eps)
        x, h = ops.add_layernorm(x, self.attn(h, self.causal), self.ln_2.weight, self.ln_2.bias, self.ln_2.eps)
        d = self.mlp(h)
        if next_ln is None:
            return x + d, None
        return ops.add_layernorm(x, d, next_ln.weight, next_ln.bias, next_ln.eps)


class _Transformer(nn.Module):
    def __init__(self, width: int, layers: int, heads: int, causal: bool):
        super().__init__()
        self.resblocks = nn.Sequential(*[_Block(width, heads, causal) for _ in range(layers)])

    def forward(self, x: Tensor) -> Tensor:
        w_dtype = self.resblocks[0].ln_1.weight.dtype
        if (x.is_cuda and not torch.is_grad_enabled() and x.shape[-1] % 8 == 0 and x.dtype == w_dtype    # not autocast over fp32 weights
                and (x.dtype == torch.bfloat16 or (FP16X3_LINEAR and x.dtype == torch.float32 and x.shape[-1] <= 2048))):
            from .. import ops                      # inference: residual add + LayerNorm in one pass (B2; fp32 twin on the fp16x3 path)
            blocks = list(self.resblocks)
            ln0 = blocks[0].ln_1
            x, h = ops.add_layernorm(x, None, ln0.weight, ln0.bias, ln0.eps)
            owned = False
            if FP16X3_LINEAR and x.dtype == torch.float32 and ACC_RESIDUAL:
                x, owned = x.clone(), True                # the stream the blocks update in place
            for i, blk in enumerate(blocks):
                x, h = blk.forward_fused(x, h, blocks[i + 1].ln_1 if i + 1 < len(blocks) else None, owned)
            return x
        return self.resblocks(x)


class _Visual(nn.Module):
    def __init__(self, cfg: CLIPConfig):
        super().__init__()
        w = cfg.v_width
        grid = cfg.image_size // cfg.patch
        self.conv1 = PatchEmbed(3, w, kernel_size=cfg.patch, stride=cfg.patch, bias=False)
        self.class_embedding = nn.Parameter(w ** -0.5 * torch.randn(w))
        self.positional_embedding = nn.Parameter(w ** -0.5 * torch.randn(grid * grid + 1, w))
        self.ln_pre = nn.LayerNorm(w)
        self.transformer = _Transformer(w, cfg.v_layers, cfg.v_heads, causal=False)
        self.ln_post = nn.LayerNorm(w)
        self.proj = nn.Parameter(w ** -0.5 * torch.randn(w, cfg.embed_dim))   # not applied on Oryon's path


class CLIP(nn.Module):
    def __init__(self, cfg: Optional[CLIPConfig] = None):
        super().__init__()
        self.cfg = cfg or CLIPConfig.vit_l14_336()
        c = self.cfg
        self.visual = _Visual(c)
        self.transformer = _Transformer(c.t_width, c.t_layers, c.t_heads, causal=True)
        self.token_embedding = nn.Embedding(c.vocab, c.t_width)
        self.positional_embedding = nn.Parameter(0.01 * torch.randn(c.ctx, c.t_width))
        self.ln_final = nn.LayerNorm(c.t_width)
        self.text_projection = nn.Parameter(c.t_width ** -0.5 * torch.randn(c.t_width, c.embed_dim))
        self.logit_scale = nn.Parameter(torch.ones([]) * math.log(1 / 0.07))

    @property
    def dtype(self):
        return self.visual.conv1.weight.dtype

    # vlm.py:45-59 (after the pre-processing transform)
    def patch_tokens(self, clip_rgb: Tensor) -> Tensor:
        v = self.visual
        x = v.conv1.tokens(clip_rgb)                                  # [B, g, g, width]
        B, g, _, W = x.shape
        x = x.view(B, g * g, W)
        cls = v.class_embedding.to(x.dtype).expand(B, 1, W)
        x = torch.cat([cls, x], dim=1) + v.positional_embedding.to(x.dtype)
        x = v.transformer(v.ln_pre(x))
        toks = v.ln_post(x[:, 1:, :])
        return toks.transpose(1, 2).reshape(B, W, g, g)

    # vlm.py:74-83
    def text_features(self, tokens: Tensor) -> Tensor:               # tokens [M, ctx] int64 -> [M, embed_dim]
        x = self.token_embedding(tokens).to(self.dtype) + self.positional_embedding.to(self.dtype)
        x = self.ln_final(self.transformer(x))
        eot = tokens.argmax(dim=-1)
        return x[torch.arange(x.shape[0], device=x.device), eot] @ self.text_projection


CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def clip_preprocess(image: Tensor, size: int) -> Tensor:
    """Resize(size, bicubic) -> CenterCrop(size) -> Normalize(CLIP mean/std) on a float batch in [0,1]
    (the three transforms vlm.py:20-21 keeps from clip's pipeline; torchvision tensor path = F.interpolate bicubic,
    align_corners=False, no antialias - irrelevant when up-sampling 224 -> 336)."""
    B, C, H, W = image.shape
    if H <= W:
        nh, nw = size, int(size * W / H)
    else:
        nh, nw = int(size * H / W), size
    x = F.interpolate(image, size=(nh, nw), mode="bicubic", align_corners=False) if (nh, nw) != (H, W) else image
    top, left = int(round((nh - size) / 2.0)), int(round((nw - size) / 2.0))
    x = x[:, :, top:top + size, left:left + size]
    mean = torch.tensor(CLIP_MEAN, device=x.device, dtype=x.dtype).view(1, 3, 1, 1)
    std = torch.tensor(CLIP_STD, device=x.device, dtype=x.dtype).view(1, 3, 1, 1)
    return (x - mean) / std
