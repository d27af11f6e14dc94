"""Thin tensor-level wrappers over the C ABI (one function per entry point of include/oryon_hip.h).

Every wrapper takes torch CUDA tensors, allocates outputs with torch (plumbing) and launches the HIP
kernel on the current torch stream.  Nothing here computes: the arithmetic lives in liboryon_hip.so."""
from __future__ import annotations

import ctypes
import functools
import weakref
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import check, lib, ptr, stream_ptr

MATCH_TILE = 128
ROW_PAD = 256      # row capacities are multiples of 256; K0 zero-fills rows [n, round_up(n, 256)) of every map - rows beyond that
                   # are NOT written (the matchers never read past ceil(n / 256) * 256)
K_PAD = 32


def _on_tensor_device(fn):
    """Run the wrapped C-ABI call with the first tensor argument's GPU as the current HIP device: the library launches on the
    stream it is handed, but kernel attributes (dynamic-LDS opt-in) and hipMemsetAsync target the CURRENT device."""
    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        for a in args:
            if isinstance(a, torch.Tensor) and a.is_cuda:
                if a.device.index != torch.cuda.current_device():
                    with torch.cuda.device(a.device):
                        return fn(*args, **kwargs)
                break
        return fn(*args, **kwargs)
    return wrapped


def round_up(x: int, m: int) -> int:
    return ((int(x) + m - 1) // m) * m


@_on_tensor_device
def round_to_f16(x: torch.Tensor) -> torch.Tensor:
    """fp32 CUDA tensor rounded to the nearest float16 value, returned as fp32 (K1' input rounding)."""
    _lib.require_gpu(x.device)
    x = x.to(torch.float32).contiguous()
    out = torch.empty_like(x)
    check(lib().oryon_round_to_f16_f32(ptr(x), ptr(out), x.numel(), stream_ptr(x.device)), "oryon_round_to_f16_f32")
    return out


@_on_tensor_device
def quick_gelu_bf16(x: torch.Tensor) -> torch.Tensor:
    """x * sigmoid(1.702 x) on a bf16 CUDA tensor in one pass (B1); used by the CLIP residual blocks at inference."""
    _lib.require_gpu(x.device)
    assert x.dtype == torch.bfloat16
    x = x.contiguous()
    y = torch.empty_like(x)
    check(lib().oryon_quick_gelu_bf16(ptr(x), ptr(y), x.numel(), stream_ptr(x.device)), "oryon_quick_gelu_bf16")
    return y


@_on_tensor_device
def add_layernorm_bf16(x: torch.Tensor, delta: Optional[torch.Tensor], weight: torch.Tensor, bias: torch.Tensor, eps: float):
    """(x + delta, LayerNorm(x + delta)) for bf16 CUDA tensors in one pass (B2); delta None -> (x, LayerNorm(x)).
    Used by the CLIP residual blocks at inference."""
    _lib.require_gpu(x.device)
    assert x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and bias.dtype == torch.bfloat16
    D = x.shape[-1]
    x = x.contiguous()
    h = torch.empty_like(x)
    if delta is None:
        s_out = x
        check(lib().oryon_add_layernorm_bf16(ptr(x), None, ptr(weight), ptr(bias), x.numel() // D, D, eps, None, ptr(h),
                                             stream_ptr(x.device)), "oryon_add_layernorm_bf16")
    else:
        assert delta.shape == x.shape and delta.dtype == torch.bfloat16
        delta = delta.contiguous()
        s_out = delta                                  # the sum overwrites the branch output (a temporary of the caller)
        check(lib().oryon_add_layernorm_bf16(ptr(x), ptr(delta), ptr(weight), ptr(bias), x.numel() // D, D, eps, ptr(s_out), ptr(h),
                                             stream_ptr(x.device)), "oryon_add_layernorm_bf16")
    return s_out, h


@_on_tensor_device
def add_layernorm_f32(x: torch.Tensor, delta: Optional[torch.Tensor], weight: torch.Tensor, bias: torch.Tensor, eps: float):
    """fp32 twin of add_layernorm_bf16 (the fp32 / fp16x3 inference path of the towers): (x + delta, LayerNorm(x + delta))."""
    _lib.require_gpu(x.device)
    assert x.dtype == torch.float32 and weight.dtype == torch.float32 and bias.dtype == torch.float32
    D = x.shape[-1]
    x = x.contiguous()
    h = torch.empty_like(x)
    if delta is None:
        s_out = x
        check(lib().oryon_add_layernorm_f32(ptr(x), None, ptr(weight), ptr(bias), x.numel() // D, D, eps, None, ptr(h),
                                            stream_ptr(x.device)), "oryon_add_layernorm_f32")
    else:
        assert delta.shape == x.shape and delta.dtype == torch.float32
        delta = delta.contiguous()
        s_out = delta
        check(lib().oryon_add_layernorm_f32(ptr(x), ptr(delta), ptr(weight), ptr(bias), x.numel() // D, D, eps, ptr(s_out), ptr(h),
                                            stream_ptr(x.device)), "oryon_add_layernorm_f32")
    return s_out, h


def add_layernorm(x, delta, weight, bias, eps):
    """Dispatch on the stream's dtype (bf16 or fp32)."""
    return (add_layernorm_bf16 if x.dtype == torch.bfloat16 else add_layernorm_f32)(x, delta, weight, bias, eps)


@_on_tensor_device
def swin_window_attention_bf16(qkv: torch.Tensor, pad_qkv: torch.Tensor, bias_t: torch.Tensor, heads: int, shift: int) -> torch.Tensor:
    """Shifted-window attention (window 7, head dim 32) on un-windowed q|k|v tokens [B,H,W,3C] bf16 -> [B,H,W,C] bf16 (B3)."""
    _lib.require_gpu(qkv.device)
    assert qkv.dtype == torch.bfloat16 and pad_qkv.dtype == torch.bfloat16 and bias_t.dtype == torch.float32
    B, H, W, C3 = qkv.shape
    C = C3 // 3
    assert C3 == 3 * C and pad_qkv.numel() == C3 and bias_t.shape == (heads, 49, 49)
    qkv = qkv.contiguous()
    out = torch.empty((B, H, W, C), dtype=torch.bfloat16, device=qkv.device)
    check(lib().oryon_swin_window_attention_bf16(ptr(qkv), ptr(pad_qkv.contiguous()), ptr(bias_t.contiguous()), B, H, W, C, heads, shift,
                                                 ptr(out), stream_ptr(qkv.device)), "oryon_swin_window_attention_bf16")
    return out


@_on_tensor_device
def swin_window_attention_f32(qkv: torch.Tensor, pad_qkv: torch.Tensor, bias_t: torch.Tensor, heads: int, shift: int) -> torch.Tensor:
    """The B3 kernel on fp32 tensors: [B,H,W,3C] fp32 -> [B,H,W,C] fp32 (fp32 evaluation of the Swin guidance tower)."""
    _lib.require_gpu(qkv.device)
    assert qkv.dtype == torch.float32 and pad_qkv.dtype == torch.float32 and bias_t.dtype == torch.float32
    B, H, W, C3 = qkv.shape
    C = C3 // 3
    assert C3 == 3 * C and pad_qkv.numel() == C3 and bias_t.shape == (heads, 49, 49)
    qkv = qkv.contiguous()
    out = torch.empty((B, H, W, C), dtype=torch.float32, device=qkv.device)
    check(lib().oryon_swin_window_attention_f32(ptr(qkv), ptr(pad_qkv.contiguous()), ptr(bias_t.contiguous()), B, H, W, C, heads, shift,
                                                ptr(out), stream_ptr(qkv.device)), "oryon_swin_window_attention_f32")
    return out


@_on_tensor_device
def fusion_class_layer(x_nhwc: torch.Tensor, text_guidance: torch.Tensor, params) -> torch.Tensor:
    """The class-aggregation layer of ImageTextFusion for T = 1 (see oryon_fusion_class_layer_f32): x [B,24,24,128] fp32 NHWC, text_guidance
    [B,128], params = the 14 fp32 parameter tensors (norm1 w/b, q w/b, k w/b, v w/b, norm2 w/b, MLP.0 w/b, MLP.2 w/b) -> [B,24,24,128]."""
    dev = _lib.require_gpu(x_nhwc.device)
    B = x_nhwc.shape[0]
    assert x_nhwc.dtype == torch.float32 and tuple(x_nhwc.shape[1:]) == (24, 24, 128) and tuple(text_guidance.shape) == (B, 128)
    x, tg = x_nhwc.contiguous(), text_guidance.to(torch.float32).contiguous()
    out = torch.empty_like(x)
    if B == 0:
        return out
    ps = [p.detach().to(torch.float32).contiguous() for p in params]
    shapes = [(128,), (128,), (128, 256), (128,), (128, 256), (128,), (128, 128), (128,), (128,), (128,), (512, 128), (512,), (128, 512), (128,)]
    assert [tuple(p.shape) for p in ps] == shapes, [tuple(p.shape) for p in ps]
    w = _lib.FusionClassWeights(*[ptr(p) for p in ps])
    import ctypes
    check(lib().oryon_fusion_class_layer_f32(ptr(x), ptr(tg), ctypes.byref(w), B, ptr(out), stream_ptr(dev)), "oryon_fusion_class_layer_f32")
    return out


_conv24_images: dict = {}


def _conv24_image(weight: torch.Tensor) -> torch.Tensor:
    """Packed fp16 hi / lo fragment image of a conv weight [cout, cin, k, k], made once per live tensor and in-place version."""
    key = id(weight)
    hit = _conv24_images.get(key)
    if hit is not None and (hit[0]() is not weight or hit[1] != (weight._version, weight.data_ptr(), tuple(weight.shape))):
        hit = None
    if hit is None:
        cout, cin, k, _ = weight.shape
        nbytes = int(lib().oryon_conv24_image_bytes(cout, cin, k))
        if nbytes <= 0:
            raise _lib.OryonError(f"oryon_conv24_f16x3: unsupported weight shape {tuple(weight.shape)}")
        w = weight.detach().to(torch.float32).contiguous()
        img = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
        check(lib().oryon_conv24_pack_f16x3(ptr(w), cout, cin, k, ptr(img), stream_ptr(w.device)), "oryon_conv24_pack_f16x3")
        if len(_conv24_images) > 64:
            _conv24_images.clear()
        ref = weakref.ref(weight, lambda _r, k_=key: _conv24_images.pop(k_, None))
        hit = _conv24_images[key] = (ref, (weight._version, weight.data_ptr(), tuple(weight.shape)), img)
    return hit[2]


def conv24_supported(x_nhwc: torch.Tensor, weight: torch.Tensor) -> bool:
    return (x_nhwc.is_cuda and x_nhwc.dtype == torch.float32 and x_nhwc.dim() == 4
            and tuple(x_nhwc.shape[1:3]) == (24, 24) and weight.dim() == 4 and weight.shape[2] == weight.shape[3] and weight.shape[2] in (3, 7)
            and weight.shape[0] % 64 == 0 and weight.shape[1] % 4 == 0 and x_nhwc.shape[3] == weight.shape[1])


@_on_tensor_device
def conv24_f16x3(x_nhwc: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], relu: bool = False) -> torch.Tensor:
    """act(conv2d(x, weight, bias, stride 1, padding k // 2)) for 24 x 24 maps: x [n,24,24,cin] fp32 NHWC, weight [cout,cin,k,k] (torch
    layout, k = 3 or 7) -> [n,24,24,cout] fp32 NHWC (ImageTextFusion's conv1 / guidance_projection; fp16x3 MFMA products, fp32-grade).
    Inference only: no autograd graph is recorded."""
    dev = _lib.require_gpu(x_nhwc.device)
    assert conv24_supported(x_nhwc, weight), (x_nhwc.shape, weight.shape)
    x = x_nhwc.contiguous()
    n, cout, cin, k = x.shape[0], weight.shape[0], weight.shape[1], weight.shape[2]
    img = _conv24_image(weight)
    b = None if bias is None else bias.detach().to(torch.float32).contiguous()
    y = torch.empty((n, 24, 24, cout), dtype=torch.float32, device=dev)
    if n == 0:
        return y
    check(lib().oryon_conv24_f16x3(ptr(x), n, cin, ptr(img), ptr(b), cout, k, int(relu), ptr(y), stream_ptr(dev)), "oryon_conv24_f16x3")
    return y


@_on_tensor_device
def fusion_window_attention(qk: torch.Tensor, v: torch.Tensor, heads: int, window: int, shift: int) -> torch.Tensor:
    """Shifted-window attention of ImageTextFusion's guided Swin blocks on un-windowed tokens: qk [B,H,W,2C] (q | k projections),
    v [B,H,W,C] fp32 -> [B,H,W,C] fp32 (roll, windows, masked softmax attention, windows back, roll back in one kernel; fp16x3 MFMA products,
    fp32-grade)."""
    _lib.require_gpu(qk.device)
    B, H, W, C2 = qk.shape
    C = C2 // 2
    assert qk.dtype == torch.float32 and v.dtype == torch.float32 and tuple(v.shape) == (B, H, W, C) and C2 == 2 * C
    qk, v = qk.contiguous(), v.contiguous()
    out = torch.empty((B, H, W, C), dtype=torch.float32, device=qk.device)
    if B == 0:
        return out
    check(lib().oryon_fusion_window_attention_f32(ptr(qk), ptr(v), B, H, W, C, heads, window, shift, ptr(out), stream_ptr(qk.device)),
          "oryon_fusion_window_attention_f32")
    return out


@_on_tensor_device
def rgb_resize_bilinear(rgb_hwc: torch.Tensor, out_hw: Tuple[int, int]) -> torch.Tensor:
    """uint8 [n,HI,WI,3] -> fp32 [n,3,HO,WO] in [0,1] (K-1: /255., CHW, bilinear align_corners=False, fp64 arithmetic)."""
    _lib.require_gpu(rgb_hwc.device)
    assert rgb_hwc.dtype == torch.uint8 and rgb_hwc.dim() == 4 and rgb_hwc.shape[3] == 3
    rgb_hwc = rgb_hwc.contiguous()
    n, HI, WI = rgb_hwc.shape[:3]
    out = torch.empty((n, 3, int(out_hw[0]), int(out_hw[1])), dtype=torch.float32, device=rgb_hwc.device)
    check(lib().oryon_rgb_resize_bilinear(ptr(rgb_hwc), n, HI, WI, int(out_hw[0]), int(out_hw[1]), ptr(out), stream_ptr(rgb_hwc.device)),
          "oryon_rgb_resize_bilinear")
    return out


@_on_tensor_device
def resize_bilinear(x: torch.Tensor, out_hw: Tuple[int, int], round_output: bool = False) -> torch.Tensor:
    """fp32 [n,HI,WI] -> fp32 [n,HO,WO], torch bilinear (align_corners=False); round_output mimics torchvision on integer images."""
    _lib.require_gpu(x.device)
    x = x.to(torch.float32).contiguous()
    n, HI, WI = x.shape
    out = torch.empty((n, int(out_hw[0]), int(out_hw[1])), dtype=torch.float32, device=x.device)
    check(lib().oryon_resize_bilinear_f32(ptr(x), n, HI, WI, int(out_hw[0]), int(out_hw[1]), int(round_output), ptr(out),
                                          stream_ptr(x.device)), "oryon_resize_bilinear_f32")
    return out


@_on_tensor_device
def roi_compact(mask: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """mask [n_maps, H, W] (or [H,W]) int32 -> (roi [n_maps, HW] int32 linear indices, count [n_maps] int32)."""
    if mask.dim() == 2:
        mask = mask[None]
    _lib.require_gpu(mask.device)
    mask = mask.to(torch.int32).contiguous()
    n_maps, HW = mask.shape[0], mask.shape[1] * mask.shape[2]
    roi = torch.empty((n_maps, HW), dtype=torch.int32, device=mask.device)
    count = torch.empty((n_maps,), dtype=torch.int32, device=mask.device)
    check(lib().oryon_roi_compact(ptr(mask), n_maps, HW, ptr(roi), ptr(count), stream_ptr(mask.device)), "oryon_roi_compact")
    return roi, count


@_on_tensor_device
def mask_from_logits(logits: torch.Tensor, threshold: float) -> torch.Tensor:
    _lib.require_gpu(logits.device)
    logits = logits.to(torch.float32).contiguous()
    out = torch.empty(logits.shape, dtype=torch.int32, device=logits.device)
    check(lib().oryon_mask_from_logits(ptr(logits), logits.numel(), float(threshold), ptr(out), stream_ptr(logits.device)),
          "oryon_mask_from_logits")
    return out


@_on_tensor_device
def mask_resize_nearest(mask: torch.Tensor, out_hw: Tuple[int, int]) -> torch.Tensor:
    """uint8 masks [n,HI,WI] -> int32 [n,HO,WO] with torch's legacy 'nearest' index rule."""
    if mask.dim() == 2:
        mask = mask[None]
    _lib.require_gpu(mask.device)
    m = mask.to(torch.uint8).contiguous()
    n, HI, WI = m.shape
    out = torch.empty((n, int(out_hw[0]), int(out_hw[1])), dtype=torch.int32, device=m.device)
    check(lib().oryon_mask_resize_nearest(ptr(m), n, HI, WI, int(out_hw[0]), int(out_hw[1]), ptr(out), stream_ptr(m.device)),
          "oryon_mask_resize_nearest")
    return out


@_on_tensor_device
def roi_subsample_(roi: torch.Tensor, count: torch.Tensor, max_keep: int, seed: int, map_key: Optional[torch.Tensor] = None) -> None:
    """In-place device-RNG subsample of every ROI list to at most max_keep entries (order preserved)."""
    _lib.require_gpu(roi.device)
    check(lib().oryon_roi_subsample(ptr(roi), ptr(count), roi.shape[0], roi.shape[1], int(max_keep), int(seed) & (2**64 - 1),
                                    ptr(map_key), stream_ptr(roi.device)), "oryon_roi_subsample")


@_on_tensor_device
def gather_normalise(feat: torch.Tensor, roi: torch.Tensor, count: torch.Tensor, rows_cap: int, c_pad: Optional[int] = None,
                     want_f16: bool = False):
    """feat [n_maps,C,H,W] fp32, roi [n_maps, stride] -> [n_maps, rows_cap, C_pad] unit rows (rows_cap % 256 == 0);
    with want_f16 also the IEEE-half copy used by the screening pass (returns (f32, f16))."""
    _lib.require_gpu(feat.device)
    assert feat.dtype == torch.float32 and feat.is_contiguous()
    n_maps, C = feat.shape[0], feat.shape[1]
    HW = feat.shape[2] * feat.shape[3]
    Cp = int(c_pad) if c_pad else round_up(C, K_PAD)
    out = torch.empty((n_maps, rows_cap, Cp), dtype=torch.float32, device=feat.device)
    out16 = torch.empty((n_maps, rows_cap, Cp), dtype=torch.float16, device=feat.device) if want_f16 else None
    check(lib().oryon_gather_normalise_f32(ptr(feat), n_maps, C, HW, ptr(roi), roi.shape[1], ptr(count), rows_cap, Cp,
                                           ptr(out), ptr(out16), stream_ptr(feat.device)), "oryon_gather_normalise_f32")
    return (out, out16) if want_f16 else out


def unpermute_k(x: torch.Tensor) -> torch.Tensor:
    """K0 stores every descriptor row with k permuted inside groups of 8 (position 8g+4h+j holds k = 8g+2j+h, the order
    the MFMA A/B operands consume 16-byte chunks in).  Returns the rows in natural k order (tests / debugging only)."""
    Cp = x.shape[-1]
    pos = torch.arange(Cp, device=x.device)
    g, r = pos // 8, pos % 8
    k_at_pos = 8 * g + 2 * (r % 4) + r // 4
    out = torch.empty_like(x)
    out[..., k_at_pos] = x
    return out


@_on_tensor_device
def match(a_hat: torch.Tensor, q_hat: torch.Tensor, n_a: torch.Tensor, n_q: torch.Tensor, threshold: float):
    """a_hat [B,cap_a,Cp], q_hat [B,cap_q,Cp] -> (min_dist [B,cap_a] f32, argmin [B,cap_a] i32, valid [B,cap_a] u8)."""
    dev = _lib.require_gpu(a_hat.device)
    B, cap_a, Cp = a_hat.shape
    cap_q = q_hat.shape[1]
    assert q_hat.shape[0] == B and q_hat.shape[2] == Cp
    min_dist = torch.empty((B, cap_a), dtype=torch.float32, device=dev)
    argmin = torch.empty((B, cap_a), dtype=torch.int32, device=dev)
    valid = torch.empty((B, cap_a), dtype=torch.uint8, device=dev)
    wsb = lib().oryon_match_workspace_bytes(B, cap_a)
    ws = torch.empty((max(wsb, 16),), dtype=torch.uint8, device=dev)
    check(lib().oryon_match_f32(ptr(a_hat), ptr(q_hat), B, Cp, cap_a, cap_q, ptr(n_a), ptr(n_q), float(threshold),
                                ptr(min_dist), ptr(argmin), ptr(valid), ptr(ws), wsb, stream_ptr(dev)), "oryon_match_f32")
    return min_dist, argmin, valid


@_on_tensor_device
def match_screened(a_hat, q_hat, a16, q16, n_a, n_q, threshold: float):
    """fp16-screened, fp32-exact matcher (K1s).  Same outputs as `match` on every row that can be valid."""
    dev = _lib.require_gpu(a_hat.device)
    B, cap_a, Cp = a_hat.shape
    cap_q = q_hat.shape[1]
    assert a16.dtype == torch.float16 and q16.dtype == torch.float16 and a16.shape == a_hat.shape and q16.shape == q_hat.shape
    min_dist = torch.empty((B, cap_a), dtype=torch.float32, device=dev)
    argmin = torch.empty((B, cap_a), dtype=torch.int32, device=dev)
    valid = torch.empty((B, cap_a), dtype=torch.uint8, device=dev)
    wsb = lib().oryon_match_screened_workspace_bytes(B, Cp, cap_a)
    ws = torch.empty((max(wsb, 16),), dtype=torch.uint8, device=dev)
    check(lib().oryon_match_screened(ptr(a_hat), ptr(q_hat), ptr(a16), ptr(q16), B, Cp, cap_a, cap_q, ptr(n_a), ptr(n_q),
                                     float(threshold), ptr(min_dist), ptr(argmin), ptr(valid), ptr(ws), wsb, stream_ptr(dev)),
          "oryon_match_screened")
    return min_dist, argmin, valid


@_on_tensor_device
def gather_normalise_q8(feat: torch.Tensor, roi: torch.Tensor, count: torch.Tensor, rows_cap: int, c_pad: int, want_f16: bool = False):
    """K0 with int8 copies: -> (rows fp32 k-permuted, rows fp16 | None, rows int8, slice_scale [n, rows_cap/16], eps_max [n])."""
    dev = _lib.require_gpu(feat.device)
    feat = feat.to(torch.float32).contiguous()
    n_maps, C, H, W = feat.shape
    assert c_pad in (256, 512) and C <= c_pad and rows_cap % ROW_PAD == 0
    out = torch.empty((n_maps, rows_cap, c_pad), dtype=torch.float32, device=dev)
    out16 = torch.empty((n_maps, rows_cap, c_pad), dtype=torch.float16, device=dev) if want_f16 else None
    out8 = torch.empty((n_maps, rows_cap, c_pad), dtype=torch.int8, device=dev)
    scale = torch.ones((n_maps, rows_cap // 16), dtype=torch.float32, device=dev)
    eps = torch.empty((n_maps,), dtype=torch.float32, device=dev)
    check(lib().oryon_gather_normalise_q8(ptr(feat), n_maps, C, H * W, ptr(roi), roi.shape[1], ptr(count), rows_cap, c_pad, ptr(out),
                                          ptr(out16), ptr(out8), ptr(scale), ptr(eps), stream_ptr(dev)), "oryon_gather_normalise_q8")
    return out, out16, out8, scale, eps


LAYOUT_NCHW, LAYOUT_NHWC = 0, 1


def map_layout(feat: torch.Tensor):
    """(tensor whose storage the C ABI can read, layout code) for a logical [n,C,H,W] descriptor map: contiguous -> NCHW,
    torch.channels_last -> NHWC (zero-copy), anything else is made contiguous first."""
    if feat.is_contiguous():
        return feat, LAYOUT_NCHW
    if feat.dim() == 4 and feat.is_contiguous(memory_format=torch.channels_last):
        return feat, LAYOUT_NHWC
    return feat.contiguous(), LAYOUT_NCHW


@_on_tensor_device
def gather_q8(feat: torch.Tensor, roi: torch.Tensor, count: torch.Tensor, rows_cap: int, c_pad: int, want_f32: bool = False,
              round_f16: bool = False):
    """K0v3 (gather8.hip): ROI rows of [n,C,H,W] fp32 maps (contiguous or channels_last) -> (rows int8 [n,rows_cap,c_pad],
    slice_scale [n,rows_cap/16], eps_max [n], row_norm [n,rows_cap], rows fp32 k-permuted [n,rows_cap,c_pad] | None)."""
    dev = _lib.require_gpu(feat.device)
    assert feat.dtype == torch.float32 and feat.dim() == 4
    feat, layout = map_layout(feat)
    n_maps, C, H, W = feat.shape
    assert c_pad in (256, 512) and C <= c_pad and rows_cap % ROW_PAD == 0
    out8 = torch.empty((n_maps, rows_cap, c_pad), dtype=torch.int8, device=dev)
    scale = torch.empty((n_maps, rows_cap // 16), dtype=torch.float32, device=dev)
    eps = torch.empty((n_maps,), dtype=torch.float32, device=dev)
    norm = torch.empty((n_maps, rows_cap), dtype=torch.float32, device=dev)
    out32 = torch.empty((n_maps, rows_cap, c_pad), dtype=torch.float32, device=dev) if want_f32 else None
    check(lib().oryon_gather_q8(feat.data_ptr(), n_maps, C, H * W, layout, ptr(roi), roi.shape[1], ptr(count), rows_cap, c_pad, ptr(out8),
                                ptr(scale), ptr(eps), ptr(norm), ptr(out32), int(bool(round_f16)), stream_ptr(dev)), "oryon_gather_q8")
    return out8, scale, eps, norm, out32


@_on_tensor_device
def gather_mx6(feat: torch.Tensor, roi: torch.Tensor, count: torch.Tensor, rows_cap: int, c_pad: int, want_f32: bool = False,
               round_f16: bool = False):
    """K0 with MX-fp6 screening operands (oryon_gather_mx6): -> (rows uint8 [n,rows_cap,c_pad] of 32-byte slots, err_max [n] fp32,
    row_norm [n,rows_cap], rows fp32 k-permuted | None)."""
    dev = _lib.require_gpu(feat.device)
    assert feat.dtype == torch.float32 and feat.dim() == 4
    feat, layout = map_layout(feat)
    n_maps, C, H, W = feat.shape
    assert c_pad in (256, 512) and C <= c_pad and rows_cap % ROW_PAD == 0
    out6 = torch.empty((n_maps, rows_cap, c_pad), dtype=torch.uint8, device=dev)
    err = torch.empty((n_maps,), dtype=torch.float32, device=dev)
    norm = torch.empty((n_maps, rows_cap), dtype=torch.float32, device=dev)
    out32 = torch.empty((n_maps, rows_cap, c_pad), dtype=torch.float32, device=dev) if want_f32 else None
    check(lib().oryon_gather_mx6(feat.data_ptr(), n_maps, C, H * W, layout, ptr(roi), roi.shape[1], ptr(count), rows_cap, c_pad, ptr(out6),
                                 ptr(err), ptr(norm), ptr(out32), int(bool(round_f16)), stream_ptr(dev)), "oryon_gather_mx6")
    return out6, err, norm, out32


@_on_tensor_device
def match_corrs_mx6(a_hat, a6, a_err, feat_q, roi_a, roi_q, q_norm, q6, q_err, n_a, n_q, threshold: float, W: int, max_corrs: int,
                    seed: int, pair_key=None, corr_rows: Optional[int] = None, n_undecided=None, round_f16: bool = False):
    """oryon_match_corrs_mx6: the lazy matcher + sampler on MX-fp6 operands; same outputs as match_corrs_i8."""
    dev = _lib.require_gpu(a_hat.device)
    feat_q, layout = map_layout(feat_q)
    B, cap_a, Cp = a_hat.shape
    cap_q = q6.shape[1]
    C_true, HW = feat_q.shape[1], feat_q.shape[2] * feat_q.shape[3]
    corr_rows = int(corr_rows or max_corrs)
    min_dist = torch.empty((B, cap_a), dtype=torch.float32, device=dev)
    argmin = torch.empty((B, cap_a), dtype=torch.int32, device=dev)
    valid = torch.empty((B, cap_a), dtype=torch.uint8, device=dev)
    corrs = torch.zeros((B, corr_rows, 4), dtype=torch.int32, device=dev)
    n_valid = torch.empty((B,), dtype=torch.int32, device=dev)
    n_sel = torch.empty((B,), dtype=torch.int32, device=dev)
    status = torch.empty((B,), dtype=torch.int32, device=dev)
    wsb = lib().oryon_match_corrs_i8_workspace_bytes(B, Cp, cap_a, cap_q, corr_rows)
    ws = torch.empty((max(wsb, 16),), dtype=torch.uint8, device=dev)
    check(lib().oryon_match_corrs_mx6(ptr(a_hat), ptr(a6), ptr(a_err), feat_q.data_ptr(), C_true, HW, layout, ptr(roi_a), roi_a.shape[1],
                                      ptr(roi_q), roi_q.shape[1], ptr(q_norm), ptr(q6), ptr(q_err), B, Cp, cap_a, cap_q, ptr(n_a), ptr(n_q),
                                      float(threshold), int(W), int(max_corrs), corr_rows, int(seed) & (2**64 - 1), ptr(pair_key),
                                      ptr(min_dist), ptr(argmin), ptr(valid), ptr(corrs), ptr(n_valid), ptr(n_sel), ptr(status),
                                      ptr(n_undecided), int(bool(round_f16)), ptr(ws), ws.numel(), stream_ptr(dev)), "oryon_match_corrs_mx6")
    return corrs, n_valid, n_sel, status, min_dist, argmin, valid


import threading

_raw_ws = {}                      # (device, stream) -> cached matcher workspace (holds room for the rarely used fp32 fall-back rows)
_raw_ws_lock = threading.Lock()   # the C entry points issue ~15 dependent launches on the workspace and ctypes releases the GIL: two host
                                  # threads that launch on ONE stream must not interleave their sequences on the shared buffer


def release_workspaces() -> None:
    """Drop the cached matcher workspaces (they pin the largest size ever requested per (device, stream) for the life of the process)."""
    with _raw_ws_lock:
        _raw_ws.clear()



@_on_tensor_device
def match_screened8_raw(a_hat, a8, a_scale, feat_q, roi_q, q_norm, q8, q_scale, q_eps, n_a, n_q, threshold: float, n_undecided=None,
                        round_f16: bool = False):
    """K1s8 on K0v3 operands: no fp32 copy of the query rows exists; the exact re-scoring reads candidates from the raw map feat_q
    ([B,C,H,W] contiguous or channels_last).  Same outputs as `match_screened8`."""
    dev = _lib.require_gpu(a_hat.device)
    feat_q, layout = map_layout(feat_q)
    B, cap_a, Cp = a_hat.shape
    cap_q = q8.shape[1]
    C_true, HW = feat_q.shape[1], feat_q.shape[2] * feat_q.shape[3]
    min_dist = torch.empty((B, cap_a), dtype=torch.float32, device=dev)
    argmin = torch.empty((B, cap_a), dtype=torch.int32, device=dev)
    valid = torch.empty((B, cap_a), dtype=torch.uint8, device=dev)
    wsb = lib().oryon_match_screened8_raw_workspace_bytes(B, Cp, cap_a, cap_q)
    # the workspace holds room for the (rarely used) fp32 fall-back rows of every pair: cached per (device, stream) instead of
    # being re-requested from the allocator on every call
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    with _raw_ws_lock:
        ws = _raw_ws.get(key)
        if ws is None or ws.numel() < wsb:
            ws = torch.empty((max(wsb, 16),), dtype=torch.uint8, device=dev)
            _raw_ws[key] = ws
        check(lib().oryon_match_screened8_raw(ptr(a_hat), ptr(a8), ptr(a_scale), feat_q.data_ptr(), C_true, HW, layout, ptr(roi_q),
                                              roi_q.shape[1], ptr(q_norm), ptr(q8), ptr(q_scale), ptr(q_eps), B, Cp, cap_a, cap_q, ptr(n_a),
                                              ptr(n_q), float(threshold), ptr(min_dist), ptr(argmin), ptr(valid), ptr(n_undecided),
                                              int(bool(round_f16)), ptr(ws), ws.numel(), stream_ptr(dev)), "oryon_match_screened8_raw")
    return min_dist, argmin, valid


@_on_tensor_device
def match_corrs_i8(a_hat, a8, a_scale, feat_q, roi_a, roi_q, q_norm, q8, q_scale, q_eps, n_a, n_q, threshold: float, W: int, max_corrs: int,
                   seed: int, pair_key=None, corr_rows: Optional[int] = None, force_eager: bool = False, n_undecided=None,
                   round_f16: bool = False):
    """Lazy K1s8 + K1b (oryon_match_corrs_i8): -> (corrs [B,corr_rows,4] i32, n_valid [B], n_sel [B], status [B], min_dist, argmin, valid).
    Same correspondences as select_corrs(match_screened8_raw(...)); min_dist / argmin are exact only on sampled rows unless force_eager."""
    dev = _lib.require_gpu(a_hat.device)
    feat_q, layout = map_layout(feat_q)
    B, cap_a, Cp = a_hat.shape
    cap_q = q8.shape[1]
    C_true, HW = feat_q.shape[1], feat_q.shape[2] * feat_q.shape[3]
    corr_rows = int(corr_rows or max_corrs)
    min_dist = torch.empty((B, cap_a), dtype=torch.float32, device=dev)
    argmin = torch.empty((B, cap_a), dtype=torch.int32, device=dev)
    valid = torch.empty((B, cap_a), dtype=torch.uint8, device=dev)
    corrs = torch.zeros((B, corr_rows, 4), dtype=torch.int32, device=dev)
    n_valid = torch.empty((B,), dtype=torch.int32, device=dev)
    n_sel = torch.empty((B,), dtype=torch.int32, device=dev)
    status = torch.empty((B,), dtype=torch.int32, device=dev)
    wsb = lib().oryon_match_corrs_i8_workspace_bytes(B, Cp, cap_a, cap_q, corr_rows)
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    with _raw_ws_lock:
        ws = _raw_ws.get(key)
        if ws is None or ws.numel() < wsb:
            ws = torch.empty((max(wsb, 16),), dtype=torch.uint8, device=dev)
            _raw_ws[key] = ws
        return _match_corrs_i8_locked(a_hat, a8, a_scale, feat_q, C_true, HW, layout, roi_a, roi_q, q_norm, q8, q_scale, q_eps, B, Cp, cap_a,
                                      cap_q, n_a, n_q, threshold, W, max_corrs, corr_rows, seed, pair_key, force_eager, min_dist, argmin, valid,
                                      corrs, n_valid, n_sel, status, n_undecided, round_f16, ws, dev)


def _match_corrs_i8_locked(a_hat, a8, a_scale, feat_q, C_true, HW, layout, roi_a, roi_q, q_norm, q8, q_scale, q_eps, B, Cp, cap_a, cap_q, n_a,
                           n_q, threshold, W, max_corrs, corr_rows, seed, pair_key, force_eager, min_dist, argmin, valid, corrs, n_valid, n_sel,
                           status, n_undecided, round_f16, ws, dev):
    check(lib().oryon_match_corrs_i8(ptr(a_hat), ptr(a8), ptr(a_scale), feat_q.data_ptr(), C_true, HW, layout, ptr(roi_a), roi_a.shape[1],
                                     ptr(roi_q), roi_q.shape[1], ptr(q_norm), ptr(q8), ptr(q_scale), ptr(q_eps), B, Cp, cap_a, cap_q,
                                     ptr(n_a), ptr(n_q), float(threshold), int(W), int(max_corrs), corr_rows, int(seed) & (2**64 - 1),
                                     ptr(pair_key), int(bool(force_eager)), ptr(min_dist), ptr(argmin), ptr(valid), ptr(corrs), ptr(n_valid),
                                     ptr(n_sel), ptr(status), ptr(n_undecided), int(bool(round_f16)), ptr(ws), ws.numel(), stream_ptr(dev)),
          "oryon_match_corrs_i8")
    return corrs, n_valid, n_sel, status, min_dist, argmin, valid


@_on_tensor_device
def match_screened8(a_hat, q_hat, a8, q8, a_scale, q_scale, q_eps, n_a, n_q, threshold: float, c_true: int, n_undecided=None):
    """int8 pre-screen + fp16 screen + exact fp32 re-scoring (K1s8).  Same outputs as `match_screened`.
    n_undecided: optional int32 [B] tensor receiving the number of anchors the int8 stage handed to the fp16 stage."""
    dev = _lib.require_gpu(a_hat.device)
    B, cap_a, Cp = a_hat.shape
    cap_q = q_hat.shape[1]
    min_dist = torch.empty((B, cap_a), dtype=torch.float32, device=dev)
    argmin = torch.empty((B, cap_a), dtype=torch.int32, device=dev)
    valid = torch.empty((B, cap_a), dtype=torch.uint8, device=dev)
    wsb = lib().oryon_match_screened8_workspace_bytes(B, Cp, cap_a, cap_q)
    ws = torch.empty((max(wsb, 16),), dtype=torch.uint8, device=dev)
    check(lib().oryon_match_screened8(ptr(a_hat), ptr(q_hat), ptr(a8), ptr(q8), ptr(a_scale), ptr(q_scale), ptr(q_eps),
                                      B, int(c_true), Cp, cap_a, cap_q, ptr(n_a), ptr(n_q), float(threshold), ptr(min_dist), ptr(argmin),
                                      ptr(valid), ptr(n_undecided), ptr(ws), wsb, stream_ptr(dev)), "oryon_match_screened8")
    return min_dist, argmin, valid


@_on_tensor_device
def select_corrs(roi_a, roi_q, n_a, n_q, argmin, valid, W: int, max_corrs: int, seed: int, pair_key=None,
                 corr_rows: Optional[int] = None):
    """Device-RNG correspondence sampling -> (corrs [B,corr_rows,4] i32, n_valid [B], n_sel [B], status [B])."""
    dev = _lib.require_gpu(roi_a.device)
    B, cap_a = argmin.shape
    corr_rows = int(corr_rows or max_corrs)
    corrs = torch.zeros((B, corr_rows, 4), dtype=torch.int32, device=dev)
    n_valid = torch.empty((B,), dtype=torch.int32, device=dev)
    n_sel = torch.empty((B,), dtype=torch.int32, device=dev)
    status = torch.empty((B,), dtype=torch.int32, device=dev)
    scratch = torch.empty((B, cap_a), dtype=torch.int32, device=dev)
    check(lib().oryon_select_corrs(ptr(roi_a), ptr(roi_q), roi_a.shape[1], roi_q.shape[1], ptr(n_a), ptr(n_q), ptr(argmin),
                                   ptr(valid), cap_a, B, int(W), int(max_corrs), corr_rows, int(seed) & (2**64 - 1),
                                   ptr(pair_key), ptr(scratch), ptr(corrs), ptr(n_valid), ptr(n_sel), ptr(status),
                                   stream_ptr(dev)), "oryon_select_corrs")
    return corrs, n_valid, n_sel, status


@_on_tensor_device
def lift_pairs(corrs: torch.Tensor, n_corr: Optional[torch.Tensor], feat_hw, depth_a: torch.Tensor, depth_q: torch.Tensor,
               cam_a: torch.Tensor, cam_q: torch.Tensor, status: Optional[torch.Tensor] = None):
    """corrs [B,n_cap,4] i32, depth_* [B,H,W] f32 (mm), cam_* [B,9] f32 -> (pcd_a, pcd_q [B,n_cap,3] metres, n_out [B])."""
    dev = _lib.require_gpu(corrs.device)
    B, n_cap = corrs.shape[0], corrs.shape[1]
    assert corrs.dtype == torch.int32 and depth_a.dtype == torch.float32 and depth_q.dtype == torch.float32
    assert cam_a.dtype == torch.float32 and cam_a.shape == (B, 9) and cam_q.shape == (B, 9)
    pa = torch.empty((B, n_cap, 3), dtype=torch.float32, device=dev)      # the kernel zeroes the rows past the lifted count
    pq = torch.empty((B, n_cap, 3), dtype=torch.float32, device=dev)
    n_out = torch.empty((B,), dtype=torch.int32, device=dev)
    check(lib().oryon_lift_pairs(ptr(corrs), ptr(n_corr), B, n_cap, int(feat_hw[0]), int(feat_hw[1]),
                                 ptr(depth_a.contiguous()), depth_a.shape[1], depth_a.shape[2],
                                 ptr(depth_q.contiguous()), depth_q.shape[1], depth_q.shape[2],
                                 ptr(cam_a.contiguous()), ptr(cam_q.contiguous()), ptr(status), ptr(pa), ptr(pq), ptr(n_out),
                                 stream_ptr(dev)), "oryon_lift_pairs")
    return pa, pq, n_out


@_on_tensor_device
def lift_points(depth: torch.Tensor, cam9: torch.Tensor, x_idx: torch.Tensor, y_idx: torch.Tensor) -> torch.Tensor:
    """depth [H,W] f32, cam9 [9] f32, pixel indices [n] -> [n,3] f32 in the depth's unit."""
    dev = _lib.require_gpu(depth.device)
    n = x_idx.shape[0]
    out = torch.empty((n, 3), dtype=torch.float32, device=dev)
    xi = x_idx.to(torch.int32).contiguous()
    yi = y_idx.to(torch.int32).contiguous()
    check(lib().oryon_lift_points(ptr(depth), depth.shape[0], depth.shape[1], ptr(cam9.contiguous()), ptr(xi), ptr(yi), n,
                                  ptr(out), stream_ptr(dev)), "oryon_lift_points")
    return out


@_on_tensor_device
def kabsch_batched(A: torch.Tensor, B: torch.Tensor, w: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[nb,m,3] x2 (+ [nb,m]) fp32 -> [nb,4,4] fp32."""
    dev = _lib.require_gpu(A.device)
    A = A.to(torch.float32).contiguous()
    B = B.to(torch.float32).contiguous()
    w = None if w is None else w.to(torch.float32).contiguous()
    nb, m = A.shape[0], A.shape[1]
    T = torch.empty((nb, 4, 4), dtype=torch.float32, device=dev)
    check(lib().oryon_kabsch_batched(ptr(A), ptr(B), ptr(w), nb, m, ptr(T), stream_ptr(dev)), "oryon_kabsch_batched")
    return T


@_on_tensor_device
def pose_metrics(pred_pose: torch.Tensor, gt_pose: torch.Tensor, model_pts: torch.Tensor, pts_offset: Optional[torch.Tensor] = None,
                 model_of_pair: Optional[torch.Tensor] = None) -> torch.Tensor:
    """pred/gt [B,4,4] (metres), model_pts [M,3] (or the concatenation of several models with pts_offset [n+1] int32 and
    model_of_pair [B] int32) -> [B,4] = (ADD, ADD-S, rotation error deg, translation error cm), all on the device (f3)."""
    dev = _lib.require_gpu(pred_pose.device)
    B = pred_pose.shape[0]
    pred = pred_pose.to(dev, torch.float32).reshape(B, 16).contiguous()
    gt = gt_pose.to(dev, torch.float32).reshape(B, 16).contiguous()
    pts = model_pts.to(dev, torch.float32).contiguous()
    if pts_offset is None:
        pts_offset = torch.tensor([0, pts.shape[0]], dtype=torch.int32, device=dev)
    off_host = pts_offset.cpu()
    n_models = off_host.numel() - 1
    max_pts = int((off_host[1:] - off_host[:-1]).max()) if n_models > 0 else 0
    ws = torch.empty((B, 2), dtype=torch.float32, device=dev)
    out = torch.empty((B, 4), dtype=torch.float32, device=dev)
    mop = None if model_of_pair is None else model_of_pair.to(dev, torch.int32).contiguous()
    off_dev = pts_offset.to(dev, torch.int32).contiguous()
    check(lib().oryon_pose_metrics(ptr(pred), ptr(gt), B, ptr(pts), ptr(off_dev), n_models, max(1, max_pts),
                                   ptr(mop), ptr(ws), ptr(out), stream_ptr(dev)), "oryon_pose_metrics")
    return out


@_on_tensor_device
def pose_bop_errors(pred_pose: torch.Tensor, gt_pose: torch.Tensor, K: torch.Tensor, model_pts_mm: torch.Tensor, pts_offset: torch.Tensor,
                    syms: torch.Tensor, sym_offset: torch.Tensor, model_of_pair: Optional[torch.Tensor] = None,
                    max_points: int = 3) -> torch.Tensor:
    """MSSD (mm) and MSPD (px) of a batch of pairs on the device -> [B,2] float64 (utils/evaluator.py:258-275 +
    bop_toolkit_lib/pose_error.py:370-427).  pred / gt [B,4,4] metres, K [B,3,3]; model_pts_mm [sum M,3] millimetres with pts_offset
    [n+1]; syms [sum S,3,4] with sym_offset [n+1]; everything is handed over as float64 (the poses are rounded to float16 inside).
    max_points = 3 is the reference's behaviour (first three model points only, see include/oryon_hip.h); 0 = all points."""
    dev = _lib.require_gpu(pred_pose.device if pred_pose.is_cuda else model_pts_mm.device)
    B = pred_pose.shape[0]
    f64 = lambda t, shape: t.to(dev, torch.float64).reshape(shape).contiguous()
    pred, gt, Kc = f64(pred_pose, (B, 16)), f64(gt_pose, (B, 16)), f64(K, (B, 9))
    pts, sy = f64(model_pts_mm, (-1, 3)), f64(syms, (-1, 12))
    po, so = pts_offset.to(torch.int32).cpu(), sym_offset.to(torch.int32).cpu()
    n_models = po.numel() - 1
    max_syms = int((so[1:] - so[:-1]).max())
    ws = torch.empty((max(1, lib().oryon_pose_bop_workspace_bytes(B, max_syms) // 8),), dtype=torch.float64, device=dev)
    out = torch.empty((B, 2), dtype=torch.float64, device=dev)
    mop = None if model_of_pair is None else model_of_pair.to(dev, torch.int32).contiguous()
    po_d, so_d = po.to(dev), so.to(dev)               # named: a temporary would be freed (and its block re-used) before the launch
    check(lib().oryon_pose_bop_errors(ptr(pred), ptr(gt), ptr(Kc), B, ptr(pts), ptr(po_d), ptr(sy), ptr(so_d), n_models, max_syms,
                                      ptr(mop), int(max_points), ptr(ws), ptr(out), stream_ptr(dev)), "oryon_pose_bop_errors")
    return out


_x3_weights = {}
# Validation mode of the fp16x3 path (backbone.enable_fp16x3(True, guard=True)): every call first checks max|activation| against the
# float16 range (one reduction + a host sync per linear - for a first run with a real checkpoint, not for throughput) and evaluates
# that layer with torch's fp32 linear when the split would overflow; weights are checked once, when they are split.
X3_GUARD = False
X3_LIMIT = 60000.0            # below float16's 65504 with room for the rounding of `hi`
X3_EXACT_WEIGHTS = True       # use the two-product kernel for weights whose fp16 split has no low half (results are bit-identical)
x3_guard_fallbacks = 0        # layers evaluated by torch because an operand left the float16 range (guard mode)


def x3_range_flag(device=None, reset: bool = True) -> bool:
    """oryon_x3_range_flag: True when an fp16x3 kernel launched on `device`'s CURRENT stream saw an out-of-range / non-finite output since
    that stream's flag was last cleared.  Synchronises that stream: call once per forward.  The flag word belongs to the (device, stream)
    pair: other streams' launches neither raise nor clear it."""
    dev = _lib.require_gpu(torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device))
    v = ctypes.c_int(0)
    with torch.cuda.device(dev):
        check(lib().oryon_x3_range_flag(ctypes.byref(v), int(bool(reset)), stream_ptr(dev)), "oryon_x3_range_flag")
    return v.value != 0


def x3_range_reset(device=None) -> None:
    """Queue a clear of the current stream's fp16x3 range flag (no synchronisation, no read-back): the start of a forward.  The first
    call on a device allocates its flag table, so that no kernel launch does."""
    dev = _lib.require_gpu(torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device))
    with torch.cuda.device(dev):
        check(lib().oryon_x3_range_flag(None, 1, stream_ptr(dev)), "oryon_x3_range_flag")


def _split_weight_f16x3(weight: torch.Tensor):
    """(hi, lo) fp16 halves of an fp32 weight, made once per live tensor and in-place version and cached.  The entry holds a weak
    reference to the tensor it was made from: an address re-used by another tensor after a free can not hit a stale split."""
    key = id(weight)
    hit = _x3_weights.get(key)
    if hit is not None and (hit[0]() is not weight or hit[1] != (weight._version, weight.data_ptr(), tuple(weight.shape))):
        hit = None
    if hit is None:
        w = weight.detach().to(torch.float32).contiguous()
        if X3_GUARD and float(w.abs().amax()) >= X3_LIMIT:
            raise _lib.OryonError(f"fp16x3 linear: a weight of magnitude {float(w.abs().amax()):.3g} does not fit the float16 split "
                                  "(|w| must stay below 65504): evaluate this model with backbone.enable_fp16x3(False)")
        hi = torch.empty(w.shape, dtype=torch.float16, device=w.device)
        lo = torch.empty(w.shape, dtype=torch.float16, device=w.device)
        check(lib().oryon_split_f16x3(ptr(w), w.numel(), ptr(hi), ptr(lo), stream_ptr(w.device)), "oryon_split_f16x3")
        # A weight that IS an fp16 value in every element (what `clip.load` leaves in the reference's CLIPEncoder - an fp16 checkpoint
        # widened with `.to(torch.float32)`, models/vlm.py:19-22 - and what every frozen layer of a checkpoint fine-tuned from it still
        # holds) has an all-zero low half: the kernel then leaves out the a_hi * w_lo products (W_lo = NULL; 16 instead of 24 MFMAs per
        # k-step, bit-identical results).  One device read-back per weight and version, here, never per call.
        if X3_EXACT_WEIGHTS and w.shape[1] >= 64 and not bool(lo.any()):
            lo = None
        if len(_x3_weights) > 4096:
            _x3_weights.clear()
        ref = weakref.ref(weight, lambda _r, k=key: _x3_weights.pop(k, None))
        hit = _x3_weights[key] = (ref, (weight._version, weight.data_ptr(), tuple(weight.shape)), hi, lo)
    return hit[2], hit[3]


def linear_f16x3_supported(x: torch.Tensor, weight: torch.Tensor) -> bool:
    return (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and not torch.is_grad_enabled()
            and weight.dim() == 2 and weight.shape[1] % 32 == 0 and x.shape[-1] == weight.shape[1] and weight.numel() < 2 ** 30
            and (weight.shape[0] % 256 == 0 or (weight.shape[0] % 128 == 0 and weight.shape[1] >= 64)))


@_on_tensor_device
def linear_f16x3(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, quick_gelu: bool = False,
                 gelu: bool = False) -> torch.Tensor:
    """act(x @ weight.T + bias) for fp32 x [..., K], weight [N, K] on the fp16 matrix pipe with error-compensated operands (B4):
    fp32-grade results (~1e-6 relative) at ~3x the fp32-MFMA rate.  act: QuickGELU (CLIP) or erf-GELU (Swin) or none.  Inference only.
    Range: |x|, |w| < 65504 or the split overflows silently to inf (X3_GUARD checks it, see above); an operand below 2^-3 in magnitude
    has its low half in float16's subnormal range, i.e. an ABSOLUTE split error of up to 2^-25 instead of the relative 2^-22."""
    dev = _lib.require_gpu(x.device)
    assert not (quick_gelu and gelu)
    K, N = weight.shape[1], weight.shape[0]
    x2 = x.reshape(-1, K).contiguous()
    if X3_GUARD and not bool(torch.isfinite(x2).all() & (x2.abs().amax() < X3_LIMIT)):
        global x3_guard_fallbacks
        x3_guard_fallbacks += 1
        y = torch.nn.functional.linear(x2, weight, bias)
        y = y * torch.sigmoid(1.702 * y) if quick_gelu else (torch.nn.functional.gelu(y) if gelu else y)
        return y.view(*x.shape[:-1], N)
    hi, lo = _split_weight_f16x3(weight)
    out = torch.empty((x2.shape[0], N), dtype=torch.float32, device=dev)
    b = None if bias is None else bias.detach().to(torch.float32).contiguous()
    check(lib().oryon_linear_f16x3(ptr(x2), x2.shape[0], K, ptr(hi), ptr(lo) if lo is not None else None, ptr(b), N, 2 if gelu else (1 if quick_gelu else 0), ptr(out),
                                   stream_ptr(dev)), "oryon_linear_f16x3")
    return out.view(*x.shape[:-1], N)


def linear_f16x3_acc_supported(weight: torch.Tensor, out: torch.Tensor) -> bool:
    """Can `out += x @ weight.T + bias` run in place (oryon_linear_f16x3_acc)?  x is any fp32 CUDA tensor [..., K] with out's row count."""
    return (out.is_cuda and out.dtype == torch.float32 and out.is_contiguous() and weight.dtype == torch.float32 and weight.dim() == 2
            and not torch.is_grad_enabled() and not X3_GUARD and weight.shape[1] % 32 == 0 and weight.shape[1] >= 64
            and weight.shape[0] % 128 == 0 and weight.numel() < 2 ** 30 and out.shape[-1] == weight.shape[0])


@_on_tensor_device
def linear_f16x3_acc(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], out: torch.Tensor) -> torch.Tensor:
    """out += x @ weight.T + bias, in place (oryon_linear_f16x3_acc): the residual update of a transformer block done by its last linear.
    Same value per element as `out + linear_f16x3(x, weight, bias)` (one fp32 addition of the finished sum)."""
    dev = _lib.require_gpu(x.device)
    K, N = weight.shape[1], weight.shape[0]
    x2 = x.reshape(-1, K).contiguous()
    assert out.is_contiguous() and out.dtype == torch.float32 and out.numel() == x2.shape[0] * N and out.data_ptr() != x2.data_ptr()
    hi, lo = _split_weight_f16x3(weight)
    b = None if bias is None else bias.detach().to(torch.float32).contiguous()
    check(lib().oryon_linear_f16x3_acc(ptr(x2), x2.shape[0], K, ptr(hi), ptr(lo) if lo is not None else None, ptr(b), N, ptr(out),
                                       stream_ptr(dev)), "oryon_linear_f16x3_acc")
    return out


@_on_tensor_device
def mha_f16x3(qkv: torch.Tensor, heads: int) -> torch.Tensor:
    """Self-attention on the packed in_proj output qkv [N, L, 3*D] fp32 (head dim 64, no mask) -> [N, L, D] fp32 (B5)."""
    dev = _lib.require_gpu(qkv.device)
    N, L, D3 = qkv.shape
    D = D3 // 3
    assert qkv.dtype == torch.float32 and D == heads * 64 and D3 == 3 * D
    qkv = qkv.contiguous()
    out = torch.empty((N, L, D), dtype=torch.float32, device=dev)
    check(lib().oryon_mha_f16x3(ptr(qkv), N, L, heads, 64, ptr(out), stream_ptr(dev)), "oryon_mha_f16x3")
    return out


@_on_tensor_device
def sample_first_gate(n_valid1: torch.Tensor, n_a1: torch.Tensor, n_a: torch.Tensor, max_corrs: int) -> torch.Tensor:
    """Per-pair anchor counts of the second matcher stage: n_a where the first stage came up short although anchors were left out, else 0."""
    dev = _lib.require_gpu(n_a.device)
    out = torch.empty_like(n_a)
    check(lib().oryon_sample_first_gate(ptr(n_valid1), ptr(n_a1), ptr(n_a), n_a.shape[0], int(max_corrs), ptr(out), stream_ptr(dev)),
          "oryon_sample_first_gate")
    return out


@_on_tensor_device
def sample_first_merge_(n_a2, corrs2, n_valid2, n_sel2, status2, corrs1, n_valid1, n_sel1, status1) -> None:
    """In place: the second stage's correspondences / counts / status replace the first stage's for the pairs that were redone."""
    dev = _lib.require_gpu(corrs1.device)
    assert corrs1.shape == corrs2.shape
    check(lib().oryon_sample_first_merge(ptr(n_a2), ptr(corrs2), ptr(n_valid2), ptr(n_sel2), ptr(status2), corrs1.shape[0], corrs1.shape[1],
                                         ptr(corrs1), ptr(n_valid1), ptr(n_sel1), ptr(status1), stream_ptr(dev)), "oryon_sample_first_merge")
