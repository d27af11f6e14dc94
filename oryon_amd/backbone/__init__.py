"""Frozen feature towers of Oryon.forward (CLIP ViT-L/14@336 image + text, Swin-B guidance): plain PyTorch modules with the reference's
state-dict names, plus the HIP inference paths behind module switches."""
import contextlib


def enable_fp16x3(flag: bool = True, guard: bool = False) -> None:
    """fp32-grade fast inference path of both towers, the fusion module's linears and the decoder (HIP kernels, csrc/decoder.hip): linears and the CLIP attention as error-compensated fp16x3 MFMA kernels (B4, B5),
    Swin window attention and residual-add + LayerNorm as single fp32 kernels (B3, B2).  Off by default: torch fp32 everywhere.
    Results stay within ~1e-5 of the fp32 evaluation (tests/test_backbone_pins.py); takes effect under torch.no_grad() on CUDA only.

    Range (round 5; per stream since round 6): the fp16x3 split needs |x| < 65504.  Every PRODUCER of a tensor the fp16x3 kernels split
    (the linears, the convolutions / up-convolutions, the residual-add + LayerNorm pass for the residual stream) raises the flag word of
    the (device, stream) it runs on when one of its values is not a finite value below 60000 - what an out-of-range operand produces in
    every product it enters; the attention kernels and the single-slab decoder convolutions carry no check of their own (their operands
    are checked outputs, and whatever they produce is consumed by a checked kernel: include/oryon_hip.h, oryon_x3_range_flag).
    `Oryon.forward` clears its stream's word before the forward and reads it ONCE after (`ops.x3_range_reset` / `ops.x3_range_flag`): a
    forward whose flag came back set is evaluated again with the torch fp32 modules (`net.Oryon.x3_range_fallbacks` counts them).
    Forwards on other streams / threads / model instances neither raise nor clear it.  Cost: one 4-byte read-back per forward.

    guard=True additionally checks every fp16x3 LINEAR's operands on the host before the call (one reduction + a sync per layer) and
    evaluates that layer with torch when they are out of range: the slow, layer-precise mode for a first run with a new checkpoint
    (`ops.x3_guard_fallbacks` counts layers).  The fused kernels stay on in both modes."""
    from . import clip, fusion, swin
    from .. import ops
    import torch
    if flag and torch.cuda.is_available():
        ops.x3_range_reset()                    # allocates the device's flag table now, so that no kernel launch ever does
    ops.X3_GUARD = bool(flag and guard)
    clip.FP16X3_LINEAR = bool(flag)
    fusion.FP16X3_LINEAR = bool(flag)           # guided Swin blocks' linears + the CLIP 1x1 projection of ImageTextFusion
    fusion.FUSED_KERNELS = bool(flag)           # window attention, whole-map convolutions, class layers of the fusion module
    fusion.HIP_DECODER = bool(flag)             # StandardDecoder.forward through oryon_decoder_forward (csrc/decoder.hip)
    swin.FUSED_F32_ATTENTION = bool(flag)


def fp16x3_enabled() -> bool:
    from . import clip, fusion, swin
    return bool(clip.FP16X3_LINEAR or fusion.FP16X3_LINEAR or fusion.FUSED_KERNELS or fusion.HIP_DECODER or swin.FUSED_F32_ATTENTION)


@contextlib.contextmanager
def fp16x3_disabled():
    """Temporarily evaluate with the torch fp32 modules (the range-flag fallback of Oryon.forward)."""
    from . import clip, fusion, swin
    from .. import ops
    saved = (ops.X3_GUARD, clip.FP16X3_LINEAR, fusion.FP16X3_LINEAR, fusion.FUSED_KERNELS, fusion.HIP_DECODER, swin.FUSED_F32_ATTENTION)
    enable_fp16x3(False)
    try:
        yield
    finally:
        (ops.X3_GUARD, clip.FP16X3_LINEAR, fusion.FP16X3_LINEAR, fusion.FUSED_KERNELS, fusion.HIP_DECODER, swin.FUSED_F32_ATTENTION) = saved
