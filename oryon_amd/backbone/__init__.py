"""Frozen feature towers of Oryon.forward (CLIP ViT-L/14@336 image + text, Swin-B guidance): plain PyTorch modules with the reference's
state-dict names, plus the HIP inference paths behind module switches."""


def enable_fp16x3(flag: bool = True, guard: bool = False) -> None:
    """fp32-grade fast inference path of both towers, the fusion module's linears and the decoder (HIP kernels, csrc/decoder.hip): linears and the CLIP attention as error-compensated fp16x3 MFMA kernels (B4, B5),
    Swin window attention and residual-add + LayerNorm as single fp32 kernels (B3, B2).  Off by default: torch fp32 everywhere.
    Results stay within ~1e-5 of the fp32 evaluation (tests/test_backbone_pins.py); takes effect under torch.no_grad() on CUDA only."""
    from . import clip, fusion, swin
    from .. import ops
    # guard: validation mode for a first run with a real checkpoint (the tests use random-init weights whose activations are O(10);
    # the released CLIP ViT-L has outlier activations): every fp16x3 linear checks its operands against the float16 range first and
    # falls back to torch's fp32 linear when the split would overflow - slow (a host sync per layer), counted in ops.x3_guard_fallbacks
    ops.X3_GUARD = bool(flag and guard)
    clip.FP16X3_LINEAR = bool(flag)
    fusion.FP16X3_LINEAR = bool(flag)           # guided Swin blocks' linears + the CLIP 1x1 projection of ImageTextFusion
    # the fusion / decoder kernels that split activations to fp16 themselves (window attention, whole-map convolutions, class layers, the
    # decoder) have no range check of their own: in guard mode they stay on the torch fp32 modules
    fusion.FUSED_KERNELS = bool(flag and not guard)
    fusion.HIP_DECODER = bool(flag and not guard)   # StandardDecoder.forward through oryon_decoder_forward (csrc/decoder.hip)
    swin.FUSED_F32_ATTENTION = bool(flag)
