"""Round 5: same-box A/B of the CLIP image tower (24 blocks, 128 images) with the residual adds folded into the blocks' last linears
(backbone.clip.ACC_RESIDUAL) and without, general and fp16-valued weights.  Usage: python tools/r5_acc_ab.py"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oryon_amd.backbone import clip as C

torch.set_grad_enabled(False)
C.FP16X3_LINEAR = True
torch.manual_seed(0)
m = C.CLIP(C.CLIPConfig.vit_l14_336()).cuda().eval()
img = torch.randn(128, 3, 336, 336, device="cuda")
for kind in ("general fp32 weights", "fp16-valued weights"):
    if kind.startswith("fp16"):
        for p in m.parameters():
            if p.dim() >= 2:
                p.copy_(p.half().float())
    for rep in range(2):
        for acc in (False, True):
            C.ACC_RESIDUAL = acc
            for _ in range(2):
                y = m.patch_tokens(img)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                y = m.patch_tokens(img)
            e1.record()
            torch.cuda.synchronize()
            print(f"{kind}: ACC_RESIDUAL={acc}: {e0.elapsed_time(e1) / 5:.2f} ms per 128 images")
