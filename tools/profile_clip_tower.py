"""encode_image only (fp16x3 path), for a rocprofv3 --kernel-trace --stats run: what the CLIP tower's time is made of."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oryon_amd.net import Oryon, default_model_args
from oryon_amd.backbone import clip as _c, swin as _s
_c.FP16X3_LINEAR = True
_s.FUSED_F32_ATTENTION = True
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
what = sys.argv[2] if len(sys.argv) > 2 else "clip"
torch.manual_seed(0)
m = Oryon(default_model_args(), "cuda").eval()
rgb = torch.rand(2 * B, 3, 224, 224, device="cuda")
with torch.no_grad():
    for _ in range(5):
        out = m.vlm.encode_image(rgb) if what == "clip" else m.get_guidance_embeds(rgb)
    torch.cuda.synchronize()
