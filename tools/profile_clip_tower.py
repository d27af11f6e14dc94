"""One part of the forward (clip | swin | head = fusion + decoder) on the fp16x3 path, for a rocprofv3 --kernel-trace --stats run:
what that part's time is made of."""
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
    if what == "head":                       # fusion + decoder only (their inputs are computed once, then 5 timed-equivalent passes)
        toks = torch.randint(1, 49000, (1, 80, 77)); toks[..., 12] = 49407; toks[..., 13:] = 0
        prompt = m.vlm.encode_tokens(toks.expand(2 * B, 80, 77).contiguous()).unsqueeze(1)
        vis, guid = m.vlm.encode_image(rgb), m.get_guidance_embeds(rgb)
        for _ in range(2):
            m.decoder(m.fusion(vis, prompt, guid), guid)
        torch.cuda.synchronize()
        print("HEAD_PASSES_START")
        for _ in range(5):
            m.decoder(m.fusion(vis, prompt, guid), guid)
    else:
        for _ in range(5):
            out = m.vlm.encode_image(rgb) if what == "clip" else m.get_guidance_embeds(rgb)
    torch.cuda.synchronize()
