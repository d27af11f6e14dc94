import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oryon_amd.net import Oryon, default_model_args
dev = "cuda"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dt = torch.bfloat16 if (len(sys.argv) > 2 and sys.argv[2] == "bf16") else torch.float32
if len(sys.argv) > 2 and sys.argv[2] == "fp16x3":
    from oryon_amd.backbone import clip as _c
    _c.FP16X3_LINEAR = True
    from oryon_amd.backbone import swin as _s
    _s.FUSED_F32_ATTENTION = True
torch.manual_seed(0)
m = Oryon(default_model_args(), dev).eval().to(dt)
rgb = torch.rand(2 * B, 3, 224, 224, device=dev, dtype=dt)
toks = torch.randint(1, 49000, (1, 80, 77)); toks[..., 12] = 49407; toks[..., 13:] = 0
def ev(): e = torch.cuda.Event(enable_timing=True); e.record(); return e
with torch.no_grad():
    prompt = m.vlm.encode_tokens(toks.expand(2 * B, 80, 77).contiguous()).unsqueeze(1).to(dt)
    for it in range(3):
        e0 = ev(); vis = m.vlm.encode_image(rgb)
        e1 = ev(); guid = m.get_guidance_embeds(rgb)
        e2 = ev(); feats = m.fusion(vis, prompt, guid)
        e3 = ev(); mask, fm = m.decoder(feats, guid)
        e4 = ev(); torch.cuda.synchronize()
        print(f"it{it} B={B} {dt}: clip {e0.elapsed_time(e1):.1f} ms | swin {e1.elapsed_time(e2):.1f} | fusion {e2.elapsed_time(e3):.1f} | decoder {e3.elapsed_time(e4):.1f} | total {e0.elapsed_time(e4):.1f}")
