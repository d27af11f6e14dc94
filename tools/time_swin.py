import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oryon_amd.backbone.swin import SwinGuidance, guidance_embeds
dev, dt = "cuda", torch.bfloat16
m = SwinGuidance().to(dev).eval().to(dt)
rgb = torch.rand(64, 3, 224, 224, device=dev, dtype=dt)
with torch.no_grad():
    for _ in range(3):
        g = guidance_embeds(m, rgb)
    torch.cuda.synchronize()
