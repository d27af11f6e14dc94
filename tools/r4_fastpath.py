"""Round-4 probe: fusion + decoder fast path (fp16x3 linears, HIP decoder) alone, 128 images - for a rocprofv3 kernel trace."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import oryon_amd
oryon_amd.configure()
from oryon_amd.backbone import enable_fp16x3
from oryon_amd.backbone.fusion import ImageTextFusion, StandardDecoder
torch.manual_seed(0)
dev = "cuda"
fu = ImageTextFusion(dev).eval()
de = StandardDecoder(dev, True, True, input_dim=128, decoder_dims=[64, 32]).eval()
n = int(os.environ.get("N_IMG", "128"))
img = torch.randn(n, 24, 24, 1024, device=dev).permute(0, 3, 1, 2)
text = torch.randn(n, 1, 80, 768, device=dev)
guid = [torch.randn(n, 24, 24, 512, device=dev).permute(0, 3, 1, 2), torch.randn(n, 48, 48, 256, device=dev).permute(0, 3, 1, 2),
        torch.randn(n, 96, 96, 128, device=dev).permute(0, 3, 1, 2)]
enable_fp16x3(os.environ.get("FAST", "1") == "1")
with torch.no_grad():
    for _ in range(2):
        de(fu(img, text, guid), guid)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        x = fu(img, text, guid)
    e1.record()
    for _ in range(5):
        de(x, guid)
    e2 = torch.cuda.Event(enable_timing=True)
    e2.record()
    torch.cuda.synchronize()
print(f"fusion {e0.elapsed_time(e1) / 5:.3f} ms  decoder {e1.elapsed_time(e2) / 5:.3f} ms")
