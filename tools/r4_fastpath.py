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


def timed(name, fn, n=5):
    with torch.no_grad():
        fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            out = fn()
        b.record()
        torch.cuda.synchronize()
    print(f"  {name:44s} {a.elapsed_time(b) / n:7.3f} ms")
    return out


if os.environ.get("SPLIT", "0") == "1":
    import torch.nn.functional as F
    B, D, H, W = img.shape
    im = timed("clip_conv (1x1, 1024->768)", lambda: fu.clip_conv(img.reshape(B, D, H * W)).reshape(B, -1, H, W) if os.environ.get("FAST", "1") != "1" else
               __import__("oryon_amd").ops.linear_f16x3(img.permute(0, 2, 3, 1).reshape(B * H * W, D), fu.clip_conv.weight.view(768, D), fu.clip_conv.bias).view(B, H, W, -1).permute(0, 3, 1, 2))
    corr = timed("normalize + cost volume einsum", lambda: torch.einsum("bchw,btpc->bpthw", F.normalize(im, dim=1), F.normalize(text, dim=-1)))
    x0 = timed("conv1 7x7 (80->128)", lambda: fu.conv1(corr.permute(0, 2, 1, 3, 4).reshape(B, 80, H, W)))
    app = timed("guidance_projection 3x3 (512->128) + ReLU", lambda: fu.guidance_projection(guid[0]))
    with torch.no_grad():
        t = text.mean(dim=-2)
        t = fu.text_guidance_projection(t / t.norm(dim=-1, keepdim=True))
        xx = x0.view(B, 1, -1, H, W).permute(0, 2, 1, 3, 4)
    y = timed("layer0.swin_block", lambda: fu.layers[0].swin_block(xx, app))
    timed("layer0.attention (class layer)", lambda: fu.layers[0].attention(y, t))
    blk = fu.layers[0].swin_block.block_2
    with torch.no_grad():
        yy = xx.permute(0, 2, 3, 4, 1).reshape(B, H * W, 128).contiguous()
        g = fu.layers[0].swin_block.guidance_norm(app.permute(0, 2, 3, 1).reshape(B, H * W, -1))
    timed("  one guided Swin block (shifted)", lambda: blk(yy, g))
