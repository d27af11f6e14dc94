"""Round-4 probe: where do the 20 ms of the decode stage set go?  Times fusion and the decoder's blocks separately (HIP events, 128 images)."""
import sys, time, torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import oryon_amd
oryon_amd.configure()
from oryon_amd.net import Oryon, default_model_args

dev = torch.device("cuda:0")
B = 64
torch.manual_seed(4321)
model = Oryon(default_model_args(), dev).eval()
gen = torch.Generator(device=dev).manual_seed(99)
rgb = torch.rand((2 * B, 3, 224, 224), generator=gen, device=dev)
toks = torch.randint(1, 49000, (1, 80, 77), generator=torch.Generator().manual_seed(7))
toks[..., 12] = 49407
toks[..., 13:] = 0
toks = toks.expand(B, 80, 77).contiguous()
with torch.no_grad():
    enc = (model.vlm.encode_image(rgb), model.get_guidance_embeds(rgb))
    prompt = model.vlm.encode_tokens(toks).unsqueeze(1)
    prompt = torch.cat([prompt, prompt])
print("enc", enc[0].shape, [g.shape for g in enc[1]], prompt.shape)


def timed(name, fn, n=3):
    with torch.no_grad():
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            out = fn()
        e1.record()
        torch.cuda.synchronize()
    print(f"{name:40s} {e0.elapsed_time(e1) / n:8.3f} ms")
    return out


fu, de = model.fusion, model.decoder
x = timed("fusion", lambda: fu(enc[0], prompt, enc[1]))
timed("decoder", lambda: de(x, enc[1]))
# fusion pieces
img = timed("  fusion.clip_conv", lambda: fu.clip_conv(enc[0].reshape(2 * B, 1024, 576)).reshape(2 * B, -1, 24, 24))
corr = torch.randn(2 * B, 80, 24, 24, device=dev)
x0 = timed("  fusion.conv1 7x7", lambda: fu.conv1(corr))
app = timed("  fusion.guidance_projection", lambda: fu.guidance_projection(enc[1][0]))
with torch.no_grad():
    t = prompt.mean(dim=-2)
    t = fu.text_guidance_projection(t / t.norm(dim=-1, keepdim=True))
    xx = x0.view(2 * B, 1, -1, 24, 24).permute(0, 2, 1, 3, 4)
for i, layer in enumerate(fu.layers):
    timed(f"  fusion.layer{i}", lambda: layer(xx, app, t))
    timed(f"    swin pair", lambda: layer.swin_block(xx, app))
# decoder pieces
g = enc[1]
pg0 = timed("  dec.guidance_proj0 (256->32 @48)", lambda: de.decoder_guidance_projection[0](g[1]))
pg1 = timed("  dec.guidance_proj1 (128->16 @96)", lambda: de.decoder_guidance_projection[1](g[2]))
y = x.permute(0, 2, 1, 3, 4).reshape(2 * B, 128, 24, 24)
y1 = timed("  dec.decoder1 (->64 @48)", lambda: de.decoder1(y, pg0))
y2 = timed("  dec.decoder2 (->32 @96)", lambda: de.decoder2(y1, pg1))
y3 = timed("  dec.decoder3 (->32 @192)", lambda: de.decoder3(y2, None))
timed("    decoder3.up", lambda: de.decoder3.up(y2))
u = de.decoder3.up(y2)
timed("    decoder3.conv", lambda: de.decoder3.conv(u))
timed("    decoder3.conv[0] conv3x3", lambda: de.decoder3.conv.double_conv[0](u))
c = de.decoder3.conv.double_conv[0](u)
timed("    decoder3.conv[1] GN", lambda: de.decoder3.conv.double_conv[1](c))
timed("  dec.head", lambda: de.head(y3))
timed("  featmap clone", lambda: y3.view(2 * B, 32, 192, 192).clone())

# the HIP decoder (csrc/decoder.hip) on the same inputs
from oryon_amd.backbone import fusion as F_
F_.enable_hip_decoder(True)
lg, fm = timed("decoder (HIP, oryon_decoder_forward)", lambda: de(x, enc[1]), n=5)
F_.enable_hip_decoder(False)
with torch.no_grad():
    lg0, fm0 = de(x, enc[1])
print("max rel diff featmap", float((fm - fm0).abs().max() / fm0.abs().max()), "logits", float((lg - lg0).abs().max() / lg0.abs().max()))

from oryon_amd.backbone import enable_fp16x3
enable_fp16x3(True)
x3 = timed("fusion (fp16x3 linears)", lambda: fu(enc[0], prompt, enc[1]), n=5)
timed("  swin pair (fp16x3)", lambda: fu.layers[0].swin_block(xx, app), n=5)
timed("fusion + decoder (fast path)", lambda: de(fu(enc[0], prompt, enc[1]), enc[1]), n=5)
enable_fp16x3(False)
print("fusion fp16x3 vs fp32 max rel", float((x3 - x).abs().max() / x.abs().max()))
