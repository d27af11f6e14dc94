"""Why are the tower's fp16x3 linears slower in place than in tools/bench_linear_x3.py?  Times one captured call against randn."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oryon_amd import ops
from oryon_amd.net import Oryon, default_model_args
from oryon_amd.backbone import clip as _c
_c.FP16X3_LINEAR = True
torch.manual_seed(0)
m = Oryon(default_model_args(), "cuda").eval()
rgb = torch.rand(128, 3, 224, 224, device="cuda")
cap = []
orig = ops.linear_f16x3
def spy(x, w, b=None, quick_gelu=False):
    if len(cap) < 8:
        cap.append((x.detach().clone(), w, b, quick_gelu, x.shape, x.stride(), x.is_contiguous()))
    return orig(x, w, b, quick_gelu=quick_gelu)
ops.linear_f16x3 = spy
def t(fn, n=50):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
with torch.no_grad():
    m.vlm.encode_image(rgb)
    ops.linear_f16x3 = orig
    for x, w, b, g, shp, st, cont in cap:
        x2 = x.reshape(-1, x.shape[-1])
        r = torch.randn_like(x2)
        print(tuple(shp), st, cont, tuple(w.shape), "gelu" if g else "", f"captured {t(lambda: orig(x, w, b, quick_gelu=g)):.3f} ms | randn {t(lambda: orig(r, w, b, quick_gelu=g)):.3f} ms",
              f"| x absmax {float(x.abs().max()):.2e} mean|x| {float(x.abs().mean()):.2e} w absmax {float(w.abs().max()):.2e}")
