"""fp16x3 linear (B4) vs torch fp32 linear at the CLIP ViT-L shapes: accuracy against an fp64 reference and time (GPU box)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oryon_amd import ops
torch.backends.cuda.matmul.allow_tf32 = False
dev = "cuda"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 128 * 577
g = torch.Generator(device=dev).manual_seed(0)
for K, N, name in ((1024, 3072, "qkv"), (1024, 1024, "out"), (1024, 4096, "fc1+gelu"), (4096, 1024, "fc2")):
    x = torch.randn(M, K, generator=g, device=dev)
    w = torch.randn(N, K, generator=g, device=dev) * K ** -0.5
    b = torch.randn(N, generator=g, device=dev)
    gelu = name.startswith("fc1")
    f32 = lambda: (lambda y: y * torch.sigmoid(1.702 * y) if gelu else y)(torch.nn.functional.linear(x, w, b))
    x3 = lambda: ops.linear_f16x3(x, w, b, quick_gelu=gelu)
    ref = torch.nn.functional.linear(x[:4096].double(), w.double(), b.double())
    if gelu:
        ref = ref * torch.sigmoid(1.702 * ref)
    e32 = float((f32()[:4096].double() - ref).abs().max() / ref.abs().max())
    ex3 = float((x3()[:4096].double() - ref).abs().max() / ref.abs().max())
    def t(fn, n=int(os.environ.get("REPS", "5"))):
        fn(); fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    t32, tx3 = t(f32), t(x3)
    fl = 2.0 * M * K * N
    print(f"{name:9s} M={M} K={K} N={N}: torch fp32 {t32:.3f} ms ({fl / t32 / 1e9:.0f} TF/s, err {e32:.1e}) | fp16x3 {tx3:.3f} ms "
          f"({fl / tx3 / 1e9:.0f} TF/s fp32-equivalent, {3 * fl / tx3 / 1e9:.0f} TF/s on the fp16 pipe, err {ex3:.1e})")
