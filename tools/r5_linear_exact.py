"""Round 5: the fp16x3 linear with fp16-exact weights (two products) against general fp32 weights (three), CLIP ViT-L shapes at the
cfg2 batch (M = 128 images x 577 tokens).  Usage: python tools/r5_linear_exact.py"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oryon_amd import ops

torch.set_grad_enabled(False)
dev = "cuda"
M = 128 * 577
g = torch.Generator(device=dev).manual_seed(0)
print("| shape (M x K -> N) | act | three products ms | TFLOP/s (fp16 pipe) | two products ms | TFLOP/s (fp16 pipe) | speed-up |")
print("|---|---|---:|---:|---:|---:|---:|")
for K, N, act in ((1024, 3072, None), (1024, 1024, None), (1024, 4096, "quick"), (4096, 1024, None)):
    x = torch.randn(M, K, generator=g, device=dev)
    w3 = torch.randn(N, K, generator=g, device=dev) * K ** -0.5
    w2 = w3.half().float()
    b = torch.randn(N, generator=g, device=dev)
    res = []
    for w, terms in ((w3, 3), (w2, 2)):
        assert (ops._split_weight_f16x3(w)[1] is None) == (terms == 2)
        for _ in range(3):
            ops.linear_f16x3(x, w, b, quick_gelu=act == "quick")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.linear_f16x3(x, w, b, quick_gelu=act == "quick")
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        res += [ms, terms * 2.0 * M * K * N / ms / 1e9]
    print(f"| {M} x {K} -> {N} | {act or '-'} | {res[0]:.3f} | {res[1]:.0f} | {res[2]:.3f} | {res[3]:.0f} | {res[0] / res[2]:.2f} |")

print()
print("| residual update, M x K -> 1024 | weights | linear + add_layernorm(x, delta) ms | linear_acc (in place) + layernorm(x) ms |")
print("|---|---|---:|---:|")
lnw, lnb = torch.ones(1024, device=dev), torch.zeros(1024, device=dev)
for K in (1024, 4096):
    h = torch.randn(M, K, generator=g, device=dev)
    x = torch.randn(M, 1024, generator=g, device=dev)
    for kind in ("fp32", "fp16-valued"):
        w = torch.randn(1024, K, generator=g, device=dev) * K ** -0.5
        if kind != "fp32":
            w = w.half().float()
        b = torch.randn(1024, generator=g, device=dev)
        ref = x + ops.linear_f16x3(h, w, b)
        got = ops.linear_f16x3_acc(h, w, b, x.clone())
        assert torch.equal(ref, got), float((ref - got).abs().max())
        res = []
        for mode in (0, 1):
            xs = x.clone()
            def run():
                if mode == 0:
                    return ops.add_layernorm(xs, ops.linear_f16x3(h, w, b), lnw, lnb, 1e-5)
                ops.linear_f16x3_acc(h, w, b, xs)
                return ops.add_layernorm(xs, None, lnw, lnb, 1e-5)
            for _ in range(3):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run()
            e1.record()
            torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / 10)
        print(f"| {M} x {K} | {kind} | {res[0]:.3f} | {res[1]:.3f} |")
