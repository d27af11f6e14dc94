"""Round 5: the Swin guidance tower's linear shapes at the cfg2 batch (128 images, 384 x 384 -> 96 x 96 tokens) on the fp16x3 linear:
time against the tensor traffic each one cannot avoid.  Usage: python tools/r5_swin_linears.py"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _devlib  # noqa: F401  (ORYON_GEMM_X3_VARIANT=1: the small-tile kernel where N % 256 == 0)
from oryon_amd import ops

torch.set_grad_enabled(False)
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
M1 = 128 * 96 * 96
shapes = [("s1 qkv", M1, 128, 384, False), ("s1 proj", M1, 128, 128, False), ("s1 fc1", M1, 128, 512, True), ("s1 fc2", M1, 512, 128, False),
          ("merge1", M1 // 4, 512, 256, False), ("s2 qkv", M1 // 4, 256, 768, False), ("s2 proj", M1 // 4, 256, 256, False),
          ("s2 fc1", M1 // 4, 256, 1024, True), ("s2 fc2", M1 // 4, 1024, 256, False), ("merge2", M1 // 16, 1024, 512, False)]
print("| linear | M x K -> N | ms | GB moved (A + C) | TB/s | TFLOP/s fp16 pipe |")
print("|---|---|---:|---:|---:|---:|")
tot = 0.0
for name, M, K, N, gelu in shapes:
    x = torch.randn(M, K, generator=g, device=dev)
    w = torch.randn(N, K, generator=g, device=dev) * K ** -0.5
    b = torch.randn(N, generator=g, device=dev)
    for _ in range(3):
        ops.linear_f16x3(x, w, b, gelu=gelu)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.linear_f16x3(x, w, b, gelu=gelu)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    gb = 4.0 * M * (K + N) / 1e9
    tot += ms * (2 if name.startswith("s") else 1)
    print(f"| {name} | {M} x {K} -> {N} | {ms:.3f} | {gb:.2f} | {gb / ms:.2f} | {6.0 * M * K * N / ms / 1e9:.0f} |")
    del x
print(f"two blocks per stage + the mergings: {tot:.2f} ms")
