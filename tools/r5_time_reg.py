"""64 registrations per call, stand-alone (development library: ORYON_PDSC_* switches live)."""
import os, sys
import _devlib  # noqa: F401
import torch
from bench import build_solver
dev = torch.device("cuda", 0)
solver = build_solver(dev)
g = torch.Generator(device=dev).manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
src = torch.rand(B, 512, 3, generator=g, device=dev)
tgt = src + 0.01 * torch.randn(B, 512, 3, generator=g, device=dev)
n = torch.full((B,), 500, dtype=torch.int32, device=dev)
status = torch.zeros(B, dtype=torch.int32, device=dev)
for _ in range(3):
    T = solver.register(src, tgt, n, status)
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        T = solver.register(src, tgt, n, status)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 20)
Tt = T[0] if isinstance(T, (tuple, list)) else T
print(f"B={B}: {sorted(ts)[2]:.3f} ms per call (min {min(ts):.3f})  pose checksum {float(Tt.double().sum()):.9f}")
