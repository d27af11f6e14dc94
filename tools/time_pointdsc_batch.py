"""PointDSC registration time against the number of pairs per call (the launches are latency-bound at 64 pairs).  GPU box."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_solver
dev = torch.device("cuda", 0)
solver = build_solver(dev)
g = torch.Generator(device=dev).manual_seed(0)
for B in ([int(a) for a in sys.argv[1:]] or [32, 64, 128, 256]):
    src = torch.rand(B, 512, 3, generator=g, device=dev)
    tgt = src + 0.01 * torch.randn(B, 512, 3, generator=g, device=dev)
    n = torch.full((B,), 500, dtype=torch.int32, device=dev)
    status = torch.zeros(B, dtype=torch.int32, device=dev)
    for _ in range(3):
        solver.register(src, tgt, n, status)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        solver.register(src, tgt, n, status)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"B={B}: {ms:.3f} ms per call, {1e3 * ms / B:.1f} us per pair")
